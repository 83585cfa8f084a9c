// gpuntt/ntt_4step/ntt_4step.cuh -- 4-Step NTT launch API for MI355X (gfx950).
//
// Drop-in for reference src/include/gpuntt/ntt_4step/ntt_4step.cuh: config structs :19-33,
// GPU_Transpose :46-49, GPU_4STEP_NTT single/RNS :278-308.
//
// Semantics (identical to the reference, N = n1 x n2 from NTTParameters4Step, cyclic only,
// 12 <= n_power <= 24):
//   GPU_4STEP_NTT takes the polynomial as an n2 x n1 row-major matrix (i.e. already
//   transposed by GPU_Transpose(in, out, n1, n2, ...)) and produces an n1 x n2 matrix:
//     rows:  n1-point transform of every row (CT forward / GS inverse, n1 table)
//            -> transposed to n1 x n2 -> element (i, j) *= W[i*n2 + j]
//            -> n2-point transform of every row (n2 table) [-> * n^-1 for INVERSE]
//   A final GPU_Transpose(out, res, n1, n2, ...) yields NTT_4STEP_CPU's flat order.
//   Tables: n1/n2 tables bit-reversed (n/2 entries), W natural generated order (N entries).
//   device_in != device_out.  Unlike the reference (which always used the legacy default
//   stream) every launch honours cfg.stream; GPU_Transpose runs on the default stream.
//   Unsupported n_power: message on stdout, no throw (reference ntt_4step.cu:2529-2532).
#pragma once

#include "gpuntt/ntt_4step/ntt_4step_cpu.cuh"

namespace gpuntt
{
    template <typename T> struct ntt4step_configuration
    {
        int n_power;
        type ntt_type;
        Ninverse<T> mod_inverse;
        stream_t stream;
    };

    template <typename T> struct ntt4step_rns_configuration
    {
        int n_power;
        type ntt_type;
        Ninverse<T>* mod_inverse;
        stream_t stream;
    };

    // per polynomial: (row x col) row-major -> (col x row) row-major
    template <typename T>
    __host__ void GPU_Transpose(T* polynomial_in, T* polynomial_out, const int row,
                                const int col, const int n_power, const int batch_size);

    template <typename T>
    __host__ void GPU_4STEP_NTT(T* device_in, T* device_out, Root<T>* n1_root_of_unity_table,
                                Root<T>* n2_root_of_unity_table, Root<T>* W_root_of_unity_table,
                                Modulus<T> modulus, ntt4step_configuration<T> cfg,
                                int batch_size);

    template <typename T>
    __host__ void GPU_4STEP_NTT(T* device_in, T* device_out, Root<T>* n1_root_of_unity_table,
                                Root<T>* n2_root_of_unity_table, Root<T>* W_root_of_unity_table,
                                Modulus<T>* modulus, ntt4step_rns_configuration<T> cfg,
                                int batch_size, int mod_count);

    // ---- extension (not in the reference) ------------------------------------------------
    // The natural-order pipeline of the reference's 4-step examples as ONE call:
    //   FORWARD  == GPU_Transpose(in, t, n1, n2) -> GPU_4STEP_NTT(t, u, FORWARD) -> GPU_Transpose(u, out, n1, n2)
    //            == NTT_4STEP_CPU<T>::ntt(in)      (example/ntt_4step/test_4step_ntt.cu:147-178)
    //   INVERSE  == intt_first_transpose(in) -> GPU_4STEP_NTT(INVERSE) -> GPU_Transpose
    //            == NTT_4STEP_CPU<T>::intt(in)     (example/ntt_4step/test_4step_intt.cu:81-179)
    // The forward direction runs in three HBM sweeps instead of five (the column transforms work
    // on the row-major input directly, the last row pass stores transposed).  device_in is used
    // as scratch (as the three-call sequence does) and must differ from device_out; every launch
    // is on cfg.stream.
    template <typename T>
    __host__ void GPU_4STEP_NTT_NaturalOrder(T* device_in, T* device_out, Root<T>* n1_root_of_unity_table,
                                             Root<T>* n2_root_of_unity_table, Root<T>* W_root_of_unity_table,
                                             Modulus<T> modulus, ntt4step_configuration<T> cfg,
                                             int batch_size);

} // namespace gpuntt
