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

} // namespace gpuntt
