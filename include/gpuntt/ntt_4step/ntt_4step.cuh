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

    // Prepared 4-step transform (extension; the 4-step counterpart of NTTPlan<T>, ntt_merge/ntt.cuh).
    // GPU_4STEP_NTT re-derives, on every call, the Shoup pairs of the caller's n1 / n2 / W tables into a
    // library-owned scratch buffer -- at 2^24 that is 128 MiB read and 256 MiB written before the first
    // sweep, most of a batch-1 call.  A plan does it once, into memory the caller owns (or the plan
    // allocates), and execute() launches the sweeps only: no allocation, no synchronisation, no
    // preparation launch, hipGraph-capturable from the first call.
    //   natural_order false: execute() == GPU_4STEP_NTT (n2 x n1 in, n1 x n2 out, in != out);
    //   natural_order true:  execute() == GPU_4STEP_NTT_NaturalOrder (device_in is scratch).
    // cfg.stream is the stream the preparation runs on; the constructor waits for it (one host wait per plan), so
    // execute() may run on any stream without a dependency on the construction stream.  batch_hint: the batch
    // size the plan will mostly run (decides the row-pass tile the n2 table is laid out for; any batch
    // size is correct).  The caller's tables must stay alive when fast_path() is false (moduli without
    // lazy headroom run GPU_4STEP_NTT / _NaturalOrder on them).
    template <typename T> class FourStepPlan
    {
      public:
        static size_t workspace_bytes(int n_power);
        FourStepPlan(Root<T>* n1_root_of_unity_table, Root<T>* n2_root_of_unity_table,
                     Root<T>* W_root_of_unity_table, Modulus<T> modulus, ntt4step_configuration<T> cfg,
                     bool natural_order, int batch_hint, void* workspace_device = nullptr);
        ~FourStepPlan();
        FourStepPlan(const FourStepPlan&) = delete;
        FourStepPlan& operator=(const FourStepPlan&) = delete;
        void execute(T* device_in, T* device_out, int batch_size, stream_t stream) const;
        bool fast_path() const;

      private:
        struct Impl;
        Impl* p_;
    };

} // namespace gpuntt
