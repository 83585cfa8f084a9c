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
//
// TABLES: whatever they say, like the reference -- mismatching tables only run slower.
//   The reference multiplies element (i, j) by W[i*n2 + j] as it stands and runs the rows through the n2 table
//   (src/lib/ntt_4step/ntt_4step.cu:1049-1058, 776-779).  The fast path here runs the transform as the ring's Merge plan
//   and derives every twiddle from n1_table and ONE row of W, which equals the reference's result exactly when the three
//   tables are those of ONE root g with g^(N/2) = -1 in the layout NTTParameters4Step generates
//   (src/lib/common/nttparameters.cu:356-444):
//       FORWARD  W[r*n2 + j] = g^(brev(r, log n1) * j)     INVERSE  W[r*n2 + c] = g^(r * brev(c, log n2))
//       n1_table[i] = (g^n2)^brev(i, log n1 - 1)           n2_table[i] = (g^n1)^brev(i, log n2 - 1)
//   (true for NTTParameters4Step on the host and GPU_GeneratePowerTable / GPU_Generate4StepW on the device).  Every
//   GPU_4STEP_NTT / GPU_4STEP_NTT_NaturalOrder call therefore VERIFIES all N + n1/2 + n2/2 table words inside its
//   preparation launch -- one modular product per word against its neighbours, no extra launch, no host
//   synchronisation -- and tables that are anything else (the reference's timing program passes random words with n1 / n2
//   swapped, benchmark/bench_4step_ntt.cu:80-90) hand the call, on the device, to the element-by-element algorithm: bit
//   for bit what the tables say.  (Who runs it: the fast kernels themselves as far as their blocks can -- all of it for
//   rings up to 2^16 and forward 2^17, the n1-point phase for every ring -- and the element-by-element Barrett kernels
//   enqueued behind the call for the rest.)  A FourStepPlan checks once, in its constructor
//   (fast_path() tells).  GPU_NTT_SetOption("check_4step_tables", "0") opts out for callers that guarantee the layout
//   above (no check, no generic launches behind the call).
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
    // GPU_4STEP_NTT rebuilds, on every call, the ring's Merge table (Shoup pairs {w^k, floor(w^k * 2^W / q)}, 2 words per
    // coefficient) from the caller's n1 table and one row of W (table contract above) into a library-owned scratch
    // buffer -- at 2^24 that is 256 MiB written before the first sweep, a quarter of a batch-1 call.  A plan does it
    // once, into memory the caller owns (or the plan allocates), and execute() launches the sweeps only: no
    // allocation, no synchronisation, no preparation launch, hipGraph-capturable from the first call.
    //   natural_order false: execute() == GPU_4STEP_NTT (n2 x n1 in, n1 x n2 out, in != out);
    //   natural_order true:  execute() == GPU_4STEP_NTT_NaturalOrder (device_in is scratch).
    // cfg.stream is the stream the preparation runs on; the constructor waits for it (one host wait per plan), so
    // execute() may run on any stream without a dependency on the construction stream.  batch_hint: the batch
    // size the plan will mostly run (decides the tile the Merge table's last stages are permuted for; any batch
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
