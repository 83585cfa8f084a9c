// gpuntt/ntt_merge/ntt.cuh -- Merge-NTT launch API for MI355X (gfx950).
//
// Drop-in for the host API of reference src/include/gpuntt/ntt_merge/ntt.cuh:
//   config structs               :31-51   (same members, same order -> designated
//                                          initialisers in caller code keep working)
//   GPU_NTT / GPU_INTT / *_Inplace, single modulus   :315-340
//   GPU_NTT / GPU_INTT / *_Inplace, RNS              :395-421
// The reference's __global__ kernel declarations and per-logN KernelConfig tables are not
// part of this header: the HIP kernels behind these entry points are private to the
// library (gpu-ntt_amd/csrc/merge_ntt.hip) and planned at run time.
//
// Semantics (identical to the reference):
//   * data: T[batch][N] row-major device memory, N = 2^n_power, 1 <= n_power <= 28
//   * GPU_NTT : natural-order in -> bit-reversed out;  GPU_INTT: bit-reversed in -> natural
//     out, scaled by cfg.mod_inverse (= N^-1 mod q; device array indexed by modulus in RNS)
//   * tables: bit-reversed powers of omega (X_N_minus, N/2 entries) or psi (X_N_plus, N
//     entries); RNS: table of modulus i starts at element i << n_power; polynomial p uses
//     modulus p % mod_count
//   * cfg.ntt_type and cfg.zero_padding are ignored (direction = function name)
//   * signed instantiations: GPU_NTT<Data64s> takes inputs in (-q, q); GPU_INTT<Data64s>
//     returns centred residues
//   * asynchronous on cfg.stream; in == out allowed.  The fast kernels read twiddles that every call
//     re-derives from the caller's table into a library-owned scratch buffer: one chain per (device, stream),
//     and one per capture while a stream is being captured into a hipGraph.  First use / a larger ring
//     allocates with hipMalloc; NOTHING is freed, reused for another chain or synchronised on before
//     GPU_NTT_ReleaseWorkspaces() -- except that the buffer of a captured call belongs to its graph and is
//     pooled for the next capture once the graph and its executables have been destroyed -- so a graph
//     captured from these calls can be replayed at any time on any stream.  A host thread holds the chain's lock from the preparation launch to its last kernel launch,
//     so threads sharing a stream are serialised there and stream order keeps their launches apart.
//     Callers that want zero allocation / preparation per call use NTTPlan<T> below (caller-owned
//     workspace, tables prepared once).
//   * throws std::invalid_argument("Invalid n_power range!") / ("Invalid ntt_layout!"),
//     HipException (alias CudaException) on a failed launch
#pragma once

#include "gpuntt/ntt_merge/ntt_cpu.cuh"

typedef std::uint32_t location_t;

namespace gpuntt
{
    template <typename T> struct ntt_configuration
    {
        int n_power;
        type ntt_type;
        NTTLayout ntt_layout;
        ReductionPolynomial reduction_poly;
        bool zero_padding;
        Ninverse<T> mod_inverse;
        stream_t stream;
    };

    template <typename T> struct ntt_rns_configuration
    {
        int n_power;
        type ntt_type;
        NTTLayout ntt_layout;
        ReductionPolynomial reduction_poly;
        bool zero_padding;
        Ninverse<T>* mod_inverse;
        stream_t stream;
    };

    // ---- public device butterflies (reference ntt.cuh:69-92): for caller kernels built on OPERATOR_GPU<T> ----
    // Cooley-Tukey: U' = U + V * root, V' = U - V * root; Gentleman-Sande: U' = U + V, V' = (U - V) * root; all mod q,
    // canonical in, canonical out.  (The library's own kernels use lazy-range butterflies with prepared twiddles.)
    template <typename T>
    __device__ __forceinline__ void CooleyTukeyUnit(T& U, T& V, const Root<T>& root, const Modulus<T>& modulus)
    {
        const T u_ = U;
        const T v_ = OPERATOR_GPU<T>::mult(V, root, modulus);
        U = OPERATOR_GPU<T>::add(u_, v_, modulus);
        V = OPERATOR_GPU<T>::sub(u_, v_, modulus);
    }

    template <typename T>
    __device__ __forceinline__ void GentlemanSandeUnit(T& U, T& V, const Root<T>& root, const Modulus<T>& modulus)
    {
        const T u_ = U;
        const T v_ = V;
        U = OPERATOR_GPU<T>::add(u_, v_, modulus);
        V = OPERATOR_GPU<T>::mult(OPERATOR_GPU<T>::sub(u_, v_, modulus), root, modulus);
    }

    // ---- single modulus (passed by value from the host) ---------------------------------
    template <typename T>
    __host__ void GPU_NTT(T* device_in, typename std::make_unsigned<T>::type* device_out,
                          Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                          Modulus<typename std::make_unsigned<T>::type> modulus,
                          ntt_configuration<typename std::make_unsigned<T>::type> cfg,
                          int batch_size);

    template <typename T>
    __host__ void GPU_INTT(typename std::make_unsigned<T>::type* device_in, T* device_out,
                           Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                           Modulus<typename std::make_unsigned<T>::type> modulus,
                           ntt_configuration<typename std::make_unsigned<T>::type> cfg,
                           int batch_size);

    template <typename T>
    __host__ void GPU_NTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                  Modulus<T> modulus, ntt_configuration<T> cfg, int batch_size);

    template <typename T>
    __host__ void GPU_INTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                   Modulus<T> modulus, ntt_configuration<T> cfg, int batch_size);

    // ---- RNS: device arrays of moduli / tables / n^-1, polynomial p -> modulus p % mod_count
    template <typename T>
    __host__ void GPU_NTT(T* device_in, typename std::make_unsigned<T>::type* device_out,
                          Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                          Modulus<typename std::make_unsigned<T>::type>* modulus,
                          ntt_rns_configuration<typename std::make_unsigned<T>::type> cfg,
                          int batch_size, int mod_count);

    template <typename T>
    __host__ void GPU_INTT(typename std::make_unsigned<T>::type* device_in, T* device_out,
                           Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                           Modulus<typename std::make_unsigned<T>::type>* modulus,
                           ntt_rns_configuration<typename std::make_unsigned<T>::type> cfg,
                           int batch_size, int mod_count);

    template <typename T>
    __host__ void GPU_NTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                  Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                  int batch_size, int mod_count);

    template <typename T>
    __host__ void GPU_INTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                   Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                   int batch_size, int mod_count);

    // ---- RNS with an indirection table (reference ntt.cuh:495-603, hosts ntt.cu:3600-3776,
    //      4281-4459).  n_power in [10, 28]; cfg.ntt_type selects FORWARD / INVERSE; `order` is a
    //      device array.
    //  Modulus_Ordered: polynomial p uses prime order[p % mod_count] -- its modulus, its table
    //                   slot (prime << n_power) and, for INVERSE, cfg.mod_inverse[prime];
    //  Poly_Ordered   : polynomial p is the one stored in slot order[p] of device_in / device_out
    //                   and uses modulus p % mod_count.
    template <typename T>
    __host__ void GPU_NTT_Modulus_Ordered(T* device_in, T* device_out, Root<T>* root_of_unity_table,
                                          Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                          int batch_size, int mod_count, int* order);
    template <typename T>
    __host__ void GPU_NTT_Modulus_Ordered_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                                  Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                                  int batch_size, int mod_count, int* order);
    template <typename T>
    __host__ void GPU_NTT_Poly_Ordered(T* device_in, T* device_out, Root<T>* root_of_unity_table,
                                       Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                       int batch_size, int mod_count, int* order);
    template <typename T>
    __host__ void GPU_NTT_Poly_Ordered_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                               Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                               int batch_size, int mod_count, int* order);

    // ---- extension (not in the reference GPU API) ------------------------------------------
    // Polynomial product in the ring Z_q[X]/(X^N -+ 1), the composition the reference's CPU
    // example checks against schoolbook multiplication (NTTCPU<T>::mult between ntt() and
    // intt(), example/ntt_merge/test_cpu_merge_ntt.cu:69-101):
    //     device_out = INTT( NTT(device_a) (.) NTT(device_b) )
    // forward_table / inverse_table as for GPU_NTT / GPU_INTT (omega tables for X_N_minus = cyclic
    // product, psi tables for X_N_plus = negacyclic product); cfg.mod_inverse = N^-1 mod q.
    // device_a and device_b are overwritten with their transforms; device_out may alias either.
    template <typename T>
    __host__ void GPU_PolyMul(T* device_a, T* device_b, T* device_out, Root<T>* forward_table,
                              Root<T>* inverse_table, Modulus<T> modulus, ntt_configuration<T> cfg,
                              int batch_size);
    // RNS: polynomial p uses modulus p % mod_count, tables at (p % mod_count) << n_power
    template <typename T>
    __host__ void GPU_PolyMul(T* device_a, T* device_b, T* device_out, Root<T>* forward_table,
                              Root<T>* inverse_table, Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                              int batch_size, int mod_count);
    // device_out[i] = device_a[i] * device_b[i] mod q (per-polynomial modulus in the RNS form); the
    // pointwise step on its own, all three pointers may alias
    template <typename T>
    __host__ void GPU_PointwiseMul(T* device_a, T* device_b, T* device_out, Modulus<T> modulus, int n_power,
                                   int batch_size, stream_t stream);
    template <typename T>
    __host__ void GPU_PointwiseMul(T* device_a, T* device_b, T* device_out, Modulus<T>* modulus, int n_power,
                                   int batch_size, int mod_count, stream_t stream);

    // ---- extension: prepared transform (no counterpart in the reference) -------------------
    // What GPU_NTT / GPU_INTT re-derive on every call -- Shoup pairs of the twiddle table in the
    // kernels' stage layout, n^-1 pairs, the choice between the fast (lazy-residue) and the generic
    // kernels -- is done ONCE here, on `stream`, into `workspace_device` (workspace_bytes() bytes,
    // caller-owned; nullptr = the plan allocates and owns it).  execute() then launches the transform
    // kernels and nothing else: no allocation, no synchronisation, no preparation launch, no go-flag
    // for RNS stacks (the moduli are host values here, so the kernel family is known), and it can be
    // captured into a hipGraph at once.  The constructor waits for `stream` before it returns
    // (one host wait per plan): execute() may run on any stream with no dependency on the construction stream.
    //   table_device       the caller's table exactly as for GPU_NTT / GPU_INTT (slot i at i << n_power);
    //                      it is only read during construction (fast path) -- the generic path keeps
    //                      reading it at execute()
    //   moduli_host        mod_count moduli; polynomial p uses modulus p % mod_count
    //   mod_inverse_host   n^-1 per modulus (INVERSE only)
    //   batch_hint         the batch size the tile plan is chosen for (any batch size runs correctly)
    // PerPolynomial layout only.  execute(): in == out allowed; io_signed selects the Data64s / Data32s
    // behaviour (signed input for FORWARD, centred output for INVERSE).
    template <typename T> class NTTPlan
    {
      public:
        static size_t workspace_bytes(int n_power, int mod_count);
        NTTPlan(const Root<T>* table_device, const Modulus<T>* moduli_host, int mod_count, int n_power,
                ReductionPolynomial reduction_poly, type ntt_type, const Ninverse<T>* mod_inverse_host,
                int batch_hint, stream_t stream, void* workspace_device = nullptr);
        ~NTTPlan();
        NTTPlan(const NTTPlan&) = delete;
        NTTPlan& operator=(const NTTPlan&) = delete;
        void execute(const void* device_in, void* device_out, int batch_size, stream_t stream,
                     bool io_signed = false) const;
        bool fast_path() const; // false: moduli without lazy headroom / ring outside the prepared range
      private:
        struct Impl;
        Impl* p_;
    };

    // Frees the library-owned twiddle scratch of the drop-in entry points (synchronises the device).  The drop-in calls
    // never free or reuse a scratch buffer on their own -- a hipGraph captured from them may be replayed at any time --
    // so call this only when no such graph will be replayed again and no call is in flight.
    void GPU_NTT_ReleaseWorkspaces();

    // Process-wide behaviour options (extension).  The library reads no environment variable.  name = value:
    //   path           default | generic | fast   kernel family for every call: the size heuristic, the element-by-element
    //                            Barrett kernels, or the fast (lazy-residue) kernels wherever they can take the call
    //   check_4step_tables 0 | 1   4-step entry points / FourStepPlan: verify all three caller tables on the device and run
    //                            the element-by-element kernels when they are not the tables of one root
    //                            (ntt_4step/ntt_4step.cuh, "TABLES"); default 1
    //   rns_predict    0 | 1     drop-in RNS calls enqueue only the lazy kernel family predicted for their stack of moduli (same
    //                            device, moduli pointer, mod_count, direction); a stack that family cannot serve is transformed
    //                            by the preparation kernel itself -- or, on rings from 2^17, by the generic kernels enqueued
    //                            behind a call nothing is known about yet (default 1); 0: every family behind the go-flag
    // An API call reads the options ONCE, when it starts: setting one from another thread never changes a call in flight.
    // Returns false for an unknown name or a value outside the sets above (the whole string must parse; nothing is silently
    // mapped to a default).  Plans keep the choice made when they were created.  (The hooks this repository's tests use --
    // forced failure paths, retired A/B switches -- are not options: gpu-ntt_amd/csrc/test_hooks.h.)
    bool GPU_NTT_SetOption(const char* name, const char* value);

} // namespace gpuntt
