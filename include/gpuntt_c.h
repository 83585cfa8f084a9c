/*
 * gpuntt_c.h -- C ABI of the MI355X-native NTT library (libgpuntt.so).
 *
 * The reference (Alisah-Ozcan/GPU-NTT) exposes this path only as C++ templates resolved
 * through explicit instantiations; it has no FFI.  This header is the flat
 * extern "C" boundary an FFI (ctypes / cgo / JNI ...) binds instead -- plain pointers and
 * sizes, no C++ or torch types -- one entry point per reference template overload:
 *
 *   gpuntt_ntt_{u32,u64}        GPU_NTT<T>  single modulus   src/include/gpuntt/ntt_merge/ntt.cuh:315-321
 *   gpuntt_intt_{u32,u64}       GPU_INTT<T> single modulus   ntt.cuh:323-329
 *   gpuntt_ntt_rns_{u32,u64}    GPU_NTT<T>  RNS              ntt.cuh:395-401
 *   gpuntt_intt_rns_{u32,u64}   GPU_INTT<T> RNS              ntt.cuh:403-409
 *       (in == out gives the *_Inplace overloads, ntt.cuh:331-340,411-421;
 *        input_signed / output_signed select the Data32s/Data64s instantiations,
 *        src/lib/ntt_merge/ntt.cu:4948-5082)
 *   gpuntt_ntt_modulus_ordered_{u32,u64} / gpuntt_ntt_poly_ordered_{u32,u64}
 *                               GPU_NTT_Modulus_Ordered / GPU_NTT_Poly_Ordered   ntt.cuh:495-603
 *   gpuntt_4step_{u32,u64}      GPU_4STEP_NTT<T> single      src/include/gpuntt/ntt_4step/ntt_4step.cuh:278-283
 *   gpuntt_4step_rns_{u32,u64}  GPU_4STEP_NTT<T> RNS         ntt_4step.cuh:301-307
 *   gpuntt_transpose_{u32,u64}  GPU_Transpose<T>             ntt_4step.cuh:46-49
 *   gpuntt_modulus_*, gpuntt_merge_params_*, gpuntt_4step_params_*
 *                               Modulus<T>, NTTParameters<T>, NTTParameters4Step<T>
 *                               src/include/gpuntt/common/modular_arith.cuh:28-57,
 *                               src/include/gpuntt/common/nttparameters.cuh:56-170
 *
 *   gpuntt_plan_*               extension NTTPlan<T> (include/gpuntt/ntt_merge/ntt.cuh): tables prepared once,
 *                               caller-owned workspace; execute = transform kernels only
 *   gpuntt_operator_gpu_*       diagnostic: the public device class OPERATOR_GPU<T>
 *                               (src/include/gpuntt/common/modular_arith.cuh:174-454) applied elementwise
 *
 * All data/table/modulus-array pointers are DEVICE pointers unless the name ends in _host.
 * Calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 * stream).  The transform entry points keep a library-owned twiddle scratch per (device, stream)
 * -- and per capture while a stream is being captured: first use / growth allocates, nothing is
 * freed, reused or synchronised on before gpuntt_release_workspaces(), so hipGraphs captured from
 * these calls stay replayable ("Scratch lifetime and hipGraphs").  gpuntt_plan_execute_* allocates nothing, never
 * synchronises and launches no preparation kernel.
 *
 * Return value: GPUNTT_OK, or a negative code with the text available from
 * gpuntt_last_error() (thread-local).  GPUNTT_ERR_INVALID_ARGUMENT corresponds to the
 * std::invalid_argument the C++ API throws (reference ntt.cu:2088-2091, 2252-2254),
 * GPUNTT_ERR_HIP to HipException/CudaException (reference common.cuh:42-50).
 */
#ifndef GPUNTT_C_H
#define GPUNTT_C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define GPUNTT_OK 0
#define GPUNTT_ERR_INVALID_ARGUMENT (-1)
#define GPUNTT_ERR_HIP (-2)
#define GPUNTT_ERR_UNKNOWN (-3)

    /* enum values of the reference (nttparameters.cuh:19-36) */
#define GPUNTT_FORWARD 0
#define GPUNTT_INVERSE 1
#define GPUNTT_PER_POLYNOMIAL 0
#define GPUNTT_PER_COEFFICIENT 1
#define GPUNTT_X_N_PLUS 0  /* negacyclic X^N + 1 */
#define GPUNTT_X_N_MINUS 1 /* cyclic     X^N - 1 */

    /* layout-identical to Modulus<Data32> / Modulus<Data64> */
    typedef struct { uint32_t value, bit, mu; } gpuntt_modulus32;
    typedef struct { uint64_t value, bit, mu; } gpuntt_modulus64;

    const char* gpuntt_last_error(void);
    int gpuntt_version(void);

    /* ---- Modulus<T>(q) ------------------------------------------------------------ */
    int gpuntt_modulus_u32(uint32_t q, gpuntt_modulus32* out_host);
    int gpuntt_modulus_u64(uint64_t q, gpuntt_modulus64* out_host);

    /* ---- Merge NTT, single modulus -------------------------------------------------- */
    int gpuntt_ntt_u32(const void* in, uint32_t* out, const uint32_t* roots,
                       gpuntt_modulus32 modulus, int n_power, int ntt_layout, int reduction_poly,
                       int input_signed, void* stream, int batch_size);
    int gpuntt_ntt_u64(const void* in, uint64_t* out, const uint64_t* roots,
                       gpuntt_modulus64 modulus, int n_power, int ntt_layout, int reduction_poly,
                       int input_signed, void* stream, int batch_size);
    int gpuntt_intt_u32(const uint32_t* in, void* out, const uint32_t* inverse_roots,
                        gpuntt_modulus32 modulus, int n_power, int ntt_layout, int reduction_poly,
                        uint32_t mod_inverse, int output_signed, void* stream, int batch_size);
    int gpuntt_intt_u64(const uint64_t* in, void* out, const uint64_t* inverse_roots,
                        gpuntt_modulus64 modulus, int n_power, int ntt_layout, int reduction_poly,
                        uint64_t mod_inverse, int output_signed, void* stream, int batch_size);

    /* ---- Merge NTT, RNS: polynomial p uses modulus p % mod_count, table at i << n_power -- */
    int gpuntt_ntt_rns_u32(const void* in, uint32_t* out, const uint32_t* roots,
                           const gpuntt_modulus32* modulus, int n_power, int ntt_layout,
                           int reduction_poly, int input_signed, void* stream, int batch_size,
                           int mod_count);
    int gpuntt_ntt_rns_u64(const void* in, uint64_t* out, const uint64_t* roots,
                           const gpuntt_modulus64* modulus, int n_power, int ntt_layout,
                           int reduction_poly, int input_signed, void* stream, int batch_size,
                           int mod_count);
    int gpuntt_intt_rns_u32(const uint32_t* in, void* out, const uint32_t* inverse_roots,
                            const gpuntt_modulus32* modulus, int n_power, int ntt_layout,
                            int reduction_poly, const uint32_t* mod_inverse, int output_signed,
                            void* stream, int batch_size, int mod_count);
    int gpuntt_intt_rns_u64(const uint64_t* in, void* out, const uint64_t* inverse_roots,
                            const gpuntt_modulus64* modulus, int n_power, int ntt_layout,
                            int reduction_poly, const uint64_t* mod_inverse, int output_signed,
                            void* stream, int batch_size, int mod_count);

    /* ---- ordered RNS entry points (reference ntt.cuh:495-603): ntt_type selects the direction,
     *      `order` is a device int array; mod_inverse (device, indexed by prime) is used for
     *      GPUNTT_INVERSE only.  n_power in [10, 28]. */
    int gpuntt_ntt_modulus_ordered_u32(const uint32_t* in, uint32_t* out, const uint32_t* roots,
                                       const gpuntt_modulus32* modulus, int n_power, int ntt_type,
                                       int reduction_poly, const uint32_t* mod_inverse, void* stream,
                                       int batch_size, int mod_count, const int* order);
    int gpuntt_ntt_modulus_ordered_u64(const uint64_t* in, uint64_t* out, const uint64_t* roots,
                                       const gpuntt_modulus64* modulus, int n_power, int ntt_type,
                                       int reduction_poly, const uint64_t* mod_inverse, void* stream,
                                       int batch_size, int mod_count, const int* order);
    int gpuntt_ntt_poly_ordered_u32(const uint32_t* in, uint32_t* out, const uint32_t* roots,
                                    const gpuntt_modulus32* modulus, int n_power, int ntt_type,
                                    int reduction_poly, const uint32_t* mod_inverse, void* stream,
                                    int batch_size, int mod_count, const int* order);
    int gpuntt_ntt_poly_ordered_u64(const uint64_t* in, uint64_t* out, const uint64_t* roots,
                                    const gpuntt_modulus64* modulus, int n_power, int ntt_type,
                                    int reduction_poly, const uint64_t* mod_inverse, void* stream,
                                    int batch_size, int mod_count, const int* order);

    /* ---- extension: polynomial product out = INTT(NTT(a) (.) NTT(b)) in Z_q[X]/(X^N -+ 1), the
     * composition of the reference's CPU example (test_cpu_merge_ntt.cu:69-101); a and b are
     * overwritten with their transforms, out may alias either; mod_inverse = N^-1 (RNS: device array) */
    int gpuntt_polymul_u32(uint32_t* a, uint32_t* b, uint32_t* out, const uint32_t* forward_table,
                           const uint32_t* inverse_table, gpuntt_modulus32 modulus, int n_power,
                           int reduction_poly, uint32_t mod_inverse, void* stream, int batch_size);
    int gpuntt_polymul_u64(uint64_t* a, uint64_t* b, uint64_t* out, const uint64_t* forward_table,
                           const uint64_t* inverse_table, gpuntt_modulus64 modulus, int n_power,
                           int reduction_poly, uint64_t mod_inverse, void* stream, int batch_size);
    int gpuntt_polymul_rns_u32(uint32_t* a, uint32_t* b, uint32_t* out, const uint32_t* forward_table,
                               const uint32_t* inverse_table, const gpuntt_modulus32* modulus,
                               int n_power, int reduction_poly, const uint32_t* mod_inverse,
                               void* stream, int batch_size, int mod_count);
    int gpuntt_polymul_rns_u64(uint64_t* a, uint64_t* b, uint64_t* out, const uint64_t* forward_table,
                               const uint64_t* inverse_table, const gpuntt_modulus64* modulus,
                               int n_power, int reduction_poly, const uint64_t* mod_inverse,
                               void* stream, int batch_size, int mod_count);

    /* ---- 4-Step NTT (cyclic, 12 <= n_power <= 24, in != out) ------------------------ */
    int gpuntt_4step_u32(const uint32_t* in, uint32_t* out, const uint32_t* n1_table,
                         const uint32_t* n2_table, const uint32_t* w_table,
                         gpuntt_modulus32 modulus, int n_power, int ntt_type, uint32_t mod_inverse,
                         void* stream, int batch_size);
    int gpuntt_4step_u64(const uint64_t* in, uint64_t* out, const uint64_t* n1_table,
                         const uint64_t* n2_table, const uint64_t* w_table,
                         gpuntt_modulus64 modulus, int n_power, int ntt_type, uint64_t mod_inverse,
                         void* stream, int batch_size);
    /* extension: the reference examples' natural-order pipeline (GPU_Transpose -> GPU_4STEP_NTT ->
     * GPU_Transpose, test_4step_ntt.cu:147-178 / test_4step_intt.cu:81-179) as one call =
     * NTT_4STEP_CPU::ntt / ::intt; `in_scratch` is overwritten, in_scratch != out */
    int gpuntt_4step_natural_u32(uint32_t* in_scratch, uint32_t* out, const uint32_t* n1_table,
                                 const uint32_t* n2_table, const uint32_t* w_table,
                                 gpuntt_modulus32 modulus, int n_power, int ntt_type,
                                 uint32_t mod_inverse, void* stream, int batch_size);
    int gpuntt_4step_natural_u64(uint64_t* in_scratch, uint64_t* out, const uint64_t* n1_table,
                                 const uint64_t* n2_table, const uint64_t* w_table,
                                 gpuntt_modulus64 modulus, int n_power, int ntt_type,
                                 uint64_t mod_inverse, void* stream, int batch_size);
    int gpuntt_4step_rns_u32(const uint32_t* in, uint32_t* out, const uint32_t* n1_table,
                             const uint32_t* n2_table, const uint32_t* w_table,
                             const gpuntt_modulus32* modulus, int n_power, int ntt_type,
                             const uint32_t* mod_inverse, void* stream, int batch_size,
                             int mod_count);
    int gpuntt_4step_rns_u64(const uint64_t* in, uint64_t* out, const uint64_t* n1_table,
                             const uint64_t* n2_table, const uint64_t* w_table,
                             const gpuntt_modulus64* modulus, int n_power, int ntt_type,
                             const uint64_t* mod_inverse, void* stream, int batch_size,
                             int mod_count);
    int gpuntt_transpose_u32(const uint32_t* in, uint32_t* out, int row, int col, int n_power,
                             int batch_size);
    int gpuntt_transpose_u64(const uint64_t* in, uint64_t* out, int row, int col, int n_power,
                             int batch_size);

    /* ---- host-side parameter / table generation (no GPU needed) --------------------
     * factors_host: {q, omega, psi} or NULL for the built-in pool.
     * info_host[8] = {q, bit, mu, omega, psi, n_inv, root_of_unity_size, n}.
     * tables are written in DEVICE order (bit-reversed), root_of_unity_size entries each. */
    int gpuntt_merge_params_u32(int logn, int reduction_poly, const uint32_t* factors_host,
                                uint64_t* info_host, uint32_t* forward_table_host,
                                uint32_t* inverse_table_host);
    int gpuntt_merge_params_u64(int logn, int reduction_poly, const uint64_t* factors_host,
                                uint64_t* info_host, uint64_t* forward_table_host,
                                uint64_t* inverse_table_host);
    /* info_host[9] = {q, bit, mu, omega, psi, n_inv, n1, n2, n}; inverse != 0 selects the
     * inverse tables; n1/n2 tables in DEVICE order (n1/2, n2/2 entries), W natural (n). */
    int gpuntt_4step_params_u32(int logn, int inverse, uint64_t* info_host,
                                uint32_t* n1_table_host, uint32_t* n2_table_host,
                                uint32_t* w_table_host);
    int gpuntt_4step_params_u64(int logn, int inverse, uint64_t* info_host,
                                uint64_t* n1_table_host, uint64_t* n2_table_host,
                                uint64_t* w_table_host);

    /* ---- extension: prepared transforms (NTTPlan<T>) ----------------------------------------
     * moduli_host[mod_count], mod_inverse_host[mod_count] (GPUNTT_INVERSE only) are HOST arrays;
     * workspace_device: gpuntt_plan_workspace_bytes_*() bytes of device memory owned by the caller, or
     * NULL (the plan allocates).  Construction runs on `stream`; execute: in == out allowed, io_signed
     * = signed input (forward) / centred output (inverse); PerPolynomial layout. */
    typedef struct gpuntt_plan gpuntt_plan;
    int gpuntt_plan_workspace_bytes_u32(int n_power, int mod_count, uint64_t* bytes_host);
    int gpuntt_plan_workspace_bytes_u64(int n_power, int mod_count, uint64_t* bytes_host);
    int gpuntt_plan_create_u32(gpuntt_plan** plan_host, const uint32_t* table, const gpuntt_modulus32* moduli_host,
                               int mod_count, int n_power, int reduction_poly, int ntt_type,
                               const uint32_t* mod_inverse_host, int batch_hint, void* workspace_device,
                               void* stream);
    int gpuntt_plan_create_u64(gpuntt_plan** plan_host, const uint64_t* table, const gpuntt_modulus64* moduli_host,
                               int mod_count, int n_power, int reduction_poly, int ntt_type,
                               const uint64_t* mod_inverse_host, int batch_hint, void* workspace_device,
                               void* stream);
    int gpuntt_plan_execute_u32(const gpuntt_plan* plan, const void* in, void* out, int batch_size, int io_signed,
                                void* stream);
    int gpuntt_plan_execute_u64(const gpuntt_plan* plan, const void* in, void* out, int batch_size, int io_signed,
                                void* stream);
    int gpuntt_plan_fast_path_u32(const gpuntt_plan* plan); /* 1 / 0, negative on error */
    int gpuntt_plan_fast_path_u64(const gpuntt_plan* plan);
    int gpuntt_plan_destroy_u32(gpuntt_plan* plan);
    int gpuntt_plan_destroy_u64(gpuntt_plan* plan);

    /* ---- extension: prepared 4-step transforms (FourStepPlan<T>, include/gpuntt/ntt_4step/ntt_4step.cuh) ----
     * The Shoup pairs of the n1 / n2 / W tables are derived once, at creation, into workspace_device
     * (gpuntt_4step_plan_workspace_bytes_*() bytes owned by the caller) or a buffer the plan allocates (NULL).
     * natural_order 0: execute == gpuntt_4step_* (n2 x n1 in, n1 x n2 out); 1: == gpuntt_4step_natural_* (`in` is
     * scratch).  in != out.  Creation runs on `stream`; execute allocates nothing and never synchronises. */
    typedef struct gpuntt_4step_plan gpuntt_4step_plan;
    int gpuntt_4step_plan_workspace_bytes_u32(int n_power, uint64_t* bytes_host);
    int gpuntt_4step_plan_workspace_bytes_u64(int n_power, uint64_t* bytes_host);
    int gpuntt_4step_plan_create_u32(gpuntt_4step_plan** plan_host, const uint32_t* n1_table, const uint32_t* n2_table,
                                     const uint32_t* w_table, gpuntt_modulus32 modulus, int n_power, int ntt_type,
                                     uint32_t mod_inverse, int natural_order, int batch_hint, void* workspace_device,
                                     void* stream);
    int gpuntt_4step_plan_create_u64(gpuntt_4step_plan** plan_host, const uint64_t* n1_table, const uint64_t* n2_table,
                                     const uint64_t* w_table, gpuntt_modulus64 modulus, int n_power, int ntt_type,
                                     uint64_t mod_inverse, int natural_order, int batch_hint, void* workspace_device,
                                     void* stream);
    int gpuntt_4step_plan_execute_u32(const gpuntt_4step_plan* plan, uint32_t* in, uint32_t* out, int batch_size,
                                      void* stream);
    int gpuntt_4step_plan_execute_u64(const gpuntt_4step_plan* plan, uint64_t* in, uint64_t* out, int batch_size,
                                      void* stream);
    int gpuntt_4step_plan_fast_path_u32(const gpuntt_4step_plan* plan); /* 1 / 0, negative on error */
    int gpuntt_4step_plan_fast_path_u64(const gpuntt_4step_plan* plan);
    int gpuntt_4step_plan_destroy_u32(gpuntt_4step_plan* plan);
    int gpuntt_4step_plan_destroy_u64(gpuntt_4step_plan* plan);

    /* ---- extension: root-of-unity tables built on the device (include/gpuntt/common/parameter_sets.hpp) ----
     * power table: out[k] = base^(bit_reversed ? bitreverse(k, log_count) : k), k < 2^log_count (log_count <= 28) --
     * with bit_reversed = 1 the device-order table GPU_NTT / GPU_INTT / the 4-step n1, n2 slots take;
     * 4-step W:  GPUNTT_FORWARD W[i*n2+j] = root^(bitreverse(i, log n1) * j), GPUNTT_INVERSE root^(bitreverse(j, log n2) * i)
     * (pass the inverse root), N = 2^n_power entries, 12 <= n_power <= 24.  Replaces the host loops of the
     * reference's src/lib/common/nttparameters.cu:356-444 and the upload. */
    int gpuntt_generate_power_table_u32(uint32_t* out, uint32_t base, gpuntt_modulus32 modulus, int log_count,
                                        int bit_reversed, void* stream);
    int gpuntt_generate_power_table_u64(uint64_t* out, uint64_t base, gpuntt_modulus64 modulus, int log_count,
                                        int bit_reversed, void* stream);
    int gpuntt_generate_4step_w_u32(uint32_t* out, uint32_t root, gpuntt_modulus32 modulus, int n_power, int ntt_type,
                                    void* stream);
    int gpuntt_generate_4step_w_u64(uint64_t* out, uint64_t root, gpuntt_modulus64 modulus, int n_power, int ntt_type,
                                    void* stream);

    /* frees the library-owned scratch buffers of the drop-in entry points (synchronises the device); only when no
     * hipGraph captured from drop-in calls will be replayed again and no call is in flight */
    int gpuntt_release_workspaces(void);

    /* process-wide tuning / test option (GPU_NTT_SetOption, include/gpuntt/ntt_merge/ntt.cuh lists the names);
     * the library itself reads no environment variable */
    int gpuntt_set_option(const char* name, const char* value);

    /* ---- diagnostic: OPERATOR_GPU<T> elementwise on device arrays -----------------------------
     * op: 0 add, 1 sub, 2 mult, 3 reduce (unsigned a), 4 reduce (a read as signed), 5 centered_reduction;
     * b is ignored by ops 3..5 */
    int gpuntt_operator_gpu_u32(int op, const uint32_t* a, const uint32_t* b, uint32_t* out,
                                gpuntt_modulus32 modulus, uint64_t count, void* stream);
    int gpuntt_operator_gpu_u64(int op, const uint64_t* a, const uint64_t* b, uint64_t* out,
                                gpuntt_modulus64 modulus, uint64_t count, void* stream);

    /* diagnostic: the normalised reciprocal floor(2^(W-1+b) / q) (b = bit length of q; 0 for q < 3 and powers of two) the
     * preparation kernels derive for every device-side modulus -- exactness is checked against integers in the tests */
    int gpuntt_debug_recip_norm_u32(const uint32_t* q, uint32_t* out, uint64_t count, void* stream);
    int gpuntt_debug_recip_norm_u64(const uint64_t* q, uint64_t* out, uint64_t count, void* stream);

    /* diagnostic: the public device butterflies CooleyTukeyUnit (gentleman_sande = 0) / GentlemanSandeUnit (1)
     * (reference src/include/gpuntt/ntt_merge/ntt.cuh:69-92) applied to the pairs (u[i], v[i]) with roots[i], in place */
    int gpuntt_butterfly_unit_u32(int gentleman_sande, uint32_t* u, uint32_t* v, const uint32_t* roots,
                                  gpuntt_modulus32 modulus, uint64_t count, void* stream);
    int gpuntt_butterfly_unit_u64(int gentleman_sande, uint64_t* u, uint64_t* v, const uint64_t* roots,
                                  gpuntt_modulus64 modulus, uint64_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPUNTT_C_H */
