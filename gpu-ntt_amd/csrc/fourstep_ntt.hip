// fourstep_ntt.hip -- 4-Step NTT entry points (gpuntt::GPU_4STEP_NTT, GPU_Transpose).
//
// Replaces reference src/lib/ntt_4step/ntt_4step.cu:36-66 (transpose), :68-2291 (kernels),
// :2293-3229 (hosts), :3292-3302 / :3606-3634 (exported instantiations).
//
// MI355X plan for N = n1 x n2 (shapes of NTTParameters4Step, nttparameters.cu:305-354): the 4-step transform IS the
// Merge transform of the ring with a transposition on the natural-order side, so the fast path runs the
// ring's own Merge plan from a Merge table rebuilt on the device out of the caller's n1 / W tables:
//   forward  in = x^T (n2 x n1): the first strided pass gathers the transposed input (fourstep_first_lazy), the rest is
//            the Merge plan as it stands; out = the bit-reversed Merge spectrum = the reference's n1 x n2 output;
//   inverse  in = that spectrum: the first contiguous pass (12 Gentleman-Sande stages) stores transposed
//            (fourstep_inv_first_lazy), the remaining stages are strided / partial row passes inside the n2-long rows;
//   rings that fit one tile (2^12 .. 2^14) take ONE launch with the transposition in LDS (fourstep_small_lazy).
// => 1 sweep to 2^14, 2 sweeps to 2^20 (2^22 forward), 3 sweeps to 2^24; no W stream, no W product.  The generic
// (Barrett) fall-back keeps the reference's two-phase form: phase 1 with the fused W multiply (merge_pass<..., FST>),
// phase 2 = the n2-point row transforms through the shared planner (launch.hpp).
//
// Extension GPU_4STEP_NTT_NaturalOrder: the reference examples' GPU_Transpose -> GPU_4STEP_NTT ->
// GPU_Transpose pipeline as one call in three sweeps (fourstep_natural_forward_lazy /
// fourstep_natural_inverse_lazy below) instead of five.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gpuntt/ntt_4step/ntt_4step.cuh"
#include "launch_impl.hpp"
#include "lazy_launch.hpp"

namespace gpuntt
{
    namespace kern
    {
        // per polynomial: (row x col) row-major -> (col x row); 32x32 tiles through padded LDS
        template <typename T>
        __global__ __launch_bounds__(256) void transpose_batch(const T* __restrict__ in,
                                                              T* __restrict__ out, int row, int col,
                                                              unsigned long long poly_elems,
                                                              const unsigned* __restrict__ skip_flag)
        {
            __shared__ T tile[32][33];
            // behind the fast natural-order kernels: "return unless the table check handed the call to the generic kernels"
            if (skip_flag != nullptr && *skip_flag != GO_GENERIC)
                return;
            const unsigned long long base = static_cast<unsigned long long>(blockIdx.z) * poly_elems;
            const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
            const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
            for (int k = 0; k < 32; k += 8)
            {
                const int r = r0 + ty + k, c = c0 + tx;
                if (r < row && c < col)
                    tile[ty + k][tx] = in[base + static_cast<unsigned long long>(r) * col + c];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 32; k += 8)
            {
                const int c = c0 + ty + k, r = r0 + tx;
                if (r < row && c < col)
                    out[base + static_cast<unsigned long long>(c) * row + r] = tile[tx][ty + k];
            }
        }
    } // namespace kern

    namespace
    {
        template <typename T>
        void transpose_on(T* in, T* out, int row, int col, int n_power, int batch_size, hipStream_t stream,
                          const unsigned* skip_flag = nullptr)
        {
            if (batch_size <= 0 || row <= 0 || col <= 0)
                return;
            // gridDim.z is limited to 65535: larger batches go out in slices
            for (int done = 0; done < batch_size; done += 65535)
            {
                const int part = (batch_size - done < 65535) ? (batch_size - done) : 65535;
                const dim3 grid((col + 31) / 32, (row + 31) / 32, part);
                const unsigned long long off = static_cast<unsigned long long>(done) << n_power;
                GPUNTT_LAUNCH((kern::transpose_batch<T>), grid, dim3(256), 0, stream, in + off, out + off, row,
                                   col, 1ull << n_power, skip_flag);
                GPUNTT_HIP_CHECK(hipGetLastError());
            }
        }
    } // namespace

    template <typename T>
    __host__ void GPU_Transpose(T* polynomial_in, T* polynomial_out, const int row, const int col,
                                const int n_power, const int batch_size)
    {
        transpose_on<T>(polynomial_in, polynomial_out, row, col, n_power, batch_size, 0);
    }

    namespace
    {
        // n1 x n2 shapes, reference src/lib/common/nttparameters.cu:305-354
        inline bool fourstep_shape(int n_power, int& log_n1, int& log_n2)
        {
            static const int l1[13] = {5, 5, 5, 6, 7, 5, 5, 5, 5, 6, 7, 7, 8};
            if (n_power < 12 || n_power > 24)
                return false;
            log_n1 = l1[n_power - 12];
            log_n2 = n_power - log_n1;
            return true;
        }

        template <typename T, bool INV>
        void fourstep_run(T* in, T* out, const T* n1_table, const T* n2_table, const T* w_table,
                          const Modulus<T>* mods, Modulus<T> mod, int mod_count, const T* ninv_arr,
                          T ninv, int n_power, int log_n1, int log_n2, int batch_size,
                          hipStream_t stream, const unsigned* skip_flag = nullptr, unsigned skip_value = 0u,
                          bool skip_phase1 = false)
        {
            kern::PassArgs<T> a{};
            a.skip_flag = skip_flag;
            a.skip_value = skip_value;
            a.in = in;
            a.out = out;
            a.roots = n1_table;
            a.mods = mods;
            a.mod = mod;
            a.ninv_arr = ninv_arr;
            a.ninv = ninv;
            a.w_table = w_table;
            a.total = static_cast<unsigned long long>(batch_size) << n_power;
            a.n = log_n1;
            a.poly_shift = n_power;
            a.root_shift = -1; // tables are shared by all moduli (reference SURVEY A.3)
            a.mod_count = mod_count;
            a.p_lo = 0;
            a.n2_log = log_n2;
            a.flags = 0;
            // behind the go-flag (one device-side modulus) this is a shadow launch: capped grid that
            // walks the tiles, like the Merge shadow launches (launch_impl.hpp)
            const unsigned long long tiles = a.total >> kern::TL;
            const unsigned grid = (skip_flag != nullptr && tiles > GPUNTT_SHADOW_GRID) ? static_cast<unsigned>(GPUNTT_SHADOW_GRID) : static_cast<unsigned>(tiles);
            if (!skip_phase1) // (skipped: the fast first kernel of the call does phase 1 itself when the veto fires)
                switch (log_n1)
                {
                    case 5:
                        host::launch_one<T, INV, true, 5, true>(a, grid, stream);
                        break;
                    case 6:
                        host::launch_one<T, INV, true, 6, true>(a, grid, stream);
                        break;
                    case 7:
                        host::launch_one<T, INV, true, 7, true>(a, grid, stream);
                        break;
                    default:
                        host::launch_one<T, INV, true, 8, true>(a, grid, stream);
                        break;
                }
            // phase 2: n2-point transforms on the batch*n1 rows of `out`, in place
            kern::PassArgs<T> b = a;
            b.in = out;
            b.roots = n2_table;
            b.w_table = nullptr;
            b.n = log_n2;
            host::run_transform<T, INV>(b, 0u, INV ? kern::F_SCALE : 0u, stream);
        }

        // FourStepPlan splits the fast paths below into their two halves: PLAN_PREPARE runs the table
        // preparation into the plan's buffer and returns, PLAN_EXECUTE skips it and launches the sweeps
        enum PlanMode
        {
            PLAN_NONE = 0,
            PLAN_PREPARE = 1,
            PLAN_EXECUTE = 2
        };
        template <typename T> struct PlanUse
        {
            PlanMode mode = PLAN_NONE;
            lazy::Tw<T>* ws = nullptr;
            int tile_log = 0; // forward reference-layout plans: tile of the ring's Merge plan the table was laid out for
            int small_tl = 0; // reference-layout plans of one-tile rings: tile of the one-launch path (0: two-phase path)
            int first_k = 0;  // forward reference-layout plans: stages of the first Merge pass (the one that reads the
                              // transposed input); tile_log is then the tile of the RING's Merge plan
            int inv_tile = 0; // inverse reference-layout plans: tile of the transposing first pass the table was permuted for
                              // (fixed at PLAN_PREPARE: the u64_big_tiles option may change before execute())
        };

        // bytes of the drop-in calls' scratch: pairs (n1 table | W | n2 table | n^-1), go-flag (16 B), normalisation constants
        template <typename T> inline size_t fourstep_ws_bytes(int log_n1, int log_n2)
        {
            const size_t pairs = (size_t(1) << log_n1) + (size_t(1) << (log_n1 + log_n2)) + (size_t(1) << log_n2) + 2;
            return sizeof(lazy::Tw<T>) * pairs + 16 + sizeof(lazy::NormConst);
        }
        // the veto word of a drop-in 4-step call on `stream` (lazy_launch.hpp: FourStepVeto); word == nullptr when the
        // fast path cannot run anyway (no scratch, option path = generic)
        template <typename T> inline host::FourStepVeto fourstep_veto(int log_n1, int log_n2, hipStream_t stream)
        {
            host::FourStepVeto v;
            if (host::forced_path() == 1 || host::lazy_workspace(stream, fourstep_ws_bytes<T>(log_n1, log_n2), true) == nullptr)
                return v;
            host::lazy_workspace_veto(stream, &v.word, &v.epoch);
            v.check = host::check_4step_tables();
            return v;
        }

        // fast path: single modulus with lazy headroom.  Workspace layout (Shoup pairs):
        //   [0, n1)  unused | [n1, n1 + N)  the ring's Merge table | [.., + n2)  unused  (layout of rounds 1-3 kept)
        // mods_dev != nullptr: the RNS overload with ONE modulus (how the reference's own example calls
        // the 4-step, test_4step_ntt.cu:126-146): modulus and n^-1 live in device memory, so the first
        // preparation kernel classifies the modulus and publishes the go-flag that the fast kernels and
        // the generic kernels (enqueued behind them with the flag as skip_flag) both test -- the same
        // dual launch as the RNS Merge calls.  `*go_flag_out` receives the flag (nullptr otherwise).
        template <typename T, bool INV>
        bool fourstep_run_lazy(T* in, T* out, const T* n1_table, const T* n2_table, const T* w_table,
                               const Modulus<T>& mod, T ninv, int n_power, int log_n1, int log_n2,
                               int batch_size, hipStream_t stream, const Modulus<T>* mods_dev = nullptr,
                               const T* ninv_dev = nullptr, const unsigned** go_flag_out = nullptr,
                               const PlanUse<T>& plan = PlanUse<T>(), int dev_family = 0, unsigned* host_state = nullptr,
                               const host::FourStepVeto& veto = host::FourStepVeto(), int* self_fallback = nullptr,
                               bool dev_self = false)
        {
            using TW = lazy::Tw<T>;
            // *self_fallback: what the fast kernels of this call do themselves when the table check vetoes them
            // (kern::F_SELF_FALLBACK): 0 nothing, 1 phase 1 of the element-by-element algorithm, 2 all of it
            if (self_fallback != nullptr)
                *self_fallback = 0;
            // eligible: drop-in calls behind which exactly ONE family of fast kernels is enqueued -- checked calls with a
            // host-side modulus, and calls with a device-side modulus for which the host predicts the default family and
            // enqueues nothing else (dev_self; the kernels then run the element-by-element algorithm whenever the call turns
            // out not to be theirs: vetoed tables, a modulus of another family or outside the fast kernels' domain).
            // path = fast-strict: a veto must leave the output untouched, the tests look for that
            const bool self_ok = plan.mode == PLAN_NONE && host::forced_path() != 3 && self_fallback != nullptr &&
                                 (mods_dev == nullptr ? (veto.check && veto.word != nullptr) : (dev_self && dev_family == 0));
            if (plan.mode != PLAN_NONE && mods_dev != nullptr)
                return false;
            // device-side modulus (the RNS overload with one modulus), dev_family:
            //    0  preparation + the kernels of the default lazy family
            //   -1  preparation only (the host predicts another family for this modulus, host::RnsGuess; host_state = the
            //       host-mapped word the preparation kernel reports the state to)
            //   8 / 4 (64-bit words)  the kernels of the 8 q / 4 q family, no preparation: they own the call when the go-flag
            //       says GO_LAZY_8Q / GO_LAZY_4Q (61- / 62-bit modulus); the table was permuted on the device for the family
            //       that will run (prep_merge_from_fourstep)
            const bool do_prep = plan.mode != PLAN_EXECUTE && dev_family <= 0;
            const bool prep_only = (dev_family == -1);
            // host-side modulus: 61- / 62-bit moduli run the same plans on the LIMIT = 8 / 4 kernels (the whole
            // documented domain of the reference, modular_arith.cuh:66-67), like the Merge entry points
            int lim = (mods_dev != nullptr && dev_family > 0) ? dev_family : 0;
            if (mods_dev == nullptr)
            {
                if (!host::modulus_fast<T>(mod) || (INV && ninv >= mod.value))
                    return false;
                lim = host::modulus_lim<T>(mod);
            }
            if (mods_dev != nullptr && INV && ninv_dev == nullptr)
                return false;
            if (host::forced_path() == 1)
                return false;
            const size_t n1 = size_t(1) << log_n1, n2 = size_t(1) << log_n2, n = size_t(1) << n_power;
            // pairs: n1 table | W | n2 table | n^-1 ; then go-flag (16 B) and the normalisation constants
            const size_t pairs = n1 + n + n2 + 2;
            auto* ws = plan.mode != PLAN_NONE
                           ? plan.ws
                           : static_cast<TW*>(host::lazy_workspace(stream, fourstep_ws_bytes<T>(log_n1, log_n2), true));
            if (ws == nullptr)
                return false; // no device memory for the scratch: the generic kernels need none
            // (the n1 / n2 regions held the stage tables of the two-phase W form of rounds 1-3; every plan now reads the
            // ring's Merge table in the W region -- the layout stays so that FourStepPlan::workspace_bytes does)
            TW* ws_w = ws + n1;
            TW* ws_ninv = ws + n1 + n + n2;
            unsigned char* tail = reinterpret_cast<unsigned char*>(ws + pairs);
            // the call's go-flag: the state half of its veto word (the preparation kernel publishes the state there and
            // checks the caller's tables, prep.hip), else -- device-side modulus, no veto -- a word of the workspace tail
            unsigned* go_flag = veto.word != nullptr ? veto.flag() : (mods_dev ? reinterpret_cast<unsigned*>(tail) : nullptr);
            // host-side modulus: the kernel families are the host's choice, the flag can only take the call away
            const unsigned vf = (go_flag != nullptr && mods_dev == nullptr) ? static_cast<unsigned>(kern::F_VETO_ONLY) : 0u;
            auto* norm_arr = mods_dev ? reinterpret_cast<lazy::NormConst*>(tail + 16) : nullptr;
            // Rings that fit one tile (2^12 .. 2^14): the 4-step transform is the Merge transform of the ring with its
            // natural-order side transposed, so ONE contiguous pass does it -- Merge table rebuilt from the caller's
            // tables into the W region of the workspace, transposition in LDS (kern::fourstep_small_lazy)
            int small_tl = plan.mode != PLAN_NONE
                               ? plan.small_tl
                               : host::fourstep_small_tile<T>(n_power, INV, static_cast<unsigned long long>(batch_size));
            if (lim != 0 && small_tl != 12)
                small_tl = 0; // the LIMIT = 8 / 4 kernels exist for 4096-coefficient tiles only
            if (small_tl != 0)
            {
                if (do_prep)
                    host::launch_prep_merge_from_fourstep<T>(n1_table, w_table, ws_w, log_n1, log_n2,
                                                             (n_power >= small_tl) ? small_tl : 0, INV, INV, mod.value, ninv,
                                                             mods_dev, (INV && mods_dev) ? ninv_dev : nullptr, ws_ninv, go_flag,
                                                             norm_arr, stream, host_state, veto, n2_table);
                if (go_flag_out != nullptr)
                    *go_flag_out = go_flag;
                if (plan.mode == PLAN_PREPARE || prep_only)
                    return true;
                kern::LazyArgsT<T> s{};
                s.in = in;
                s.out = out;
                s.tw = ws_w;
                s.mods = mods_dev;
                s.q = mod.value;
                s.q_bit = mod.bit;
                s.q_mu = mod.mu;
                s.ninv = TW{0, 0};
                if (INV && mods_dev == nullptr)
                    s.ninv = TW{ninv, host::shoup_host(ninv, mod.value)};
                if (INV && mods_dev != nullptr)
                    s.ninv_arr = ws_ninv;
                s.go_flag = go_flag;
                s.norm = lazy::norm_const_of(mod.value, mod.bit);
                s.norm_arr = norm_arr;
                s.total = static_cast<unsigned long long>(batch_size) << n_power;
                s.n = n_power;
                s.poly_shift = n_power;
                s.mod_count = 1;
                s.flags = vf;
                // ONE fast kernel is enqueued, and when the table check takes the call away it runs the element-by-element
                // algorithm on its tile itself (kern::fourstep_tile_generic) -- nothing behind the call
                if (self_ok && go_flag != nullptr && log_n1 == kern::XP_L1)
                {
                    s.flags |= kern::F_SELF_FALLBACK;
                    s.fs_n1 = n1_table;
                    s.fs_n2 = n2_table;
                    s.fs_w = w_table;
                    *self_fallback = 2;
                }
                if constexpr (sizeof(T) == 8)
                {
                    if (lim == 8)
                        host::launch_fourstep_lim<INV, 8>(2, log_n1, s, stream);
                    else if (lim == 4)
                        host::launch_fourstep_lim<INV, 4>(2, log_n1, s, stream);
                    else
                        host::launch_fourstep_small_lazy<T, INV>(small_tl, n_power, s, stream);
                }
                else
                    host::launch_fourstep_small_lazy<T, INV>(small_tl, n_power, s, stream);
                return true;
            }
            // Forward, rings larger than a tile: the Merge plan of the ring itself (GPU_4STEP_NTT(x^T) == MergeNTT_w(x)),
            // its first strided pass reading the transposed input (kern::fourstep_first_lazy): in[(c << l1) | r] holds the
            // tile's columns x n1 rows as one run per value of the stage bits below r, so a first pass of k1 >= l1 stages
            // gathers 2^(k1 - l1) coalesced runs, and everything behind it is the Merge transform as it stands with the
            // ring's Merge table (2^17 .. 2^22: the same two sweeps) -- the W matrix is never streamed.
            if constexpr (!INV)
            {
                {
                    int k1 = plan.first_k;
                    const int tlf = plan.mode != PLAN_NONE
                                        ? plan.tile_log
                                        : host::fourstep_fwd_tile<T>(n_power, log_n1, static_cast<unsigned long long>(batch_size),
                                                                     lim, k1);
                    if (do_prep)
                        host::launch_prep_merge_from_fourstep<T>(n1_table, w_table, ws_w, log_n1, log_n2, tlf, false, false,
                                                                 mod.value, T(0), mods_dev, nullptr, nullptr, go_flag,
                                                                 norm_arr, stream, host_state, veto, n2_table);
                    if (go_flag_out != nullptr)
                        *go_flag_out = go_flag;
                    if (plan.mode == PLAN_PREPARE || prep_only)
                        return true;
                    kern::LazyArgsT<T> f{};
                    f.in = in;
                    f.out = out;
                    f.tw = ws_w;
                    f.mods = mods_dev;
                    f.q = mod.value;
                    f.q_bit = mod.bit;
                    f.q_mu = mod.mu;
                    f.ninv = TW{0, 0};
                    f.go_flag = go_flag;
                    f.norm = lazy::norm_const_of(mod.value, mod.bit);
                    f.norm_arr = norm_arr;
                    f.n2_log = log_n1; // row stride of the transposed side
                    f.total = static_cast<unsigned long long>(batch_size) << n_power;
                    f.n = n_power;
                    f.p_lo = n_power - k1;
                    f.poly_shift = n_power;
                    f.mod_count = 1;
                    f.flags = host::lazy_order_flags() | vf;
                    if (self_ok && go_flag != nullptr)
                    {
                        // the gathering first kernel does phase 1 of the element-by-element algorithm when vetoed: only the
                        // n2-point row transforms of the generic kernels are enqueued behind the call
                        f.flags |= kern::F_SELF_FALLBACK;
                        f.fs_n1 = n1_table;
                        f.fs_n2 = n2_table;
                        f.fs_w = w_table;
                        *self_fallback = 1;
                    }
                    {
                        // consecutive sweeps walk the batch in opposite directions, the LAST one forwards
                        const int pn = n_power - k1;
                        const host::Plan rp = host::make_plan_tl(pn, tlf, tlf == 12 ? host::lazy_contig_k(pn) : tlf);
                        if ((rp.count & 1) != 0 && host::lazy_reverse_passes())
                            f.flags |= kern::F_REVERSE;
                    }
                    if constexpr (sizeof(T) == 8)
                    {
                        if (lim == 8)
                            host::launch_fourstep_lim<false, 8>(1, k1, f, stream);
                        else if (lim == 4)
                            host::launch_fourstep_lim<false, 4>(1, k1, f, stream);
                        else
                            host::launch_fourstep_first_lazy<T>(k1, f, stream);
                    }
                    else
                        host::launch_fourstep_first_lazy<T>(k1, f, stream);
                    kern::LazyArgsT<T> r = f;
                    r.in = out;
                    r.flags = host::lazy_order_flags() | vf;
                    r.lim = lim;
                    if constexpr (sizeof(T) == 8)
                        if (lim == 0 && mods_dev == nullptr && host::lazy_lim31_enabled() && host::lazy_lim31_modulus(mod.value))
                            r.lim = 31;
                    if constexpr (sizeof(T) == 4)
                        if (mods_dev == nullptr && host::lazy_lim31_enabled() && host::lazy_wide_modulus32(mod.value))
                            r.lim = 8;
                    {
                        // rings 2^14 .. 2^17: ONE contiguous pass is left and the n2-long rows of `out` fit its tiles -- the
                        // pass runs as an instantiation of its own that does the n2-point phase of the element-by-element
                        // algorithm when the call is not the fast kernels' (launch_fourstep_fwd_last_lazy): nothing behind
                        const int pn = n_power - k1;
                        const host::Plan rp = host::make_plan_tl(pn, tlf, tlf == 12 ? host::lazy_contig_k(pn) : tlf);
                        if (self_ok && go_flag != nullptr && tlf == 12 && rp.count == 1 && (pn == 9 || pn == 11) && log_n2 <= 12)
                        {
                            kern::LazyArgsT<T> a = r; // what run_transform_lazy hands that pass
                            a.p_lo = 0;
                            a.batch = 0;
                            a.flags |= kern::F_SELF_FALLBACK;
                            if constexpr (sizeof(T) == 8)
                            {
                                if (r.lim == 8)
                                    host::launch_fourstep_fwd_last_lazy<T, 8>(pn, a, stream);
                                else if (r.lim == 4)
                                    host::launch_fourstep_fwd_last_lazy<T, 4>(pn, a, stream);
                                else if (r.lim == 31)
                                    host::launch_fourstep_fwd_last_lazy<T, 31>(pn, a, stream);
                                else
                                    host::launch_fourstep_fwd_last_lazy<T, 0>(pn, a, stream);
                            }
                            else
                            {
                                if (r.lim == 8)
                                    host::launch_fourstep_fwd_last_lazy<T, 8>(pn, a, stream);
                                else
                                    host::launch_fourstep_fwd_last_lazy<T, 0>(pn, a, stream);
                            }
                            *self_fallback = 2;
                            return true;
                        }
                    }
                    host::run_transform_lazy<T, false>(r, 0u, 0u, stream, tlf, n_power - k1);
                    return true;
                }
            }
            // Inverse, rings from 2^15: the ring's inverse Merge plan with the transposition on its FIRST pass
            // (kern::fourstep_inv_first_lazy: 12 contiguous Gentleman-Sande stages on the spectrum as it lies, stored
            // transposed), then every remaining stage inside the n2-long rows of `out` -- strided inverse passes of an
            // n2-point ring reading a prefix of the same table (2^15 / 2^16, n2 = 512: one partial contiguous pass, eight
            // rows per tile), n^-1 folded into its slot 1.  No W stream, no W product, 2^17 .. 2^20 in two sweeps instead
            // of three -- and 2^21 / 2^22 (32-bit: 2^20 .. 2^22) as well, on the big first tile of the ring's Merge plan
            // (host::fourstep_inv_tile).
            if constexpr (INV)
            {
                int k_a = 0, k_b = 0;
                const int tli = (plan.mode == PLAN_EXECUTE && plan.inv_tile != 0) ? plan.inv_tile
                                                                                  : host::fourstep_inv_tile<T>(n_power, lim, log_n1);
                if (host::fourstep_inv_merge_split(n_power, log_n1, k_a, k_b, tli))
                {
                    if (do_prep)
                        host::launch_prep_merge_from_fourstep<T>(n1_table, w_table, ws_w, log_n1, log_n2, tli, true, true,
                                                                 mod.value, ninv, mods_dev, mods_dev ? ninv_dev : nullptr,
                                                                 ws_ninv, go_flag, norm_arr, stream, host_state, veto, n2_table);
                    if (go_flag_out != nullptr)
                        *go_flag_out = go_flag;
                    if (plan.mode == PLAN_PREPARE || prep_only)
                        return true;
                    const bool rows512 = (k_a == 0); // 2^15 / 2^16: one partial contiguous pass over the 512-long rows
                    const int passes = k_b != 0 ? 3 : 2;
                    const bool rev = host::lazy_reverse_passes();
                    kern::LazyArgsT<T> f{};
                    f.in = in;
                    f.out = out;
                    f.tw = ws_w;
                    f.mods = mods_dev;
                    f.q = mod.value;
                    f.q_bit = mod.bit;
                    f.q_mu = mod.mu;
                    f.ninv = TW{0, 0};
                    f.go_flag = go_flag;
                    f.norm = lazy::norm_const_of(mod.value, mod.bit);
                    f.norm_arr = norm_arr;
                    f.n2_log = log_n2; // row stride of the transposed side
                    f.total = static_cast<unsigned long long>(batch_size) << n_power;
                    f.n = n_power;
                    f.poly_shift = n_power;
                    f.mod_count = 1;
                    // from 2^20 the per-lane twiddles of the pass are tens of MiB per polynomial: poly-minor block order
                    f.batch = (n_power >= 20 && batch_size >= 2) ? batch_size : 0;
                    f.flags = host::lazy_order_flags() | vf;
                    const bool self_inv = self_ok && go_flag != nullptr && lim == 0;
                    if (self_inv)
                    {
                        // vetoed: the transposing first kernel does phase 1 of the element-by-element algorithm, the row
                        // pass of the rings 2^14 .. 2^16 phase 2 (nothing behind the call); larger rings keep the generic
                        // row transforms behind them
                        f.flags |= kern::F_SELF_FALLBACK;
                        f.fs_n1 = n1_table;
                        f.fs_n2 = n2_table;
                        f.fs_w = w_table;
                        *self_fallback = rows512 ? 2 : 1;
                    }
                    if (rev && ((passes - 1) & 1) != 0) // the last sweep walks the batch forwards, the one before it backwards, ...
                        f.flags |= kern::F_REVERSE;
                    bool wide32 = false;
                    if constexpr (sizeof(T) == 4)
                        wide32 = mods_dev == nullptr && host::lazy_lim31_enabled() && host::lazy_wide_modulus32(mod.value);
                    if constexpr (sizeof(T) == 4)
                    {
                        if (wide32)
                            host::launch_fourstep_inv_first_lazy<T, 8>(log_n1, f, stream, tli);
                        else
                            host::launch_fourstep_inv_first_lazy<T, 0>(log_n1, f, stream, tli);
                    }
                    else
                    {
                        if (lim == 8)
                            host::launch_fourstep_inv_first_lazy<T, 8>(log_n1, f, stream);
                        else if (lim == 4)
                            host::launch_fourstep_inv_first_lazy<T, 4>(log_n1, f, stream);
                        else
                            host::launch_fourstep_inv_first_lazy<T, 0>(log_n1, f, stream, tli);
                    }

                    kern::LazyArgsT<T> r = f;
                    r.in = out;
                    r.n = log_n2;
                    r.poly_shift = log_n2;
                    r.batch = 0;
                    if (mods_dev == nullptr)
                        r.ninv = TW{ninv, host::shoup_host(ninv, mod.value)};
                    else
                        r.ninv_arr = ws_ninv;
                    if (rows512)
                    {
                        r.flags = host::lazy_order_flags() | vf | (self_inv ? static_cast<unsigned>(kern::F_SELF_FALLBACK) : 0u);
                        if constexpr (sizeof(T) == 4)
                        {
                            if (wide32)
                                host::launch_fourstep_inv_rows_lazy<T, 8>(log_n2, 12 - log_n1, r, stream);
                            else
                                host::launch_fourstep_inv_rows_lazy<T, 0>(log_n2, 12 - log_n1, r, stream);
                        }
                        else
                        {
                            if (lim == 8)
                                host::launch_fourstep_inv_rows_lazy<T, 8>(log_n2, 12 - log_n1, r, stream);
                            else if (lim == 4)
                                host::launch_fourstep_inv_rows_lazy<T, 4>(log_n2, 12 - log_n1, r, stream);
                            else
                                host::launch_fourstep_inv_rows_lazy<T, 0>(log_n2, 12 - log_n1, r, stream);
                        }
                        return true;
                    }
                    const host::Pass pa{false, k_a, tli - log_n1};
                    const host::Pass pb{false, k_b, tli - log_n1 + k_a};
                    // 32-bit rings behind a 16384-coefficient first tile: the strided pass on that tile as well, like the
                    // ring's Merge plan (2^22: 100 against 139 us per 2^26 coefficients)
                    const int stl = (sizeof(T) == 4 && tli == 14) ? 14 : 12;
                    for (int i = 1; i < passes; i++)
                    {
                        kern::LazyArgsT<T> x = r;
                        const host::Pass& p = (i == 1) ? pa : pb;
                        x.p_lo = p.p_lo;
                        x.flags = host::lazy_order_flags() | vf;
                        if (rev && ((passes - 1 - i) & 1) != 0)
                            x.flags |= kern::F_REVERSE;
                        const bool last = (i == passes - 1);
                        if constexpr (sizeof(T) == 4)
                        {
                            if (wide32)
                                host::launch_pass_lazy_u32w<true>(p, stl, false, last, x, stream);
                            else
                                host::launch_pass_lazy<T, true>(p, stl, false, last, x, stream);
                        }
                        else
                        {
                            if (lim == 8)
                                host::launch_pass_lazy_lim<true, 8>(p, false, last, x, stream);
                            else if (lim == 4)
                                host::launch_pass_lazy_lim<true, 4>(p, false, last, x, stream);
                            else
                                host::launch_pass_lazy<T, true>(p, 12, false, last, x, stream);
                        }
                    }
                    return true;
                }
            }
            // (every ring of the reference's range, 2^12 .. 2^24, has one of the plans above)
            return false;
        }

        // Natural-order transforms (extension) in Merge form.  With x the natural-order polynomial,
        // NTT_4STEP_CPU::ntt(x) is the TRANSPOSE of the bit-reversed Merge spectrum of x (read as n1 x n2), so:
        //   forward  1. STRIDED Merge pass over the top log2(n1) index bits, in place on `in` (lazy out);
        //            2. n2 > 512: STRIDED Merge pass over index bits [8, log2 n2), in place (lazy);
        //            3. the low K <= 9 stages on 2^K-column runs of 2^(12-K) consecutive rows, stored transposed
        //               into `out` (canonical): out[c * n1 + r] = row r, column c.
        //   inverse  the same sweeps backwards: transposed load + low K Gentleman-Sande stages into `out`, the
        //            strided passes in place on `out`, N^-1 folded into the final stage; `in` is left intact.
        // All twiddles come from the ring's Merge table (rebuilt from the caller's 4-step tables into the W region of
        // the workspace, plain stage layout): no W stream, no W product.  `in` is overwritten by the forward transform
        // (the reference's own three-call sequence ping-pongs through it too).
        template <typename T>
        void natural_args(kern::LazyArgsT<T>& a, const lazy::Tw<T>* table, const Modulus<T>& mod, int n_power, int log_n1,
                          int log_n2, int batch_size)
        {
            a.tw = table;
            a.mods = nullptr;
            a.q = mod.value;
            a.q_bit = mod.bit;
            a.q_mu = mod.mu;
            a.ninv_arr = nullptr;
            a.ninv = lazy::Tw<T>{0, 0};
            a.go_flag = nullptr;
            a.norm = lazy::norm_const_of(mod.value, mod.bit);
            a.norm_arr = nullptr;
            a.n2_log = log_n1;  // row stride of the column-major (n2 x n1) side
            a.row_log = log_n2; // row stride of the row-major (n1 x n2) side
            a.batch = 0;        // plain block order
            a.total = static_cast<unsigned long long>(batch_size) << n_power;
            a.n = n_power;
            a.poly_shift = n_power;
            a.mod_count = 1;
            a.p_lo = 0;
            a.flags = host::lazy_order_flags();
        }

        template <typename T>
        bool fourstep_natural_forward_lazy(T* in, T* out, const T* n1_table, const T* n2_table,
                                           const T* w_table, const Modulus<T>& mod, int n_power, int log_n1,
                                           int log_n2, int batch_size, hipStream_t stream,
                                           const PlanUse<T>& plan = PlanUse<T>(),
                                           const host::FourStepVeto& veto = host::FourStepVeto())
        {
            using TW = lazy::Tw<T>;
            // 61- / 62-bit moduli (64-bit words): the same sweeps on the 4 q kernels (round 4; Barrett kernels between two
            // transposes before)
            if (!host::modulus_fast<T>(mod))
                return false;
            const bool wide = host::modulus_lim<T>(mod) != 0;
            if (host::forced_path() == 1)
                return false;
            const size_t n1 = size_t(1) << log_n1;
            auto* ws = plan.mode != PLAN_NONE
                           ? plan.ws
                           : static_cast<TW*>(host::lazy_workspace(stream, fourstep_ws_bytes<T>(log_n1, log_n2), true));
            if (ws == nullptr)
                return false; // no device memory for the scratch: the generic kernels need none
            TW* ws_merge = ws + n1; // the W region of the workspace holds the ring's Merge table
            // rings that fill one tile: ONE launch (contiguous Merge pass, the transposition in LDS); the table then
            // carries the per-tile permutation of its last three stages
            int small_tl = plan.mode != PLAN_NONE
                               ? plan.small_tl
                               : host::fourstep_small_tile<T>(n_power, false, static_cast<unsigned long long>(batch_size));
            if (wide && small_tl != 12)
                small_tl = 0; // the 4 q kernels exist for 4096-coefficient tiles only
            if (plan.mode != PLAN_EXECUTE)
                host::launch_prep_merge_from_fourstep<T>(n1_table, w_table, ws_merge, log_n1, log_n2, small_tl, false, false,
                                                         mod.value, T(0), nullptr, nullptr, nullptr, nullptr, nullptr, stream,
                                                         nullptr, veto, n2_table);
            if (plan.mode == PLAN_PREPARE)
                return true;

            kern::LazyArgsT<T> a{};
            natural_args<T>(a, ws_merge, mod, n_power, log_n1, log_n2, batch_size);
            if (veto.word != nullptr && plan.mode == PLAN_NONE)
            {
                a.go_flag = veto.flag();
                a.flags |= kern::F_VETO_ONLY;
            }
            if (small_tl != 0)
            {
                a.in = in;
                a.out = out;
                if constexpr (sizeof(T) == 8)
                    if (wide)
                    {
                        host::launch_fourstep_lim<false, 4>(3, log_n1, a, stream);
                        return true;
                    }
                host::launch_fourstep_small_lazy<T, false>(small_tl, n_power, a, stream, true);
                return true;
            }
            auto strided = [&](const host::Pass& p, bool first) {
                if constexpr (sizeof(T) == 8)
                    if (wide)
                        return host::launch_pass_lazy_lim<false, 4>(p, first, false, a, stream);
                host::launch_pass_lazy<T, false>(p, 12, first, false, a, stream);
            };
            a.in = in;
            a.out = in;
            // 1. top log2(n1) stages, canonical in, lazy out
            a.p_lo = log_n2;
            strided(host::Pass{false, log_n1, log_n2}, true);
            // 2. index bits [8, log2 n2)
            const int k_last = (log_n2 > 9) ? 8 : log_n2;
            if (log_n2 > 9)
            {
                a.p_lo = k_last;
                strided(host::Pass{false, log_n2 - k_last, k_last}, false);
            }
            // 3. low stages + transposed store (big rings: poly-minor block order, the batch shares the table in L2)
            a.in = in;
            a.out = out;
            a.p_lo = 0;
            a.batch = (n_power >= 20 && batch_size >= 2) ? batch_size : 0;
            if constexpr (sizeof(T) == 8)
                if (wide)
                {
                    host::launch_fourstep_nat_last_lazy<T, 4>(k_last, a, stream);
                    return true;
                }
            host::launch_fourstep_nat_last_lazy<T>(k_last, a, stream);
            return true;
        }

        template <typename T>
        bool fourstep_natural_inverse_lazy(T* in, T* out, const T* n1_table, const T* n2_table,
                                           const T* w_table, const Modulus<T>& mod, T ninv, int n_power,
                                           int log_n1, int log_n2, int batch_size, hipStream_t stream,
                                           const PlanUse<T>& plan = PlanUse<T>(),
                                           const host::FourStepVeto& veto = host::FourStepVeto())
        {
            using TW = lazy::Tw<T>;
            if (!host::modulus_fast<T>(mod) || ninv >= mod.value)
                return false;
            const bool wide = host::modulus_lim<T>(mod) != 0; // 61- / 62-bit moduli: the 4 q kernels
            if (host::forced_path() == 1)
                return false;
            const size_t n1 = size_t(1) << log_n1;
            auto* ws = plan.mode != PLAN_NONE
                           ? plan.ws
                           : static_cast<TW*>(host::lazy_workspace(stream, fourstep_ws_bytes<T>(log_n1, log_n2), true));
            if (ws == nullptr)
                return false; // no device memory for the scratch: the generic kernels need none
            TW* ws_merge = ws + n1;
            int small_tl = plan.mode != PLAN_NONE
                               ? plan.small_tl
                               : host::fourstep_small_tile<T>(n_power, true, static_cast<unsigned long long>(batch_size), true);
            if (wide && small_tl != 12)
                small_tl = 0;
            // inverse Merge table of the ring, N^-1 folded into the single twiddle of the final stage (slot 1)
            if (plan.mode != PLAN_EXECUTE)
                host::launch_prep_merge_from_fourstep<T>(n1_table, w_table, ws_merge, log_n1, log_n2, small_tl, true, true,
                                                         mod.value, ninv, nullptr, nullptr, nullptr, nullptr, nullptr, stream,
                                                         nullptr, veto, n2_table);
            if (plan.mode == PLAN_PREPARE)
                return true;

            kern::LazyArgsT<T> a{};
            natural_args<T>(a, ws_merge, mod, n_power, log_n1, log_n2, batch_size);
            if (veto.word != nullptr && plan.mode == PLAN_NONE)
            {
                a.go_flag = veto.flag();
                a.flags |= kern::F_VETO_ONLY;
            }
            if (small_tl != 0)
            {
                a.in = in;
                a.out = out;
                a.ninv = TW{ninv, host::shoup_host(ninv, mod.value)};
                if constexpr (sizeof(T) == 8)
                    if (wide)
                    {
                        host::launch_fourstep_lim<true, 4>(3, log_n1, a, stream);
                        return true;
                    }
                host::launch_fourstep_small_lazy<T, true>(small_tl, n_power, a, stream, true);
                return true;
            }
            // 1. transposed load + the low stages, row-major `out`, lazy
            a.in = in;
            a.out = out;
            const int k_first = (log_n2 > 9) ? 8 : log_n2;
            a.batch = (n_power >= 20 && batch_size >= 2) ? batch_size : 0; // poly-minor: the batch shares the table in L2
            bool first_done = false;
            if constexpr (sizeof(T) == 8)
                if (wide)
                {
                    host::launch_fourstep_nat_first_inv_lazy<T, 4>(k_first, a, stream);
                    first_done = true;
                }
            if (!first_done)
                host::launch_fourstep_nat_first_inv_lazy<T>(k_first, a, stream);
            a.batch = 0;
            auto strided = [&](const host::Pass& p, bool last) {
                if constexpr (sizeof(T) == 8)
                    if (wide)
                        return host::launch_pass_lazy_lim<true, 4>(p, false, last, a, stream);
                host::launch_pass_lazy<T, true>(p, 12, false, last, a, stream);
            };
            // 2. index bits [8, log2 n2), in place
            a.in = out;
            if (log_n2 > 9)
            {
                a.p_lo = k_first;
                strided(host::Pass{false, log_n2 - k_first, k_first}, false);
            }
            // 3. top log2(n1) stages with N^-1: canonical, natural order
            a.p_lo = log_n2;
            a.ninv = TW{ninv, host::shoup_host(ninv, mod.value)};
            strided(host::Pass{false, log_n1, log_n2}, true);
            return true;
        }

        // ---- dispatch of the two GPU_4STEP_NTT overloads ------------------------------------------------------------------
        // DESIGN.md 3.8 is this function as a table (tests/dispatch_rows.py; walked on the GPU by
        // tests/test_gpu_dispatch_table.py).  A ROUTE enqueues the fast kernels that can take the call and answers one
        // question: what, if anything, has to sit behind them?
        //   nothing   the fast kernels serve the call whatever the preparation kernel finds (they are their own fall-back up
        //             to 2^16, 2^17 forward), or the caller opted out of the table check, or a test hook asked for it
        //   generic   the element-by-element kernels (fourstep_run), skipped on the device unless the call's go-flag says
        //             the fast kernels did not take it (table check vetoed; modulus of another family than predicted)
        struct Behind
        {
            bool generic = true;              // enqueue fourstep_run at all
            const unsigned* flag = nullptr;   // device word it tests (nullptr: runs unconditionally)
            unsigned skip_value = 0u;         // 0: return when *flag != GO_GENERIC; s: return when *flag == s (PassArgs::skip_value)
            bool skip_phase1 = false;         // the fast first kernel already did the n1-point phase of a vetoed call
            static Behind nothing() { return Behind{false, nullptr, 0u, false}; }
        };
        template <typename T> struct FourStepCall
        {
            T *in, *out;
            const T *n1_table, *n2_table, *w_table;
            const Modulus<T>* mods; // device-side modulus (RNS overload) or nullptr
            Modulus<T> mod;         // host-side modulus
            int mod_count;
            const T* ninv_arr;
            T ninv;
            int n_power, l1, l2;
            bool inverse;
            int batch_size;
            hipStream_t stream;
        };

        // test hook path = generic-capped, RNS overload with one modulus: the generic kernels exactly as they run behind a
        // go-flag that names them (capped grid walking the tiles)
        template <typename T> Behind route_generic_capped(const FourStepCall<T>& c)
        {
            auto* flag = static_cast<unsigned*>(host::lazy_workspace(c.stream, 16));
            GPUNTT_HIP_CHECK(hipMemsetAsync(flag, 0, 16, c.stream));
            return Behind{true, flag, 0u, false};
        }

        // RNS overload, ONE modulus in device memory (the reference examples' calling style, test_4step_ntt.cu:126-146): the
        // lazy family this modulus needed last time (host::RnsGuess; unknown: every family), the preparation kernel
        // classifies the modulus, checks the tables and publishes the go-flag
        template <typename T> Behind route_device_modulus(const FourStepCall<T>& c)
        {
            const host::FourStepVeto veto = fourstep_veto<T>(c.l1, c.l2, c.stream);
            const host::RnsGuess guess = host::rns_guess(c.mods, 1, static_cast<int>(sizeof(T)) | 0x40, c.inverse, nullptr,
                                                         true); // (0x40: the 4-step entry keeps its own prediction slot)
            const bool only_default = !guess.all_families && guess.state == kern::GO_LAZY;
            const unsigned* flag = nullptr;
            int self_fb = 0; // what the default family's kernels can do themselves: 2 everything, 1 the n1-point phase
            auto enqueue = [&](int family, const unsigned** flag_out) {
                // only_default: the ONE family enqueued for the call -- its kernels may be their own fall-back
                int* self = (family == 0 && only_default) ? &self_fb : nullptr;
                if (!c.inverse)
                    return fourstep_run_lazy<T, false>(c.in, c.out, c.n1_table, c.n2_table, c.w_table, Modulus<T>(), T(0), c.n_power,
                                                       c.l1, c.l2, c.batch_size, c.stream, c.mods, c.ninv_arr, flag_out, PlanUse<T>(),
                                                       family, guess.state_out, veto, self, self != nullptr);
                return fourstep_run_lazy<T, true>(c.in, c.out, c.n1_table, c.n2_table, c.w_table, Modulus<T>(), T(0), c.n_power, c.l1,
                                                  c.l2, c.batch_size, c.stream, c.mods, c.ninv_arr, flag_out, PlanUse<T>(), family,
                                                  guess.state_out, veto, self, self != nullptr);
            };
            // the first enqueue prepares the table and publishes the flag; with it the default family unless another is predicted
            enqueue((guess.all_families || only_default) ? 0 : -1, &flag);
            if (flag == nullptr)
                return Behind{}; // no scratch / path = generic: the generic kernels are the call
            if (self_fb == 2)
                return Behind::nothing();
            if constexpr (sizeof(T) == 8)
                for (int fam : {8, 4})
                    if (guess.all_families || guess.state == (fam == 8 ? kern::GO_LAZY_8Q : kern::GO_LAZY_4Q))
                        enqueue(fam, nullptr);
            if (host::forced_path() == 3)
                return Behind::nothing(); // test hook fast-strict: the lazy families must own the call
            Behind b{true, flag, 0u, self_fb == 1};
            if (!guess.all_families)
            {
                // ONE family in front: the generic kernels run for every state but that family's (GO_GENERIC predicted: always)
                if (guess.state == kern::GO_GENERIC)
                    b.flag = nullptr;
                else
                    b.skip_value = guess.state;
            }
            return b;
        }

        // host-side modulus: the host picks the family; only the table check can take the call away from it
        template <typename T> Behind route_host_modulus(const FourStepCall<T>& c)
        {
            const host::FourStepVeto veto = fourstep_veto<T>(c.l1, c.l2, c.stream);
            int self_fb = 0;
            const bool done =
                !c.inverse ? fourstep_run_lazy<T, false>(c.in, c.out, c.n1_table, c.n2_table, c.w_table, c.mod, c.ninv, c.n_power, c.l1,
                                                         c.l2, c.batch_size, c.stream, nullptr, nullptr, nullptr, PlanUse<T>(), 0, nullptr,
                                                         veto, &self_fb)
                           : fourstep_run_lazy<T, true>(c.in, c.out, c.n1_table, c.n2_table, c.w_table, c.mod, c.ninv, c.n_power, c.l1,
                                                        c.l2, c.batch_size, c.stream, nullptr, nullptr, nullptr, PlanUse<T>(), 0, nullptr,
                                                        veto, &self_fb);
            if (!done)
            {
                if (host::forced_path() == 3) // test hook, like the Merge entry points
                    throw std::invalid_argument("fast path unavailable for this call (path = fast-strict)");
                return Behind{}; // modulus outside the fast kernels' domain, tiny job, no scratch, path = generic
            }
            if (!veto.check || host::forced_path() == 3 || self_fb == 2)
                return Behind::nothing();
            return Behind{true, veto.flag(), 0u, self_fb == 1}; // "return unless the state is GO_GENERIC"
        }

        template <typename T>
        void fourstep_dispatch(T* in, T* out, const T* n1_table, const T* n2_table, const T* w_table,
                               const Modulus<T>* mods, Modulus<T> mod, int mod_count,
                               const T* ninv_arr, T ninv, int n_power, type ntt_type, int batch_size,
                               hipStream_t stream)
        {
            int l1 = 0, l2 = 0;
            if ((ntt_type != FORWARD && ntt_type != INVERSE) || !fourstep_shape(n_power, l1, l2))
            {
                // reference behaviour: report on stdout and return (ntt_4step.cu:2529-2532)
                std::printf("This ring size is not supported!\n");
                return;
            }
            if (batch_size <= 0)
                return;
            if ((static_cast<unsigned long long>(batch_size) << n_power) >> kern::TL > 0x7fffffffull)
                throw std::invalid_argument("batch_size * N too large for one launch");
            const FourStepCall<T> c{in, out, n1_table, n2_table, w_table, mods, mod, mod_count, ninv_arr, ninv, n_power, l1, l2,
                                    ntt_type == INVERSE, batch_size, stream};
            // overload x modulus location -> route (mod_count > 1 shares one set of tables between its moduli like the
            // reference, ntt_4step.cu:81-82, 111-114: only the element-by-element kernels compute that)
            const Behind behind = (mods == nullptr)                                     ? route_host_modulus<T>(c)
                                  : (mod_count == 1 && host::forced_path() == 4)        ? route_generic_capped<T>(c)
                                  : (mod_count == 1)                                    ? route_device_modulus<T>(c)
                                                                                        : Behind{};
            if (!behind.generic)
                return;
            if (ntt_type == FORWARD)
                fourstep_run<T, false>(in, out, n1_table, n2_table, w_table, mods, mod, mod_count, ninv_arr, ninv, n_power, l1, l2,
                                       batch_size, stream, behind.flag, behind.skip_value, behind.skip_phase1);
            else
                fourstep_run<T, true>(in, out, n1_table, n2_table, w_table, mods, mod, mod_count, ninv_arr, ninv, n_power, l1, l2,
                                      batch_size, stream, behind.flag, behind.skip_value, behind.skip_phase1);
        }
    } // namespace

    template <typename T>
    __host__ void GPU_4STEP_NTT(T* device_in, T* device_out, Root<T>* n1_root_of_unity_table,
                                Root<T>* n2_root_of_unity_table, Root<T>* W_root_of_unity_table,
                                Modulus<T> modulus, ntt4step_configuration<T> cfg, int batch_size)
    {
        host::WorkspaceScope ws_scope; // scratch lock held until the last launch of this call
        fourstep_dispatch<T>(device_in, device_out, n1_root_of_unity_table, n2_root_of_unity_table,
                             W_root_of_unity_table, nullptr, modulus, 1, nullptr, cfg.mod_inverse,
                             cfg.n_power, cfg.ntt_type, batch_size, cfg.stream);
    }

    template <typename T>
    __host__ void GPU_4STEP_NTT(T* device_in, T* device_out, Root<T>* n1_root_of_unity_table,
                                Root<T>* n2_root_of_unity_table, Root<T>* W_root_of_unity_table,
                                Modulus<T>* modulus, ntt4step_rns_configuration<T> cfg,
                                int batch_size, int mod_count)
    {
        host::WorkspaceScope ws_scope; // scratch lock held until the last launch of this call
        if (mod_count <= 0 || modulus == nullptr)
            throw std::invalid_argument("Invalid mod_count!");
        fourstep_dispatch<T>(device_in, device_out, n1_root_of_unity_table, n2_root_of_unity_table,
                             W_root_of_unity_table, modulus, Modulus<T>(), mod_count,
                             cfg.mod_inverse, static_cast<T>(0), cfg.n_power, cfg.ntt_type,
                             batch_size, cfg.stream);
    }

    // Extension: the whole natural-order pipeline of the reference's examples in
    // one call.  FORWARD == GPU_Transpose(in,t,n1,n2) ; GPU_4STEP_NTT(t,u,FORWARD) ;
    // GPU_Transpose(u,out,n1,n2) == NTT_4STEP_CPU::ntt, computed in three sweeps instead of five
    // (no transpose sweeps).  INVERSE == intt_first_transpose ; GPU_4STEP_NTT(INVERSE) ;
    // GPU_Transpose == NTT_4STEP_CPU::intt (composed from the existing passes).  device_in is
    // used as scratch and holds no meaningful data afterwards; device_in != device_out.
    template <typename T>
    __host__ void GPU_4STEP_NTT_NaturalOrder(T* device_in, T* device_out, Root<T>* n1_root_of_unity_table,
                                             Root<T>* n2_root_of_unity_table, Root<T>* W_root_of_unity_table,
                                             Modulus<T> modulus, ntt4step_configuration<T> cfg, int batch_size)
    {
        host::WorkspaceScope ws_scope; // scratch lock held until the last launch of this call
        int l1 = 0, l2 = 0;
        if ((cfg.ntt_type != FORWARD && cfg.ntt_type != INVERSE) || !fourstep_shape(cfg.n_power, l1, l2))
        {
            std::printf("This ring size is not supported!\n");
            return;
        }
        if (device_in == device_out)
            throw std::invalid_argument("GPU_4STEP_NTT_NaturalOrder needs distinct buffers");
        if (batch_size <= 0)
            return;
        if ((static_cast<unsigned long long>(batch_size) << cfg.n_power) >> kern::TL > 0x7fffffffull)
            throw std::invalid_argument("batch_size * N too large for one launch");
        const int n1 = 1 << l1, n2 = 1 << l2;
        // the fast sweeps, and behind the call's veto word the reference examples' own composition on the generic kernels
        // (every launch of it returns at once unless the table check handed the call over; prep.hip)
        const host::FourStepVeto veto = fourstep_veto<T>(l1, l2, cfg.stream);
        const bool fwd = (cfg.ntt_type == FORWARD);
        const bool done = fwd ? fourstep_natural_forward_lazy<T>(device_in, device_out, n1_root_of_unity_table,
                                                                 n2_root_of_unity_table, W_root_of_unity_table, modulus,
                                                                 cfg.n_power, l1, l2, batch_size, cfg.stream, PlanUse<T>(), veto)
                              : fourstep_natural_inverse_lazy<T>(device_in, device_out, n1_root_of_unity_table,
                                                                 n2_root_of_unity_table, W_root_of_unity_table, modulus,
                                                                 cfg.mod_inverse, cfg.n_power, l1, l2, batch_size, cfg.stream,
                                                                 PlanUse<T>(), veto);
        if (done && (!veto.check || host::forced_path() == 3))
            return;
        const unsigned* skip = done ? veto.flag() : nullptr;
        // forward: GPU_Transpose(in, out, n1, n2); inverse: NTT_4STEP_CPU::intt_first_transpose, flat[i*n2+j] = x[i + j*n1]
        transpose_on<T>(device_in, device_out, fwd ? n1 : n2, fwd ? n2 : n1, cfg.n_power, batch_size, cfg.stream, skip);
        if (fwd)
            fourstep_run<T, false>(device_out, device_in, n1_root_of_unity_table, n2_root_of_unity_table, W_root_of_unity_table,
                                   nullptr, modulus, 1, nullptr, cfg.mod_inverse, cfg.n_power, l1, l2, batch_size, cfg.stream,
                                   skip, 0u);
        else
            fourstep_run<T, true>(device_out, device_in, n1_root_of_unity_table, n2_root_of_unity_table, W_root_of_unity_table,
                                  nullptr, modulus, 1, nullptr, cfg.mod_inverse, cfg.n_power, l1, l2, batch_size, cfg.stream,
                                  skip, 0u);
        transpose_on<T>(device_in, device_out, n1, n2, cfg.n_power, batch_size, cfg.stream, skip);
    }

    // ------------------------------------------------------------------ FourStepPlan ----
    template <typename T> struct FourStepPlan<T>::Impl
    {
        T *n1_table = nullptr, *n2_table = nullptr, *w_table = nullptr;
        Modulus<T> mod{};
        T ninv = 0;
        int n = 0, l1 = 0, l2 = 0;
        bool inverse = false, natural = false, fast = false, owns_ws = false;
        PlanUse<T> use{};
    };

    template <typename T> size_t FourStepPlan<T>::workspace_bytes(int n_power)
    {
        int l1 = 0, l2 = 0;
        if (!fourstep_shape(n_power, l1, l2))
            throw std::invalid_argument("Invalid n_power range!");
        return sizeof(lazy::Tw<T>) * ((size_t(1) << l1) + (size_t(1) << n_power) + (size_t(1) << l2) + 2);
    }

    template <typename T>
    FourStepPlan<T>::FourStepPlan(Root<T>* n1_root_of_unity_table, Root<T>* n2_root_of_unity_table,
                                  Root<T>* W_root_of_unity_table, Modulus<T> modulus, ntt4step_configuration<T> cfg,
                                  bool natural_order, int batch_hint, void* workspace_device)
        : p_(nullptr)
    {
        int l1 = 0, l2 = 0;
        if (!fourstep_shape(cfg.n_power, l1, l2))
            throw std::invalid_argument("Invalid n_power range!");
        if (cfg.ntt_type != FORWARD && cfg.ntt_type != INVERSE)
            throw std::invalid_argument("Invalid ntt_type!");
        if (n1_root_of_unity_table == nullptr || n2_root_of_unity_table == nullptr || W_root_of_unity_table == nullptr)
            throw std::invalid_argument("null pointer argument");
        if (batch_hint < 1)
            batch_hint = 1;
        Impl* p = new Impl();
        p_ = p;
        try
        {
            p->n1_table = n1_root_of_unity_table;
            p->n2_table = n2_root_of_unity_table;
            p->w_table = W_root_of_unity_table;
            p->mod = modulus;
            p->ninv = cfg.mod_inverse;
            p->n = cfg.n_power;
            p->l1 = l1;
            p->l2 = l2;
            p->inverse = (cfg.ntt_type == INVERSE);
            p->natural = natural_order;
            if (workspace_device != nullptr)
                p->use.ws = static_cast<lazy::Tw<T>*>(workspace_device);
            else
            {
                void* mem = nullptr;
                GPUNTT_HIP_CHECK(hipMalloc(&mem, workspace_bytes(cfg.n_power)));
                p->use.ws = static_cast<lazy::Tw<T>*>(mem);
                p->owns_ws = true;
            }
            p->use.mode = PLAN_PREPARE;
            if (!p->inverse && !natural_order)
            {
                // forward: the Merge plan of the ring, first pass with the transposed gather
                p->use.tile_log = host::fourstep_fwd_tile<T>(p->n, l1, static_cast<unsigned long long>(batch_hint),
                                                             host::modulus_lim<T>(modulus), p->use.first_k);
            }
            p->use.small_tl =
                host::fourstep_small_tile<T>(p->n, p->inverse, static_cast<unsigned long long>(batch_hint), natural_order);
            if (p->inverse && !natural_order)
                p->use.inv_tile = host::fourstep_inv_tile<T>(p->n, host::modulus_lim<T>(modulus), l1);
            // the plan's own veto word: the head of the (unused) n1 region of its workspace.  The preparation kernel checks
            // the caller's three tables once, here (option check_4step_tables); tables that are not those of one root make
            // the plan a generic one -- execute() then runs the element-by-element kernels, like the drop-in call would
            host::FourStepVeto veto;
            veto.word = reinterpret_cast<unsigned long long*>(p->use.ws);
            veto.check = host::check_4step_tables();
            GPUNTT_HIP_CHECK(hipMemsetAsync(veto.word, 0xff, 16, cfg.stream));
            // the eligibility checks of the fast paths decide (modulus width, n^-1 canonical, option "path")
            if (natural_order)
                p->fast = p->inverse ? fourstep_natural_inverse_lazy<T>(nullptr, nullptr, p->n1_table, p->n2_table,
                                                                        p->w_table, p->mod, p->ninv, p->n, l1, l2, 1,
                                                                        cfg.stream, p->use, veto)
                                     : fourstep_natural_forward_lazy<T>(nullptr, nullptr, p->n1_table, p->n2_table,
                                                                        p->w_table, p->mod, p->n, l1, l2, 1, cfg.stream,
                                                                        p->use, veto);
            else
                p->fast = p->inverse ? fourstep_run_lazy<T, true>(nullptr, nullptr, p->n1_table, p->n2_table, p->w_table,
                                                                  p->mod, p->ninv, p->n, l1, l2, 1, cfg.stream, nullptr,
                                                                  nullptr, nullptr, p->use, 0, nullptr, veto)
                                     : fourstep_run_lazy<T, false>(nullptr, nullptr, p->n1_table, p->n2_table,
                                                                   p->w_table, p->mod, p->ninv, p->n, l1, l2, 1,
                                                                   cfg.stream, nullptr, nullptr, nullptr, p->use, 0, nullptr,
                                                                   veto);
            p->use.mode = PLAN_EXECUTE;
            // complete when the constructor returns: execute() may run on any stream (one host wait per plan)
            unsigned long long verdict = 0;
            GPUNTT_HIP_CHECK(hipMemcpyAsync(&verdict, veto.word, sizeof(verdict), hipMemcpyDeviceToHost, cfg.stream));
            GPUNTT_HIP_CHECK(hipStreamSynchronize(cfg.stream));
            if (p->fast && static_cast<unsigned>(verdict) == kern::GO_GENERIC)
                p->fast = false;
        }
        catch (...)
        {
            if (p->owns_ws && p->use.ws != nullptr)
                (void) hipFree(p->use.ws);
            delete p;
            p_ = nullptr;
            throw;
        }
    }

    template <typename T> FourStepPlan<T>::~FourStepPlan()
    {
        if (p_ != nullptr)
        {
            if (p_->owns_ws && p_->use.ws != nullptr)
                (void) hipFree(p_->use.ws);
            delete p_;
        }
    }

    template <typename T> bool FourStepPlan<T>::fast_path() const { return p_->fast; }

    template <typename T>
    void FourStepPlan<T>::execute(T* device_in, T* device_out, int batch_size, stream_t stream) const
    {
        const Impl& p = *p_;
        if (batch_size <= 0)
            return;
        if (device_in == nullptr || device_out == nullptr)
            throw std::invalid_argument("null pointer argument");
        if (device_in == device_out)
            throw std::invalid_argument("FourStepPlan::execute needs distinct buffers");
        if ((static_cast<unsigned long long>(batch_size) << p.n) >> kern::TL > 0x7fffffffull)
            throw std::invalid_argument("batch_size * N too large for one launch");
        if (!p.fast)
        {
            ntt4step_configuration<T> cfg = {p.n, p.inverse ? INVERSE : FORWARD, p.ninv, stream};
            if (p.natural)
                GPU_4STEP_NTT_NaturalOrder<T>(device_in, device_out, p.n1_table, p.n2_table, p.w_table, p.mod, cfg,
                                              batch_size);
            else
                GPU_4STEP_NTT<T>(device_in, device_out, p.n1_table, p.n2_table, p.w_table, p.mod, cfg, batch_size);
            return;
        }
        bool ok;
        if (p.natural)
            ok = p.inverse ? fourstep_natural_inverse_lazy<T>(device_in, device_out, nullptr, nullptr, nullptr, p.mod,
                                                              p.ninv, p.n, p.l1, p.l2, batch_size, stream, p.use)
                           : fourstep_natural_forward_lazy<T>(device_in, device_out, nullptr, nullptr, nullptr, p.mod,
                                                              p.n, p.l1, p.l2, batch_size, stream, p.use);
        else
            ok = p.inverse ? fourstep_run_lazy<T, true>(device_in, device_out, nullptr, nullptr, nullptr, p.mod, p.ninv,
                                                        p.n, p.l1, p.l2, batch_size, stream, nullptr, nullptr, nullptr,
                                                        p.use)
                           : fourstep_run_lazy<T, false>(device_in, device_out, nullptr, nullptr, nullptr, p.mod, p.ninv,
                                                         p.n, p.l1, p.l2, batch_size, stream, nullptr, nullptr, nullptr,
                                                         p.use);
        if (!ok)
            throw std::runtime_error("FourStepPlan: prepared path refused (option path changed since creation?)");
    }

    template class FourStepPlan<Data32>;
    template class FourStepPlan<Data64>;

    template __host__ void GPU_4STEP_NTT_NaturalOrder<Data32>(Data32*, Data32*, Root<Data32>*, Root<Data32>*,
                                                              Root<Data32>*, Modulus<Data32>,
                                                              ntt4step_configuration<Data32>, int);
    template __host__ void GPU_4STEP_NTT_NaturalOrder<Data64>(Data64*, Data64*, Root<Data64>*, Root<Data64>*,
                                                              Root<Data64>*, Modulus<Data64>,
                                                              ntt4step_configuration<Data64>, int);
    template __host__ void GPU_Transpose<Data32>(Data32*, Data32*, const int, const int, const int,
                                                 const int);
    template __host__ void GPU_Transpose<Data64>(Data64*, Data64*, const int, const int, const int,
                                                 const int);
    template __host__ void GPU_4STEP_NTT<Data32>(Data32*, Data32*, Root<Data32>*, Root<Data32>*,
                                                 Root<Data32>*, Modulus<Data32>,
                                                 ntt4step_configuration<Data32>, int);
    template __host__ void GPU_4STEP_NTT<Data64>(Data64*, Data64*, Root<Data64>*, Root<Data64>*,
                                                 Root<Data64>*, Modulus<Data64>,
                                                 ntt4step_configuration<Data64>, int);
    template __host__ void GPU_4STEP_NTT<Data32>(Data32*, Data32*, Root<Data32>*, Root<Data32>*,
                                                 Root<Data32>*, Modulus<Data32>*,
                                                 ntt4step_rns_configuration<Data32>, int, int);
    template __host__ void GPU_4STEP_NTT<Data64>(Data64*, Data64*, Root<Data64>*, Root<Data64>*,
                                                 Root<Data64>*, Modulus<Data64>*,
                                                 ntt4step_rns_configuration<Data64>, int, int);
} // namespace gpuntt
