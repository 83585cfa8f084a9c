// merge_ntt.hip -- host entry points of the Merge NTT (gpuntt::GPU_NTT / GPU_INTT /
// *_Inplace, single-modulus and RNS overloads) for MI355X.
//
// Replaces reference src/lib/ntt_merge/ntt.cu:2076-3097 (hosts) and :4948-5082 (explicit
// instantiations = the exported symbol set).  Behaviour kept: argument checks and exception
// types, asynchronous launches on cfg.stream with a hipGetLastError() check after each,
// in == out allowed, first pass reads `device_in` and later passes run in place on
// `device_out`, cfg.ntt_type / cfg.zero_padding ignored.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gpuntt/ntt_merge/ntt.cuh"
#include "launch.hpp"
#include "lazy_launch.hpp"

namespace gpuntt
{
    namespace
    {
        template <typename TU>
        kern::PassArgs<TU> base_args(const void* in, TU* out, const TU* roots, int n_power,
                                     ReductionPolynomial poly, int batch_size)
        {
            kern::PassArgs<TU> a{};
            a.in = in;
            a.out = out;
            a.roots = roots;
            a.mods = nullptr;
            a.ninv_arr = nullptr;
            a.ninv = 0;
            a.w_table = nullptr;
            a.total = static_cast<unsigned long long>(batch_size < 0 ? 0 : batch_size) << n_power;
            a.n = n_power;
            a.poly_shift = n_power;
            a.root_shift = n_power;
            a.mod_count = 1;
            a.p_lo = 0;
            a.n2_log = 0;
            a.flags = (poly == ReductionPolynomial::X_N_plus) ? kern::F_NEGACYCLIC : 0u;
            return a;
        }

        inline void check_layout_and_range(NTTLayout layout, int n_power)
        {
            switch (layout)
            {
                case PerPolynomial:
                    if (n_power <= 0 || n_power >= 29)
                        throw std::invalid_argument("Invalid n_power range!");
                    break;
                case PerCoefficient:
                    if (n_power <= 0 || n_power >= 10)
                        throw std::invalid_argument("Invalid n_power range!");
                    break;
                default:
                    throw std::invalid_argument("Invalid ntt_layout!");
            }
        }

        // ---- fast path (lazy residues + prepared Shoup twiddles) ------------------------
        // Used for calls whose moduli leave the lazy headroom: bit <= 60 (Data64) / bit <= 30 (Data32).
        //   single modulus: the host sees Modulus<T>::bit and picks the path;
        //   RNS: the moduli live in device memory, so the twiddle-prep kernel classifies them and
        //        publishes a four-state go-flag (generic / default lazy range / 8 q / 4 q range); every family is
        //        enqueued, each returning at once when the flag names another one (run_transform_lazy_rns).
        // Tiny single-modulus rings (below 2^5 / 2^11), small RNS jobs and RNS stacks of rings of 2 .. 8
        // coefficients use the generic kernels only.
        // option path = generic | fast overrides the size heuristic (testing / A-B timing; host::set_option);
        // moduli without the headroom always take the generic kernels.
        using host::forced_path;

        template <typename TU> inline bool lazy_eligible(int n_power, int batch_size, int mod_count)
        {
            // (RNS stacks of rings below one tile: the tile mixes moduli -- per wave from 1024 coefficients, per lane down
            // to 16, kern::merge_pass_lazy_vqc; rings of 2 .. 8 coefficients with mod_count > 1 keep the generic kernels)
            const bool can = n_power <= host::LAZY_MAX_N_POWER &&
                             !(mod_count > 1 && n_power < host::LAZY_MIN_RNS_N_POWER) &&
                             (static_cast<unsigned long long>(mod_count) << n_power) <= (1ull << 28); // 4 GiB table
            const int fp = forced_path();
            if (fp == 3 && !can)
                throw std::invalid_argument("fast path unavailable for this call (path = fast-strict)");
            if (!can || fp == 1)
                return false;
            if (fp == 2 || fp == 3)
                return true;
            // Single modulus: measured at batch = 1 (profiles/batch1_r01.txt) the prepared-twiddle
            // kernels win from 2^5 (64-bit) / 2^11 (32-bit) upwards even though they cost one
            // extra launch; below that one generic launch is the whole job.
            if (mod_count == 1 && n_power >= (sizeof(TU) == 8 ? 5 : 11))
                return true;
            // RNS stacks: preparation launch + the transform against one launch of the Barrett kernels -- from one tile of
            // coefficients the fast kernels win (4 polynomials of 2^12 with 4 primes: 13 us against 32 us on the generic
            // kernels, which round 4's threshold of 2^15 coefficients still chose; profiles/r05_small_dropin.txt)
            // (tiny single-modulus rings -- below the sizes above -- keep round 4's threshold of 2^15 coefficients)
            return (static_cast<unsigned long long>(batch_size) << n_power) >= (mod_count > 1 ? (1ull << 12) : (1ull << 15)) &&
                   batch_size >= 2;
        }

        // drop-in calls: eligible AND the per-(device, stream) scratch for the prepared twiddles can be had.  A device
        // without room for it (several streams, nearly full HBM) gets the generic kernels, which need none.
        template <typename TU>
        inline bool lazy_eligible(int n_power, int batch_size, int mod_count, hipStream_t stream)
        {
            if (!lazy_eligible<TU>(n_power, batch_size, mod_count))
                return false;
            const size_t entries = (static_cast<size_t>(mod_count) << n_power) + mod_count;
            const size_t norm_bytes = (sizeof(lazy::NormConst) * static_cast<size_t>(mod_count) + 15u) & ~size_t(15);
            if (host::lazy_workspace(stream, sizeof(lazy::Tw<TU>) * entries + 16 + norm_bytes, true) != nullptr)
                return true;
            if (forced_path() == 3)
                throw std::invalid_argument("fast path unavailable for this call: no device memory for the twiddle scratch");
            return false;
        }

        // single-modulus calls and NTTPlan see their moduli on the host: 64-bit words take 61-bit moduli on the
        // LIMIT = 8 kernels (one range correction per stage) and 62-bit moduli on the LIMIT = 4 kernels (products
        // corrected to [0, 2q)) -- the whole documented domain of the reference (modular_arith.cuh:66-67)
        template <typename TU> inline bool fast_modulus(const Modulus<TU>& m)
        {
            const TU max_bit = (sizeof(TU) == 8) ? TU(62) : TU(lazy::Mod<TU>::MAX_BIT);
            return m.value >= 3 && m.bit <= max_bit;
        }
        template <typename TU> inline int needs_lim(const Modulus<TU>& m)
        {
            if (sizeof(TU) != 8)
                return 0;
            return m.bit == TU(62) ? 4 : (m.bit == TU(61) ? 8 : 0);
        }

        // mods == nullptr: single modulus `m`; else device array of mod_count moduli (+ optional
        // device array of n^-1 values whose Shoup pairs are prepared alongside the twiddles)
        template <typename TU>
        inline kern::LazyArgsT<TU> lazy_args(const void* in, TU* out, const TU* roots, const Modulus<TU>& m,
                                            const Modulus<TU>* mods, int mod_count, const TU* ninv_dev,
                                            int n_power, ReductionPolynomial poly, int batch_size,
                                            hipStream_t stream, const int* mod_order = nullptr,
                                            const TU* ninv_single = nullptr, const host::RnsGuess* guess = nullptr,
                                            unsigned io_flags = 0u, const int* poly_order = nullptr,
                                            const TU* mul_in = nullptr)
        {
            using TW = lazy::Tw<TU>;
            const bool neg = (poly == ReductionPolynomial::X_N_plus);
            int lim = (mods == nullptr) ? needs_lim<TU>(m) : 0;
            const bool inverse = ninv_dev != nullptr || ninv_single != nullptr;
            const int tl = lim ? 12 : host::lazy_tile_log_merge<TU>(n_power, inverse, static_cast<unsigned long long>(batch_size));
            // forward, host-side modulus with 31 q < 2^64: the LIMIT = 31 kernels
            if constexpr (sizeof(TU) == 8)
                if (lim == 0 && mods == nullptr && !inverse && host::lazy_lim31_enabled() &&
                    host::lazy_lim31_modulus(m.value))
                    lim = 31;
            // 32-bit words, host-side modulus below 2^29: the LIMIT = 8 kernels (both directions)
            if constexpr (sizeof(TU) == 4)
                if (mods == nullptr && host::lazy_lim31_enabled() && host::lazy_wide_modulus32(m.value))
                    lim = 8;
            const int perm_tile_log = (n_power >= tl) ? tl : 0;
            const size_t entries = (static_cast<size_t>(mod_count) << n_power) + mod_count;
            // workspace: twiddle pairs | n^-1 pairs | go-flag | per-modulus normalisation constants
            const size_t norm_bytes = (sizeof(lazy::NormConst) * static_cast<size_t>(mod_count) + 15u) & ~size_t(15);
            const size_t tail = 16 + norm_bytes;
            auto* ws = static_cast<TW*>(host::lazy_workspace(stream, sizeof(TW) * entries + tail));
            TW* ws_ninv = ws + (static_cast<size_t>(mod_count) << n_power);
            unsigned char* tail_p = reinterpret_cast<unsigned char*>(ws + entries);
            unsigned* go_flag = mods ? reinterpret_cast<unsigned*>(tail_p) : nullptr;
            auto* norm_arr = mods ? reinterpret_cast<lazy::NormConst*>(tail_p + 16) : nullptr;
            // option lim31 is read ONCE per call: the preparation kernel may name the 31 q family only if
            // run_transform_lazy_rns enqueues it (two reads could disagree when another thread flips the option)
            const bool allow_31q = mods != nullptr && host::lazy_lim31_enabled();
            // drop-in RNS call: the family the host enqueues behind the preparation launch (0: every family) and the
            // fall-back the preparation kernel runs itself when the stack does not fit it (kern::SlowArgs)
            unsigned family = 0u;
            kern::SlowArgs<TU> slow{};
            int perm = perm_tile_log;
            if (mods != nullptr && guess != nullptr)
            {
                family = guess->all_families ? 0u : guess->state;
                if (sizeof(TU) == 8 && (family == kern::GO_LAZY_8Q || family == kern::GO_LAZY_4Q) && perm > 12)
                    perm = 12; // those families run on 4096-coefficient tiles
                slow.in = in;
                slow.out = out;
                slow.mul_in = mul_in;
                slow.poly_order = poly_order;
                slow.polys = static_cast<unsigned long long>(batch_size);
                slow.col_log = -1;
                slow.flags = io_flags & (kern::F_SIGNED_IN | kern::F_SCALE | kern::F_CENTERED);
                slow.inverse = inverse ? 1 : 0;
                // fast-strict: the lazy families must own the call; shadow_generic: the generic kernels behind the call do
                slow.enabled = (forced_path() == 3 || guess->shadow_generic) ? 0 : 1;
            }
            host::launch_prep<TU>(roots, ws, mods, m.value, mod_count, n_power, neg, perm, ninv_dev,
                                  ninv_dev ? ws_ninv : nullptr, go_flag, norm_arr, stream, mod_order, ninv_single,
                                  ninv_dev != nullptr, (mods && guess) ? guess->state_out : nullptr, allow_31q, family,
                                  (mods != nullptr && guess != nullptr) ? &slow : nullptr);
            kern::LazyArgsT<TU> a{};
            a.in = in;
            a.out = out;
            a.tw = ws;
            a.mods = mods;
            a.q = m.value;
            a.q_bit = m.bit;
            a.q_mu = m.mu;
            a.ninv_arr = ninv_dev ? ws_ninv : nullptr;
            a.ninv = TW{0, 0};
            a.go_flag = go_flag;
            a.lim = lim;
            a.host_allow_31q = allow_31q ? 1 : 0;
            a.mod_order = mod_order;
            a.poly_order = poly_order;
            a.mul_in = mul_in;
            a.norm = lazy::norm_const_of(m.value, m.bit);
            a.norm_arr = norm_arr;
            a.total = static_cast<unsigned long long>(batch_size) << n_power;
            a.n = n_power;
            a.poly_shift = n_power;
            a.mod_count = mod_count;
            a.p_lo = 0;
            a.flags = 0u;
            return a;
        }

        // Drop-in RNS calls (moduli in device memory): the ONE lazy family the host predicts for the stack (host::RnsGuess),
        // or -- all_families -- every family behind the exact go-flag state.  The preparation kernel in front of them
        // publishes the flag and is itself the fall-back for a stack the enqueued family cannot serve (kern::SlowArgs):
        // there is no generic launch behind these calls.
        template <typename TU, bool INV>
        inline void run_transform_lazy_rns(const kern::LazyArgsT<TU>& la, unsigned in_flags, unsigned out_flags,
                                           hipStream_t stream, const host::RnsGuess& guess)
        {
            if (guess.all_families || guess.state == kern::GO_LAZY)
                host::run_transform_lazy<TU, INV>(la, in_flags, out_flags, stream);
            if constexpr (sizeof(TU) == 8)
            {
                kern::LazyArgsT<TU> wide = la;
                if constexpr (!INV)
                    if (la.host_allow_31q != 0 && (guess.all_families || guess.state == kern::GO_LAZY_31Q))
                    {
                        wide.lim = 31; // every modulus has 31 q < 2^64 (the reference's pool primes): forward transforms
                        host::run_transform_lazy<TU, INV>(wide, in_flags, out_flags, stream);
                    }
                if (guess.all_families || guess.state == kern::GO_LAZY_8Q)
                {
                    wide.lim = 8; // widest modulus 61 bit
                    host::run_transform_lazy<TU, INV>(wide, in_flags, out_flags, stream);
                }
                if (guess.all_families || guess.state == kern::GO_LAZY_4Q)
                {
                    wide.lim = 4; // widest modulus 62 bit
                    host::run_transform_lazy<TU, INV>(wide, in_flags, out_flags, stream);
                }
            }
        }

        // The preparation kernel's own fall-back (kern::SlowArgs) walks ONE polynomial per block, stage by stage through
        // global memory: milliseconds for FHE-sized calls, but ~40 ms per polynomial of 2^20 and most of a second at 2^24
        // (ADVICE r5).  So when NOTHING is known about a stack (first call with this moduli buffer, or its last call found
        // moduli outside the domain) and the ring is large -- or the call is being captured into a graph, whose replays
        // cannot learn -- the call casts the wide net instead: every lazy family behind the exact go-flag state and the
        // generic kernels behind them, the preparation kernel only classifies.  A few skipped launches (~6 us each) on a
        // call of >= 100 us, once per stack.
        constexpr int RNS_WIDE_NET_MIN_N_POWER = 17;
        inline bool cast_wide_net(host::RnsGuess& guess, int n_power, hipStream_t stream)
        {
            if (!guess.unsure || forced_path() == 3)
                return false;
            bool wide = n_power >= RNS_WIDE_NET_MIN_N_POWER;
            if (!wide)
            {
                hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(stream, &st) != hipSuccess)
                    (void) hipGetLastError();
                wide = st != hipStreamCaptureStatusNone;
            }
            if (wide)
            {
                guess.all_families = true;
                guess.shadow_generic = true;
            }
            return wide;
        }

        // option path = generic-capped (test hook): a device word holding kern::GO_GENERIC, so that the generic kernels of an
        // RNS call run exactly as they do behind a go-flag that names them -- capped grid, blocks walking the tiles.  (No
        // modulus of the documented domain reaches that state any more: 61- / 62-bit primes run the 4 q lazy family.)
        inline const unsigned* zeroed_flag(hipStream_t stream)
        {
            auto* flag = static_cast<unsigned*>(host::lazy_workspace(stream, 16));
            GPUNTT_HIP_CHECK(hipMemsetAsync(flag, 0, 16, stream));
            return flag;
        }

        // ---- PerCoefficient (column-wise) layout ------------------------------------------
        // The matrix is N rows (coefficient index) x batch columns (polynomials), row-major; every
        // column is transformed (reference ForwardCoreTranspose / InverseCoreTranspose,
        // ntt.cu:1554-2074, hosts :2228-2250; checked like test_merge_ntt.cu:343-474).  With a
        // power-of-two batch the element (c, x) sits at flat index (c << log2 batch) | x, i.e. the
        // columns are exactly a STRIDED tile pass over one virtual ring of N * batch coefficients
        // whose stage bits start at bit log2(batch) -- no transpose, rows stay coalesced.
        // Fast kernels for the PerCoefficient layout (round 3), single modulus with lazy headroom: the same strided passes
        // over the virtual ring of N * batch coefficients, run by the lazy-residue strided kernels from the prepared
        // (Shoup) table of the N-ring -- the twiddle index of a stage only looks at the index bits above it, which
        // are row (coefficient) bits here, so the N-ring's stage layout serves unchanged; no permuted stages (there
        // is no contiguous pass).  Returns false when the call must take the generic kernels.
        template <typename TU, bool INV>
        bool run_percoefficient_lazy(const void* in, TU* out, const TU* roots, const Modulus<TU>& modulus, TU ninv,
                                     int n_power, ReductionPolynomial poly, int batch_size, unsigned in_flags,
                                     unsigned out_flags, hipStream_t stream)
        {
            using TW = lazy::Tw<TU>;
            if (batch_size <= 0 || (batch_size & (batch_size - 1)) != 0)
                return false; // (the generic path reports the error)
            if (forced_path() == 1 || !fast_modulus<TU>(modulus) || (INV && ninv >= modulus.value))
                return false;
            const bool wide = needs_lim<TU>(modulus) != 0; // 61- / 62-bit modulus (64-bit words): the 4 q strided kernels
            int log_w = 0;
            while ((1 << log_w) < batch_size)
                log_w++;
            const int nv = n_power + log_w;
            if (nv < 12 || nv > 30 || n_power > host::LAZY_MAX_N_POWER)
                return false;
            host::Plan pl{};
            const int np = (n_power + 7) / 8;
            int top = n_power;
            for (int i = 0; i < np; i++)
            {
                const int k = n_power / np + ((i < n_power % np) ? 1 : 0);
                top -= k;
                pl.pass[pl.count++] = host::Pass{false, k, log_w + top};
                if (12 - k > log_w + top)
                    return false; // a tile row would be wider than the matrix row: small-matrix kernel (generic)
            }
            const size_t entries = (size_t(1) << n_power) + 1;
            auto* ws = static_cast<TW*>(host::lazy_workspace(stream, sizeof(TW) * entries + 16, true));
            if (ws == nullptr)
                return false;
            const bool neg = (poly == ReductionPolynomial::X_N_plus);
            host::launch_prep<TU>(roots, ws, nullptr, modulus.value, 1, n_power, neg, 0, nullptr, nullptr, nullptr, nullptr,
                                  stream, nullptr, INV ? &ninv : nullptr, false);
            kern::LazyArgsT<TU> a{};
            a.tw = ws;
            a.q = modulus.value;
            a.q_bit = modulus.bit;
            a.q_mu = modulus.mu;
            a.ninv = TW{0, 0};
            if (INV)
                a.ninv = TW{ninv, host::shoup_host(ninv, modulus.value)};
            a.norm = lazy::norm_const_of(modulus.value, modulus.bit);
            a.total = 1ull << nv;
            a.n = nv;
            a.poly_shift = nv;
            a.mod_count = 1;
            const void* src = in;
            for (int i = 0; i < pl.count; i++)
            {
                const host::Pass& p = INV ? pl.pass[pl.count - 1 - i] : pl.pass[i];
                kern::LazyArgsT<TU> b = a;
                b.in = src;
                b.out = out;
                b.p_lo = p.p_lo;
                if (i == 0)
                    b.flags |= in_flags;
                if (i == pl.count - 1)
                    b.flags |= out_flags;
                bool launched = false;
                if constexpr (sizeof(TU) == 8)
                    if (wide)
                    {
                        host::launch_pass_lazy_lim<INV, 4>(p, i == 0, i == pl.count - 1, b, stream);
                        launched = true;
                    }
                if (!launched)
                    host::launch_pass_lazy<TU, INV>(p, 12, i == 0, i == pl.count - 1, b, stream);
                src = out;
            }
            return true;
        }

        // PerCoefficient layout with an RNS stack (column c = a polynomial of modulus c % mod_count, table slot and n^-1 of
        // that modulus: reference ForwardCoreTranspose / InverseCoreTranspose, ntt.cu:1693-1835, 1957-2074) on the lazy
        // kernels with PER-LANE moduli (kern::merge_pass_lazy_vq): the lanes of a wave hold different columns, so q, -q, the
        // twiddles and n^-1 are vector operands; one family covers the documented domain (64-bit: the 4 q range, <= 62 bit).
        // The moduli live in device memory: the preparation kernel classifies them, *go_flag_out receives the flag the
        // generic kernels behind this call must test.  false: the call must take the generic kernels alone.
        template <typename TU, bool INV>
        bool run_percoefficient_lazy_rns(const void* in, TU* out, const TU* roots, const Modulus<TU>* mods_dev, int mod_count,
                                         const TU* ninv_dev, int n_power, ReductionPolynomial poly, int batch_size,
                                         unsigned in_flags, unsigned out_flags, hipStream_t stream,
                                         const unsigned** go_flag_out)
        {
            using TW = lazy::Tw<TU>;
            if (batch_size <= 0 || (batch_size & (batch_size - 1)) != 0 || mod_count < 1)
                return false; // (the generic path reports the error)
            if (forced_path() == 1 || forced_path() == 4 || (INV && ninv_dev == nullptr))
                return false;
            int log_w = 0;
            while ((1 << log_w) < batch_size)
                log_w++;
            const int nv = n_power + log_w;
            if (nv < 12 || nv > 30)
                return false;
            host::Plan pl{};
            const int np = (n_power + 7) / 8;
            int top = n_power;
            for (int i = 0; i < np; i++)
            {
                const int k = n_power / np + ((i < n_power % np) ? 1 : 0);
                top -= k;
                pl.pass[pl.count++] = host::Pass{false, k, log_w + top};
                if (12 - k > log_w + top)
                    return false; // a tile row would be wider than the matrix row: small-matrix kernel (generic)
                // A pass of k < 4 stages (n_power 1 .. 3) keeps its register window on tile bits 8 .. 11 while the stage
                // bits start at 12 - k: register bits 0 .. 3 - k are COLUMN bits, a thread holds columns c, c + 256, ...
                // and the kernel picks ONE modulus per thread (pass_body VQ) -- right only when every such column has the
                // modulus of column c, i.e. when mod_count divides 256.  Anything else takes the generic kernels
                if (k < 4 && mod_count > 1 && (256 % mod_count) != 0)
                    return false;
            }
            const size_t entries = (static_cast<size_t>(mod_count) << n_power) + mod_count;
            const size_t norm_bytes = (sizeof(lazy::NormConst) * static_cast<size_t>(mod_count) + 15u) & ~size_t(15);
            auto* ws = static_cast<TW*>(host::lazy_workspace(stream, sizeof(TW) * entries + 16 + norm_bytes, true));
            if (ws == nullptr)
                return false;
            TW* ws_ninv = ws + (static_cast<size_t>(mod_count) << n_power);
            unsigned char* tail_p = reinterpret_cast<unsigned char*>(ws + entries);
            unsigned* go_flag = reinterpret_cast<unsigned*>(tail_p);
            auto* norm_arr = reinterpret_cast<lazy::NormConst*>(tail_p + 16);
            const bool neg = (poly == ReductionPolynomial::X_N_plus);
            // the per-lane-modulus kernels serve every stack of the documented domain; one with a modulus outside it is
            // transformed by the preparation kernel itself, column by column (kern::SlowArgs::col_log) -- nothing behind the call
            kern::SlowArgs<TU> slow{};
            slow.in = in;
            slow.out = out;
            slow.polys = static_cast<unsigned long long>(batch_size);
            slow.col_log = log_w;
            slow.flags = (in_flags | out_flags) & (kern::F_SIGNED_IN | kern::F_SCALE | kern::F_CENTERED);
            slow.inverse = INV ? 1 : 0;
            slow.enabled = forced_path() == 3 ? 0 : 1;
            host::launch_prep<TU>(roots, ws, mods_dev, TU(0), mod_count, n_power, neg, 0, INV ? ninv_dev : nullptr,
                                  INV ? ws_ninv : nullptr, go_flag, norm_arr, stream, nullptr, nullptr, INV, nullptr, false, 0u,
                                  &slow);
            kern::LazyArgsT<TU> a{};
            a.tw = ws;
            a.mods = mods_dev;
            a.ninv = TW{0, 0};
            a.ninv_arr = INV ? ws_ninv : nullptr;
            a.norm_arr = norm_arr;
            a.go_flag = go_flag;
            a.total = 1ull << nv;
            a.n = nv;
            a.poly_shift = nv;
            a.mod_count = mod_count;
            a.col_log = log_w;
            a.mod_shift = n_power;
            const void* src = in;
            for (int i = 0; i < pl.count; i++)
            {
                const host::Pass& p = INV ? pl.pass[pl.count - 1 - i] : pl.pass[i];
                kern::LazyArgsT<TU> b = a;
                b.in = src;
                b.out = out;
                b.p_lo = p.p_lo;
                if (i == 0)
                    b.flags |= in_flags;
                if (i == pl.count - 1)
                    b.flags |= out_flags;
                host::launch_pass_lazy_vq<TU, INV>(p, i == 0, i == pl.count - 1, b, stream);
                src = out;
            }
            *go_flag_out = go_flag;
            return true;
        }

        template <typename TU, bool INV>
        void run_percoefficient(kern::PassArgs<TU> a, int n_power, int batch_size, unsigned in_flags,
                                unsigned out_flags, hipStream_t stream)
        {
            if (batch_size <= 0)
                return;
            if ((batch_size & (batch_size - 1)) != 0)
                throw std::invalid_argument("PerCoefficient batch_size must be a power of two!");
            int log_w = 0;
            while ((1 << log_w) < batch_size)
                log_w++;
            const int nv = n_power + log_w; // virtual ring
            a.n = nv;
            a.poly_shift = nv;
            a.root_shift = -1;
            a.total = 1ull << nv;
            if (a.mods != nullptr && a.mod_count > 1)
            {
                // RNS: column c is a polynomial of modulus c % mod_count with its own table slot and n^-1
                // (the intent of reference ForwardCoreTranspose / InverseCoreTranspose, ntt.cu:1693-1835,
                // 1957-2074: mod_index = batch_index % mod_count, roots at mod_index << log_row)
                a.flags |= kern::F_MULTI | kern::F_COLMOD;
                a.n2_log = log_w;
                a.root_shift = n_power;
            }
            // stages split into strided passes of <= 8, highest first (forward order)
            host::Plan pl{};
            const int np = (n_power + 7) / 8;
            int top = n_power;
            bool fits = (nv >= kern::TL);
            for (int i = 0; i < np; i++)
            {
                const int k = n_power / np + ((i < n_power % np) ? 1 : 0);
                top -= k;
                pl.pass[pl.count++] = host::Pass{false, k, log_w + top};
                if (kern::TL - k > log_w + top)
                    fits = false; // a tile row would be wider than the matrix row
            }
            if (!fits)
            {
                a.flags |= in_flags | out_flags;
                host::launch_column_small<TU, INV>(a, n_power, log_w, stream);
                return;
            }
            const void* src = a.in;
            for (int i = 0; i < pl.count; i++)
            {
                kern::PassArgs<TU> b = a;
                const host::Pass& p = INV ? pl.pass[pl.count - 1 - i] : pl.pass[i];
                b.in = src;
                b.p_lo = p.p_lo;
                if (i == 0)
                    b.flags |= in_flags;
                if (i == pl.count - 1)
                    b.flags |= out_flags;
                host::launch_pass<TU, INV>(p, b, stream);
                src = a.out;
            }
        }

        template <typename TU> inline void set_multi(kern::PassArgs<TU>& a)
        {
            if (a.mods != nullptr && a.mod_count > 1 && a.poly_shift < kern::TL)
                a.flags |= kern::F_MULTI;
        }
    } // namespace

    // ---------------------------------------------------------------- single modulus ----
    template <typename T>
    __host__ void GPU_NTT(T* device_in, typename std::make_unsigned<T>::type* device_out,
                          Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                          Modulus<typename std::make_unsigned<T>::type> modulus,
                          ntt_configuration<typename std::make_unsigned<T>::type> cfg,
                          int batch_size)
    {
        host::WorkspaceScope ws_scope; // scratch lock held until the last launch of this call
        using TU = typename std::make_unsigned<T>::type;
        check_layout_and_range(cfg.ntt_layout, cfg.n_power);
        const unsigned in_flags = std::is_signed<T>::value ? kern::F_SIGNED_IN : 0u;
        if (cfg.ntt_layout == PerCoefficient)
        {
            if (run_percoefficient_lazy<TU, false>(device_in, device_out, root_of_unity_table, modulus, TU(0), cfg.n_power,
                                                   cfg.reduction_poly, batch_size, in_flags, 0u, cfg.stream))
                return;
            if (forced_path() == 3)
                throw std::invalid_argument("fast path unavailable for this call (path = fast-strict)");
            kern::PassArgs<TU> a = base_args<TU>(device_in, device_out, root_of_unity_table, cfg.n_power,
                                                 cfg.reduction_poly, batch_size);
            a.mod = modulus;
            run_percoefficient<TU, false>(a, cfg.n_power, batch_size, in_flags, 0u, cfg.stream);
            return;
        }
        if (batch_size > 0 && fast_modulus<TU>(modulus) && lazy_eligible<TU>(cfg.n_power, batch_size, 1, cfg.stream))
        {
            kern::LazyArgsT<TU> la =
                lazy_args<TU>(device_in, device_out, root_of_unity_table, modulus, nullptr, 1, nullptr,
                              cfg.n_power, cfg.reduction_poly, batch_size, cfg.stream);
            host::run_transform_lazy<TU, false>(la, in_flags, 0u, cfg.stream);
            return;
        }
        kern::PassArgs<TU> a = base_args<TU>(device_in, device_out, root_of_unity_table, cfg.n_power,
                                             cfg.reduction_poly, batch_size);
        a.mod = modulus;
        host::run_transform<TU, false>(a, in_flags, 0u, cfg.stream);
    }

    template <typename T>
    __host__ void GPU_INTT(typename std::make_unsigned<T>::type* device_in, T* device_out,
                           Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                           Modulus<typename std::make_unsigned<T>::type> modulus,
                           ntt_configuration<typename std::make_unsigned<T>::type> cfg,
                           int batch_size)
    {
        host::WorkspaceScope ws_scope; // scratch lock held until the last launch of this call
        using TU = typename std::make_unsigned<T>::type;
        check_layout_and_range(cfg.ntt_layout, cfg.n_power);
        const unsigned out_flags =
            kern::F_SCALE | (std::is_signed<T>::value ? kern::F_CENTERED : 0u);
        if (cfg.ntt_layout == PerCoefficient)
        {
            if (run_percoefficient_lazy<TU, true>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table, modulus,
                                                  cfg.mod_inverse, cfg.n_power, cfg.reduction_poly, batch_size, 0u, out_flags,
                                                  cfg.stream))
                return;
            if (forced_path() == 3)
                throw std::invalid_argument("fast path unavailable for this call (path = fast-strict)");
            kern::PassArgs<TU> a =
                base_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table,
                              cfg.n_power, cfg.reduction_poly, batch_size);
            a.mod = modulus;
            a.ninv = cfg.mod_inverse;
            run_percoefficient<TU, true>(a, cfg.n_power, batch_size, 0u, out_flags, cfg.stream);
            return;
        }
        if (batch_size > 0 && fast_modulus<TU>(modulus) && lazy_eligible<TU>(cfg.n_power, batch_size, 1, cfg.stream) &&
            cfg.mod_inverse < modulus.value)
        {
            kern::LazyArgsT<TU> la =
                lazy_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table, modulus,
                              nullptr, 1, nullptr, cfg.n_power, cfg.reduction_poly, batch_size, cfg.stream,
                              nullptr, &cfg.mod_inverse);
            la.ninv = lazy::Tw<TU>{cfg.mod_inverse, host::shoup_host(cfg.mod_inverse, modulus.value)};
            host::run_transform_lazy<TU, true>(la, 0u, out_flags, cfg.stream);
            return;
        }
        kern::PassArgs<TU> a =
            base_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table,
                          cfg.n_power, cfg.reduction_poly, batch_size);
        a.mod = modulus;
        a.ninv = cfg.mod_inverse;
        host::run_transform<TU, true>(a, 0u, out_flags, cfg.stream);
    }

    template <typename T>
    __host__ void GPU_NTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                  Modulus<T> modulus, ntt_configuration<T> cfg, int batch_size)
    {
        GPU_NTT<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size);
    }

    template <typename T>
    __host__ void GPU_INTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                   Modulus<T> modulus, ntt_configuration<T> cfg, int batch_size)
    {
        GPU_INTT<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size);
    }

    // --------------------------------------------------------------------------- RNS ----
    template <typename T>
    __host__ void GPU_NTT(T* device_in, typename std::make_unsigned<T>::type* device_out,
                          Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                          Modulus<typename std::make_unsigned<T>::type>* modulus,
                          ntt_rns_configuration<typename std::make_unsigned<T>::type> cfg,
                          int batch_size, int mod_count)
    {
        host::WorkspaceScope ws_scope; // scratch lock held until the last launch of this call
        using TU = typename std::make_unsigned<T>::type;
        check_layout_and_range(cfg.ntt_layout, cfg.n_power);
        if (mod_count <= 0 || modulus == nullptr)
            throw std::invalid_argument("Invalid mod_count!");
        const unsigned in_flags = std::is_signed<T>::value ? kern::F_SIGNED_IN : 0u;
        const unsigned* skip_flag = nullptr;
        if (cfg.ntt_layout == PerCoefficient)
        {
            kern::PassArgs<TU> a = base_args<TU>(device_in, device_out, root_of_unity_table, cfg.n_power,
                                                 cfg.reduction_poly, batch_size);
            a.mods = modulus;
            a.mod_count = mod_count;
            const bool lazy = run_percoefficient_lazy_rns<TU, false>(device_in, device_out, root_of_unity_table, modulus,
                                                                     mod_count, nullptr, cfg.n_power, cfg.reduction_poly,
                                                                     batch_size, in_flags, 0u, cfg.stream, &skip_flag);
            if (lazy)
                return; // (a stack outside the domain: the preparation kernel's own fall-back)
            if (forced_path() == 3)
                throw std::invalid_argument("fast path unavailable for this call (path = fast-strict)");
            run_percoefficient<TU, false>(a, cfg.n_power, batch_size, in_flags, 0u, cfg.stream);
            return;
        }
        if (batch_size > 0 && forced_path() == 4)
            skip_flag = zeroed_flag(cfg.stream); // test hook: the generic kernels as they run behind a go-flag that names them
        else if (batch_size > 0 && lazy_eligible<TU>(cfg.n_power, batch_size, mod_count, cfg.stream))
        {
            // preparation (classifies the stack; its own fall-back when the stack does not fit) + the predicted lazy family
            host::RnsGuess guess = host::rns_guess(modulus, mod_count, static_cast<int>(sizeof(TU)), false);
            const bool wide_net = cast_wide_net(guess, cfg.n_power, cfg.stream);
            kern::LazyArgsT<TU> la =
                lazy_args<TU>(device_in, device_out, root_of_unity_table, Modulus<TU>(), modulus, mod_count,
                              nullptr, cfg.n_power, cfg.reduction_poly, batch_size, cfg.stream, nullptr, nullptr,
                              &guess, in_flags);
            run_transform_lazy_rns<TU, false>(la, in_flags, 0u, cfg.stream, guess);
            if (!wide_net)
                return;
            skip_flag = la.go_flag; // the generic kernels below run iff the preparation kernel published GO_GENERIC
        }
        kern::PassArgs<TU> a = base_args<TU>(device_in, device_out, root_of_unity_table, cfg.n_power,
                                             cfg.reduction_poly, batch_size);
        a.mods = modulus;
        a.mod_count = mod_count;
        a.skip_flag = skip_flag;
        set_multi(a);
        host::run_transform<TU, false>(a, in_flags, 0u, cfg.stream);
    }

    template <typename T>
    __host__ void GPU_INTT(typename std::make_unsigned<T>::type* device_in, T* device_out,
                           Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                           Modulus<typename std::make_unsigned<T>::type>* modulus,
                           ntt_rns_configuration<typename std::make_unsigned<T>::type> cfg,
                           int batch_size, int mod_count)
    {
        host::WorkspaceScope ws_scope; // scratch lock held until the last launch of this call
        using TU = typename std::make_unsigned<T>::type;
        check_layout_and_range(cfg.ntt_layout, cfg.n_power);
        if (mod_count <= 0 || modulus == nullptr)
            throw std::invalid_argument("Invalid mod_count!");
        const unsigned out_flags =
            kern::F_SCALE | (std::is_signed<T>::value ? kern::F_CENTERED : 0u);
        const unsigned* skip_flag = nullptr;
        if (cfg.ntt_layout == PerCoefficient)
        {
            kern::PassArgs<TU> a =
                base_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table,
                              cfg.n_power, cfg.reduction_poly, batch_size);
            a.mods = modulus;
            a.mod_count = mod_count;
            a.ninv_arr = cfg.mod_inverse;
            const bool lazy = run_percoefficient_lazy_rns<TU, true>(device_in, reinterpret_cast<TU*>(device_out),
                                                                    root_of_unity_table, modulus, mod_count, cfg.mod_inverse,
                                                                    cfg.n_power, cfg.reduction_poly, batch_size, 0u, out_flags,
                                                                    cfg.stream, &skip_flag);
            if (lazy)
                return;
            if (forced_path() == 3)
                throw std::invalid_argument("fast path unavailable for this call (path = fast-strict)");
            run_percoefficient<TU, true>(a, cfg.n_power, batch_size, 0u, out_flags, cfg.stream);
            return;
        }
        if (batch_size > 0 && forced_path() == 4)
            skip_flag = zeroed_flag(cfg.stream); // test hook: the generic kernels as they run behind a go-flag that names them
        else if (batch_size > 0 && cfg.mod_inverse != nullptr &&
            lazy_eligible<TU>(cfg.n_power, batch_size, mod_count, cfg.stream))
        {
            host::RnsGuess guess = host::rns_guess(modulus, mod_count, static_cast<int>(sizeof(TU)), true);
            const bool wide_net = cast_wide_net(guess, cfg.n_power, cfg.stream);
            kern::LazyArgsT<TU> la =
                lazy_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table,
                              Modulus<TU>(), modulus, mod_count, cfg.mod_inverse, cfg.n_power,
                              cfg.reduction_poly, batch_size, cfg.stream, nullptr, nullptr, &guess, out_flags);
            run_transform_lazy_rns<TU, true>(la, 0u, out_flags, cfg.stream, guess);
            if (!wide_net)
                return;
            skip_flag = la.go_flag; // the generic kernels below run iff the preparation kernel published GO_GENERIC
        }
        kern::PassArgs<TU> a =
            base_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table,
                          cfg.n_power, cfg.reduction_poly, batch_size);
        a.mods = modulus;
        a.mod_count = mod_count;
        a.ninv_arr = cfg.mod_inverse;
        a.skip_flag = skip_flag;
        set_multi(a);
        host::run_transform<TU, true>(a, 0u, out_flags, cfg.stream);
    }

    template <typename T>
    __host__ void GPU_NTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                  Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                  int batch_size, int mod_count)
    {
        GPU_NTT<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size,
                   mod_count);
    }

    template <typename T>
    __host__ void GPU_INTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                   Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                   int batch_size, int mod_count)
    {
        GPU_INTT<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size,
                    mod_count);
    }

    // ------------------------------------------------- pointwise product / PolyMul ----
    namespace kern
    {
        // out[i] = a[i] * b[i] mod q_p, p = (i >> n) % mod_count; Barrett with the caller's
        // {value, bit, mu} (reference OPERATOR_GPU::mult, modular_arith.cuh:312-339); HBM-bound:
        // 16-byte accesses, grid-stride
        template <typename T, int V>
        __global__ __launch_bounds__(256) void pointwise_mul(const T* a, const T* b, T* out,
                                                             const Modulus<T>* __restrict__ mods, Modulus<T> mod,
                                                             int mod_count, int n, unsigned long long total,
                                                             const unsigned* __restrict__ skip_flag, unsigned skip_value)
        {
            // GPU_PolyMul, RNS form: the fast forward kernels already multiplied on their final store (same test as the
            // generic transform in front of this launch, merge_kernels.hpp: PassArgs::skip_value)
            if (skip_flag != nullptr)
            {
                const unsigned st = *skip_flag;
                if (skip_value == 0u ? (st != 0u) : (st == skip_value))
                    return;
            }
            // V elements per access: 16 bytes when the buffers are 16-byte aligned, else 1 element
            struct alignas(V * sizeof(T)) Vec
            {
                T x[V];
            };
            const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 256ull * V;
            for (unsigned long long i = (blockIdx.x * 256ull + threadIdx.x) * V; i < total; i += stride)
            {
                Modulus<T> md = mod;
                if (mods != nullptr)
                    md = mods[(i >> n) % static_cast<unsigned>(mod_count)];
                const dev::ModCtx<T> m{md.value, md.bit, md.mu};
                const Vec va = *reinterpret_cast<const Vec*>(a + i);
                const Vec vb = *reinterpret_cast<const Vec*>(b + i);
                Vec vo;
#pragma unroll
                for (int k = 0; k < V; k++)
                    vo.x[k] = m.mul(va.x[k], vb.x[k]);
                *reinterpret_cast<Vec*>(out + i) = vo;
            }
        }
    } // namespace kern

    namespace
    {
        template <typename T>
        void pointwise_launch(T* a, T* b, T* out, const Modulus<T>* mods, Modulus<T> mod, int mod_count,
                              int n_power, int batch_size, hipStream_t stream, const unsigned* skip_flag = nullptr,
                              unsigned skip_value = 0u)
        {
            if (n_power <= 0 || n_power >= 29)
                throw std::invalid_argument("Invalid n_power range!");
            if (batch_size <= 0)
                return;
            const unsigned long long total = static_cast<unsigned long long>(batch_size) << n_power;
            constexpr int VW = 16 / sizeof(T);
            // a 16-byte group must stay inside one polynomial (one modulus) and be aligned
            const bool wide = (n_power >= (sizeof(T) == 8 ? 1 : 2)) &&
                              ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                                reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
            const unsigned long long per_block = 256ull * (wide ? VW : 1);
            unsigned long long blocks = (total + per_block - 1) / per_block;
            if (blocks > 16384)
                blocks = 16384; // 64 blocks per CU, grid-stride beyond
            if (wide)
                GPUNTT_LAUNCH((kern::pointwise_mul<T, VW>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                                   stream, a, b, out, mods, mod, mod_count, n_power, total, skip_flag, skip_value);
            else
                GPUNTT_LAUNCH((kern::pointwise_mul<T, 1>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                                   stream, a, b, out, mods, mod, mod_count, n_power, total, skip_flag, skip_value);
            GPUNTT_HIP_CHECK(hipGetLastError());
        }
    } // namespace

    template <typename T>
    __host__ void GPU_PointwiseMul(T* device_a, T* device_b, T* device_out, Modulus<T> modulus, int n_power,
                                   int batch_size, stream_t stream)
    {
        pointwise_launch<T>(device_a, device_b, device_out, nullptr, modulus, 1, n_power, batch_size, stream);
    }
    template <typename T>
    __host__ void GPU_PointwiseMul(T* device_a, T* device_b, T* device_out, Modulus<T>* modulus, int n_power,
                                   int batch_size, int mod_count, stream_t stream)
    {
        if (mod_count <= 0 || modulus == nullptr)
            throw std::invalid_argument("Invalid mod_count!");
        pointwise_launch<T>(device_a, device_b, device_out, modulus, Modulus<T>(), mod_count, n_power, batch_size,
                            stream);
    }

    // out = INTT(NTT(a) . NTT(b)).  The first operand is transformed in place; the second one's forward
    // transform multiplies by it on its final store (fast kernels, LazyArgsT::mul_in), so the
    // pointwise step costs one extra read instead of a kernel of its own; the generic kernels (wide
    // moduli, tiny rings) are followed by pointwise_mul instead.  `out` may alias either operand:
    // the operand that shares its buffer with `out` is the one transformed last.
    template <typename T>
    __host__ void GPU_PolyMul(T* device_a, T* device_b, T* device_out, Root<T>* forward_table,
                              Root<T>* inverse_table, Modulus<T> modulus, ntt_configuration<T> cfg,
                              int batch_size)
    {
        host::WorkspaceScope ws_scope; // scratch lock held until the last launch of this call
        check_layout_and_range(PerPolynomial, cfg.n_power);
        if (batch_size <= 0)
            return;
        ntt_configuration<T> f = cfg;
        f.ntt_type = FORWARD;
        f.ntt_layout = PerPolynomial;
        if (device_a == device_b)
        {
            // squaring: one forward transform, pointwise square, inverse (transforming the shared
            // buffer twice would compute INTT(NTT(NTT(a)) . NTT(a)))
            GPU_NTT<T>(device_a, device_a, forward_table, modulus, f, batch_size);
            pointwise_launch<T>(device_a, device_a, device_out, nullptr, modulus, 1, cfg.n_power, batch_size,
                                cfg.stream);
            f.ntt_type = INVERSE;
            GPU_INTT<T>(device_out, device_out, inverse_table, modulus, f, batch_size);
            return;
        }
        T* first = (device_out == device_a) ? device_b : device_a;  // transformed in place
        T* second = (device_out == device_a) ? device_a : device_b; // transformed into device_out
        GPU_NTT<T>(first, first, forward_table, modulus, f, batch_size);
        if (modulus.value >= 3 && modulus.bit <= T(lazy::Mod<T>::MAX_BIT) &&
            lazy_eligible<T>(cfg.n_power, batch_size, 1, cfg.stream))
        {
            kern::LazyArgsT<T> la = lazy_args<T>(second, device_out, forward_table, modulus, nullptr, 1, nullptr,
                                                 cfg.n_power, cfg.reduction_poly, batch_size, cfg.stream);
            la.mul_in = first;
            host::run_transform_lazy<T, false>(la, 0u, 0u, cfg.stream);
        }
        else
        {
            GPU_NTT<T>(second, device_out, forward_table, modulus, f, batch_size);
            pointwise_launch<T>(first, device_out, device_out, nullptr, modulus, 1, cfg.n_power, batch_size,
                                cfg.stream);
        }
        f.ntt_type = INVERSE;
        GPU_INTT<T>(device_out, device_out, inverse_table, modulus, f, batch_size);
    }
    template <typename T>
    __host__ void GPU_PolyMul(T* device_a, T* device_b, T* device_out, Root<T>* forward_table,
                              Root<T>* inverse_table, Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                              int batch_size, int mod_count)
    {
        host::WorkspaceScope ws_scope; // scratch lock held until the last launch of this call
        check_layout_and_range(PerPolynomial, cfg.n_power);
        if (mod_count <= 0 || modulus == nullptr)
            throw std::invalid_argument("Invalid mod_count!");
        if (batch_size <= 0)
            return;
        ntt_rns_configuration<T> f = cfg;
        f.ntt_type = FORWARD;
        f.ntt_layout = PerPolynomial;
        if (device_a == device_b)
        {
            GPU_NTT<T>(device_a, device_a, forward_table, modulus, f, batch_size, mod_count);
            pointwise_launch<T>(device_a, device_a, device_out, modulus, Modulus<T>(), mod_count, cfg.n_power,
                                batch_size, cfg.stream);
            f.ntt_type = INVERSE;
            GPU_INTT<T>(device_out, device_out, inverse_table, modulus, f, batch_size, mod_count);
            return;
        }
        T* first = (device_out == device_a) ? device_b : device_a;
        T* second = (device_out == device_a) ? device_a : device_b;
        GPU_NTT<T>(first, first, forward_table, modulus, f, batch_size, mod_count);
        // moduli live on the device: the predicted lazy family multiplies on its final store; a stack it cannot serve is
        // transformed AND multiplied by the preparation kernel's own fall-back (kern::SlowArgs::mul_in)
        if (lazy_eligible<T>(cfg.n_power, batch_size, mod_count, cfg.stream))
        {
            const host::RnsGuess guess = host::rns_guess(modulus, mod_count, static_cast<int>(sizeof(T)), false);
            kern::LazyArgsT<T> la = lazy_args<T>(second, device_out, forward_table, Modulus<T>(), modulus, mod_count,
                                                 nullptr, cfg.n_power, cfg.reduction_poly, batch_size, cfg.stream, nullptr,
                                                 nullptr, &guess, 0u, nullptr, first);
            run_transform_lazy_rns<T, false>(la, 0u, 0u, cfg.stream, guess);
        }
        else
        {
            kern::PassArgs<T> a = base_args<T>(second, device_out, forward_table, cfg.n_power, cfg.reduction_poly, batch_size);
            a.mods = modulus;
            a.mod_count = mod_count;
            set_multi(a);
            host::run_transform<T, false>(a, 0u, 0u, cfg.stream);
            pointwise_launch<T>(first, device_out, device_out, modulus, Modulus<T>(), mod_count, cfg.n_power, batch_size,
                                cfg.stream);
        }
        f.ntt_type = INVERSE;
        GPU_INTT<T>(device_out, device_out, inverse_table, modulus, f, batch_size, mod_count);
    }

#define GPUNTT_INST_POLYMUL(T)                                                                                    \
    template __host__ void GPU_PolyMul<T>(T*, T*, T*, Root<T>*, Root<T>*, Modulus<T>, ntt_configuration<T>, int);  \
    template __host__ void GPU_PolyMul<T>(T*, T*, T*, Root<T>*, Root<T>*, Modulus<T>*, ntt_rns_configuration<T>,   \
                                          int, int);                                                              \
    template __host__ void GPU_PointwiseMul<T>(T*, T*, T*, Modulus<T>, int, int, stream_t);                        \
    template __host__ void GPU_PointwiseMul<T>(T*, T*, T*, Modulus<T>*, int, int, int, stream_t);
    GPUNTT_INST_POLYMUL(Data32)
    GPUNTT_INST_POLYMUL(Data64)
#undef GPUNTT_INST_POLYMUL

    // --------------------------------------------------------------------- NTTPlan ----
    // Prepared transform (extension): everything lazy_args() derives per call is derived once.
    template <typename T> struct NTTPlan<T>::Impl
    {
        const T* table = nullptr;
        std::vector<Modulus<T>> moduli;
        std::vector<T> ninv;
        int mod_count = 1, n = 0, tile_log = 12, lim = 0;
        ReductionPolynomial poly = ReductionPolynomial::X_N_minus;
        bool inverse = false, fast = false, owns_ws = false;
        unsigned char* ws = nullptr;
        // workspace layout
        lazy::Tw<T>* tw = nullptr;
        lazy::Tw<T>* ninv_pairs = nullptr;
        unsigned* go_flag = nullptr;
        lazy::NormConst* norm_arr = nullptr;
        Modulus<T>* mods_dev = nullptr;
        T* ninv_dev = nullptr;
    };

    namespace
    {
        inline size_t up16(size_t v) { return (v + 15u) & ~size_t(15); }
        template <typename T> struct PlanLayout
        {
            size_t tw, ninv_pairs, go_flag, norm, mods, ninv, total;
            PlanLayout(int n_power, int mod_count)
            {
                const size_t mc = static_cast<size_t>(mod_count);
                size_t off = 0;
                tw = off;
                off += up16(sizeof(lazy::Tw<T>) * (mc << n_power));
                ninv_pairs = off;
                off += up16(sizeof(lazy::Tw<T>) * mc);
                go_flag = off;
                off += 16;
                norm = off;
                off += up16(sizeof(lazy::NormConst) * mc);
                mods = off;
                off += up16(sizeof(Modulus<T>) * mc);
                ninv = off;
                off += up16(sizeof(T) * mc);
                total = off;
            }
        };
    } // namespace

    template <typename T> size_t NTTPlan<T>::workspace_bytes(int n_power, int mod_count)
    {
        if (n_power <= 0 || n_power >= 29 || mod_count <= 0)
            throw std::invalid_argument("Invalid n_power range!");
        return PlanLayout<T>(n_power, mod_count).total;
    }

    template <typename T>
    NTTPlan<T>::NTTPlan(const Root<T>* table_device, const Modulus<T>* moduli_host, int mod_count, int n_power,
                        ReductionPolynomial reduction_poly, type ntt_type, const Ninverse<T>* mod_inverse_host,
                        int batch_hint, stream_t stream, void* workspace_device)
        : p_(nullptr)
    {
        if (n_power <= 0 || n_power >= 29)
            throw std::invalid_argument("Invalid n_power range!");
        if (mod_count <= 0 || moduli_host == nullptr || table_device == nullptr)
            throw std::invalid_argument("Invalid mod_count!");
        if (ntt_type != FORWARD && ntt_type != INVERSE)
            throw std::invalid_argument("Invalid ntt_type!");
        const bool inverse = (ntt_type == INVERSE);
        if (inverse && mod_inverse_host == nullptr)
            throw std::invalid_argument("Invalid mod_inverse!");
        Impl* p = new Impl();
        p_ = p;
        try
        {
            p->table = table_device;
            p->moduli.assign(moduli_host, moduli_host + mod_count);
            if (inverse)
                p->ninv.assign(mod_inverse_host, mod_inverse_host + mod_count);
            p->mod_count = mod_count;
            p->n = n_power;
            p->poly = reduction_poly;
            p->inverse = inverse;
            if (batch_hint < 1)
                batch_hint = 1;
            bool fast = lazy_eligible<T>(n_power, batch_hint > 1 ? batch_hint : 2, mod_count);
            for (int i = 0; i < mod_count; i++)
            {
                const Modulus<T>& m = p->moduli[i];
                if (!fast_modulus<T>(m))
                    fast = false;
                const int l = needs_lim<T>(m); // the widest modulus decides the lazy range of the whole stack
                if (l != 0 && (p->lim == 0 || l < p->lim))
                    p->lim = l;
                if (inverse && p->ninv[i] >= m.value)
                    fast = false;
            }
            p->fast = fast;
            p->tile_log = p->lim ? 12 : host::lazy_tile_log_merge<T>(n_power, inverse, static_cast<unsigned long long>(batch_hint));
            if constexpr (sizeof(T) == 8)
            {
                // forward plans whose every modulus has 31 q < 2^64: the LIMIT = 31 kernels
                bool wide = fast && p->lim == 0 && !inverse && host::lazy_lim31_enabled();
                for (int i = 0; wide && i < mod_count; i++)
                    wide = host::lazy_lim31_modulus(p->moduli[i].value);
                if (wide)
                    p->lim = 31;
            }
            else
            {
                bool wide = fast && host::lazy_lim31_enabled();
                for (int i = 0; wide && i < mod_count; i++)
                    wide = host::lazy_wide_modulus32(p->moduli[i].value);
                if (wide)
                    p->lim = 8;
            }
            const PlanLayout<T> lay(n_power, mod_count);
            if (workspace_device != nullptr)
                p->ws = static_cast<unsigned char*>(workspace_device);
            else
            {
                void* mem = nullptr;
                GPUNTT_HIP_CHECK(hipMalloc(&mem, lay.total));
                p->ws = static_cast<unsigned char*>(mem);
                p->owns_ws = true;
            }
            p->tw = reinterpret_cast<lazy::Tw<T>*>(p->ws + lay.tw);
            p->ninv_pairs = reinterpret_cast<lazy::Tw<T>*>(p->ws + lay.ninv_pairs);
            p->go_flag = reinterpret_cast<unsigned*>(p->ws + lay.go_flag);
            p->norm_arr = reinterpret_cast<lazy::NormConst*>(p->ws + lay.norm);
            p->mods_dev = reinterpret_cast<Modulus<T>*>(p->ws + lay.mods);
            p->ninv_dev = reinterpret_cast<T*>(p->ws + lay.ninv);
            // device copies of the moduli / n^-1 values (the host vectors live as long as the plan)
            GPUNTT_HIP_CHECK(hipMemcpyAsync(p->mods_dev, p->moduli.data(), sizeof(Modulus<T>) * mod_count,
                                            hipMemcpyHostToDevice, stream));
            if (inverse)
                GPUNTT_HIP_CHECK(hipMemcpyAsync(p->ninv_dev, p->ninv.data(), sizeof(T) * mod_count,
                                                hipMemcpyHostToDevice, stream));
            if (fast)
            {
                const bool neg = (reduction_poly == ReductionPolynomial::X_N_plus);
                const int perm_tile_log = (n_power >= p->tile_log) ? p->tile_log : 0;
                if (mod_count == 1)
                    host::launch_prep<T>(table_device, p->tw, nullptr, p->moduli[0].value, 1, n_power, neg,
                                         perm_tile_log, nullptr, nullptr, nullptr, nullptr, stream, nullptr,
                                         inverse ? &p->ninv[0] : nullptr, false);
                else
                    host::launch_prep<T>(table_device, p->tw, p->mods_dev, T(0), mod_count, n_power, neg,
                                         perm_tile_log, inverse ? p->ninv_dev : nullptr,
                                         inverse ? p->ninv_pairs : nullptr, p->go_flag, p->norm_arr, stream,
                                         nullptr, nullptr, inverse);
            }
            // the plan is complete when the constructor returns: execute() may run on ANY stream without an
            // external dependency on the construction stream (one host wait, once per plan)
            GPUNTT_HIP_CHECK(hipStreamSynchronize(stream));
        }
        catch (...)
        {
            if (p->owns_ws && p->ws != nullptr)
                (void) hipFree(p->ws);
            delete p;
            p_ = nullptr;
            throw;
        }
    }

    template <typename T> NTTPlan<T>::~NTTPlan()
    {
        if (p_ != nullptr)
        {
            if (p_->owns_ws && p_->ws != nullptr)
                (void) hipFree(p_->ws);
            delete p_;
        }
    }

    template <typename T> bool NTTPlan<T>::fast_path() const { return p_->fast; }

    template <typename T>
    void NTTPlan<T>::execute(const void* device_in, void* device_out, int batch_size, stream_t stream,
                             bool io_signed) const
    {
        const Impl& p = *p_;
        if (batch_size <= 0)
            return;
        if (device_in == nullptr || device_out == nullptr)
            throw std::invalid_argument("null pointer argument");
        const unsigned in_flags = (!p.inverse && io_signed) ? kern::F_SIGNED_IN : 0u;
        const unsigned out_flags = p.inverse ? (kern::F_SCALE | (io_signed ? kern::F_CENTERED : 0u)) : 0u;
        T* out = static_cast<T*>(device_out);
        if (p.fast)
        {
            kern::LazyArgsT<T> a{};
            a.in = device_in;
            a.out = out;
            a.tw = p.tw;
            const Modulus<T>& m0 = p.moduli[0];
            if (p.mod_count > 1)
            {
                a.mods = p.mods_dev;
                a.norm_arr = p.norm_arr;
                a.ninv_arr = p.inverse ? p.ninv_pairs : nullptr;
            }
            else
            {
                a.q = m0.value;
                a.q_bit = m0.bit;
                a.q_mu = m0.mu;
                a.norm = lazy::norm_const_of(m0.value, m0.bit);
                if (p.inverse)
                    a.ninv = lazy::Tw<T>{p.ninv[0], host::shoup_host(p.ninv[0], m0.value)};
            }
            a.total = static_cast<unsigned long long>(batch_size) << p.n;
            a.n = p.n;
            a.poly_shift = p.n;
            a.mod_count = p.mod_count;
            a.lim = p.lim;
            if (p.inverse)
                host::run_transform_lazy<T, true>(a, in_flags, out_flags, stream, p.tile_log);
            else
                host::run_transform_lazy<T, false>(a, in_flags, out_flags, stream, p.tile_log);
            return;
        }
        kern::PassArgs<T> a = base_args<T>(device_in, out, p.table, p.n, p.poly, batch_size);
        if (p.mod_count > 1)
        {
            a.mods = p.mods_dev;
            a.mod_count = p.mod_count;
            a.ninv_arr = p.inverse ? p.ninv_dev : nullptr;
            set_multi(a);
        }
        else
        {
            a.mod = p.moduli[0];
            a.ninv = p.inverse ? p.ninv[0] : T(0);
        }
        if (p.inverse)
            host::run_transform<T, true>(a, in_flags, out_flags, stream);
        else
            host::run_transform<T, false>(a, in_flags, out_flags, stream);
    }

    template class NTTPlan<Data32>;
    template class NTTPlan<Data64>;

    void GPU_NTT_ReleaseWorkspaces() { host::release_workspaces(); }
    bool GPU_NTT_SetOption(const char* name, const char* value) { return host::set_option(name, value); }

    // ---------------------------------------------------------------- ordered RNS ----
    namespace
    {
        template <typename T>
        void ordered_run(T* device_in, T* device_out, Root<T>* table, Modulus<T>* modulus,
                         const ntt_rns_configuration<T>& cfg, int batch_size, int mod_count,
                         const int* mod_order, const int* poly_order)
        {
            host::WorkspaceScope ws_scope;
            // reference ntt.cu:3607-3610 / 4288-4291
            if (cfg.n_power <= 9 || cfg.n_power >= 29)
                throw std::invalid_argument("Invalid n_power range!");
            if (mod_count <= 0 || modulus == nullptr || (mod_order == nullptr && poly_order == nullptr))
                throw std::invalid_argument("Invalid mod_count!");
            if (cfg.ntt_type != FORWARD && cfg.ntt_type != INVERSE)
                return; // reference: `default: break`
            if (batch_size <= 0)
                return;
            const bool inv = (cfg.ntt_type == INVERSE);
            const unsigned* ordered_skip_flag = nullptr;
            // fast path: whole tiles inside one polynomial
            if (cfg.n_power >= host::lazy_tile_log<T>(cfg.n_power) &&
                lazy_eligible<T>(cfg.n_power, batch_size, mod_count, cfg.stream) && (!inv || cfg.mod_inverse != nullptr))
            {
                host::RnsGuess guess = host::rns_guess(modulus, mod_count, static_cast<int>(sizeof(T)), inv, mod_order);
                const bool wide_net = cast_wide_net(guess, cfg.n_power, cfg.stream);
                const unsigned io_flags = inv ? static_cast<unsigned>(kern::F_SCALE) : 0u;
                kern::LazyArgsT<T> la =
                    lazy_args<T>(device_in, device_out, table, Modulus<T>(), modulus, mod_count,
                                 inv ? cfg.mod_inverse : nullptr, cfg.n_power, cfg.reduction_poly,
                                 batch_size, cfg.stream, mod_order, nullptr, &guess, io_flags, poly_order);
                if (inv)
                    run_transform_lazy_rns<T, true>(la, 0u, kern::F_SCALE, cfg.stream, guess);
                else
                    run_transform_lazy_rns<T, false>(la, 0u, 0u, cfg.stream, guess);
                if (!wide_net)
                    return;
                ordered_skip_flag = la.go_flag; // the generic kernels below run iff the preparation kernel published GO_GENERIC
            }
            kern::PassArgs<T> a = base_args<T>(device_in, device_out, table, cfg.n_power,
                                               cfg.reduction_poly, batch_size);
            a.mods = modulus;
            a.mod_count = mod_count;
            a.ninv_arr = cfg.mod_inverse;
            a.mod_order = mod_order;
            a.poly_order = poly_order;
            a.skip_flag = ordered_skip_flag;
            set_multi(a);
            if (inv)
                host::run_transform<T, true>(a, 0u, kern::F_SCALE, cfg.stream);
            else
                host::run_transform<T, false>(a, 0u, 0u, cfg.stream);
        }
    } // namespace

    template <typename T>
    __host__ void GPU_NTT_Modulus_Ordered(T* device_in, T* device_out, Root<T>* root_of_unity_table,
                                          Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                          int batch_size, int mod_count, int* order)
    {
        ordered_run<T>(device_in, device_out, root_of_unity_table, modulus, cfg, batch_size, mod_count,
                       order, nullptr);
    }
    template <typename T>
    __host__ void GPU_NTT_Modulus_Ordered_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                                  Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                                  int batch_size, int mod_count, int* order)
    {
        ordered_run<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size,
                       mod_count, order, nullptr);
    }
    template <typename T>
    __host__ void GPU_NTT_Poly_Ordered(T* device_in, T* device_out, Root<T>* root_of_unity_table,
                                       Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                       int batch_size, int mod_count, int* order)
    {
        ordered_run<T>(device_in, device_out, root_of_unity_table, modulus, cfg, batch_size, mod_count,
                       nullptr, order);
    }
    template <typename T>
    __host__ void GPU_NTT_Poly_Ordered_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                               Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                               int batch_size, int mod_count, int* order)
    {
        ordered_run<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size,
                       mod_count, nullptr, order);
    }

#define GPUNTT_INSTANTIATE_ORDERED(T)                                                          \
    template __host__ void GPU_NTT_Modulus_Ordered<T>(T*, T*, Root<T>*, Modulus<T>*,           \
                                                      ntt_rns_configuration<T>, int, int, int*); \
    template __host__ void GPU_NTT_Modulus_Ordered_Inplace<T>(T*, Root<T>*, Modulus<T>*,       \
                                                              ntt_rns_configuration<T>, int, int, \
                                                              int*);                           \
    template __host__ void GPU_NTT_Poly_Ordered<T>(T*, T*, Root<T>*, Modulus<T>*,              \
                                                   ntt_rns_configuration<T>, int, int, int*);  \
    template __host__ void GPU_NTT_Poly_Ordered_Inplace<T>(T*, Root<T>*, Modulus<T>*,          \
                                                           ntt_rns_configuration<T>, int, int, int*);
    GPUNTT_INSTANTIATE_ORDERED(Data32)
    GPUNTT_INSTANTIATE_ORDERED(Data64)
#undef GPUNTT_INSTANTIATE_ORDERED

    // ------------------------------------------------- exported symbol set ----
#define GPUNTT_INSTANTIATE(T, TU)                                                              \
    template __host__ void GPU_NTT<T>(T*, TU*, Root<TU>*, Modulus<TU>, ntt_configuration<TU>,  \
                                      int);                                                    \
    template __host__ void GPU_INTT<T>(TU*, T*, Root<TU>*, Modulus<TU>, ntt_configuration<TU>, \
                                       int);                                                   \
    template __host__ void GPU_NTT<T>(T*, TU*, Root<TU>*, Modulus<TU>*,                        \
                                      ntt_rns_configuration<TU>, int, int);                    \
    template __host__ void GPU_INTT<T>(TU*, T*, Root<TU>*, Modulus<TU>*,                       \
                                       ntt_rns_configuration<TU>, int, int);

    GPUNTT_INSTANTIATE(Data32, Data32)
    GPUNTT_INSTANTIATE(Data64, Data64)
    GPUNTT_INSTANTIATE(Data32s, Data32)
    GPUNTT_INSTANTIATE(Data64s, Data64)
#undef GPUNTT_INSTANTIATE

#define GPUNTT_INSTANTIATE_INPLACE(T)                                                          \
    template __host__ void GPU_NTT_Inplace<T>(T*, Root<T>*, Modulus<T>, ntt_configuration<T>,  \
                                              int);                                            \
    template __host__ void GPU_INTT_Inplace<T>(T*, Root<T>*, Modulus<T>, ntt_configuration<T>, \
                                               int);                                           \
    template __host__ void GPU_NTT_Inplace<T>(T*, Root<T>*, Modulus<T>*,                       \
                                              ntt_rns_configuration<T>, int, int);             \
    template __host__ void GPU_INTT_Inplace<T>(T*, Root<T>*, Modulus<T>*,                      \
                                               ntt_rns_configuration<T>, int, int);

    GPUNTT_INSTANTIATE_INPLACE(Data32)
    GPUNTT_INSTANTIATE_INPLACE(Data64)
#undef GPUNTT_INSTANTIATE_INPLACE

} // namespace gpuntt
