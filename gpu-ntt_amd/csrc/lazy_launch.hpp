// lazy_launch.hpp -- host side of the fast 64-bit path: twiddle preparation, workspace and
// pass dispatch for merge_pass_lazy (merge_lazy_kernels.hpp).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "launch.hpp"
#include "merge_lazy_kernels.hpp"

namespace gpuntt
{
    namespace host
    {
        // largest ring of the fast kernels = the reference's own limit (reference ntt.cu:2088-2091).  The prepared
        // table is 2 words x N per modulus (4 GiB of scratch at 2^28 in 64-bit words -- 1.4 % of the part's HBM);
        // rings above 2^24 take two strided passes + the contiguous pass like 2^24 does (the reference switches
        // to its grid-swapped ForwardCore_ / InverseCore_ there, ntt.cu:763-1084, 1320-1552)
        constexpr int LAZY_MAX_N_POWER = 28;

        // tile size (log2) used by the fast kernels for element type T and ring size 2^n:
        // 64-bit: 4096 coefficients (32 KiB of LDS).  32-bit: 16384 coefficients (64 KiB) for rings
        // of 2^14, 8192 for 2^13, which makes them a single HBM sweep.  Larger 32-bit rings take the tile
        // that needs fewer sweeps (2^20..2^22: 16384) and, at equal sweep count, the 4096 tile, whose
        // kernels are 10-20 % faster per stage (4 blocks per CU, strided pass carrying up to 8
        // stages): measured in profiles/u32_tile_ab_r01.txt, r04_u32_tile_big_rings_ab.txt, r04_u32_ring13_tile_ab.txt.
        // (The A/B switches of rounds 1-4 -- u32_tile, u64_big_tiles, u32_ring13_batch, contig_k, xcd_order, reverse, lim31 --
        // are retired: their questions are closed, the constants below are what they settled on.)
        constexpr int lazy_u32_tile_override() { return 0; }
        // largest ring (log2) that 64-bit calls may transform inside one big tile
        constexpr int lazy_u64_big_tiles() { return 14; }

        // EXPERIMENT kept behind a test hook (two_sweep_big; rounds 5 and 6): 64-bit rings 2^23 / 2^24, forward, in TWO sweeps --
        // one strided pass of 9 / 10 stages and the 14-stage contiguous pass, both on 16384-coefficient tiles -- instead of
        // three.  Bit-exact, a third less traffic, and no faster: profiles/r06_two_sweep_big_pmc.txt names the limiter.
        bool lazy_two_sweep_big();
        // 32-bit ring 2^13: every call takes the 8192-coefficient tile
        constexpr unsigned long long lazy_u32_small_batch() { return 0x7fffffffull; }
        // `inverse` and `polys` (transforms in the call) must be the same wherever one call asks:
        // the twiddle preparation lays the table out for the tile the passes will use
        template <typename T> inline int lazy_tile_log(int n, bool inverse = false, unsigned long long polys = 0)
        {
            if (sizeof(T) == 8)
            {
                // a 64-bit ring of 2^13 fits one 8192-coefficient tile (68 KiB of LDS, two blocks per CU):
                // one HBM sweep instead of two, 0.45 -> 0.33 ms per 2^26 coefficients.  2^14 in one
                // 16384-coefficient tile (136 KiB, ONE block per CU) gains 7 % forward on a full chip and
                // loses 9 % inverse (its first round waits for per-lane twiddles with nothing else
                // resident): forward calls of at least 256 transforms only.
                const int big = lazy_u64_big_tiles();
                if (n == 13 && big >= 13)
                    return 13;
                if (n == 14 && big >= 14 && !inverse && polys >= 256)
                    return 14;
                // 2^21 / 2^22: an 8-stage strided pass on 4096-coefficient tiles + the big contiguous tile = two sweeps
                // instead of three (2^22 inverse since round 5: 0.631 against 0.640-0.647 ms per 2^26 coefficients)
                if (n == 21 && big >= 13)
                    return 13;
                if (n == 22 && big >= 14)
                    return 14;
                if ((n == 23 || n == 24) && !inverse && lazy_two_sweep_big())
                    return 14;
                // (2^23 / 2^24 in two sweeps -- a strided pass of 9 / 10 stages + the 14-stage contiguous pass on
                // 16384-coefficient tiles -- was built and measured in round 5: bit-exact, -0.8 % at 2^24 x 64, +3 % at
                // 2^24 x 4 and 2^23 x 8, profiles/r05_two_sweep_big_ab.txt: not kept)
                return 12;
            }
            if (n <= 12)
                return 12;
            // 32-bit ring 2^13: a 8192-coefficient tile of its own instead of half of a 16384-coefficient one -- equal or
            // faster at every batch size (tools/ab_u32_ring13.py, round 4: batch 16 10.6 vs 12.2 us, 8192 inverse 0.137 vs
            // 0.146 ms, 16384 0.265 vs 0.282 ms; batch 1 was slower than the ring TWICE its size).  The
            // kernel is the one the one-launch 4-step of this ring runs on.  Option u32_ring13_batch = N restricts it to
            // calls of at most N polynomials (0: never); polys = 0 asks "for any batch" (eligibility checks).
            {
                const unsigned long long small = lazy_u32_small_batch();
                if (n == 13 && small != 0 && (polys == 0 ? small >= 0x7fffffffull : polys <= small))
                    return 13;
            }
            if (n <= 14)
                return 14;
            const int forced = lazy_u32_tile_override();
            if (forced != 0)
                return forced;
            return (n >= 20 && n <= 22) ? 14 : 12;
        }

        // The 32-coefficients-per-lane geometry of the 32-bit single-sweep rings (merge_e32_kernels.hpp, lazy_e32.hip): bit n
        // of the mask set = Merge transforms of the ring 2^n run on it (option u32_e32; rings 2^12 .. 2^15).  The ring 2^15
        // then takes ONE sweep on a 32768-coefficient tile instead of two on 4096-coefficient ones; 4-step plans keep
        // lazy_tile_log.  The prepared table of a tile is the same for both geometries.
        unsigned lazy_e32_mask();
        template <bool INV> void launch_ring_e32(int n, int lim, const kern::LazyArgsT<uint32_t>& a, hipStream_t stream);
        // the same geometry as the contiguous pass of a larger ring's plan (tile_log stages on a 4096- / 16384-coefficient tile:
        // forward the last pass, inverse the first); mask bit 16 of u32_e32
        template <bool INV> void launch_tile_e32(int tile_log, int lim, const kern::LazyArgsT<uint32_t>& a, hipStream_t stream);
        template <typename T> inline int lazy_tile_log_merge(int n, bool inverse = false, unsigned long long polys = 0)
        {
            if (sizeof(T) == 4 && n == 15 && ((lazy_e32_mask() >> 15) & 1u) != 0u)
                return 15;
            // 32-bit ring 2^23: one strided pass of 8 stages (16384-coefficient tiles) + the 15-stage contiguous pass on the
            // 32768-coefficient tile of the second geometry = TWO sweeps instead of three
            // ... and 2^24 with a strided pass of 9 stages (rows of 32 coefficients = 128-byte runs)
            if (sizeof(T) == 4 && (n == 23 || n == 24) && ((lazy_e32_mask() >> 16) & 1u) != 0u)
                return 15;
            return lazy_tile_log<T>(n, inverse, polys);
        }

        // Library-owned scratch for the prepared twiddles of the drop-in calls: one chain of buffers per (device, stream)
        // for eager calls, one per (device, stream, capture) for calls made while the stream is being captured into a
        // hipGraph; stream-ordered reuse inside a chain.  A buffer is never freed or synchronised on while anything can
        // still read it: growth allocates a new buffer and retires the old one; an eager chain's buffers live until
        // GPU_NTT_ReleaseWorkspaces(), a capture chain's belong to the captured graph and are pooled when it dies (prep.hip).
        // Inside a WorkspaceScope (every public entry point opens one) the calling thread keeps the chain's lock until
        // the scope ends, i.e. from the preparation launch to the last kernel launch of the call: two host threads on
        // one stream cannot interleave A.prep, B.prep, A.kernels.
        // or_null: nullptr instead of a HipException when the device has no memory left for the buffer (the entry
        // points then run the generic kernels, which need no scratch).
        void* lazy_workspace(hipStream_t stream, size_t bytes, bool or_null = false);
        // veto word of the chain lazy_workspace(stream, ...) just served + a fresh epoch for it (4-step table check)
        void lazy_workspace_veto(hipStream_t stream, unsigned long long** word, unsigned* epoch);
        struct WorkspaceScope
        {
            WorkspaceScope();
            ~WorkspaceScope();
            WorkspaceScope(const WorkspaceScope&) = delete;
            WorkspaceScope& operator=(const WorkspaceScope&) = delete;
        };
        void release_workspaces();

        // fills ws[0 .. mod_count*N) with Shoup pairs of the caller's table (device order) and
        // ws_ninv[0 .. mod_count) with the pairs of n^-1 (RNS only); perm_tile_log > 0 permutes
        // the distance-1/2/4 stages for tiles of that size.  Inverse transforms fold n^-1 into the
        // twiddle of their final stage: fold_ninv_single points at the host value (single
        // modulus), fold_ninv_rns takes it from ninv_arr.
        template <typename T>
        void launch_prep(const T* roots, lazy::Tw<T>* ws, const Modulus<T>* mods, T q, int mod_count, int n,
                         bool negacyclic, int perm_tile_log, const T* ninv_arr, lazy::Tw<T>* ws_ninv,
                         unsigned* go_flag, lazy::NormConst* norm_arr, hipStream_t stream,
                         const int* mod_order = nullptr, const T* fold_ninv_single = nullptr,
                         bool fold_ninv_rns = false, unsigned* host_state = nullptr, bool allow_31q = false,
                         unsigned family = 0u, const kern::SlowArgs<T>* slow = nullptr);

        // Drop-in RNS calls keep their moduli in device memory; the preparation kernel classifies them (go-flag states,
        // merge_lazy_kernels.hpp).  The host PREDICTS the lazy family from what the same stack -- same device, moduli
        // pointer, mod_order, mod_count, direction -- needed before: the preparation kernel writes the state to a host-mapped
        // word (`state_out`), read here WITHOUT any synchronisation (an older value is as good).  ONLY the predicted family is
        // enqueued; it also serves every narrower stack (kern::family_rank), so the prediction widens at once and narrows
        // only after 16 calls in a row that needed less.  A stack the enqueued family cannot serve (first call of a stack
        // with a 61- / 62-bit prime, moduli rewritten in place with wider ones, a captured graph replayed after the moduli
        // changed, moduli outside the documented domain) is transformed by the preparation kernel ITSELF (kern::SlowArgs:
        // slow, never wrong) -- no generic launch behind any Merge call.  all_families: every lazy family behind the exact
        // state (option rns_predict = 0, path = fast-strict, a full prediction table); only the 4-step entry points keep the
        // generic kernels behind their calls (table check).
        struct RnsGuess
        {
            unsigned state;      // predicted go-flag state (kern::GO_*); meaningful when !all_families
            bool all_families;
            unsigned* state_out; // device pointer of the host-mapped word for the preparation kernel, or nullptr
            bool unsure = false;         // nothing is known about this stack yet (first call with this moduli buffer), or its last
                                         // call found moduli outside the lazy families' domain
            bool shadow_generic = false; // set by the entry point: the generic kernels are enqueued behind the call, so the
                                         // preparation kernel must not transform the batch itself
        };
        // order: the mod_order array of a *_Modulus_Ordered call (a different subset of the stack may classify differently), else nullptr
        // exact (the 4-step entry point, whose kernels match the go-flag state exactly and keep the generic kernels behind
        // them): predict the state seen last, as it stands
        RnsGuess rns_guess(const void* moduli_device, int mod_count, int word_bytes, bool inverse, const void* order = nullptr,
                           bool exact = false);
        extern template void launch_prep<uint64_t>(const uint64_t*, lazy::Tw64*, const Modulus<uint64_t>*,
                                                   uint64_t, int, int, bool, int, const uint64_t*,
                                                   lazy::Tw64*, unsigned*, lazy::NormConst*, hipStream_t, const int*,
                                                   const uint64_t*, bool, unsigned*, bool, unsigned,
                                                   const kern::SlowArgs<uint64_t>*);
        extern template void launch_prep<uint32_t>(const uint32_t*, lazy::Tw32*, const Modulus<uint32_t>*,
                                                   uint32_t, int, int, bool, int, const uint32_t*,
                                                   lazy::Tw32*, unsigned*, lazy::NormConst*, hipStream_t, const int*,
                                                   const uint32_t*, bool, unsigned*, bool, unsigned,
                                                   const kern::SlowArgs<uint32_t>*);

        // 4-step transform of a ring that fits one tile in one launch: 64-bit 2^12 (tile 12), 2^13 (tile 13), 2^14 forward
        // calls of at least 256 transforms (tile 14); 32-bit 2^12 (tile 12), 2^13 (tile 13), 2^14 (tile 14).  a.tw = Merge table of the ring
        // (launch_prep_merge_from_fourstep).  fourstep_small_tile: the tile such a call runs on, 0 = ring too large
        template <typename T>
        inline int fourstep_small_tile(int n_power, bool inverse, unsigned long long polys, bool natural = false)
        {
            // the ring must fill the tile: the 32-bit ring 2^13 would share a 16384-coefficient tile between two
            // polynomials, and that kernel spills at the tile's 64-VGPR budget -- it gets a 8192-coefficient tile of
            // its own, which no Merge plan uses
            if (sizeof(T) == 4 && n_power == 13)
                return 13;
            // (64-bit 2^14, inverse: two sweeps like the ring's Merge plan -- transposing first pass + one partial row pass,
            // fourstep_inv_merge_split -- 0.438-0.440 ms per 2^26 coefficients against 0.452 for the one-tile kernel
            // rounds 3 and 4 ran there, which is gone)
            const int tl = lazy_tile_log<T>(n_power, inverse, polys);
            return (n_power >= 12 && n_power == tl) ? tl : 0;
        }
        template <typename T, bool INV>
        void launch_fourstep_small_lazy(int tile_log, int n, const kern::LazyArgsT<T>& a, hipStream_t stream,
                                        bool natural = false);
        extern template void launch_fourstep_small_lazy<uint64_t, false>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t, bool);
        extern template void launch_fourstep_small_lazy<uint64_t, true>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t, bool);
        extern template void launch_fourstep_small_lazy<uint32_t, false>(int, int, const kern::LazyArgsT<uint32_t>&, hipStream_t, bool);
        extern template void launch_fourstep_small_lazy<uint32_t, true>(int, int, const kern::LazyArgsT<uint32_t>&, hipStream_t, bool);
        // forward 4-step, rings 2^14 .. 2^17: the single contiguous pass behind the first kernel, able to be its own fall-back
        // (lazy_launch_impl.hpp); LIMSEL 0 / 31 / 8 / 4 (64-bit), 0 / 8 (32-bit)
        template <typename T, int LIMSEL> void launch_fourstep_fwd_last_lazy(int k, const kern::LazyArgsT<T>& a, hipStream_t stream);
        extern template void launch_fourstep_fwd_last_lazy<uint64_t, 0>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_fwd_last_lazy<uint64_t, 31>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_fwd_last_lazy<uint64_t, 8>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_fwd_last_lazy<uint64_t, 4>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_fwd_last_lazy<uint32_t, 0>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        extern template void launch_fourstep_fwd_last_lazy<uint32_t, 8>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        // forward 4-step, first pass of the ring's Merge plan reading the transposed input (kern::fourstep_first_lazy):
        // k = 5 .. 8 stages, k >= log2 n1 = a.n2_log
        template <typename T, int LIMSEL = 0>
        void launch_fourstep_first_lazy(int k, const kern::LazyArgsT<T>& a, hipStream_t stream);
        extern template void launch_fourstep_first_lazy<uint64_t>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_first_lazy<uint32_t>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        // inverse 4-step in Merge form (kern::fourstep_inv_first_lazy): the 12-stage contiguous first pass of the ring's
        // inverse Merge plan, stored transposed (log_n1 = 5 .. 8); LIMSEL = 8: 32-bit words, moduli below 2^29
        template <typename T, int LIMSEL = 0>
        void launch_fourstep_inv_first_lazy(int log_n1, const kern::LazyArgsT<T>& a, hipStream_t stream, int tile_log = 12);
        extern template void launch_fourstep_inv_first_lazy<uint64_t, 0>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t, int);
        extern template void launch_fourstep_inv_first_lazy<uint32_t, 0>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t, int);
        extern template void launch_fourstep_inv_first_lazy<uint32_t, 8>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t, int);
        extern template void launch_fourstep_inv_first_lazy<uint64_t, 8>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t, int);
        extern template void launch_fourstep_inv_first_lazy<uint64_t, 4>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t, int);
        // rings 2^13 .. 2^16 (n2 = 256 / 512): the remaining 1 .. 4 stages are the top stages of the n2-long rows -- one
        // partial contiguous pass (kern::PassSched SKIP = 12 - log_n1 = 7 / 6 / 5), 16 / 8 rows per tile
        template <typename T, int LIMSEL = 0>
        void launch_fourstep_inv_rows_lazy(int log_n2, int skip, const kern::LazyArgsT<T>& a, hipStream_t stream);
        extern template void launch_fourstep_inv_rows_lazy<uint64_t, 0>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_inv_rows_lazy<uint32_t, 0>(int, int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        extern template void launch_fourstep_inv_rows_lazy<uint32_t, 8>(int, int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        extern template void launch_fourstep_inv_rows_lazy<uint64_t, 8>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_inv_rows_lazy<uint64_t, 4>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        // Stage split of the strided row passes behind it: s = n - 12 stages on the bits above the first pass, as one pass
        // (s <= 8) or two; the first of them starts at row bit 12 - log_n1 and keeps 2^(12 - k) contiguous words per tile
        // row, so k >= log_n1.  false: the shape has no such plan (2^15, 2^16: fewer stages left than log_n1)
        // tl: tile of the first pass (fourstep_inv_tile: 12, or the big tile of the ring's inverse Merge plan)
        inline bool fourstep_inv_merge_split(int n_power, int log_n1, int& k_a, int& k_b, int tl = 12)
        {
            if (tl >= 13)
            {
                // the big tile does tl stages, ONE strided pass does the rest (two sweeps, like the ring's Merge plan);
                // that pass starts at row bit tl - log_n1 and wants runs of at least 2^4 contiguous words there
                k_a = n_power - tl;
                k_b = 0;
                return k_a >= 1 && k_a <= 8 && tl - log_n1 >= 4 && log_n1 >= 4 && log_n1 <= 9;
            }
            const int s = n_power - 12;
            const int l2 = n_power - log_n1, skip = 12 - log_n1;
            if (n_power >= 13 && n_power <= 16 && ((l2 == 9 && skip >= 5 && skip <= 7) || (l2 == 8 && skip == 7)))
            {
                k_a = k_b = 0; // one partial contiguous row pass (launch_fourstep_inv_rows_lazy)
                return true;
            }
            if (n_power < 17 || n_power > LAZY_MAX_N_POWER || s < log_n1 || s > 16)
                return false;
            if (s <= 8)
            {
                k_a = s;
                k_b = 0;
                return true;
            }
            k_a = (s + 1) / 2;
            if (k_a < log_n1)
                k_a = log_n1;
            k_b = s - k_a;
            return k_b >= 1 && k_a <= 8;
        }
        // tile of the inverse 4-step's first pass = the tile of the ring's inverse Merge plan where that plan takes a
        // big one and the transposing kernel exists for the reference's n1 of the ring (64-bit 2^21: 8192, 2^22: 16384;
        // 32-bit 2^20 .. 2^22: 16384 -- two sweeps instead of three); everything else, and the 61- / 62-bit families, 4096
        template <typename T> inline int fourstep_inv_tile(int n_power, int lim, int log_n1)
        {
            if (sizeof(T) == 8)
                return lim != 0 ? 12 : (n_power == 21 && log_n1 == 6) ? 13 : (n_power == 22 && log_n1 == 7) ? 14 : 12;
            return (n_power >= 20 && n_power <= 22 && log_n1 == n_power - 15) ? 14 : 12;
        }
        // natural-order 4-step (extension) in Merge form: strided Merge passes + one transposing row pass
        template <typename T, int LIMSEL = 0>
        void launch_fourstep_nat_last_lazy(int k, const kern::LazyArgsT<T>& a, hipStream_t stream);
        extern template void launch_fourstep_nat_last_lazy<uint64_t, 0>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_nat_last_lazy<uint32_t, 0>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        extern template void launch_fourstep_nat_last_lazy<uint64_t, 4>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        template <typename T, int LIMSEL = 0>
        void launch_fourstep_nat_first_inv_lazy(int k, const kern::LazyArgsT<T>& a, hipStream_t stream);
        extern template void launch_fourstep_nat_first_inv_lazy<uint64_t, 0>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_nat_first_inv_lazy<uint32_t, 0>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        extern template void launch_fourstep_nat_first_inv_lazy<uint64_t, 4>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        // The veto word of a 4-step call (prep.hip: publish_state): the preparation kernel publishes the call's go-flag state
        // through it and -- check -- verifies ALL of the caller's three tables against the one-root structure the fast
        // path relies on; any mismatch publishes kern::GO_GENERIC, the fast kernels return and the element-by-element
        // Barrett kernels behind them serve the call.  word == nullptr: no veto (option check_4step_tables = 0, plans
        // after their construction).  The kernels read the low half of the word as their go-flag.
        struct FourStepVeto
        {
            unsigned long long* word = nullptr;
            unsigned epoch = 0;
            bool check = false;
            unsigned* flag() const { return reinterpret_cast<unsigned*>(word); } // little endian: the state half
        };
        // Merge table of the 4-step ring (bit-reversed powers of its root), rebuilt from the caller's 4-step tables
        // straight into the Merge kernels' stage layout (prep.hip: prep_merge_from_fourstep)
        template <typename T>
        void launch_prep_merge_from_fourstep(const T* n1_table, const T* w_table, lazy::Tw<T>* ws, int log_n1, int log_n2,
                                             int perm_tile_log, bool inverse, bool fold, T q, T ninv,
                                             const Modulus<T>* mods, const T* ninv_dev, lazy::Tw<T>* ws_ninv,
                                             unsigned* go_flag, lazy::NormConst* norm_arr, hipStream_t stream,
                                             unsigned* host_state = nullptr, const FourStepVeto& veto = FourStepVeto(),
                                             const T* n2_table = nullptr);
        extern template void launch_prep_merge_from_fourstep<uint64_t>(const uint64_t*, const uint64_t*, lazy::Tw64*, int,
                                                                       int, int, bool, bool, uint64_t, uint64_t,
                                                                       const Modulus<uint64_t>*, const uint64_t*,
                                                                       lazy::Tw64*, unsigned*, lazy::NormConst*, hipStream_t,
                                                                       unsigned*, const FourStepVeto&, const uint64_t*);
        extern template void launch_prep_merge_from_fourstep<uint32_t>(const uint32_t*, const uint32_t*, lazy::Tw32*, int,
                                                                       int, int, bool, bool, uint32_t, uint32_t,
                                                                       const Modulus<uint32_t>*, const uint32_t*,
                                                                       lazy::Tw32*, unsigned*, lazy::NormConst*, hipStream_t,
                                                                       unsigned*, const FourStepVeto&, const uint32_t*);
        // diagnostic: the preparation kernels' normalised reciprocal floor(2^(W-1+b) / q) of every q[i] (0 for q < 3 / powers of two)
        template <typename T> void debug_recip_norm(const T* q, T* out, unsigned long long count, hipStream_t stream);
        // host Shoup companion floor(w * 2^W / q)
        inline uint64_t shoup_host(uint64_t w, uint64_t q)
        {
            return static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 64) / q);
        }
        inline uint32_t shoup_host(uint32_t w, uint32_t q)
        {
            return static_cast<uint32_t>((static_cast<uint64_t>(w) << 32) / q);
        }

        // pass list for the fast kernels: like make_plan but with the tile size as a parameter
        inline Plan make_plan_tl(int n, int tl, int contig_k, int max_strided = 8)
        {
            Plan pl{};
            pl.count = 0;
            if (n <= tl)
            {
                pl.pass[pl.count++] = Pass{true, n, 0};
                return pl;
            }
            if (contig_k > tl)
                contig_k = tl;
            if (contig_k < tl - 4)
                contig_k = tl - 4;
            const int s = n - contig_k;
            const int np = (s + max_strided - 1) / max_strided;
            int top = n;
            for (int i = 0; i < np; i++)
            {
                const int k = s / np + ((i < s % np) ? 1 : 0);
                top -= k;
                pl.pass[pl.count++] = Pass{false, k, top};
            }
            pl.pass[pl.count++] = Pass{true, contig_k, 0};
            return pl;
        }

        // process-wide tuning / test options (prep.hip); the library reads no environment variable
        bool set_option(const char* name, const char* value);
        // test hooks (csrc/test_hooks.h; not in the public headers): path = fast-strict | generic-capped, no_scratch,
        // rns_force_fallback, u32_e32 -- and everything set_option takes
        bool set_test_hook(const char* name, const char* value);
        void launch_log_start();
        std::string launch_log_take(); // space-separated kernel names ("merge_pass_lazy:31 prep_twiddles ..."), stops the log
        void scratch_stats(unsigned long long out[6]); // test hook: graph-owned scratch of capture chains (prep.hip)
        // 0 size heuristic, 1 generic kernels, 2 fast, 3 fast-strict (a call the fast kernels cannot take throws),
        // 4 generic-capped (4-step RNS overload: generic kernels on the capped shadow grid)
        int forced_path();
        // stages handled by the contiguous pass of the two-pass plans on 4096-coefficient tiles (8..12)
        int lazy_contig_k(int n);
        // the same for whole Merge transforms, by word size and direction: the split was swept in round 5 for every ring
        // 2^13 .. 2^20 (profiles/r05_stage_split_sweep.txt: strided stages 4 .. 8); where another split than the rule
        // "six strided stages" wins by more than the noise it is listed here (32-bit inverse 2^15: 9 + 6 instead of 10 + 5,
        // 0.177 against 0.211 ms per 2^26 coefficients; the others 1-4 %).  The 4-step plans keep lazy_contig_k.
        template <typename T> inline int lazy_contig_k_merge(int n, bool inverse)
        {
            if (sizeof(T) == 4)
            {
                if (!inverse && n >= 15 && n <= 17)
                    return n - 7; // 8 / 9 / 10 contiguous stages behind 7 strided ones
                if (inverse && n == 15)
                    return 9;
            }
            else
            {
                if (!inverse && n == 15)
                    return 8;
                if (inverse && (n == 15 || n == 16))
                    return 11;
                if (inverse && n == 17)
                    return 12;
            }
            return lazy_contig_k(n);
        }

        // consecutive passes of one transform walk the batch in opposite directions (Infinity Cache reuse of the hand-off)
        constexpr bool lazy_reverse_passes() { return true; }
        bool check_4step_tables();        // option check_4step_tables (default on)
        // forward 4-step in Merge form: stages of the first pass (the one that reads the transposed input) -- the first
        // strided pass of the ring's Merge plan on tile `tl`, widened to log2 n1 where that is larger (5 .. 8)
        inline int fourstep_first_k(int n_power, int log_n1, int tl)
        {
            const Plan pl = make_plan_tl(n_power, tl, tl == 12 ? lazy_contig_k(n_power) : tl);
            const int k0 = (pl.count >= 2 && !pl.pass[0].contig) ? pl.pass[0].k : 0;
            const int k = k0 > log_n1 ? k0 : log_n1;
            return k > 8 ? 8 : k;
        }

        // forward 4-step in Merge form: tile of the ring's Merge plan and stages of the gathering first pass.  A 16384-coefficient
        // tile whose remaining low stages (n - k1) would be fewer than 13 has no lazy-input contiguous kernel (only K = 13 /
        // 14 exist on that tile): such rings -- only reachable through the u32_tile = 14 tuning option, 32-bit 2^15 .. 2^17 --
        // keep the 4096-coefficient tile
        template <typename T>
        inline int fourstep_fwd_tile(int n_power, int log_n1, unsigned long long polys, int lim, int& k1)
        {
            int tl = lim != 0 ? 12 : lazy_tile_log<T>(n_power, false, polys);
            if (sizeof(T) == 8 && n_power >= 23)
                tl = 12; // (the two-sweep experiment of the Merge rings 2^23 / 2^24 does not extend to the 4-step form)
            // 32-bit ring 2^23: the gathering first pass takes 8 stages, the other 15 are ONE contiguous pass on the
            // 32768-coefficient tile of the second geometry -- two sweeps instead of three (2^24 would need a 9-stage gather)
            if (sizeof(T) == 4 && n_power == 23 && log_n1 <= 8 && ((lazy_e32_mask() >> 16) & 1u) != 0u)
                tl = 15;
            k1 = fourstep_first_k(n_power, log_n1, tl);
            if (tl == 14 && n_power > tl && n_power - k1 < 13)
            {
                tl = 12;
                k1 = fourstep_first_k(n_power, log_n1, tl);
            }
            return tl;
        }

        // grid of a lazy kernel: one block per tile (the capped, tile-walking grids of the 8 q / 4 q families went with
        // kern::WalksTiles in round 5)
        template <typename T, int LIMSEL> inline unsigned lazy_grid_cap(unsigned long long tiles, const unsigned*)
        {
            return static_cast<unsigned>(tiles);
        }

        // in_first: the pass reads canonical input (first pass of the transform)
        template <typename T, bool INV>
        void launch_pass_lazy(const Pass& p, int tile_log, bool in_first, bool last,
                              const kern::LazyArgsT<T>& a, hipStream_t stream);
        extern template void launch_pass_lazy<uint64_t, false>(const Pass&, int, bool, bool,
                                                               const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_pass_lazy<uint64_t, true>(const Pass&, int, bool, bool,
                                                              const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_pass_lazy<uint32_t, false>(const Pass&, int, bool, bool,
                                                               const kern::LazyArgsT<uint32_t>&, hipStream_t);
        extern template void launch_pass_lazy<uint32_t, true>(const Pass&, int, bool, bool,
                                                              const kern::LazyArgsT<uint32_t>&, hipStream_t);

        // strided passes with per-lane moduli (PerCoefficient layout with an RNS stack; lazy_vq.hip)
        template <typename T, bool INV>
        void launch_pass_lazy_vq(const Pass& p, bool in_first, bool last, const kern::LazyArgsT<T>& a, hipStream_t stream);
        extern template void launch_pass_lazy_vq<uint64_t, false>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_pass_lazy_vq<uint64_t, true>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_pass_lazy_vq<uint32_t, false>(const Pass&, bool, bool, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        extern template void launch_pass_lazy_vq<uint32_t, true>(const Pass&, bool, bool, const kern::LazyArgsT<uint32_t>&, hipStream_t);

        // RNS stacks of rings of 2^4 .. 2^9 coefficients: the single contiguous pass with per-lane moduli
        // (kern::merge_pass_lazy_vqc, lazy_vqc_*.hip); a.lim = 0 / 31 / 8 / 4 picks the lazy range like run_transform_lazy does
        template <typename T, bool INV> void launch_small_rns_lazy(int n, const kern::LazyArgsT<T>& a, hipStream_t stream);
        template <> void launch_small_rns_lazy<uint64_t, false>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        template <> void launch_small_rns_lazy<uint64_t, true>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_small_rns_lazy<uint32_t, false>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        extern template void launch_small_rns_lazy<uint32_t, true>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        // smallest ring an RNS stack (mod_count > 1) runs on the fast kernels: below it the 16 coefficients of a thread
        // would span polynomials of different moduli
        constexpr int LAZY_MIN_RNS_N_POWER = kern::R;

        template <bool INV, int LIMSEL>
        void launch_pass_lazy_lim(const Pass& p, bool in_first, bool last, const kern::LazyArgsT<uint64_t>& a,
                                  hipStream_t stream);
        extern template void launch_pass_lazy_lim<false, 8>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_pass_lazy_lim<true, 8>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_pass_lazy_lim<false, 4>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        template <bool INV, int LIMSEL>
        void launch_fourstep_lim(int what, int log_n1, const kern::LazyArgsT<uint64_t>& a, hipStream_t stream);
        extern template void launch_fourstep_lim<false, 8>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_lim<true, 8>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_lim<false, 4>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        extern template void launch_fourstep_lim<true, 4>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        // 64-bit words: lazy range a host-side modulus needs (0: the default 16 q kernels, 8: 61 bit, 4: 62 bit),
        // and whether the fast kernels can take it at all -- the documented domain of the reference
        // (src/include/gpuntt/common/modular_arith.cuh:66-67)
        template <typename TU> inline bool modulus_fast(const Modulus<TU>& m)
        {
            const TU max_bit = (sizeof(TU) == 8) ? TU(62) : TU(lazy::Mod<TU>::MAX_BIT);
            return m.value >= 3 && m.bit <= max_bit;
        }
        template <typename TU> inline int modulus_lim(const Modulus<TU>& m)
        {
            if (sizeof(TU) != 8)
                return 0;
            return m.bit == TU(62) ? 4 : (m.bit == TU(61) ? 8 : 0);
        }
        // LIMIT = 31 forward kernels on any 64-bit tile size (4096 / 8192 / 16384 coefficients)
        void launch_pass_lazy31(const Pass& p, int tile_log, bool in_first, bool last, const kern::LazyArgsT<uint64_t>& a,
                                hipStream_t stream);
        // forward transforms of a modulus with 31 q < 2^64 take the LIMIT = 31 kernels (32-bit: q < 2^29 the LIMIT = 8 ones)
        constexpr bool lazy_lim31_enabled() { return true; }
        constexpr unsigned lazy_order_flags() { return 0u; } // XCD-aware poly-minor block order
        // 32-bit words, every modulus of the call below 2^29: the LIMIT = 8 kernels (both directions, both tiles);
        // switched together with the 31 q range (GPUNTT_LIM31=0)
        template <bool INV>
        void launch_pass_lazy_u32w(const Pass& p, int tile_log, bool in_first, bool last,
                                   const kern::LazyArgsT<uint32_t>& a, hipStream_t stream);
        template <>
        void launch_pass_lazy_u32w<false>(const Pass&, int, bool, bool, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        template <>
        void launch_pass_lazy_u32w<true>(const Pass&, int, bool, bool, const kern::LazyArgsT<uint32_t>&, hipStream_t);
        inline bool lazy_wide_modulus32(uint32_t q) { return q >= 3 && q < (1u << 29); }
        inline bool lazy_lim31_modulus(uint64_t q) { return q >= 3 && q <= 0xffffffffffffffffull / 31; }
        extern template void launch_pass_lazy_lim<true, 4>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);

        // Range bound (units of q) a forward pass of k stages hands over when it reads values below `bound` -- the
        // run-time twin of kern::PassSched for Cooley-Tukey passes (every register carries the same bound there).
        // The last contiguous pass of a two-pass plan is instantiated for the bound it really receives (25 q behind
        // six strided stages, not the range limit): one round of range corrections less.
        inline int fwd_bound_after(int bound, int k, int limit, int tb)
        {
            for (int s = 0; s < k; s++)
                bound = lazy::ct_plan(bound, limit, tb).out;
            return bound;
        }

        // forced_tl: tile size the twiddle table was prepared for (NTTPlan); 0 = choose from the batch
        // low_stages > 0 (forward only): run only the stages on the low `low_stages` index bits of the ring -- the
        // values come lazy (below 16 q) from a pass that did the stages above them (forward 4-step: the gather pass, fourstep_first_lazy)
        template <typename T, bool INV>
        inline void run_transform_lazy(kern::LazyArgsT<T> base, unsigned first_in_flags,
                                       unsigned last_out_flags, hipStream_t stream, int forced_tl = 0, int low_stages = 0)
        {
            const int tl = (sizeof(T) == 8 && base.lim && base.lim != 31)
                               ? 12
                               : (forced_tl ? forced_tl : lazy_tile_log_merge<T>(base.n, INV, base.total >> base.n));
            const int pn = (!INV && low_stages > 0) ? low_stages : base.n;
            const bool partial = pn != base.n;
            // 32-bit rings that fill their tile: the 32-coefficients-per-lane kernels (one polynomial per block).  A table
            // laid out for the 32768-coefficient tile has no other kernel
            if constexpr (sizeof(T) == 4)
                if (!partial && base.n == tl && tl >= 12 && tl <= 15 &&
                    (tl == 15 || ((lazy_e32_mask() >> tl) & 1u) != 0u) && (base.total & ((1ull << tl) - 1ull)) == 0ull)
                {
                    kern::LazyArgsT<uint32_t> a = base;
                    a.p_lo = 0;
                    a.flags |= first_in_flags | last_out_flags;
                    return launch_ring_e32<INV>(tl, base.lim, a, stream);
                }
            // RNS stack of rings below one tile: the tile holds polynomials of different moduli.  From 1024 coefficients a
            // wave lies in one polynomial and the ordinary kernels pick the modulus per wave (merge_pass_lazy); rings of
            // 16 .. 512 coefficients take the per-lane-modulus form of the same pass
            if (base.mods != nullptr && base.mod_count > 1 && base.n < tl && base.n - kern::R < 6)
            {
                if (base.n < LAZY_MIN_RNS_N_POWER || partial)
                    throw std::invalid_argument("internal: RNS ring too small for the fast path");
                kern::LazyArgsT<T> a = base;
                a.p_lo = 0;
                a.flags |= first_in_flags | last_out_flags;
                return launch_small_rns_lazy<T, INV>(base.n, a, stream);
            }
            // (64-bit rings 2^23 / 2^24 on 16384-coefficient tiles, experiment: ONE strided pass of 9 / 10 stages)
            const bool big2 = sizeof(T) == 8 && tl == 14 && pn >= 23 && !partial;
            const Plan pl = make_plan_tl(pn, tl, tl == 12 ? (partial ? lazy_contig_k(pn) : lazy_contig_k_merge<T>(pn, INV)) : tl,
                                         big2 ? 10 : (sizeof(T) == 4 && tl == 15) ? 9 : 8);
            const void* src = base.in;
            int fwd_bound = partial ? 16 : 1; // range bound of the values in flight (forward, 31 q range: see fwd_bound_after)
            for (int i = 0; i < pl.count; i++)
            {
                Pass p = INV ? pl.pass[pl.count - 1 - i] : pl.pass[i];
                if (!INV && sizeof(T) == 8 && base.lim == 31)
                {
                    p.in_b = fwd_bound;
                    fwd_bound = fwd_bound_after(fwd_bound, p.k, 31, 4);
                }
                kern::LazyArgsT<T> a = base;
                a.in = src;
                a.p_lo = p.p_lo;
                a.flags |= lazy_order_flags();
                const bool first = (i == 0) && !partial; // reads canonical input
                if (i == 0)
                    a.flags |= first_in_flags;
                if (i == pl.count - 1)
                    a.flags |= last_out_flags;
                // the last pass walks the batch forwards, the one before it backwards, ... (Infinity Cache reuse
                // of the hand-off; GPUNTT_NO_REVERSE=1 switches it off for A/B timing)
                if (((pl.count - 1 - i) & 1) != 0 && lazy_reverse_passes())
                    a.flags |= kern::F_REVERSE;
                // 64-bit: only the contiguous pass runs on a big tile
                // (32-bit plans on the 32768-coefficient tile: their strided pass runs on 16384-coefficient tiles)
                const int tlp = (sizeof(T) == 8 && !p.contig && p.k <= 8) ? 12 : (sizeof(T) == 4 && tl == 15 && !p.contig) ? 14 : tl;
                // rings from 2^20: the per-lane twiddles of a contiguous pass are tens of MiB per
                // polynomial -- poly-minor block order lets a batch share them through L2
                const unsigned long long polys = base.total >> base.n;
                a.batch = (p.contig && base.n >= 20 && base.n >= tlp && polys >= 2 && polys <= 0x7fffffffull &&
                           (polys << base.n) == base.total)
                              ? static_cast<int>(polys)
                              : 0;
                if constexpr (sizeof(T) == 4)
                {
                    // the full-tile contiguous pass of a 32-bit plan (forward: last, on lazy input; inverse: first) on the
                    // 32-coefficients-per-lane geometry (GPU_PolyMul's fused product included)
                    // (partial: the low stages of a forward 4-step behind its gathering first pass -- lazy input as well)
                    const bool edge = INV ? (i == 0 && pl.count > 1) : (i == pl.count - 1 && (pl.count > 1 || partial));
                    if (p.contig && p.k == tlp && (tlp == 12 || tlp == 14 || tlp == 15) && edge &&
                        (tlp == 15 || ((lazy_e32_mask() >> 16) & 1u) != 0u))
                    {
                        launch_tile_e32<INV>(tlp, base.lim, a, stream);
                        src = base.out;
                        continue;
                    }
                }
                if constexpr (sizeof(T) == 8)
                {
                    if (base.lim == 8)
                        launch_pass_lazy_lim<INV, 8>(p, first, i == pl.count - 1, a, stream);
                    else if (base.lim == 4)
                        launch_pass_lazy_lim<INV, 4>(p, first, i == pl.count - 1, a, stream);
                    else if (base.lim == 31)
                    {
                        if constexpr (!INV)
                            launch_pass_lazy31(p, tlp, first, i == pl.count - 1, a, stream);
                        else
                            throw std::invalid_argument("internal: the 31 q range serves forward transforms only");
                    }
                    else
                        launch_pass_lazy<T, INV>(p, tlp, first, i == pl.count - 1, a, stream);
                }
                else
                {
                    if (base.lim == 8)
                        launch_pass_lazy_u32w<INV>(p, tlp, first, i == pl.count - 1, a, stream);
                    else
                        launch_pass_lazy<T, INV>(p, tlp, first, i == pl.count - 1, a, stream);
                }
                src = base.out;
            }
        }
    } // namespace host
} // namespace gpuntt
