// lazy_launch_impl.hpp -- kernel dispatch of the fast path (included by lazy_*.hip only)
#pragma once

#include "lazy_launch.hpp"

namespace gpuntt
{
    namespace host
    {
        template <typename T, int TLOG, bool INV, bool CONTIG, int K, int IN_BOUND, bool LAST, int LIMSEL = 0>
        inline void launch_lazy_one(const kern::LazyArgsT<T>& a, unsigned grid, hipStream_t stream)
        {
            GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::merge_pass_lazy<T, TLOG, INV, CONTIG, K, IN_BOUND, LAST, LIMSEL>), dim3(grid),
                               dim3(kern::LTile<TLOG>::NT), 0, stream, a);
            GPUNTT_HIP_CHECK(hipGetLastError());
        }

        // Instantiated pass shapes (everything run_transform_lazy can ask for):
        //   single pass       CONTIG K = n <= TLOG              (IN 1, last)   [TLOG 14: K = 13, 14 only]
        //   forward, n > TLOG STRIDED K 1..8 (IN 1 | LIMIT, not last) + CONTIG K (IN LIMIT, last)
        //   inverse, n > TLOG CONTIG K (IN 1, not last) + STRIDED K 1..8 (IN LIMIT / 2, last | not last)
        //   with CONTIG K in 8..12 for 4096-coefficient tiles and K = 14 for 16384-coefficient tiles
        template <typename T, int TLOG, bool INV, int LIMSEL = 0>
        void dispatch_tl(const Pass& p, bool in_first, bool last, const kern::LazyArgsT<T>& a,
                         hipStream_t stream)
        {
            constexpr int LIM = lazy::Mod<T, LIMSEL>::LIMIT;
            constexpr int TILE = kern::LTile<TLOG>::TILE;
            const unsigned long long tiles = (a.total + TILE - 1) >> TLOG;
            if (tiles == 0)
                return;
            if (tiles > 0x7fffffffull)
                throw std::invalid_argument("batch_size * N too large for one launch");
            const unsigned grid = lazy_grid_cap<T, LIMSEL>(tiles, a.go_flag);
#define GPUNTT_ONE(CONTIG_, K_, IN_, LAST_)                                                      \
    return launch_lazy_one<T, TLOG, INV, CONTIG_, K_, IN_, LAST_, LIMSEL>(a, grid, stream)
            if constexpr (sizeof(T) == 8 && TLOG >= 13)
            {
                // 64-bit big tiles: contiguous passes over the whole tile only -- the single pass of a
                // ring that fits, or the contiguous end of a two-pass plan (2^21, 2^22) whose strided
                // pass runs on 4096-coefficient tiles
                if (p.contig && p.k == TLOG)
                {
                    if (in_first && last)
                        GPUNTT_ONE(true, TLOG, 1, true);
                    if constexpr (!INV)
                    {
                        if (!in_first && last)
                            GPUNTT_ONE(true, TLOG, LIM, true);
                    }
                    else
                    {
                        // inverse: the contiguous first pass of the two-sweep plans (2^21 on the 8192 tile, 2^22 on the 16384 one)
                        if (in_first && !last)
                            GPUNTT_ONE(true, TLOG, 1, false);
                    }
                }
                // EXPERIMENT (test hook two_sweep_big): the single strided pass of 9 / 10 stages in front of the 14-stage
                // contiguous pass (rings 2^23 / 2^24, forward): rows of 32 / 16 coefficients = 256 / 128-byte runs
                if constexpr (TLOG == 14 && !INV)
                    if (!p.contig && in_first && !last)
                    {
                        if (p.k == 9)
                            GPUNTT_ONE(false, 9, 1, false);
                        if (p.k == 10)
                            GPUNTT_ONE(false, 10, 1, false);
                    }
                throw std::invalid_argument("internal: unsupported 64-bit big-tile pass");
            }
            else
            {
            if (p.contig)
            {
                if (in_first && last)
                {
                    if constexpr (TLOG == 12)
                        switch (p.k)
                        {
                            case 1: GPUNTT_ONE(true, 1, 1, true);
                            case 2: GPUNTT_ONE(true, 2, 1, true);
                            case 3: GPUNTT_ONE(true, 3, 1, true);
                            case 4: GPUNTT_ONE(true, 4, 1, true);
                            case 5: GPUNTT_ONE(true, 5, 1, true);
                            case 6: GPUNTT_ONE(true, 6, 1, true);
                            case 7: GPUNTT_ONE(true, 7, 1, true);
                            case 8: GPUNTT_ONE(true, 8, 1, true);
                            case 9: GPUNTT_ONE(true, 9, 1, true);
                            case 10: GPUNTT_ONE(true, 10, 1, true);
                            case 11: GPUNTT_ONE(true, 11, 1, true);
                            case 12: GPUNTT_ONE(true, 12, 1, true);
                            default: break;
                        }
                    else if constexpr (TLOG == 13)
                    {
                        if (p.k == 13)
                            GPUNTT_ONE(true, 13, 1, true);
                    }
                    else
                        switch (p.k)
                        {
                            case 13: GPUNTT_ONE(true, 13, 1, true);
                            case 14: GPUNTT_ONE(true, 14, 1, true);
                            default: break;
                        }
                }
                else if (!INV && !in_first && last)
                {
                    if constexpr (!INV)
                    {
                        if constexpr (TLOG == 12)
                            switch (p.k)
                            {
                                case 8: GPUNTT_ONE(true, 8, LIM, true);
                                case 9: GPUNTT_ONE(true, 9, LIM, true);
                                case 10:
                                    // 31 q range behind a strided pass that hands over <= 27 q (two-pass plans
                                    // of 2^14 .. 2^16: 17 / 21 / 25 q): three rounds of range corrections, not four
                                    if constexpr (LIM == 31)
                                        if (p.in_b > 0 && p.in_b <= 27)
                                            GPUNTT_ONE(true, 10, 27, true);
                                    GPUNTT_ONE(true, 10, LIM, true);
                                case 11: GPUNTT_ONE(true, 11, LIM, true);
                                case 12: GPUNTT_ONE(true, 12, LIM, true);
                                default: break;
                            }
                        else if constexpr (TLOG == 14)
                        {
                            if (p.k == 14)
                                GPUNTT_ONE(true, 14, LIM, true);
                            // two rows of 2^13 per tile, lazy input: the row pass of the 32-bit 4-step ring 2^18
                            if (p.k == 13)
                                GPUNTT_ONE(true, 13, LIM, true);
                        }
                    }
                }
                else if (INV && in_first && !last)
                {
                    if constexpr (INV)
                    {
                        if constexpr (TLOG == 12)
                            switch (p.k)
                            {
                                case 8: GPUNTT_ONE(true, 8, 1, false);
                                case 9: GPUNTT_ONE(true, 9, 1, false);
                                case 10: GPUNTT_ONE(true, 10, 1, false);
                                case 11: GPUNTT_ONE(true, 11, 1, false);
                                case 12: GPUNTT_ONE(true, 12, 1, false);
                                default: break;
                            }
                        else if constexpr (TLOG == 14)
                            if (p.k == 14)
                                GPUNTT_ONE(true, 14, 1, false);
                    }
                }
                throw std::invalid_argument("internal: unsupported contiguous pass in the fast path");
            }
            // strided passes exist only for rings larger than the tile
            if constexpr (TLOG == 12 || TLOG == 14)
            {
                if constexpr (!INV)
                {
                    if (!last)
                        switch (p.k * 2 + (in_first ? 1 : 0))
                        {
#define GPUNTT_CASE(KK)                                                                          \
    case KK * 2 + 1: GPUNTT_ONE(false, KK, 1, false);                                            \
    case KK * 2: GPUNTT_ONE(false, KK, LIM, false);
                            GPUNTT_CASE(1)
                            GPUNTT_CASE(2)
                            GPUNTT_CASE(3)
                            GPUNTT_CASE(4)
                            GPUNTT_CASE(5)
                            GPUNTT_CASE(6)
                            GPUNTT_CASE(7)
                            GPUNTT_CASE(8)
#undef GPUNTT_CASE
                            default: break;
                        }
                    // 32-bit ring 2^24 in two sweeps: ONE strided pass of 9 stages on 16384-coefficient tiles (rows of 32
                    // coefficients) in front of the 15-stage contiguous pass of the 32-coefficients-per-lane geometry
                    if constexpr (sizeof(T) == 4 && TLOG == 14)
                        if (!last && in_first && p.k == 9)
                            GPUNTT_ONE(false, 9, 1, false);
                    // strided passes that END a forward transform / BEGIN an inverse one: the PerCoefficient layout
                    // (rows = coefficients, every stage is a strided stage; default lazy range and, for 61- / 62-bit moduli
                    // in 64-bit words, the 4 q range; 4096-coefficient tiles)
                    if constexpr ((LIMSEL == 0 || (LIMSEL == 4 && sizeof(T) == 8)) && TLOG == 12)
                        if (last)
                            switch (p.k * 2 + (in_first ? 1 : 0))
                            {
#define GPUNTT_CASE(KK)                                                                          \
    case KK * 2 + 1: GPUNTT_ONE(false, KK, 1, true);                                             \
    case KK * 2: GPUNTT_ONE(false, KK, LIM, true);
                                GPUNTT_CASE(1)
                                GPUNTT_CASE(2)
                                GPUNTT_CASE(3)
                                GPUNTT_CASE(4)
                                GPUNTT_CASE(5)
                                GPUNTT_CASE(6)
                                GPUNTT_CASE(7)
                                GPUNTT_CASE(8)
#undef GPUNTT_CASE
                                default: break;
                            }
                }
                else
                {
                    if constexpr ((LIMSEL == 0 || (LIMSEL == 4 && sizeof(T) == 8)) && TLOG == 12)
                        if (in_first)
                            switch (p.k * 2 + (last ? 1 : 0))
                            {
#define GPUNTT_CASE(KK)                                                                          \
    case KK * 2 + 1: GPUNTT_ONE(false, KK, 1, true);                                             \
    case KK * 2: GPUNTT_ONE(false, KK, 1, false);
                                GPUNTT_CASE(1)
                                GPUNTT_CASE(2)
                                GPUNTT_CASE(3)
                                GPUNTT_CASE(4)
                                GPUNTT_CASE(5)
                                GPUNTT_CASE(6)
                                GPUNTT_CASE(7)
                                GPUNTT_CASE(8)
#undef GPUNTT_CASE
                                default: break;
                            }
                    if constexpr (sizeof(T) == 4 && TLOG == 14)
                        if (!in_first && last && p.k == 9) // (the inverse twin of the 9-stage pass of the 32-bit ring 2^24)
                            GPUNTT_ONE(false, 9, LIM / 2, true);
                    if (!in_first)
                        switch (p.k * 2 + (last ? 1 : 0))
                        {
                    // an inverse pass hands over values below LIMIT / 2 (lazy.hpp: gs_plan corrects the sums)
#define GPUNTT_CASE(KK)                                                                          \
    case KK * 2 + 1: GPUNTT_ONE(false, KK, LIM / 2, true);                                       \
    case KK * 2: GPUNTT_ONE(false, KK, LIM / 2, false);
                            GPUNTT_CASE(1)
                            GPUNTT_CASE(2)
                            GPUNTT_CASE(3)
                            GPUNTT_CASE(4)
                            GPUNTT_CASE(5)
                            GPUNTT_CASE(6)
                            GPUNTT_CASE(7)
                            GPUNTT_CASE(8)
#undef GPUNTT_CASE
                            default: break;
                        }
                }
            }
            throw std::invalid_argument("internal: unsupported strided pass in the fast path");
            }
#undef GPUNTT_ONE
        }

        // forward 4-step: first strided pass of the Merge plan with the transposed gather, k stages
        template <typename T, int LIMSEL>
        void launch_fourstep_first_lazy(int k, const kern::LazyArgsT<T>& a, hipStream_t stream)
        {
            const unsigned long long tiles = a.total >> 12;
            if (tiles == 0)
                return;
            if (tiles > 0x7fffffffull)
                throw std::invalid_argument("batch_size * N too large for one launch");
            const unsigned grid = lazy_grid_cap<T, LIMSEL>(tiles, a.go_flag);
            switch (k)
            {
#define GPUNTT_CASE(KK)                                                                                               \
    case KK:                                                                                                           \
        GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::fourstep_first_lazy<T, KK, LIMSEL>), dim3(grid), dim3(kern::LTile<12>::NT), 0, stream, a); \
        break;
                GPUNTT_CASE(5)
                GPUNTT_CASE(6)
                GPUNTT_CASE(7)
                GPUNTT_CASE(8)
#undef GPUNTT_CASE
                default:
                    throw std::invalid_argument("internal: bad 4-step first pass");
            }
            GPUNTT_HIP_CHECK(hipGetLastError());
        }

        // inverse 4-step: first (contiguous, 12-stage) pass of the ring's inverse Merge plan with the transposed store
        template <typename T, int LIMSEL>
        void launch_fourstep_inv_first_lazy(int log_n1, const kern::LazyArgsT<T>& a, hipStream_t stream, int tile_log)
        {
            const unsigned long long tiles = a.total >> tile_log;
            if (tiles == 0)
                return;
            if (tiles > 0x7fffffffull)
                throw std::invalid_argument("batch_size * N too large for one launch");
            const unsigned grid = lazy_grid_cap<T, LIMSEL>(tiles, a.go_flag);
            // big first tiles (fourstep_inv_tile): the shapes the reference's launch table gives those rings
#define GPUNTT_BIG(TLG, KK)                                                                                               \
    if (tile_log == TLG && log_n1 == KK)                                                                                   \
    {                                                                                                                      \
        GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::fourstep_inv_first_lazy<T, KK, LIMSEL, TLG>), dim3(grid), dim3(kern::LTile<TLG>::NT), 0,  \
                           stream, a);                                                                                     \
        GPUNTT_HIP_CHECK(hipGetLastError());                                                                               \
        return;                                                                                                            \
    }
            if constexpr (sizeof(T) == 8 && LIMSEL == 0)
            {
                GPUNTT_BIG(13, 6) // 2^21 = 64 x 32768
                GPUNTT_BIG(14, 7) // 2^22 = 128 x 32768
            }
            if constexpr (sizeof(T) == 4)
            {
                GPUNTT_BIG(14, 5) // 2^20 = 32 x 32768
                GPUNTT_BIG(14, 6) // 2^21
                GPUNTT_BIG(14, 7) // 2^22
            }
#undef GPUNTT_BIG
            if (tile_log != 12)
                throw std::invalid_argument("internal: bad 4-step tile");
            switch (log_n1)
            {
#define GPUNTT_CASE(KK)                                                                                                   \
    case KK:                                                                                                               \
        GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::fourstep_inv_first_lazy<T, KK, LIMSEL>), dim3(grid), dim3(kern::LTile<12>::NT), 0, stream, a); \
        break;
                GPUNTT_CASE(5)
                GPUNTT_CASE(6)
                GPUNTT_CASE(7)
                GPUNTT_CASE(8)
#undef GPUNTT_CASE
                default:
                    throw std::invalid_argument("internal: bad 4-step n1");
            }
            GPUNTT_HIP_CHECK(hipGetLastError());
        }

        // inverse 4-step of the rings 2^13 .. 2^16 (n2 = 256 / 512): the stages the first pass left -- row bits
        // [skip, log2 n2) of the rows of `out`, 16 / 8 rows per tile, one register round, no LDS; last pass of the transform
        template <typename T, int LIMSEL>
        void launch_fourstep_inv_rows_lazy(int log_n2, int skip, const kern::LazyArgsT<T>& a, hipStream_t stream)
        {
            constexpr int LIM = lazy::Mod<T, LIMSEL>::LIMIT;
            const unsigned long long tiles = a.total >> 12;
            if (tiles == 0)
                return;
            if (tiles > 0x7fffffffull)
                throw std::invalid_argument("batch_size * N too large for one launch");
            const unsigned grid = lazy_grid_cap<T, LIMSEL>(tiles, a.go_flag);
#define GPUNTT_ROWS(K_, S_)                                                                                                \
    GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::merge_pass_lazy<T, 12, true, true, K_, LIM / 2, true, LIMSEL, S_>), dim3(grid),               \
                       dim3(kern::LTile<12>::NT), 0, stream, a)
            if (log_n2 == 9 && skip == 5)
                GPUNTT_ROWS(9, 5);
            else if (log_n2 == 9 && skip == 6)
                GPUNTT_ROWS(9, 6);
            else if (log_n2 == 9 && skip == 7)
                GPUNTT_ROWS(9, 7);
            else if (log_n2 == 8 && skip == 7)
                GPUNTT_ROWS(8, 7);
            else
                throw std::invalid_argument("internal: bad 4-step row pass");
#undef GPUNTT_ROWS
            GPUNTT_HIP_CHECK(hipGetLastError());
        }

        // forward 4-step, rings 2^14 .. 2^17 (n2 <= 4096): the ONE contiguous pass behind the gathering first kernel as
        // instantiations of its own (SKIP = 1: a marker, merge_lazy_kernels.hpp) -- exactly the kernel run_transform_lazy
        // would launch for that pass, plus the n2-point phase of the element-by-element algorithm for vetoed calls
        // (kern::F_SELF_FALLBACK): nothing is enqueued behind such a call.  k = stages of the pass (9 or 11)
        template <typename T, int LIMSEL>
        void launch_fourstep_fwd_last_lazy(int k, const kern::LazyArgsT<T>& a, hipStream_t stream)
        {
            constexpr int LIM = lazy::Mod<T, LIMSEL>::LIMIT;
            const unsigned long long tiles = a.total >> 12;
            if (tiles == 0)
                return;
            if (tiles > 0x7fffffffull)
                throw std::invalid_argument("batch_size * N too large for one launch");
            const unsigned grid = lazy_grid_cap<T, LIMSEL>(tiles, a.go_flag);
            if (k == 9)
                GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::merge_pass_lazy<T, 12, false, true, 9, LIM, true, LIMSEL, 1>), dim3(grid),
                                   dim3(kern::LTile<12>::NT), 0, stream, a);
            else if (k == 11)
                GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::merge_pass_lazy<T, 12, false, true, 11, LIM, true, LIMSEL, 1>), dim3(grid),
                                   dim3(kern::LTile<12>::NT), 0, stream, a);
            else
                throw std::invalid_argument("internal: bad forward 4-step last pass");
            GPUNTT_HIP_CHECK(hipGetLastError());
        }

        // 4-step transform of a ring that fits one tile: the whole transform in one launch (fourstep_small_lazy);
        // natural: the natural-order extension (spectrum side in NTT_4STEP_CPU order)
        template <typename T, bool INV, bool NAT>
        void launch_fourstep_small_impl(int tile_log, int n, const kern::LazyArgsT<T>& a, hipStream_t stream)
        {
            const unsigned long long tiles = (a.total + (1ull << tile_log) - 1) >> tile_log;
            if (tiles == 0)
                return;
            if (tiles > 0x7fffffffull)
                throw std::invalid_argument("batch_size * N too large for one launch");
            const unsigned grid = static_cast<unsigned>(tiles);
#define GPUNTT_SMALL(TL_, K_)                                                                                          \
    GPUNTT_LAUNCH_FAMILY(0, (kern::fourstep_small_lazy<T, TL_, INV, K_, 0, NAT>), dim3(grid), dim3(kern::LTile<TL_>::NT), 0, \
                       stream, a)
            if constexpr (sizeof(T) == 8)
            {
                if (tile_log == 12 && n == 12)
                    GPUNTT_SMALL(12, 12);
                else if (tile_log == 13 && n == 13)
                    GPUNTT_SMALL(13, 13);
                else if (tile_log == 14 && n == 14 && !INV)
                {
                    if constexpr (!INV) // (the inverse of that ring runs in two sweeps: fourstep_small_tile)
                        GPUNTT_SMALL(14, 14);
                }
                else
                    throw std::invalid_argument("internal: no one-tile 4-step kernel for this ring");
            }
            else
            {
                if (tile_log == 12 && n == 12)
                    GPUNTT_SMALL(12, 12);
                else if (tile_log == 13 && n == 13)
                    GPUNTT_SMALL(13, 13);
                else if (tile_log == 14 && n == 14)
                    GPUNTT_SMALL(14, 14);
                else
                    throw std::invalid_argument("internal: no one-tile 4-step kernel for this ring");
            }
#undef GPUNTT_SMALL
            GPUNTT_HIP_CHECK(hipGetLastError());
        }
        template <typename T, bool INV>
        void launch_fourstep_small_lazy(int tile_log, int n, const kern::LazyArgsT<T>& a, hipStream_t stream, bool natural)
        {
            if (natural)
                launch_fourstep_small_impl<T, INV, true>(tile_log, n, a, stream);
            else
                launch_fourstep_small_impl<T, INV, false>(tile_log, n, a, stream);
        }

        // natural-order forward 4-step (fourstep_ntt.hip): the transposing last row pass, k = 7 .. 9 low stages of the
        // ring, lazy input from the strided passes above it.  LIMSEL = 4: 64-bit words with a 61- / 62-bit modulus
        template <typename T, int LIMSEL>
        void launch_fourstep_nat_last_lazy(int k, const kern::LazyArgsT<T>& a, hipStream_t stream)
        {
            constexpr int TLOG = 12;
            constexpr int LIM = lazy::Mod<T, LIMSEL>::LIMIT;
            const unsigned grid = static_cast<unsigned>(a.total >> TLOG);
#define GPUNTT_ONE(KK)                                                                            \
    GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::fourstep_nat_last_lazy<T, TLOG, KK, LIM, LIMSEL>), dim3(grid),       \
                       dim3(kern::LTile<TLOG>::NT), 0, stream, a)
            if (k == 7)
                GPUNTT_ONE(7);
            else if (k == 8)
                GPUNTT_ONE(8);
            else if (k == 9)
                GPUNTT_ONE(9);
            else
                throw std::invalid_argument("internal: bad natural-order 4-step row pass");
#undef GPUNTT_ONE
            GPUNTT_HIP_CHECK(hipGetLastError());
        }

        // natural-order inverse 4-step: the transposing first row pass (k = 7 .. 9 low stages of the ring)
        template <typename T, int LIMSEL>
        void launch_fourstep_nat_first_inv_lazy(int k, const kern::LazyArgsT<T>& a, hipStream_t stream)
        {
            constexpr int TLOG = 12;
            const unsigned grid = static_cast<unsigned>(a.total >> TLOG);
            switch (k)
            {
#define GPUNTT_CASE(KK)                                                                          \
    case KK:                                                                                      \
        GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::fourstep_nat_first_inv_lazy<T, TLOG, KK, LIMSEL>), dim3(grid),  \
                           dim3(kern::LTile<TLOG>::NT), 0, stream, a);                            \
        break;
                GPUNTT_CASE(7)
                GPUNTT_CASE(8)
                GPUNTT_CASE(9)
#undef GPUNTT_CASE
                default:
                    throw std::invalid_argument("internal: bad natural-order 4-step row pass");
            }
            GPUNTT_HIP_CHECK(hipGetLastError());
        }
        // 4-step kernels for 64-bit words with a 61- / 62-bit modulus (LIMIT = 8 / 4): what = 1 first pass of the forward
        // Merge plan (log_n1 = its stage count), 2 the one-launch 2^12 ring, 3 the same in natural order (LIMIT = 4 only)
        template <bool INV, int LIMSEL>
        void launch_fourstep_lim(int what, int log_n1, const kern::LazyArgsT<uint64_t>& a, hipStream_t stream)
        {
            if constexpr (!INV)
                if (what == 1)
                    return launch_fourstep_first_lazy<uint64_t, LIMSEL>(log_n1, a, stream);
            if (what == 2 || what == 3)
            {
                const unsigned long long tiles = a.total >> 12;
                if (tiles == 0)
                    return;
                if (tiles > 0x7fffffffull)
                    throw std::invalid_argument("batch_size * N too large for one launch");
                if (what == 2)
                    GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::fourstep_small_lazy<uint64_t, 12, INV, 12, LIMSEL>),
                                       dim3(lazy_grid_cap<uint64_t, LIMSEL>(tiles, a.go_flag)),
                                       dim3(kern::LTile<12>::NT), 0, stream, a);
                else if constexpr (LIMSEL == 4) // natural-order extension (4 q family only)
                    GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::fourstep_small_lazy<uint64_t, 12, INV, 12, LIMSEL, true>),
                                       dim3(static_cast<unsigned>(tiles)), dim3(kern::LTile<12>::NT), 0, stream, a);
                else
                    throw std::invalid_argument("internal: bad 4-step kernel selector");
                GPUNTT_HIP_CHECK(hipGetLastError());
                return;
            }
            throw std::invalid_argument("internal: bad 4-step kernel selector");
        }

        // 64-bit words with a 61- / 62-bit modulus: LIMIT = 8 / 4 kernels, 4096-coefficient tiles only
        template <bool INV, int LIMSEL>
        void launch_pass_lazy_lim(const Pass& p, bool in_first, bool last, const kern::LazyArgsT<uint64_t>& a,
                                  hipStream_t stream)
        {
            return dispatch_tl<uint64_t, 12, INV, LIMSEL>(p, in_first, last, a, stream);
        }

        template <typename T, bool INV>
        void launch_pass_lazy(const Pass& p, int tile_log, bool in_first, bool last,
                              const kern::LazyArgsT<T>& a, hipStream_t stream)
        {
            if (tile_log == 12)
                return dispatch_tl<T, 12, INV>(p, in_first, last, a, stream);
            if (tile_log == 14 && (sizeof(T) == 4 || p.contig || p.k > 8))
                return dispatch_tl<T, 14, INV>(p, in_first, last, a, stream);
            // 8192-coefficient tile: 64-bit contiguous passes (2^13, 2^21); 32-bit: the single pass of the ring 2^13
            if (tile_log == 13 && p.contig)
                return dispatch_tl<T, 13, INV>(p, in_first, last, a, stream);
            throw std::invalid_argument("internal: unsupported tile size in the fast path");
        }
    } // namespace host
} // namespace gpuntt
