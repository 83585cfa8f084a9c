// lazy_launch_impl.hpp -- kernel dispatch of the fast 64-bit path (included by lazy_*.hip only)
#pragma once

#include "lazy_launch.hpp"

namespace gpuntt
{
    namespace host
    {
        template <bool INV, bool CONTIG, int K, int IN_BOUND, bool LAST>
        inline void launch_lazy_one(const kern::LazyArgs& a, unsigned grid, hipStream_t stream)
        {
            hipLaunchKernelGGL((kern::merge_pass_lazy<INV, CONTIG, K, IN_BOUND, LAST, LAZY_LIMIT>), dim3(grid),
                               dim3(kern::NT), 0, stream, a);
            GPUNTT_HIP_CHECK(hipGetLastError());
        }

        // instantiated combinations (see lazy_launch.hpp::run_transform_lazy):
        //   forward: STRIDED (IN 1 | 16, not last), CONTIG K=12 (IN 16, last), CONTIG K<=12 (IN 1, last)
        //   inverse: CONTIG K=12 (IN 1, not last), CONTIG K<=12 (IN 1, last), STRIDED (IN 16, last | not)
        template <bool INV>
        void launch_pass_lazy(const Pass& p, int in_bound, bool last, const kern::LazyArgs& a, hipStream_t stream)
        {
            const unsigned long long tiles = (a.total + kern::TILE - 1) >> kern::TL;
            if (tiles == 0)
                return;
            if (tiles > 0x7fffffffull)
                throw std::invalid_argument("batch_size * N too large for one launch");
            const unsigned grid = static_cast<unsigned>(tiles);
            constexpr int LIM = LAZY_LIMIT;
            if (p.contig)
            {
                if constexpr (!INV)
                {
                    if (in_bound != 1 && last)
                        switch (p.k)
                        {
                            case 8: return launch_lazy_one<false, true, 8, LIM, true>(a, grid, stream);
                            case 9: return launch_lazy_one<false, true, 9, LIM, true>(a, grid, stream);
                            case 10: return launch_lazy_one<false, true, 10, LIM, true>(a, grid, stream);
                            case 11: return launch_lazy_one<false, true, 11, LIM, true>(a, grid, stream);
                            case 12: return launch_lazy_one<false, true, 12, LIM, true>(a, grid, stream);
                            default: throw std::invalid_argument("internal: bad lazy contiguous pass");
                        }
                }
                else
                {
                    if (in_bound == 1 && !last)
                        switch (p.k)
                        {
                            case 8: return launch_lazy_one<true, true, 8, 1, false>(a, grid, stream);
                            case 9: return launch_lazy_one<true, true, 9, 1, false>(a, grid, stream);
                            case 10: return launch_lazy_one<true, true, 10, 1, false>(a, grid, stream);
                            case 11: return launch_lazy_one<true, true, 11, 1, false>(a, grid, stream);
                            case 12: return launch_lazy_one<true, true, 12, 1, false>(a, grid, stream);
                            default: throw std::invalid_argument("internal: bad lazy contiguous pass");
                        }
                }
                if (in_bound != 1 || !last)
                    throw std::invalid_argument("internal: unsupported lazy contiguous pass");
                switch (p.k)
                {
#define GPUNTT_CASE(KK)                                                                          \
    case KK:                                                                                      \
        return launch_lazy_one<INV, true, KK, 1, true>(a, grid, stream);
                    GPUNTT_CASE(1)
                    GPUNTT_CASE(2)
                    GPUNTT_CASE(3)
                    GPUNTT_CASE(4)
                    GPUNTT_CASE(5)
                    GPUNTT_CASE(6)
                    GPUNTT_CASE(7)
                    GPUNTT_CASE(8)
                    GPUNTT_CASE(9)
                    GPUNTT_CASE(10)
                    GPUNTT_CASE(11)
                    GPUNTT_CASE(12)
#undef GPUNTT_CASE
                    default:
                        throw std::invalid_argument("internal: bad contiguous pass size");
                }
            }
            else
            {
                if constexpr (!INV)
                {
                    if (last)
                        throw std::invalid_argument("internal: forward strided pass cannot be last");
                    switch (p.k * 2 + (in_bound == 1 ? 1 : 0))
                    {
#define GPUNTT_CASE(KK)                                                                          \
    case KK * 2 + 1:                                                                              \
        return launch_lazy_one<false, false, KK, 1, false>(a, grid, stream);                      \
    case KK * 2:                                                                                  \
        return launch_lazy_one<false, false, KK, LIM, false>(a, grid, stream);
                        GPUNTT_CASE(1)
                        GPUNTT_CASE(2)
                        GPUNTT_CASE(3)
                        GPUNTT_CASE(4)
                        GPUNTT_CASE(5)
                        GPUNTT_CASE(6)
                        GPUNTT_CASE(7)
                        GPUNTT_CASE(8)
#undef GPUNTT_CASE
                        default:
                            throw std::invalid_argument("internal: bad strided pass size");
                    }
                }
                else
                {
                    if (in_bound == 1)
                        throw std::invalid_argument("internal: inverse strided pass cannot be first");
                    switch (p.k * 2 + (last ? 1 : 0))
                    {
#define GPUNTT_CASE(KK)                                                                          \
    case KK * 2 + 1:                                                                              \
        return launch_lazy_one<true, false, KK, LIM, true>(a, grid, stream);                      \
    case KK * 2:                                                                                  \
        return launch_lazy_one<true, false, KK, LIM, false>(a, grid, stream);
                        GPUNTT_CASE(1)
                        GPUNTT_CASE(2)
                        GPUNTT_CASE(3)
                        GPUNTT_CASE(4)
                        GPUNTT_CASE(5)
                        GPUNTT_CASE(6)
                        GPUNTT_CASE(7)
                        GPUNTT_CASE(8)
#undef GPUNTT_CASE
                        default:
                            throw std::invalid_argument("internal: bad strided pass size");
                    }
                }
            }
        }
    } // namespace host
} // namespace gpuntt
