// lazy_e32.hip -- instantiates the 32-coefficients-per-lane single-sweep kernels of the 32-bit rings 2^12 .. 2^15.
#include "lazy_launch.hpp"
#include "merge_e32_kernels.hpp"
namespace gpuntt { namespace host {
template <bool INV> void launch_ring_e32(int n, int lim, const kern::LazyArgsT<uint32_t>& a, hipStream_t stream)
{
    const unsigned long long polys = a.total >> n;
    if (polys == 0)
        return;
    if (polys > 0x7fffffffull)
        throw std::invalid_argument("batch_size * N too large for one launch");
    const dim3 grid(static_cast<unsigned>(polys));
#define GPUNTT_E32(TL_)                                                                                                  \
    case TL_:                                                                                                            \
        if (lim == 8)                                                                                                    \
            GPUNTT_LAUNCH_FAMILY(8, (kern::merge_ring_e32<TL_, INV, 8>), grid, dim3(kern::ETile<TL_>::NT), 0, stream, a);     \
        else                                                                                                             \
            GPUNTT_LAUNCH_FAMILY(0, (kern::merge_ring_e32<TL_, INV, 0>), grid, dim3(kern::ETile<TL_>::NT), 0, stream, a);     \
        break;
    switch (n)
    {
        GPUNTT_E32(12)
        GPUNTT_E32(13)
        GPUNTT_E32(14)
        GPUNTT_E32(15)
        default:
            throw std::invalid_argument("internal: no 32-coefficients-per-lane kernel for this ring");
    }
#undef GPUNTT_E32
    GPUNTT_HIP_CHECK(hipGetLastError());
}
// the contiguous pass of a LARGER ring on the same geometry (kern::merge_ring_e32<..., PART = true>): forward = the last pass of
// the plan, inverse = the first; tiles of 4096 or 16384 coefficients, blocks in merge_pass_lazy's order
template <bool INV> void launch_tile_e32(int tile_log, int lim, const kern::LazyArgsT<uint32_t>& a, hipStream_t stream)
{
    const unsigned long long tiles = a.total >> tile_log;
    if (tiles == 0)
        return;
    if (tiles > 0x7fffffffull)
        throw std::invalid_argument("batch_size * N too large for one launch");
    const dim3 grid(static_cast<unsigned>(tiles));
#define GPUNTT_E32P(TL_)                                                                                                       \
    case TL_:                                                                                                                  \
        if (lim == 8)                                                                                                          \
            GPUNTT_LAUNCH_FAMILY(8, (kern::merge_ring_e32<TL_, INV, 8, true>), grid, dim3(kern::ETile<TL_>::NT), 0, stream, a); \
        else                                                                                                                   \
            GPUNTT_LAUNCH_FAMILY(0, (kern::merge_ring_e32<TL_, INV, 0, true>), grid, dim3(kern::ETile<TL_>::NT), 0, stream, a); \
        break;
    switch (tile_log)
    {
        GPUNTT_E32P(12)
        GPUNTT_E32P(14)
        GPUNTT_E32P(15)
        default:
            throw std::invalid_argument("internal: no 32-coefficients-per-lane pass for this tile");
    }
#undef GPUNTT_E32P
    GPUNTT_HIP_CHECK(hipGetLastError());
}
template void launch_tile_e32<false>(int, int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
template void launch_tile_e32<true>(int, int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
template void launch_ring_e32<false>(int, int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
template void launch_ring_e32<true>(int, int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
} }
