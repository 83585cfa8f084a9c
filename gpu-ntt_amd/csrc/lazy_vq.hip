// lazy_vq.hip -- instantiates the strided lazy kernels with per-lane moduli (PerCoefficient layout with an RNS stack).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
// Pass shapes of the PerCoefficient plans (n_power <= 9, strided passes of <= 8 stages, merge_ntt.hip):
//   one pass (n <= 8): K = n, canonical in, last;   n = 9: forward 5 (first) + 4 (last), inverse 4 (first) + 5 (last)
template <typename T, bool INV>
void launch_pass_lazy_vq(const Pass& p, bool in_first, bool last, const kern::LazyArgsT<T>& a, hipStream_t stream)
{
    constexpr int LIMIT = lazy::Mod<T, kern::VqLim<T>::LIM, true>::LIMIT;
    const unsigned long long tiles = a.total >> 12;
    if (tiles == 0)
        return;
    if (tiles > 0x7fffffffull)
        throw std::invalid_argument("batch_size * N too large for one launch");
    const unsigned grid = static_cast<unsigned>(tiles);
#define GPUNTT_VQ(K_, IN_, LAST_)                                                                                     \
    do                                                                                                                \
    {                                                                                                                 \
        GPUNTT_LAUNCH((kern::merge_pass_lazy_vq<T, INV, K_, IN_, LAST_>), dim3(grid), dim3(kern::LTile<12>::NT), \
                           0, stream, a);                                                                             \
        GPUNTT_HIP_CHECK(hipGetLastError());                                                                          \
        return;                                                                                                       \
    } while (0)
    if (p.contig)
        throw std::invalid_argument("internal: per-lane-modulus kernels are strided passes");
    if (in_first && last)
        switch (p.k)
        {
            case 1: GPUNTT_VQ(1, 1, true);
            case 2: GPUNTT_VQ(2, 1, true);
            case 3: GPUNTT_VQ(3, 1, true);
            case 4: GPUNTT_VQ(4, 1, true);
            case 5: GPUNTT_VQ(5, 1, true);
            case 6: GPUNTT_VQ(6, 1, true);
            case 7: GPUNTT_VQ(7, 1, true);
            case 8: GPUNTT_VQ(8, 1, true);
            default: break;
        }
    if constexpr (!INV)
    {
        if (in_first && !last && p.k == 5)
            GPUNTT_VQ(5, 1, false);
        if (!in_first && last && p.k == 4)
            GPUNTT_VQ(4, LIMIT, true);
    }
    else
    {
        if (in_first && !last && p.k == 4)
            GPUNTT_VQ(4, 1, false);
        if (!in_first && last && p.k == 5)
            GPUNTT_VQ(5, LIMIT / 2, true);
    }
#undef GPUNTT_VQ
    throw std::invalid_argument("internal: unsupported per-lane-modulus pass");
}
template void launch_pass_lazy_vq<uint64_t, false>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_pass_lazy_vq<uint64_t, true>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_pass_lazy_vq<uint32_t, false>(const Pass&, bool, bool, const kern::LazyArgsT<uint32_t>&, hipStream_t);
template void launch_pass_lazy_vq<uint32_t, true>(const Pass&, bool, bool, const kern::LazyArgsT<uint32_t>&, hipStream_t);
} }
