// lazy_vqc_u64_inv.hip -- RNS stacks of rings of 2^4 .. 2^9 coefficients, 64-bit words, inverse: the single contiguous pass with per-lane moduli
// (kern::merge_pass_lazy_vqc; reference ForwardCoreLowRing / InverseCoreLowRing RNS forms, src/lib/ntt_merge/ntt.cu:116-219, 326-433).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
namespace
{
    template <typename T, bool INV, int LIMSEL>
    void small_rns_k(int n, const kern::LazyArgsT<T>& a, unsigned grid, hipStream_t stream)
    {
        switch (n)
        {
#define GPUNTT_CASE(KK)                                                                                                \
    case KK:                                                                                                            \
        GPUNTT_LAUNCH_FAMILY(LIMSEL, (kern::merge_pass_lazy_vqc<T, INV, KK, LIMSEL>), dim3(grid), dim3(kern::LTile<12>::NT), 0, stream, a); \
        break;
            GPUNTT_CASE(4)
            GPUNTT_CASE(5)
            GPUNTT_CASE(6)
            GPUNTT_CASE(7)
            GPUNTT_CASE(8)
            GPUNTT_CASE(9)
#undef GPUNTT_CASE
            default:
                throw std::invalid_argument("internal: no per-lane-modulus kernel for this ring");
        }
        GPUNTT_HIP_CHECK(hipGetLastError());
    }
} // namespace
template <>
void launch_small_rns_lazy<uint64_t, true>(int n, const kern::LazyArgsT<uint64_t>& a, hipStream_t stream)
{
    const unsigned long long tiles = (a.total + 4095ull) >> 12;
    if (tiles == 0)
        return;
    if (tiles > 0x7fffffffull)
        throw std::invalid_argument("batch_size * N too large for one launch");
    if (a.lim == 8)
        return small_rns_k<uint64_t, true, 8>(n, a, lazy_grid_cap<uint64_t, 8>(tiles, a.go_flag), stream);
    if (a.lim == 4)
        return small_rns_k<uint64_t, true, 4>(n, a, lazy_grid_cap<uint64_t, 4>(tiles, a.go_flag), stream);
    // default range; a.lim = 31: the launch stands in for the 31 q family (same kernels, other go-flag state)
    return small_rns_k<uint64_t, true, 0>(n, a, static_cast<unsigned>(tiles), stream);
}
} }
