// lazy_u64_fwd31.hip -- instantiates the fwd fast-path kernels for uint64_t with the LIMIT = 31 lazy range (31 q < 2^64).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
void launch_pass_lazy31(const Pass& p, int tile_log, bool in_first, bool last, const kern::LazyArgsT<uint64_t>& a,
                        hipStream_t stream)
{
    if (tile_log == 12)
        return dispatch_tl<uint64_t, 12, false, 31>(p, in_first, last, a, stream);
    if (tile_log == 13 && p.contig)
        return dispatch_tl<uint64_t, 13, false, 31>(p, in_first, last, a, stream);
    if (tile_log == 14 && (p.contig || p.k > 8))
        return dispatch_tl<uint64_t, 14, false, 31>(p, in_first, last, a, stream);
    throw std::invalid_argument("internal: unsupported tile size in the fast path");
}
template void launch_fourstep_fwd_last_lazy<uint64_t, 31>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
} }
