// lazy_u32_fwd8.hip -- instantiates the fwd fast-path kernels for uint32_t with the LIMIT = 8 lazy range (q < 2^29).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template <>
void launch_pass_lazy_u32w<false>(const Pass& p, int tile_log, bool in_first, bool last, const kern::LazyArgsT<uint32_t>& a,
                               hipStream_t stream)
{
    if (tile_log == 12)
        return dispatch_tl<uint32_t, 12, false, 8>(p, in_first, last, a, stream);
    if (tile_log == 14)
        return dispatch_tl<uint32_t, 14, false, 8>(p, in_first, last, a, stream);
    if (tile_log == 13 && p.contig)
        return dispatch_tl<uint32_t, 13, false, 8>(p, in_first, last, a, stream);
    throw std::invalid_argument("internal: unsupported tile size in the fast path");
}
template void launch_fourstep_fwd_last_lazy<uint32_t, 8>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
} }
