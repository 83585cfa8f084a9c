// lazy_u32_fwd.hip -- instantiates the fwd fast-path kernels for uint32_t (lazy residues).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy<uint32_t, false>(const Pass&, int, bool, bool, const kern::LazyArgsT<uint32_t>&, hipStream_t);
template void launch_fourstep_small_lazy<uint32_t, false>(int, int, const kern::LazyArgsT<uint32_t>&, hipStream_t, bool);
template void launch_fourstep_first_lazy<uint32_t>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
template void launch_fourstep_nat_last_lazy<uint32_t, 0>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
template void launch_fourstep_fwd_last_lazy<uint32_t, 0>(int, const kern::LazyArgsT<uint32_t>&, hipStream_t);
} }
