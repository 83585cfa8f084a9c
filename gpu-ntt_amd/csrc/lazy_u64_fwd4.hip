// lazy_u64_fwd4.hip -- instantiates the fwd fast-path kernels for uint64_t with the LIMIT = 4 lazy range (62-bit moduli).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy_lim<false, 4>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_lim<false, 4>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_nat_last_lazy<uint64_t, 4>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_fwd_last_lazy<uint64_t, 4>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
} }
