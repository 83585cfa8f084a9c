// lazy_u64_fwd8.hip -- instantiates the fwd fast-path kernels for uint64_t with the LIMIT = 8 lazy range (61-bit moduli).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy_lim<false, 8>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_lim<false, 8>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_fwd_last_lazy<uint64_t, 8>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
} }
