// lazy_u64_fwd31.hip -- instantiates the fwd fast-path kernels for uint64_t with the LIMIT = 31 lazy range (31 q < 2^64).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy_lim<false, 31>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
} }
