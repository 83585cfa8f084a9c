// lazy_u64_fwd.hip -- instantiates the fwd fast-path kernels for uint64_t (lazy residues).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy<uint64_t, false>(const Pass&, int, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_small_lazy<uint64_t, false>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t, bool);
template void launch_fourstep_first_lazy<uint64_t>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_nat_last_lazy<uint64_t, 0>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_fwd_last_lazy<uint64_t, 0>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
} }
