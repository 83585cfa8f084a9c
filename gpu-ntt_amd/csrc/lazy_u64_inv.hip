// lazy_u64_inv.hip -- instantiates the inv fast-path kernels for uint64_t (lazy residues).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy<uint64_t, true>(const Pass&, int, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_small_lazy<uint64_t, true>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t, bool);
template void launch_fourstep_nat_first_inv_lazy<uint64_t, 0>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_inv_first_lazy<uint64_t, 0>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t, int);
template void launch_fourstep_inv_rows_lazy<uint64_t, 0>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
} }
