// lazy_u64_inv4.hip -- instantiates the inv fast-path kernels for uint64_t with the LIMIT = 4 lazy range (62-bit moduli).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy_lim<true, 4>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_lim<true, 4>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_nat_first_inv_lazy<uint64_t, 4>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_inv_first_lazy<uint64_t, 4>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t, int);
template void launch_fourstep_inv_rows_lazy<uint64_t, 4>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
} }
