// lazy_u64_inv8.hip -- instantiates the inv fast-path kernels for uint64_t with the LIMIT = 8 lazy range (61-bit moduli).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy_lim<true, 8>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_lim<true, 8>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
template void launch_fourstep_inv_first_lazy<uint64_t, 8>(int, const kern::LazyArgsT<uint64_t>&, hipStream_t, int);
template void launch_fourstep_inv_rows_lazy<uint64_t, 8>(int, int, const kern::LazyArgsT<uint64_t>&, hipStream_t);
} }
