// lazy_inv.hip -- instantiates the inverse fast-path kernels (64-bit, lazy residues).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy<true>(const Pass&, int, bool, const kern::LazyArgs&, hipStream_t);
} }
