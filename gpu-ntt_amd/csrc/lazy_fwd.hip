// lazy_fwd.hip -- instantiates the forward fast-path kernels (64-bit, lazy residues).
#include "lazy_launch_impl.hpp"
namespace gpuntt { namespace host {
template void launch_pass_lazy<false>(const Pass&, int, bool, const kern::LazyArgs&, hipStream_t);
} }
