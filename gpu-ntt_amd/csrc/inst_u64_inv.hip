// inst_u64_inv.hip -- instantiates the Data64 inverse (Gentleman-Sande) tile-pass kernel family.
#include "launch_impl.hpp"

namespace gpuntt
{
    namespace host
    {
        template void launch_pass<Data64, true>(const Pass&, const kern::PassArgs<Data64>&, hipStream_t);
        template void launch_column_small<Data64, true>(const kern::PassArgs<Data64>&, int, int, hipStream_t);
    }
} // namespace gpuntt
