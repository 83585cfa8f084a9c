// inst_u32_inv.hip -- instantiates the Data32 inverse (Gentleman-Sande) tile-pass kernel family.
#include "launch_impl.hpp"

namespace gpuntt
{
    namespace host
    {
        template void launch_pass<Data32, true>(const Pass&, const kern::PassArgs<Data32>&, hipStream_t);
        template void launch_column_small<Data32, true>(const kern::PassArgs<Data32>&, int, int, hipStream_t);
    }
} // namespace gpuntt
