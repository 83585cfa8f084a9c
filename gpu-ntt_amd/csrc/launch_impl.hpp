// launch_impl.hpp -- definition of host::launch_pass (kernel dispatch on the compile-time
// pass shape).  Included only by the inst_*.hip translation units.
#pragma once

#include "launch.hpp"

namespace gpuntt
{
    namespace host
    {
        template <typename T, bool INV, bool CONTIG, int K, bool FST = false>
        inline void launch_one(const kern::PassArgs<T>& a, unsigned grid, hipStream_t stream)
        {
            GPUNTT_LAUNCH((kern::merge_pass<T, INV, CONTIG, K, FST>), dim3(grid), dim3(kern::NT),
                               0, stream, a);
            GPUNTT_HIP_CHECK(hipGetLastError());
        }

        template <typename T, bool INV>
        void launch_pass(const Pass& p, const kern::PassArgs<T>& a, hipStream_t stream)
        {
            const unsigned long long tiles = (a.total + kern::TILE - 1) >> kern::TL;
            if (tiles == 0)
                return;
            if (tiles > 0x7fffffffull)
                throw std::invalid_argument("batch_size * N too large for one launch");
            // shadow launch of an RNS call (the fast kernels usually own it): capped grid, see merge_pass
            const unsigned grid = (a.skip_flag != nullptr && tiles > GPUNTT_SHADOW_GRID) ? static_cast<unsigned>(GPUNTT_SHADOW_GRID) : static_cast<unsigned>(tiles);
            if (p.contig)
            {
                switch (p.k)
                {
#define GPUNTT_CASE(KK)                                                                          \
    case KK:                                                                                      \
        launch_one<T, INV, true, KK>(a, grid, stream);                                            \
        break;
                    GPUNTT_CASE(1)
                    GPUNTT_CASE(2)
                    GPUNTT_CASE(3)
                    GPUNTT_CASE(4)
                    GPUNTT_CASE(5)
                    GPUNTT_CASE(6)
                    GPUNTT_CASE(7)
                    GPUNTT_CASE(8)
                    GPUNTT_CASE(9)
                    GPUNTT_CASE(10)
                    GPUNTT_CASE(11)
                    GPUNTT_CASE(12)
#undef GPUNTT_CASE
                    default:
                        throw std::invalid_argument("internal: bad contiguous pass size");
                }
            }
            else
            {
                switch (p.k)
                {
#define GPUNTT_CASE(KK)                                                                          \
    case KK:                                                                                      \
        launch_one<T, INV, false, KK>(a, grid, stream);                                           \
        break;
                    GPUNTT_CASE(1)
                    GPUNTT_CASE(2)
                    GPUNTT_CASE(3)
                    GPUNTT_CASE(4)
                    GPUNTT_CASE(5)
                    GPUNTT_CASE(6)
                    GPUNTT_CASE(7)
                    GPUNTT_CASE(8)
#undef GPUNTT_CASE
                    default:
                        throw std::invalid_argument("internal: bad strided pass size");
                }
            }
        }


        template <typename T, bool INV>
        void launch_column_small(const kern::PassArgs<T>& a, int n, int log_w, hipStream_t stream)
        {
            // one block owns all N rows of a group of columns; 2^cols_log columns keep >= 256 butterflies
            int cols_log = 9 - n;
            if (cols_log < 0)
                cols_log = 0;
            if (cols_log > log_w)
                cols_log = log_w;
            const unsigned grid = 1u << (log_w - cols_log);
            GPUNTT_LAUNCH((kern::column_ntt_small<T, INV>), dim3(grid), dim3(256), 0, stream, a, n, log_w,
                               cols_log);
            GPUNTT_HIP_CHECK(hipGetLastError());
        }
    } // namespace host
} // namespace gpuntt
