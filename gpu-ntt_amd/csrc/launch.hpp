// launch.hpp -- host-side pass planner and launcher shared by the Merge and 4-Step entry
// points.  A transform of length 2^n over `rows` independent rows is executed as
//   forward : [STRIDED passes, highest stage bits first] + one CONTIG pass (<= 12 stages)
//   inverse : one CONTIG pass + [STRIDED passes, lowest first]
// which replaces the reference's per-logN KernelConfig tables
// (reference src/include/gpuntt/ntt_merge/ntt.cuh:606-797).
#pragma once

#include <hip/hip_runtime.h>

#include <stdexcept>

#include "gpuntt/common/common.cuh"
#include "merge_kernels.hpp"

// blocks of a generic kernel launched BEHIND fast kernels (it returns at once unless the go-flag hands it the call): the
// blocks walk the tiles.  2 blocks per CU keep the part busy when the launch does own the call
#ifndef GPUNTT_SHADOW_GRID
#define GPUNTT_SHADOW_GRID 512
#endif

namespace gpuntt
{
    namespace host
    {
        // test hook (prep.hip): the launch log.  family: the lazy range of a fast kernel (0 default, 31, 8, 4), -1 = none
        void note_launch(const char* kernel_expr, int family = -1);
    } // namespace host
} // namespace gpuntt
// EVERY kernel launch of the library: reports the kernel to the launch log (off unless a test switched it on: one relaxed
// atomic load), then launches
#define GPUNTT_LAUNCH(kernel, ...)                                                                                       \
    do                                                                                                                   \
    {                                                                                                                    \
        ::gpuntt::host::note_launch(#kernel);                                                                            \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);                                                                         \
    } while (0)
#define GPUNTT_LAUNCH_FAMILY(family, kernel, ...)                                                                        \
    do                                                                                                                   \
    {                                                                                                                    \
        ::gpuntt::host::note_launch(#kernel, family);                                                                    \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);                                                                         \
    } while (0)

namespace gpuntt
{
    namespace host
    {
        struct Pass
        {
            bool contig;
            int k;    // stages
            int p_lo; // STRIDED: lowest global stage position
            int in_b = 0; // fast path, forward: range bound (units of q) of the values this pass reads, 0 = not tracked
        };

        struct Plan
        {
            Pass pass[8];
            int count;
        };

        // forward order; the inverse runs the same list backwards.  `contig_k` = stages done by
        // the contiguous pass when n > 12 (8..12): 12 minimises LDS exchanges, smaller values move
        // butterflies into the (memory-bound) strided passes.
        inline Plan make_plan(int n, int contig_k = kern::TL)
        {
            Plan pl{};
            pl.count = 0;
            if (n <= kern::TL)
            {
                pl.pass[pl.count++] = Pass{true, n, 0};
                return pl;
            }
            if (contig_k > kern::TL)
                contig_k = kern::TL;
            if (contig_k < 8)
                contig_k = 8;
            const int s = n - contig_k;         // stages above the contiguous pass
            const int np = (s + 7) / 8;         // STRIDED passes of at most 8 stages
            int top = n;                        // stage positions [top-1 .. ] still to cover
            for (int i = 0; i < np; i++)
            {
                const int k = s / np + ((i < s % np) ? 1 : 0);
                top -= k;
                pl.pass[pl.count++] = Pass{false, k, top};
            }
            pl.pass[pl.count++] = Pass{true, contig_k, 0};
            return pl;
        }

        // one pass = one kernel launch; defined in launch_impl.hpp and explicitly instantiated per
        // (element type, direction) in inst_*.hip so the kernel family compiles in parallel
        template <typename T, bool INV>
        void launch_pass(const Pass& p, const kern::PassArgs<T>& a, hipStream_t stream);
        extern template void launch_pass<Data32, false>(const Pass&, const kern::PassArgs<Data32>&, hipStream_t);
        extern template void launch_pass<Data32, true>(const Pass&, const kern::PassArgs<Data32>&, hipStream_t);
        extern template void launch_pass<Data64, false>(const Pass&, const kern::PassArgs<Data64>&, hipStream_t);
        extern template void launch_pass<Data64, true>(const Pass&, const kern::PassArgs<Data64>&, hipStream_t);

        // small-matrix column transform (merge_kernels.hpp::column_ntt_small), defined in inst_*.hip
        template <typename T, bool INV>
        void launch_column_small(const kern::PassArgs<T>& a, int n, int log_w, hipStream_t stream);
        extern template void launch_column_small<Data32, false>(const kern::PassArgs<Data32>&, int, int, hipStream_t);
        extern template void launch_column_small<Data32, true>(const kern::PassArgs<Data32>&, int, int, hipStream_t);
        extern template void launch_column_small<Data64, false>(const kern::PassArgs<Data64>&, int, int, hipStream_t);
        extern template void launch_column_small<Data64, true>(const kern::PassArgs<Data64>&, int, int, hipStream_t);

        // Runs the whole pass list.  `base` carries pointers, moduli, n, poly_shift,
        // root_shift, total and the direction-independent flags; this routine fills the
        // per-pass fields.  first_in_flags apply to the first pass only (signed input),
        // last_out_flags to the last pass only (scale / centred output).
        template <typename T, bool INV>
        inline void run_transform(kern::PassArgs<T> base, unsigned first_in_flags,
                                  unsigned last_out_flags, hipStream_t stream)
        {
            const Plan pl = make_plan(base.n);
            const void* src = base.in;
            for (int i = 0; i < pl.count; i++)
            {
                const Pass& p = INV ? pl.pass[pl.count - 1 - i] : pl.pass[i];
                kern::PassArgs<T> a = base;
                a.in = src;
                a.p_lo = p.p_lo;
                if (i == 0)
                    a.flags |= first_in_flags;
                if (i == pl.count - 1)
                    a.flags |= last_out_flags;
                launch_pass<T, INV>(p, a, stream);
                src = base.out; // later passes run in place on the output
            }
        }
    } // namespace host
} // namespace gpuntt
