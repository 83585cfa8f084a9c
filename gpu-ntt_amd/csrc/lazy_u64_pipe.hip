// lazy_u64_pipe.hip -- the persistent, software-pipelined 10-stage contiguous pass (merge_pipe_kernels.hpp):
// grid sizing and launch.
#include "lazy_launch.hpp"
#include "merge_pipe_kernels.hpp"

namespace gpuntt
{
    namespace host
    {
        int device_cu_count();

        int lazy_pipe_env()
        {
            static const int v = [] {
                const char* e = std::getenv("GPUNTT_PIPE");
                return e ? std::atoi(e) : 0; // 0 (default): never; 1: whenever the shape allows; 2: only when
                                             // every workgroup gets at least two tiles to pipeline
            }();
            return v;
        }

        // OFF unless GPUNTT_PIPE is set: measured on MI355X (profiles/r02_pipelined_contig_pass.md) the pass takes
        // 279 us against 267 us for the one-tile-per-workgroup kernel at C2 -- the part sits at its 1400 W power
        // cap during these kernels, so hiding the load latency buys nothing: the energy per call is unchanged.
        // Polynomial stride of a workgroup (= polynomials in flight), 0 = the call does not qualify:
        // two resident workgroups per CU, one per (tile position, polynomial lane); the stride must be a
        // multiple of mod_count so that a workgroup keeps its modulus
        unsigned lazy_pipe_stride(const kern::LazyArgsT<uint64_t>& a)
        {
            const int env = lazy_pipe_env();
            if (env <= 0 || a.lim != 0 || a.mul_in != nullptr || a.poly_order != nullptr || a.batch != 0 ||
                a.n < 13 || a.n > 16)
                return 0;
            const unsigned long long polys = a.total >> a.n;
            if ((polys << a.n) != a.total || polys > 0x7fffffffull)
                return 0;
            // GPUNTT_PIPE=3: the variant without prefetch, three resident workgroups per CU
            unsigned stride = static_cast<unsigned>((env == 3 ? 3 : 2) * device_cu_count()) >> (a.n - 12);
            const unsigned mc = a.mods != nullptr ? static_cast<unsigned>(a.mod_count) : 1u;
            stride -= stride % mc;
            if (stride == 0)
                return 0;
            if (env == 2 && polys < 2ull * stride)
                return 0;
            if (polys < stride)
            {
                if (polys % mc != 0)
                    return 0;
                stride = static_cast<unsigned>(polys); // small batch: one tile per workgroup
            }
            return stride;
        }

        template <bool INV>
        bool launch_contig_pipe(const Pass& p, bool in_first, bool last, const kern::LazyArgsT<uint64_t>& a,
                                hipStream_t stream)
        {
            if (!p.contig || p.k != 10 || in_first != INV || last != !INV)
                return false;
            const unsigned stride = lazy_pipe_stride(a);
            if (stride == 0)
                return false;
            const unsigned grid = stride << (a.n - 12);
            const bool prefetch = lazy_pipe_env() != 3;
            if constexpr (!INV)
            {
                if (prefetch)
                    hipLaunchKernelGGL((kern::merge_contig_pipe<uint64_t, false, 10, lazy::Mod<uint64_t>::LIMIT, true, true>),
                                       dim3(grid), dim3(256), 0, stream, a);
                else
                    hipLaunchKernelGGL((kern::merge_contig_pipe<uint64_t, false, 10, lazy::Mod<uint64_t>::LIMIT, true, false>),
                                       dim3(grid), dim3(256), 0, stream, a);
            }
            else
            {
                if (prefetch)
                    hipLaunchKernelGGL((kern::merge_contig_pipe<uint64_t, true, 10, 1, false, true>), dim3(grid), dim3(256),
                                       0, stream, a);
                else
                    hipLaunchKernelGGL((kern::merge_contig_pipe<uint64_t, true, 10, 1, false, false>), dim3(grid),
                                       dim3(256), 0, stream, a);
            }
            GPUNTT_HIP_CHECK(hipGetLastError());
            return true;
        }
        template bool launch_contig_pipe<false>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
        template bool launch_contig_pipe<true>(const Pass&, bool, bool, const kern::LazyArgsT<uint64_t>&, hipStream_t);
    } // namespace host
} // namespace gpuntt
