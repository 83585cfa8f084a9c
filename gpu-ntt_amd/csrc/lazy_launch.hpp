// lazy_launch.hpp -- host side of the fast 64-bit path: twiddle preparation, workspace and
// pass dispatch for merge_pass_lazy (merge_lazy_kernels.hpp).
#pragma once

#include <hip/hip_runtime.h>

#include "launch.hpp"
#include "merge_lazy_kernels.hpp"

namespace gpuntt
{
    namespace host
    {
        constexpr int LAZY_LIMIT = 16;      // q < 2^60  =>  16 q < 2^64
        constexpr int LAZY_MAX_BIT = 60;    // Modulus<T>::bit bound for the fast path
        constexpr int LAZY_MAX_N_POWER = 24; // prepared table = 16 B * N per modulus

        // per-(device, stream) scratch for prepared twiddles; grows on demand, stream-ordered reuse
        void* lazy_workspace(hipStream_t stream, size_t bytes);

        // fills ws[0 .. mod_count*N) with Shoup pairs of the caller's table (device order) and
        // ws_ninv[0 .. mod_count) with the pairs of n^-1 (RNS only)
        void launch_prep(const uint64_t* roots, lazy::Tw64* ws, const Modulus<uint64_t>* mods, uint64_t q,
                         int mod_count, int n, bool negacyclic, bool perm_low, const uint64_t* ninv_arr,
                         lazy::Tw64* ws_ninv, hipStream_t stream);

        // host Shoup companion floor(w * 2^64 / q)
        inline uint64_t shoup_host(uint64_t w, uint64_t q)
        {
            return static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 64) / q);
        }

        // stages handled by the contiguous pass of the fast path (GPUNTT_CONTIG_K overrides, 8..12)
        int lazy_contig_k(int n);

        template <bool INV> void launch_pass_lazy(const Pass& p, int in_bound, bool last, const kern::LazyArgs& a,
                                                 hipStream_t stream);
        extern template void launch_pass_lazy<false>(const Pass&, int, bool, const kern::LazyArgs&, hipStream_t);
        extern template void launch_pass_lazy<true>(const Pass&, int, bool, const kern::LazyArgs&, hipStream_t);

        template <bool INV>
        inline void run_transform_lazy(kern::LazyArgs base, unsigned first_in_flags, unsigned last_out_flags,
                                       hipStream_t stream)
        {
            const Plan pl = make_plan(base.n, lazy_contig_k(base.n));
            const void* src = base.in;
            for (int i = 0; i < pl.count; i++)
            {
                const Pass& p = INV ? pl.pass[pl.count - 1 - i] : pl.pass[i];
                kern::LazyArgs a = base;
                a.in = src;
                a.p_lo = p.p_lo;
                if (i == 0)
                    a.flags |= first_in_flags;
                if (i == pl.count - 1)
                    a.flags |= last_out_flags;
                launch_pass_lazy<INV>(p, i == 0 ? 1 : LAZY_LIMIT, i == pl.count - 1, a, stream);
                src = base.out;
            }
        }
    } // namespace host
} // namespace gpuntt
