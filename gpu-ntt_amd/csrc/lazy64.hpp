// lazy64.hpp -- lazy-range 64-bit modular arithmetic for the fast Merge-NTT kernels (gfx950).
//
// Measured on MI355X (tools/ubench_int.hip, profiles/ubench_int_r01.txt): v_mad_u64_u32 /
// v_mul_{lo,hi}_u32 issue at ~4.3-5.1 cycles per wave, the same class as most VALU ops, while
// 64-bit compare+select sequences cost 4 instructions.  The butterfly is therefore built to
// minimise *instructions*, not multiplies:
//
//   * twiddles come with a precomputed Shoup quotient w' = floor(w * 2^64 / q) (twiddle-prep
//     kernel, prep.hip), so  x*w mod q  =  x*w - qh*q  with  qh ~ hi64(x*w')  -- 9 multiply-class
//     instructions, no shifts, no compare;
//   * qh drops the low partial products (error <= 3), so the product lands in [0, 4q);
//   * values are kept in [0, B*q) with B tracked at COMPILE TIME per register (bound.hpp);
//     a conditional subtraction is emitted only where U + 4q could overflow LIMIT*q < 2^64.
//     For q < 2^60 (LIMIT = 16) that is one correction per two stages on the U input.
//
// Every kernel output is normalised to the canonical residue in [0, q), which is unique, so
// the transform is bit-identical to the reference's Barrett code.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace gpuntt
{
    namespace lazy
    {
        struct alignas(16) Tw64
        {
            uint64_t w;  // twiddle, canonical
            uint64_t wp; // floor(w * 2^64 / q)
        };

        __device__ __forceinline__ uint32_t lo32(uint64_t v) { return static_cast<uint32_t>(v); }
        __device__ __forceinline__ uint32_t hi32(uint64_t v) { return static_cast<uint32_t>(v >> 32); }

        struct Mod64
        {
            uint64_t q;
            uint64_t qneg; // 2^64 - q

            __device__ __forceinline__ uint64_t kq(int k) const { return q * static_cast<uint64_t>(k); }

            // x * w  (mod q), any x < 2^64, result in [0, 4q)
            __device__ __forceinline__ uint64_t mul(uint64_t x, const Tw64& t) const
            {
                const uint32_t x0 = lo32(x), x1 = hi32(x);
                const uint32_t h1 = __umulhi(x1, lo32(t.wp)), h2 = __umulhi(x0, hi32(t.wp));
                const uint64_t qh = static_cast<uint64_t>(x1) * hi32(t.wp) + h1 + h2;
                return x * t.w + qh * qneg;
            }

            // if (x >= k*q) x -= k*q
            template <int K> __device__ __forceinline__ uint64_t csub(uint64_t x) const
            {
                const uint64_t m = kq(K);
                return (x >= m) ? (x - m) : x;
            }

            // [0, B*q) -> [0, q)
            template <int B> __device__ __forceinline__ uint64_t normalize(uint64_t x) const
            {
                if constexpr (B > 8)
                    x = csub<8>(x);
                if constexpr (B > 4)
                    x = csub<4>(x);
                if constexpr (B > 2)
                    x = csub<2>(x);
                if constexpr (B > 1)
                    x = csub<1>(x);
                return x;
            }
        };

        // ---- compile-time range bookkeeping (units of q) --------------------------------
        constexpr int TB = 4; // mul() output bound

        constexpr int ceil_pow2(int v)
        {
            int p = 1;
            while (p < v)
                p <<= 1;
            return p;
        }
        // conditional subtraction amount that halves a bound b (b <= 2k): k = ceil_pow2(b) / 2
        constexpr int csub_k(int b) { return ceil_pow2(b) / 2; }

        // Cooley-Tukey (forward) butterfly plan for input bounds (bu, bv) under LIMIT:
        //   U' = U + T, V' = U - T + 4q  with  T = V*w in [0, 4q)
        struct CtPlan
        {
            int ku;  // 0, or conditional-subtract k*q from U first
            int out; // bound of both outputs
        };
        constexpr CtPlan ct_plan(int bu, int limit)
        {
            CtPlan p{0, 0};
            if (bu + TB > limit)
            {
                p.ku = csub_k(bu);
                bu = p.ku;
            }
            p.out = bu + TB;
            return p;
        }

        // Gentleman-Sande (inverse) butterfly plan:
        //   U' = U + V, V' = (U - V + c*q) * w  with  c >= bound(V), output V' in [0, 4q)
        struct GsPlan
        {
            int ku, kv; // conditional subtractions applied first (0 = none)
            int c;      // offset multiple
            int out_u;  // bound of U'
        };
        constexpr GsPlan gs_plan(int bu, int bv, int limit)
        {
            GsPlan p{0, 0, 0, 0};
            // at most one correction per operand is needed once bounds are <= limit
            if (bu + bv > limit || bu + ceil_pow2(bv) > limit)
            {
                if (bu >= bv)
                {
                    p.ku = csub_k(bu);
                    bu = p.ku;
                }
                else
                {
                    p.kv = csub_k(bv);
                    bv = p.kv;
                }
            }
            if (bu + bv > limit || bu + ceil_pow2(bv) > limit)
            {
                if (p.ku == 0 && bu >= bv)
                {
                    p.ku = csub_k(bu);
                    bu = p.ku;
                }
                else if (p.kv == 0)
                {
                    p.kv = csub_k(bv);
                    bv = p.kv;
                }
            }
            p.c = ceil_pow2(bv);
            p.out_u = bu + bv;
            return p;
        }
    } // namespace lazy
} // namespace gpuntt
