// lazy.hpp -- lazy-range 64-bit and 32-bit modular arithmetic for the fast Merge-NTT kernels (gfx950).
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
//   * values are kept in [0, B*q) with B tracked at COMPILE TIME per register (PassSched);
//     a conditional subtraction is emitted only where U + 4q could overflow LIMIT*q < 2^64.
//     For q < 2^60 (LIMIT = 16) that is one correction per two stages on the U input.
//   * 32-bit moduli (q < 2^30): exact one-instruction quotient, products in [0, 2q), LIMIT 4,
//     corrections are a subtract + unsigned min.
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
        // twiddle + precomputed quotient floor(w * 2^W / q), W = 8 * sizeof(T)
        template <typename T> struct Tw;
        template <> struct alignas(16) Tw<uint64_t>
        {
            uint64_t w;
            uint64_t wp;
        };
        template <> struct alignas(8) Tw<uint32_t>
        {
            uint32_t w;
            uint32_t wp;
        };
        using Tw64 = Tw<uint64_t>;
        using Tw32 = Tw<uint32_t>;

        __device__ __forceinline__ uint32_t lo32(uint64_t v) { return static_cast<uint32_t>(v); }
        __device__ __forceinline__ uint32_t hi32(uint64_t v) { return static_cast<uint32_t>(v >> 32); }

        // LIM = 0: the default lazy range of the word size (16 q for 64-bit words, q < 2^60; 4 q for 32-bit words,
        // q < 2^30); LIM = 31: 64-bit words with 31 q < 2^64 (forward transforms); LIM = 8: 64-bit words with 61-bit moduli (8 q < 2^64: one range correction per stage);
        // LIM = 4: 62-bit moduli (4 q < 2^64: products corrected to [0, 2q), the 32-bit scheme in 64-bit words)
        // VQ: the modulus differs per LANE (PerCoefficient layout with an RNS stack: column c uses modulus c % mod_count, and
        // the lanes of a wave hold different columns) -- q, -q and every multiple of them live in vector registers, and so
        // do the twiddles; the instruction sequences are the same, only the operand classes change
        template <typename T, int LIM = 0, bool VQ = false> struct Mod;

        // constants of the one-multiply final normalisation (64-bit only): for x < 16 q
        //   k = ((x >> sh) * M) >> (32 + c)  is floor(x / q) or one less, so x - k*q is in [0, 2q)
        // with sh = max(bit - 27, 0), qt = (q >> sh) + [sh > 0], c = bit - 1 - sh, M = floor(2^(32+c) / qt)
        // Moduli of 48 bits and more take the estimate from the HIGH WORD of x alone (sh = 32: no 64-bit shift;
        // qt = (q >> 32) + 1 still has >= 15 bits, so k * 2^-15 < 1 for k < 32 and the estimate stays
        // floor(x / q) or one less): `hi` = 1, same formulas.
        struct NormConst
        {
            uint32_t sh, c, M, hi;
        };
        __host__ __device__ inline NormConst make_norm_const(uint64_t q, uint64_t bit)
        {
            NormConst n{0, 0, 0, 0};
            if (q < 3 || bit < 2 || bit > 61)
                return n;
            n.hi = bit >= 48 ? 1u : 0u;
            n.sh = n.hi ? 32u : (bit > 27 ? static_cast<uint32_t>(bit - 27) : 0u);
            const uint64_t qt = (q >> n.sh) + (n.sh > 0 ? 1u : 0u); // exact when nothing is shifted out
            n.c = static_cast<uint32_t>(bit - 1 - n.sh);
            const uint64_t m = (1ull << (32 + n.c)) / qt;
            n.M = m > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(m);
            return n;
        }

        // 32-bit words: x < 2^32 -> [0, 2q) with k = hi32(x * M), M = floor(2^32 / q) (k is floor(x / q) or one less), the
        // same record with sh = c = 0; 64-bit words: make_norm_const
        template <typename T> __host__ __device__ inline NormConst norm_const_of(T q, T bit)
        {
            if constexpr (sizeof(T) == 4)
                return NormConst{0u, 0u, q >= 3u ? static_cast<uint32_t>(0x100000000ull / q) : 0u, 0u};
            else
                return make_norm_const(static_cast<uint64_t>(q), static_cast<uint64_t>(bit));
        }

        // d = a * b + c (32 x 32 + 64 -> 64) pinned to ONE v_mad_u64_u32.  SB: b is wave-uniform and is
        // read straight from a scalar register (a VOP3 instruction may name one).  The carry-out lands
        // in a dead scalar pair.  Written as asm because the compiler narrows a 64-bit product whose
        // high word is dead into v_mul_lo_u32 + v_add3_u32 chains (one more instruction per two terms).
        template <bool SB> __device__ __forceinline__ uint64_t mad32(uint32_t a, uint32_t b, uint64_t c)
        {
            uint64_t d, cy;
            if constexpr (SB)
                asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(b), "v"(c));
            else
                asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
            return d;
        }
        template <bool SB> __device__ __forceinline__ uint64_t mad32z(uint32_t a, uint32_t b)
        {
            uint64_t d, cy;
            if constexpr (SB)
                asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "s"(b));
            else
                asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "v"(b));
            return d;
        }

        // a * b - 1 (mod 2^64): the accumulator starts from the inline constant -1 (mulc below)
        template <bool SB> __device__ __forceinline__ uint64_t mad32m1(uint32_t a, uint32_t b)
        {
            uint64_t d, cy;
            if constexpr (SB)
                asm("v_mad_u64_u32 %0, %1, %2, %3, -1" : "=v"(d), "=s"(cy) : "v"(a), "s"(b));
            else
                asm("v_mad_u64_u32 %0, %1, %2, %3, -1" : "=v"(d), "=s"(cy) : "v"(a), "v"(b));
            return d;
        }
        // ~w + u  =  u - w - 1 (mod 2^32) as ONE instruction, v_xad_u32 (w ^ -1) + u: the sum u + v of a value u and the
        // COMPLEMENT w = ~v of a value v
        // (plain C: the compiler selects v_xad_u32 for it, and an asm statement costs an s_nop of the hazard recogniser)
        __device__ __forceinline__ uint32_t xad_not(uint32_t w, uint32_t u) { return (w ^ 0xffffffffu) + u; }

        // ---- 64-bit: sloppy-quotient Shoup product in [0, 4q); LIMIT 16 needs q < 2^60, LIMIT 8 q < 2^61 ------
        template <int LIM, bool VQ = false> struct Mod64
        {
            static constexpr int TB = (LIM == 4) ? 2 : 4; // product bound (units of q)
            static constexpr int LIMIT = LIM;              // lazy values stay below LIMIT * q < 2^64
            static constexpr int MAX_BIT = (LIM == 16 || LIM == 31) ? 60 : (LIM == 8 ? 61 : 62);
            uint64_t q;
            uint64_t qneg; // 2^64 - q
            NormConst nc;
            uint32_t zero; // 0, pinned to v127 where the quotient chain needs a zero high word (mul_acc_raw)
            uint32_t one;  // 1, opaque to the optimiser (mul_acc_raw)

            __device__ __forceinline__ void set(uint64_t modulus, const NormConst& n)
            {
                q = modulus;
                qneg = 0 - modulus;
                nc = n;
                // opaque to the optimiser: a constant 0 would be re-materialised (one v_mov per use)
                asm("v_mov_b32 %0, 0" : "=v"(zero));
                asm("v_mov_b32 %0, 1" : "=v"(one));
            }
            // x < 32 q  ->  [0, 2q): quotient estimate from the top bits, one 32 x 64 multiply-subtract.
            // HI: the modulus has >= 48 bits (nc.hi), the top bits are the high word as it stands
            template <bool HI = false> __device__ __forceinline__ uint64_t reduce_2q(uint64_t x) const
            {
                const uint32_t xt = HI ? hi32(x) : static_cast<uint32_t>(x >> nc.sh);
                const uint32_t k = __umulhi(xt, nc.M) >> nc.c;
                const uint64_t r = static_cast<uint64_t>(k) * lo32(qneg) + x; // v_mad_u64_u32
                return r + (static_cast<uint64_t>(k * hi32(qneg)) << 32);
            }
            __device__ __forceinline__ uint64_t kq(int k) const { return q * static_cast<uint64_t>(k); }
            __device__ __forceinline__ bool hi_norm() const { return nc.hi != 0u; } // wave-uniform

            // acc + x * w - qh * q  (mod 2^64)  =  acc + T  with  T = x * w (mod q) + {0..3} q  in [0, 4q),
            // for any x < 2^64: 11 instructions, the 64-bit add of the butterfly included --
            //   qh ~ hi64(x * w')      2 v_mul_hi_u32 + 2 v_mad_u64_u32 (low partial products dropped: 3 short at most)
            //   cross terms (word 1)   4 v_mad_u64_u32 chained through one accumulator whose low word is the sum
            //   words 0..1             2 v_mad_u64_u32 chained from `acc`, 1 v_add_u32 for the cross sum
            // UNI: the twiddle is wave-uniform (scalar registers).  ZERO: acc = 0.
            template <bool UNI, bool ZERO = false>
            __device__ __forceinline__ uint64_t mul_acc(uint64_t x, const Tw64& t, uint64_t acc) const
            {
                if constexpr (LIM == 4)
                {
                    // 62-bit moduli: the sloppy product lies in [0, 4q) < 2^64; one correction brings it to [0, 2q)
                    const uint64_t p = csub<2>(mul_acc_raw<UNI, true>(x, t, 0));
                    return ZERO ? p : acc + p;
                }
                else
                    return mul_acc_raw<UNI, ZERO>(x, t, acc);
            }
            template <bool UNI, bool ZERO>
            __device__ __forceinline__ uint64_t mul_acc_raw(uint64_t x, const Tw64& t, uint64_t acc) const
            {
                const uint32_t x0 = lo32(x), x1 = hi32(x);
                const uint32_t h2 = __umulhi(x0, hi32(t.wp));
                // qh = x1 * wp1 + hi32(x1 * wp0): the 64-bit addend {h1, 0} is the fixed pair v[126:127] whose
                // high half holds `zero` for the whole kernel (the "{v127}" operand), so the zero extension of
                // h1 costs nothing -- the compiler's own form re-creates the pair with a v_mov per butterfly
                // (80 of 1640 VALU instructions per wave in the 10-stage pass).  Every 64-bit kernel is built
                // for 4 waves per SIMD = a 128-VGPR budget, so v126 / v127 exist.
                uint64_t qh, carry;
                if constexpr (UNI)
                    asm("v_mul_hi_u32 v126, %2, %3\n\tv_mad_u64_u32 %0, %1, %2, %4, v[126:127]"
                        : "=v"(qh), "=s"(carry)
                        : "v"(x1), "s"(lo32(t.wp)), "s"(hi32(t.wp)), "{v127}"(zero)
                        : "v126");
                else
                    asm("v_mul_hi_u32 v126, %2, %3\n\tv_mad_u64_u32 %0, %1, %2, %4, v[126:127]"
                        : "=v"(qh), "=s"(carry)
                        : "v"(x1), "v"(lo32(t.wp)), "v"(hi32(t.wp)), "{v127}"(zero)
                        : "v126");
                // The multiply-adds by the twiddle and by h2 are plain C++: the compiler selects v_mad_u64_u32 itself for
                // zext(a) * zext(b) + c64 and -- unlike behind an inline-asm statement, where it assumes a forwarding hazard
                // and puts an s_nop in front of every dependent instruction -- knows that nothing has to wait (register-
                // only loop, whole chain in C++: 71.2 -> 67.8 SIMD cycles per butterfly at 4 waves per SIMD, 76.6 -> 71.1
                // at 2, 108.9 -> 84.2 at 1; tools/ubench_bfly, round 5).  `one` is an opaque 1 (+ h2 as a multiply-add:
                // adding a 32-bit value to a 64-bit one otherwise costs a zero-extending move plus a 64-bit add).
                // The three multiply-adds by the words of -q stay asm (they also keep the cross-term accumulator whole: with
                // only its low word live the compiler narrows the chain to v_mul_lo_u32 + v_add3_u32): the
                // compiler's divergence analysis loses the modulus of some kernels to vector registers and then multiplies
                // by a 64-bit word (two multiply-adds and two moves each).
                qh = static_cast<uint64_t>(h2) * one + qh;
                uint64_t c = static_cast<uint64_t>(x0) * hi32(t.w);
                c = static_cast<uint64_t>(x1) * lo32(t.w) + c;
                c = mad32<!VQ>(lo32(qh), hi32(qneg), c);
                c = mad32<!VQ>(hi32(qh), lo32(qneg), c);
                uint64_t a = ZERO ? static_cast<uint64_t>(x0) * lo32(t.w) : static_cast<uint64_t>(x0) * lo32(t.w) + acc;
                // the cross sum goes into the accumulator's high word BEFORE the last multiply-add, so the result
                // leaves the chain as one 64-bit register pair (with the add last, the compiler started the
                // butterfly's 64-bit subtraction on the halves: a third instruction in a third of the butterflies)
                uint32_t ah;
                asm("v_add_u32 %0, %1, %2" : "=v"(ah) : "v"(hi32(a)), "v"(lo32(c)));
                a = (static_cast<uint64_t>(ah) << 32) | lo32(a);
                return mad32<!VQ>(lo32(qh), lo32(qneg), a);
            }
            // 2 x + k (k wave-uniform unless VQ): one v_lshl_add_u64
            __device__ __forceinline__ uint64_t shl1_add(uint64_t x, uint64_t k) const
            {
                uint64_t d;
                if constexpr (VQ)
                    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(d) : "v"(x), "v"(k));
                else
                    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(d) : "v"(x), "s"(k));
                return d;
            }
            // x * w  (mod q), any x < 2^64, result in [0, 4q)
            template <bool UNI = false> __device__ __forceinline__ uint64_t mul(uint64_t x, const Tw64& t) const
            {
                return mul_acc<UNI, true>(x, t, 0);
            }

            // if (x >= k*q) x -= k*q   -- 4 instructions: v_lshl_add_u64 with the negated constant
            // (kept opaque so it is not re-canonicalised into a carry-chained subtract, which
            // costs an extra select and VCC hazard nops), one 64-bit compare, two selects
            template <int K> __device__ __forceinline__ uint64_t csub(uint64_t x) const
            {
                const uint64_t m = kq(K);
                const uint64_t negm = 0 - m;
                uint64_t d;
                if constexpr (VQ)
                    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(x), "v"(negm));
                else
                    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(x), "s"(negm));
                return (x >= m) ? d : x;
            }
        };

        template <> struct Mod<uint64_t, 0, false> : Mod64<16>
        {
        };
        // 31 q < 2^64 (every prime of the reference's 64-bit pools: 2^59 + small): twice the headroom, forward
        // transforms correct the range every fourth stage instead of every second (host-side switch, lim = 31)
        template <> struct Mod<uint64_t, 31, false> : Mod64<31>
        {
        };
        template <> struct Mod<uint64_t, 8, false> : Mod64<8>
        {
        };
        template <> struct Mod<uint64_t, 4, false> : Mod64<4>
        {
        };
        // per-lane moduli.  PerCoefficient layout (strided passes): one family for the whole documented domain (<= 62 bit),
        // the 4 q range.  RNS stacks of rings below one tile (contiguous passes, several polynomials per wave): the same
        // three ranges as the uniform kernels, chosen by the go-flag / the plan
        template <> struct Mod<uint64_t, 4, true> : Mod64<4, true>
        {
        };
        template <> struct Mod<uint64_t, 0, true> : Mod64<16, true>
        {
        };
        template <> struct Mod<uint64_t, 8, true> : Mod64<8, true>
        {
        };

        // 64-bit register pair whose low word is `lo` and whose high word is anything at all (no instruction): the
        // accumulator a 32-bit multiply-add chain starts from
        __device__ __forceinline__ uint64_t pair_lo(uint32_t lo)
        {
            // (an empty asm per use: every pair gets a high half of its OWN that no instruction writes --
            // __builtin_nondeterministic_value is ONE frozen value per function, which the compiler then copies into the
            // odd register of every pair: 99 v_mov per wave in the single-sweep 32-bit kernel, round 5)
            uint32_t hi;
            asm("" : "=v"(hi) : "v"(lo));
            return (static_cast<uint64_t>(hi) << 32) | lo;
        }

        // ---- 32-bit: exact-quotient Shoup product in [0, 2q); q < 2^30 => LIMIT 4, q < 2^29 => LIMIT 8 --------
        // The product is ONE v_mul_hi_u32 (quotient) and TWO v_mad_u64_u32 chained through a 64-bit accumulator whose
        // high word is dead -- x * w, then + qh * (2^32 - q) -- and the chain starts from the butterfly's own U
        // (mul_acc), so U + T costs three instructions where mul_hi / mul_lo / mul_lo / sub / add costs five:
        // v_mad_u64_u32 issues at the full rate on gfx950 (tools/ubench_bfly32: forward butterfly 30.8 -> 22.6 SIMD
        // cycles, inverse 31.9 -> 28.0; profiles/ubench_bfly32_r05.txt).  VQ: per-lane moduli (q in a vector register).
        template <int LIM, bool VQ = false> struct Mod32
        {
            static constexpr int TB = 2;
            static constexpr int LIMIT = LIM;
            static constexpr int MAX_BIT = (LIM == 8) ? 29 : 30;
            uint32_t q;
            uint32_t qneg; // 2^32 - q
            uint32_t M;    // floor(2^32 / q) (norm_const_of)

            __device__ __forceinline__ void set(uint32_t modulus, const NormConst& n)
            {
                q = modulus;
                qneg = 0u - modulus;
                M = n.M;
            }
            // any x < 2^32 -> [0, 2q): v_mul_hi_u32 + v_mad_u64_u32
            template <bool HI = false> __device__ __forceinline__ uint32_t reduce_2q(uint32_t x) const
            {
                return lo32(mad32<!VQ>(__umulhi(x, M), qneg, pair_lo(x)));
            }
            __device__ __forceinline__ uint32_t kq(int k) const { return q * static_cast<uint32_t>(k); }
            __device__ __forceinline__ bool hi_norm() const { return false; }

            // UNI: the twiddle is wave-uniform (scalar registers)
            template <bool UNI = false> __device__ __forceinline__ uint32_t mul(uint32_t x, const Tw32& t) const
            {
                const uint32_t qh = __umulhi(x, t.wp);
                return lo32(mad32<!VQ>(qh, qneg, mad32z<UNI>(x, t.w)));
            }
            // The COMPLEMENT ~r of the product r = x * w (mod q) + {0, 1} q, for the same three instructions: with
            // wneg = 2^32 - w,  qh * q + (x * wneg - 1)  =  -(x * w - qh * q) - 1  (mod 2^32).  A Gentleman-Sande butterfly
            // whose second operand arrives complemented is two instructions instead of three (merge_e32_kernels.hpp).
            template <bool UNI = false> __device__ __forceinline__ uint32_t mulc(uint32_t x, uint32_t wneg, uint32_t wp) const
            {
                const uint32_t qh = __umulhi(x, wp);
                return lo32(mad32<!VQ>(qh, q, mad32m1<UNI>(x, wneg)));
            }
            // the conditional subtraction of k q on a complemented value: ~min(v, v - kq) = max(~v, ~v + kq)
            template <int K> __device__ __forceinline__ uint32_t csub_c(uint32_t w) const
            {
                const uint32_t d = w + kq(K);
                return d > w ? d : w;
            }
            // acc + T, T = x * w (mod q) + {0, 1} q
            template <bool UNI, bool ZERO = false>
            __device__ __forceinline__ uint32_t mul_acc(uint32_t x, const Tw32& t, uint32_t acc) const
            {
                const uint32_t qh = __umulhi(x, t.wp);
                const uint64_t a = ZERO ? mad32z<UNI>(x, t.w) : mad32<UNI>(x, t.w, pair_lo(acc));
                return lo32(mad32<!VQ>(qh, qneg, a));
            }
            // 2 x + k (k wave-uniform unless VQ): one v_lshl_add_u32
            __device__ __forceinline__ uint32_t shl1_add(uint32_t x, uint32_t k) const
            {
                uint32_t d;
                if constexpr (VQ)
                    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(d) : "v"(x), "v"(k));
                else
                    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(d) : "v"(x), "s"(k));
                return d;
            }

            // x < 2*k*q:  min(x, x - k*q) as unsigned (the difference wraps above x when x < k*q)
            template <int K> __device__ __forceinline__ uint32_t csub(uint32_t x) const
            {
                const uint32_t d = x - kq(K);
                return d < x ? d : x;
            }
        };
        template <> struct Mod<uint32_t, 0, false> : Mod32<4>
        {
        };
        template <> struct Mod<uint32_t, 0, true> : Mod32<4, true> // per-lane moduli
        {
        };
        // moduli below 2^29 (the reference's 32-bit pools: 469762049, ...): twice the headroom, a range correction
        // every third forward stage instead of every stage (host-side switch, lim = 8; both directions)
        template <> struct Mod<uint32_t, 8, false> : Mod32<8>
        {
        };

        // [0, B*q) -> [0, q)
        // HI (64-bit words): the caller found nc.hi set (wave-uniform) -- reduce_2q reads the high word only
        template <int B, bool HI = false, typename MM, typename T> __device__ __forceinline__ T normalize(const MM& m, T x)
        {
            if constexpr (sizeof(T) == 8 && B > 4)
                return m.template csub<1>(m.template reduce_2q<HI>(x)); // 9-10 instructions instead of 4 x 4
            if constexpr (sizeof(T) == 4 && B > 4)
                return m.template csub<1>(m.template reduce_2q<HI>(x)); // 4 instructions instead of 3 x 2
            if constexpr (B > 8)
                x = m.template csub<8>(x);
            if constexpr (B > 4)
                x = m.template csub<4>(x);
            if constexpr (B > 2)
                x = m.template csub<2>(x);
            if constexpr (B > 1)
                x = m.template csub<1>(x);
            return x;
        }

        // ---- compile-time range bookkeeping (units of q) --------------------------------
        constexpr int ceil_pow2(int v)
        {
            int p = 1;
            while (p < v)
                p <<= 1;
            return p;
        }
        // conditional subtraction amount that halves a bound b (b <= 2k): k = ceil_pow2(b) / 2
        constexpr int csub_k(int b) { return ceil_pow2(b) / 2; }

        // Cooley-Tukey (forward) butterfly plan for an input bound bu under `limit`:
        //   U' = U + T, V' = U - T + tb*q  with  T = V*w in [0, tb*q)
        struct CtPlan
        {
            int ku;  // 0, or conditional-subtract k*q from U first
            int out; // bound of both outputs
        };
        constexpr CtPlan ct_plan(int bu, int limit, int tb)
        {
            CtPlan p{0, 0};
            if (bu + tb > limit)
            {
                p.ku = csub_k(bu);
                bu = p.ku;
            }
            p.out = bu + tb;
            return p;
        }

        // Gentleman-Sande (inverse) butterfly plan:
        //   S = U + V, U' = S (- ko*q if S >= ko*q), V' = (U - V + c*q) * w  with  c >= bound(V), V' in [0, tb*q)
        // Stored values are kept at or below limit/2 by correcting the SUM: one conditional subtraction per
        // butterfly whose sum can pass limit/2, none on its inputs (inputs above limit/2 only come from a
        // conservative hand-off bound and are halved first).  Pairs of products (4 + 4 = 8 <= limit/2) need
        // none at all, so a saturated schedule spends 0.5 corrections per butterfly -- correcting the inputs
        // instead (round 1) needed both of them for every pair of sums: 0.89 per butterfly over a 2^16 transform.
        struct GsPlan
        {
            int ku, kv; // conditional subtractions applied to the inputs first (0 = none)
            int c;      // offset multiple
            int ko;     // conditional subtraction applied to the sum (0 = none)
            int out_u;  // bound of U'
        };
        constexpr GsPlan gs_plan(int bu, int bv, int limit)
        {
            GsPlan p{0, 0, 0, 0, 0};
            const int half = limit / 2;
            if (bu > half)
            {
                p.ku = csub_k(bu);
                bu = p.ku;
            }
            if (bv > half)
            {
                p.kv = csub_k(bv);
                bv = p.kv;
            }
            p.c = ceil_pow2(bv); // U + c q - V < (bu + c) q <= limit q
            int out = bu + bv;   // <= limit q: no wrap
            if (out > half)
            {
                p.ko = csub_k(out);
                out = p.ko;
            }
            p.out_u = out;
            return p;
        }
    } // namespace lazy
} // namespace gpuntt
