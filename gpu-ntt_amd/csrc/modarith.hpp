// modarith.hpp -- device-side modular arithmetic cores used by the NTT kernels (gfx950).
//
// Arithmetic of the GENERIC kernels (merge_kernels.hpp): the reference's {value, bit, mu}
// Barrett contract (reference src/include/gpuntt/common/modular_arith.cuh:312-339), which needs
// nothing but the caller's Modulus<T> and the caller's plain twiddle table.  The fast kernels use
// the precomputed-quotient arithmetic of lazy.hpp instead; both produce canonical residues in
// [0, q), hence bit-identical transforms.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "gpuntt/common/modular_arith.cuh"

namespace gpuntt
{
    namespace dev
    {
        __device__ __forceinline__ uint32_t mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
        __device__ __forceinline__ uint64_t mulhi(uint64_t a, uint64_t b) { return __umul64hi(a, b); }

        template <typename T> __device__ __forceinline__ T add_mod(T a, T b, T q)
        {
            T s = a + b;
            return (s >= q) ? (s - q) : s;
        }
        template <typename T> __device__ __forceinline__ T sub_mod(T a, T b, T q)
        {
            T d = a - b;
            return (a < b) ? (d + q) : d;
        }

        // a*b mod q for a, b < q, canonical result; {q, bit, mu} as in Modulus<T>.
        __device__ __forceinline__ uint32_t barrett_mul(uint32_t a, uint32_t b, uint32_t q,
                                                        uint32_t bit, uint32_t mu)
        {
            uint64_t z = static_cast<uint64_t>(a) * b;
            uint64_t w = z >> (bit - 2);
            w = static_cast<uint64_t>(static_cast<uint32_t>(w)) * mu;
            w >>= (bit + 3);
            uint32_t r = static_cast<uint32_t>(z) - static_cast<uint32_t>(w) * q;
            return (r >= q) ? (r - q) : r;
        }
        __device__ __forceinline__ uint64_t barrett_mul(uint64_t a, uint64_t b, uint64_t q,
                                                        uint64_t bit, uint64_t mu)
        {
            const uint64_t zlo = a * b, zhi = __umul64hi(a, b);
            const int s1 = static_cast<int>(bit) - 2;
            uint64_t w = (s1 == 0) ? zlo : ((zlo >> s1) | (zhi << (64 - s1)));
            const uint64_t plo = w * mu, phi = __umul64hi(w, mu);
            const int s2 = static_cast<int>(bit) + 3;
            w = (s2 >= 64) ? (phi >> (s2 - 64)) : ((plo >> s2) | (phi << (64 - s2)));
            uint64_t r = zlo - w * q;
            return (r >= q) ? (r - q) : r;
        }

        // per-polynomial modulus context held in registers / SGPRs
        template <typename T> struct ModCtx
        {
            T q, bit, mu;
            __device__ __forceinline__ T mul(T a, T b) const { return barrett_mul(a, b, q, bit, mu); }
            __device__ __forceinline__ T add(T a, T b) const { return add_mod(a, b, q); }
            __device__ __forceinline__ T sub(T a, T b) const { return sub_mod(a, b, q); }
        };

        // Cooley-Tukey butterfly: (U, V) -> (U + V*w, U - V*w)   (reference ntt.cuh:69-78)
        template <typename T> __device__ __forceinline__ void ct_butterfly(T& U, T& V, T w, const ModCtx<T>& m)
        {
            T v = m.mul(V, w);
            T u = U;
            U = m.add(u, v);
            V = m.sub(u, v);
        }
        // Gentleman-Sande butterfly: (U, V) -> (U + V, (U - V)*w)   (reference ntt.cuh:80-92)
        template <typename T> __device__ __forceinline__ void gs_butterfly(T& U, T& V, T w, const ModCtx<T>& m)
        {
            T u = U, v = V;
            U = m.add(u, v);
            V = m.mul(m.sub(u, v), w);
        }
    } // namespace dev
} // namespace gpuntt
