// modarith.hpp -- device-side modular arithmetic cores used by the NTT kernels (gfx950).
//
// Arithmetic of the GENERIC kernels (merge_kernels.hpp): the public OPERATOR_GPU<T> -- the reference's
// {value, bit, mu} Barrett contract (reference src/include/gpuntt/common/modular_arith.cuh:312-339),
// which needs nothing but the caller's Modulus<T> and the caller's plain twiddle table.  The fast kernels use
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

        // per-polynomial modulus context held in registers / SGPRs.  The arithmetic IS the public
        // device class OPERATOR_GPU<T> (include/gpuntt/common/modular_arith.cuh, the reference's
        // modular_arith.cuh:174-454 surface): one implementation, exercised by every test that runs
        // the generic kernels.
        template <typename T> struct ModCtx
        {
            T q, bit, mu;
            __device__ __forceinline__ Modulus<T> modulus() const
            {
                Modulus<T> m;
                m.value = q;
                m.bit = bit;
                m.mu = mu;
                return m;
            }
            __device__ __forceinline__ T mul(T a, T b) const { return OPERATOR_GPU<T>::mult(a, b, modulus()); }
            __device__ __forceinline__ T add(T a, T b) const { return OPERATOR_GPU<T>::add(a, b, modulus()); }
            __device__ __forceinline__ T sub(T a, T b) const { return OPERATOR_GPU<T>::sub(a, b, modulus()); }
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
