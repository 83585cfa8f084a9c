// gpuntt/common/modular_arith.cuh -- Modulus<T> and Barrett-compatible modular arithmetic.
//
// Same public surface as the reference header of this include path
// (src/include/gpuntt/common/modular_arith.cuh): typedefs, Modulus<T>{value,bit,mu},
// OPERATOR<T> (host) and OPERATOR_GPU<T> (device), Root<T>, Ninverse<T>.
// Results are the canonical residues the reference's Barrett code produces for every
// supported modulus (<= 30 bit for Data32, <= 62 bit for Data64; reference :66-67).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <stdexcept>
#include <cstdint>
#include <type_traits>

typedef std::int32_t Data32s;
typedef std::uint32_t Data32;
typedef std::uint32_t Root32;
typedef std::uint32_t Ninverse32;

typedef std::int64_t Data64s;
typedef std::uint64_t Data64;
typedef std::uint64_t Root64;
typedef std::uint64_t Ninverse64;

namespace gpuntt_detail
{
    template <typename T> struct wide;
    template <> struct wide<Data32> { using type = Data64; };
    template <> struct wide<Data64> { using type = unsigned __int128; };

    template <typename T> __host__ __device__ constexpr int bit_length(T v)
    {
        int b = 0;
        while (v) { ++b; v >>= 1; }
        return b;
    }
} // namespace gpuntt_detail

// Layout contract (reference :28-57): three T words {value, bit, mu};
// bit = (T)(log2((double) q) + 1), mu = floor(2^(2*bit+1) / q).  `bit` is evaluated in double exactly as
// the reference does (:44-47): a modulus q >= 2^53 within rounding distance below a power of two (2^60 - 107,
// say) converts to that power of two and gets bit = 61, not its exact length 60.  Every kernel family accepts
// the over-stated width (Barrett shifts, lazy ranges and path selection all key on `bit`), and a caller that
// compares or serialises the three words sees the reference's values.
// ONE divergence from the reference's constructor, at the edge of / outside its documented domain (:66-67: "does not
// work modulus higher than 30 bit for Data32 ... 62 bit for Data64"): a modulus whose mu = floor(2^(2*bit+1) / q) does
// not fit the word is REFUSED with std::invalid_argument, where the reference stores the truncated quotient and
// computes wrong Barrett products with it from then on.  That concerns (a) Data64: a 61-bit prime within double
// rounding of 2^61 (it gets bit = 62); (b) Data32: EVERY modulus of 2^30 or more (bit >= 31 makes 2^(2*bit+1) / q >=
// 2^32) -- such a Modulus32 constructs in the reference (unusable there too) and throws here.  Callers that only
// want the three words of an out-of-domain modulus can fill the struct through the default constructor.
template <typename T1> struct Modulus
{
    T1 value;
    T1 bit;
    T1 mu;

    __host__ Modulus(T1 mod) : value(mod), bit(0), mu(0)
    {
        using T2 = typename gpuntt_detail::wide<T1>::type;
        if (mod != 0)
        {
            bit = static_cast<T1>(std::log2(static_cast<double>(mod)) + 1);
            const unsigned sh = static_cast<unsigned>(2 * bit + 1);
            if (sh < sizeof(T2) * 8) // (the reference shifts past the word for bit = 64: undefined; mu stays 0 here)
            {
                const T2 quot = (static_cast<T2>(1) << sh) / mod;
                // an over-stated width at the top of the domain (a 61-bit prime within double rounding of 2^61 gets
                // bit = 62) makes 2^(2 bit + 1) / q pass the word: the reference stores the truncated value and its
                // Barrett product is wrong from then on -- refused here instead of computed with
                if ((quot >> (sizeof(T1) * 8)) != 0)
                    throw std::invalid_argument("Invalid modulus! (2^(2*bit+1) / q does not fit the word)");
                mu = static_cast<T1>(quot);
            }
        }
    }
    __host__ __device__ Modulus() : value(0), bit(0), mu(0) {}
};

typedef Modulus<Data32> Modulus32;
typedef Modulus<Data64> Modulus64;

namespace modular_operation_cpu
{
    // host-side exact modular arithmetic (reference :62-158)
    template <typename T1> class BarrettOperations
    {
        using T2 = typename gpuntt_detail::wide<T1>::type;

      public:
        static __host__ T1 add(const T1& a, const T1& b, const Modulus<T1>& m)
        {
            T1 s = a + b;
            return (s >= m.value) ? (s - m.value) : s;
        }
        static __host__ T1 sub(const T1& a, const T1& b, const Modulus<T1>& m)
        {
            T1 d = a + m.value - b;
            return (d >= m.value) ? (d - m.value) : d;
        }
        static __host__ T1 mult(const T1& a, const T1& b, const Modulus<T1>& m)
        {
            return static_cast<T1>((static_cast<T2>(a) * static_cast<T2>(b)) % m.value);
        }
        static __host__ T1 exp(T1 base, T1 exponent, const Modulus<T1>& m)
        {
            T1 result = 1 % m.value;
            base = static_cast<T1>(base % m.value);
            while (exponent)
            {
                if (exponent & 1)
                    result = mult(result, base, m);
                base = mult(base, base, m);
                exponent >>= 1;
            }
            return result;
        }
        static __host__ T1 modinv(T1 a, const Modulus<T1>& m) // prime modulus: a^(q-2)
        {
            return exp(a, m.value - 2, m);
        }
        static __host__ T1 reduce(const T1& a, const Modulus<T1>& m) { return a % m.value; }
    };
} // namespace modular_operation_cpu

template <typename T> using OPERATOR = modular_operation_cpu::BarrettOperations<T>;
typedef OPERATOR<Data32> OPERATOR32;
typedef OPERATOR<Data64> OPERATOR64;

template <typename T>
using Root = typename std::conditional<std::is_same<T, Data32>::value, Root32, Root64>::type;
template <typename T>
using Ninverse =
    typename std::conditional<std::is_same<T, Data32>::value, Ninverse32, Ninverse64>::type;

namespace modular_operation_gpu
{
    // device-side arithmetic for caller kernels (reference :174-454).  mult() follows the
    // {value,bit,mu} Barrett contract so it needs no per-modulus precomputation; the
    // library's own NTT kernels use a cheaper precomputed-quotient form internally.
    template <typename T1> class BarrettOperations
    {
      public:
        static __device__ __forceinline__ T1 add(const T1& a, const T1& b, const Modulus<T1>& m)
        {
            T1 s = a + b;
            return (s >= m.value) ? (s - m.value) : s;
        }
        static __device__ __forceinline__ T1 sub(const T1& a, const T1& b, const Modulus<T1>& m)
        {
            T1 d = a + m.value - b;
            return (d >= m.value) ? (d - m.value) : d;
        }
        static __device__ __forceinline__ T1 mult(const T1& a, const T1& b,
                                                  const Modulus<T1>& m)
        {
            if constexpr (std::is_same<T1, Data32>::value)
            {
                Data64 z = static_cast<Data64>(a) * b;
                Data64 w = z >> (m.bit - 2);
                w = static_cast<Data64>(static_cast<Data32>(w)) * m.mu;
                w >>= (m.bit + 3);
                z -= static_cast<Data64>(static_cast<Data32>(w)) * m.value;
                Data32 r = static_cast<Data32>(z);
                return (r >= m.value) ? (r - m.value) : r;
            }
            else
            {
                const Data64 zlo = a * b, zhi = __umul64hi(a, b);
                const int s1 = static_cast<int>(m.bit) - 2; // 0 <= s1 <= 60
                Data64 w = (s1 == 0) ? zlo : ((zlo >> s1) | (zhi << (64 - s1)));
                const Data64 plo = w * m.mu, phi = __umul64hi(w, m.mu);
                const int s2 = static_cast<int>(m.bit) + 3; // 6 <= s2 <= 65
                w = (s2 >= 64) ? (phi >> (s2 - 64)) : ((plo >> s2) | (phi << (64 - s2)));
                Data64 r = zlo - w * m.value;
                return (r >= m.value) ? (r - m.value) : r;
            }
        }
        static __device__ __forceinline__ T1 reduce(const T1& a, const Modulus<T1>& m)
        {
            return mult(a, static_cast<T1>(1), m);
        }
        // signed input in (-q, q) -> [0, q)          (reference :372-385)
        static __device__ __forceinline__ T1
        reduce(const typename std::make_signed<T1>::type& a, const Modulus<T1>& m)
        {
            return (a < 0) ? static_cast<T1>(m.value + static_cast<T1>(a)) : static_cast<T1>(a);
        }
        // [0, q) -> centred representative, v > q/2 maps to v - q   (reference :389-405)
        static __device__ __forceinline__ typename std::make_signed<T1>::type
        centered_reduction(const T1& a, const Modulus<T1>& m)
        {
            using S = typename std::make_signed<T1>::type;
            return (a > (m.value >> 1)) ? static_cast<S>(a - m.value) : static_cast<S>(a);
        }
    };
} // namespace modular_operation_gpu

template <typename T> using OPERATOR_GPU = modular_operation_gpu::BarrettOperations<T>;
typedef OPERATOR_GPU<Data32> OPERATOR_GPU_32;
typedef OPERATOR_GPU<Data64> OPERATOR_GPU_64;
