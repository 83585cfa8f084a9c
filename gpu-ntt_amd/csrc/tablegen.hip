// tablegen.hip -- root-of-unity tables built on the device (extension).
//
// The reference generates every table on the host and uploads it: NTTParameters<T> powers of psi / omega
// (src/lib/common/nttparameters.cu:120-189), NTTParameters4Step<T> n1 / n2 powers and the N-entry W matrix
// W[i*n2 + j] = root^(bitreverse(i) * j), inverse W[i*n2 + j] = root^-(bitreverse(j) * i) (:356-444, N modular
// exponentiations on one core, minutes at 2^24 in the reference, 2.6 s with this library's running products).
// Here every entry is root^e with e < 2^28 written as the product of the squares root^(2^k) of its set bits:
// the host makes the <= 28 squares with OPERATOR<T>::mult, a thread multiplies ~e.bit_count() of them with
// OPERATOR_GPU<T>::mult (canonical operands and results), so the words equal the host tables exactly.
#include <stdexcept>

#include "gpuntt/common/common.cuh"
#include "gpuntt/common/modular_arith.cuh"
#include "gpuntt/common/parameter_sets.hpp"
#include "launch.hpp"

namespace gpuntt
{
    namespace
    {
        constexpr int MAX_EXP_BITS = 28;
        template <typename T> struct Squares
        {
            T p[MAX_EXP_BITS]; // p[k] = base^(2^k)
        };

        template <typename T> __device__ __forceinline__ T power(const Squares<T>& sq, unsigned e, const Modulus<T>& m)
        {
            T r = 1;
            bool first = true;
#pragma unroll
            for (int k = 0; k < MAX_EXP_BITS; k++)
                if ((e >> k) & 1u)
                {
                    r = first ? sq.p[k] : OPERATOR_GPU<T>::mult(r, sq.p[k], m);
                    first = false;
                }
            return r;
        }

        __device__ __forceinline__ unsigned brev(unsigned v, int bits)
        {
            return bits == 0 ? 0u : (__brev(v) >> (32 - bits));
        }

        template <typename T>
        __global__ __launch_bounds__(256) void power_table(T* out, Squares<T> sq, Modulus<T> m, int log_count,
                                                          int bit_reversed)
        {
            const unsigned k = blockIdx.x * 256u + threadIdx.x;
            if (k >= (1u << log_count))
                return;
            out[k] = power(sq, bit_reversed ? brev(k, log_count) : k, m);
        }

        // forward: e = brev(i, l1) * j; inverse: e = brev(j, l2) * i  (i = row < n1, j = column < n2)
        template <typename T>
        __global__ __launch_bounds__(256) void fourstep_w_table(T* out, Squares<T> sq, Modulus<T> m, int l1, int l2,
                                                               int inverse)
        {
            const unsigned f = blockIdx.x * 256u + threadIdx.x; // f = i * n2 + j < 2^24
            const unsigned i = f >> l2, j = f & ((1u << l2) - 1u);
            const unsigned e = inverse ? brev(j, l2) * i : brev(i, l1) * j;
            out[f] = power(sq, e, m);
        }

        template <typename T> Squares<T> make_squares(T base, const Modulus<T>& m)
        {
            if (m.value < 2 || base >= m.value)
                throw std::invalid_argument("Invalid root / modulus!");
            Squares<T> sq;
            T cur = base;
            for (int k = 0; k < MAX_EXP_BITS; k++)
            {
                sq.p[k] = cur;
                cur = OPERATOR<T>::mult(cur, cur, m);
            }
            return sq;
        }
    } // namespace

    template <typename T>
    void GPU_GeneratePowerTable(T* device_out, T base, Modulus<T> modulus, int log_count, bool bit_reversed,
                                stream_t stream)
    {
        if (device_out == nullptr)
            throw std::invalid_argument("null pointer argument");
        if (log_count < 0 || log_count > MAX_EXP_BITS)
            throw std::invalid_argument("Invalid table size!");
        const Squares<T> sq = make_squares<T>(base, modulus);
        const unsigned count = 1u << log_count;
        GPUNTT_LAUNCH((power_table<T>), dim3((count + 255u) / 256u), dim3(256), 0, stream, device_out, sq, modulus,
                           log_count, bit_reversed ? 1 : 0);
        GPUNTT_HIP_CHECK(hipGetLastError());
    }

    template <typename T>
    void GPU_Generate4StepW(T* device_W, T root, Modulus<T> modulus, int n_power, type ntt_type, stream_t stream)
    {
        // n1 x n2 shapes, reference src/lib/common/nttparameters.cu:305-354
        static const int l1[13] = {5, 5, 5, 6, 7, 5, 5, 5, 5, 6, 7, 7, 8};
        if (device_W == nullptr)
            throw std::invalid_argument("null pointer argument");
        if (n_power < 12 || n_power > 24)
            throw std::invalid_argument("Invalid n_power range!");
        if (ntt_type != FORWARD && ntt_type != INVERSE)
            throw std::invalid_argument("Invalid ntt_type!");
        const int log_n1 = l1[n_power - 12], log_n2 = n_power - log_n1;
        const Squares<T> sq = make_squares<T>(root, modulus);
        GPUNTT_LAUNCH((fourstep_w_table<T>), dim3((1u << n_power) / 256u), dim3(256), 0, stream, device_W, sq,
                           modulus, log_n1, log_n2, ntt_type == INVERSE ? 1 : 0);
        GPUNTT_HIP_CHECK(hipGetLastError());
    }

    template void GPU_GeneratePowerTable<Data32>(Data32*, Data32, Modulus<Data32>, int, bool, stream_t);
    template void GPU_GeneratePowerTable<Data64>(Data64*, Data64, Modulus<Data64>, int, bool, stream_t);
    template void GPU_Generate4StepW<Data32>(Data32*, Data32, Modulus<Data32>, int, type, stream_t);
    template void GPU_Generate4StepW<Data64>(Data64*, Data64, Modulus<Data64>, int, type, stream_t);
} // namespace gpuntt
