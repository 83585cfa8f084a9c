// ntt_cpu.cpp -- host-side transforms shipped with the library (NTTCPU<T>, NTT_4STEP_CPU<T>,
// schoolbook_poly_multiplication<T>): the same public classes the reference library exports
// (src/lib/ntt_merge/ntt_cpu.cu:10-185, src/lib/ntt_4step/ntt_4step_cpu.cu:10-299) so caller
// code that builds expected values with them keeps working.  They are plain host utilities;
// no GPU entry point ever falls back to them.
#include <stdexcept>

#include "gpuntt/ntt_4step/ntt_4step_cpu.cuh"
#include "gpuntt/ntt_merge/ntt_cpu.cuh"

namespace gpuntt
{
    namespace
    {
        // in-place radix-2 CT (natural in, bit-reversed out); tw(s, i) returns the twiddle of
        // group i at stage s (m = 2^s groups)
        template <typename T, typename TW>
        void ct_inplace(T* a, int logn, const Modulus<T>& q, TW tw)
        {
            const size_t n = size_t(1) << logn;
            size_t t = n;
            for (size_t m = 1; m < n; m <<= 1)
            {
                t >>= 1;
                for (size_t i = 0; i < m; i++)
                {
                    const T S = tw(m, i);
                    T* lo = a + 2 * i * t;
                    T* hi = lo + t;
                    for (size_t j = 0; j < t; j++)
                    {
                        const T U = lo[j];
                        const T V = OPERATOR<T>::mult(hi[j], S, q);
                        lo[j] = OPERATOR<T>::add(U, V, q);
                        hi[j] = OPERATOR<T>::sub(U, V, q);
                    }
                }
            }
        }

        // in-place radix-2 GS (bit-reversed in, natural out, unscaled)
        template <typename T, typename TW>
        void gs_inplace(T* a, int logn, const Modulus<T>& q, TW tw)
        {
            const size_t n = size_t(1) << logn;
            size_t t = 1;
            for (size_t m = n; m > 1; m >>= 1)
            {
                const size_t h = m >> 1;
                for (size_t i = 0; i < h; i++)
                {
                    const T S = tw(h, i);
                    T* lo = a + 2 * i * t;
                    T* hi = lo + t;
                    for (size_t j = 0; j < t; j++)
                    {
                        const T U = lo[j];
                        const T V = hi[j];
                        lo[j] = OPERATOR<T>::add(U, V, q);
                        hi[j] = OPERATOR<T>::mult(OPERATOR<T>::sub(U, V, q), S, q);
                    }
                }
                t <<= 1;
            }
        }

        template <typename T> std::vector<T> transposed(const std::vector<T>& v, int rows, int cols)
        {
            std::vector<T> o(v.size());
            for (int i = 0; i < rows; i++)
                for (int j = 0; j < cols; j++)
                    o[size_t(j) * rows + i] = v[size_t(i) * cols + j];
            return o;
        }

        inline int ilog2(size_t v)
        {
            int l = 0;
            while ((size_t(1) << l) < v)
                l++;
            return l;
        }
    } // namespace

    template <typename T>
    std::vector<T> schoolbook_poly_multiplication(std::vector<T> a, std::vector<T> b,
                                                  Modulus<T> modulus,
                                                  ReductionPolynomial reduction_poly)
    {
        const size_t len = a.size();
        std::vector<T> wide(2 * len, 0);
        for (size_t i = 0; i < len; i++)
            for (size_t j = 0; j < len; j++)
                wide[i + j] = OPERATOR<T>::add(wide[i + j], OPERATOR<T>::mult(a[i], b[j], modulus),
                                               modulus);
        std::vector<T> r(len);
        if (reduction_poly == ReductionPolynomial::X_N_minus)
            for (size_t i = 0; i < len; i++)
                r[i] = OPERATOR<T>::add(wide[i], wide[i + len], modulus);
        else if (reduction_poly == ReductionPolynomial::X_N_plus)
            for (size_t i = 0; i < len; i++)
                r[i] = OPERATOR<T>::sub(wide[i], wide[i + len], modulus);
        else
            throw std::runtime_error("Poly reduction type is not supported!");
        return r;
    }

    template std::vector<Data32> schoolbook_poly_multiplication<Data32>(std::vector<Data32>,
                                                                        std::vector<Data32>,
                                                                        Modulus<Data32>,
                                                                        ReductionPolynomial);
    template std::vector<Data64> schoolbook_poly_multiplication<Data64>(std::vector<Data64>,
                                                                        std::vector<Data64>,
                                                                        Modulus<Data64>,
                                                                        ReductionPolynomial);

    // --------------------------------------------------------------------- NTTCPU ----
    template <typename T> NTTCPU<T>::NTTCPU(NTTParameters<T> parameters_) : parameters(parameters_) {}

    template <typename T> std::vector<T> NTTCPU<T>::mult(std::vector<T>& input1, std::vector<T>& input2)
    {
        std::vector<T> out(static_cast<size_t>(parameters.n));
        for (size_t i = 0; i < out.size(); i++)
            out[i] = OPERATOR<T>::mult(input1[i], input2[i], parameters.modulus);
        return out;
    }

    template <typename T> std::vector<T> NTTCPU<T>::ntt(std::vector<T>& input)
    {
        std::vector<T> out = input;
        const auto& tab = parameters.forward_root_of_unity_table;
        const int logn = parameters.logn;
        if (parameters.poly_reduction == ReductionPolynomial::X_N_minus)
            ct_inplace<T>(out.data(), logn, parameters.modulus,
                          [&](size_t, size_t i) { return tab[bitreverse(int(i), logn - 1)]; });
        else
            ct_inplace<T>(out.data(), logn, parameters.modulus,
                          [&](size_t m, size_t i) { return tab[bitreverse(int(m + i), logn)]; });
        return out;
    }

    template <typename T> std::vector<T> NTTCPU<T>::intt(std::vector<T>& input)
    {
        std::vector<T> out = input;
        const auto& tab = parameters.inverse_root_of_unity_table;
        const int logn = parameters.logn;
        if (parameters.poly_reduction == ReductionPolynomial::X_N_minus)
            gs_inplace<T>(out.data(), logn, parameters.modulus,
                          [&](size_t, size_t i) { return tab[bitreverse(int(i), logn - 1)]; });
        else
            gs_inplace<T>(out.data(), logn, parameters.modulus,
                          [&](size_t h, size_t i) { return tab[bitreverse(int(h + i), logn)]; });
        for (auto& x : out)
            x = OPERATOR<T>::mult(x, parameters.n_inv, parameters.modulus);
        return out;
    }

    template class NTTCPU<Data32>;
    template class NTTCPU<Data64>;

    // -------------------------------------------------------------- NTT_4STEP_CPU ----
    template <typename T>
    NTT_4STEP_CPU<T>::NTT_4STEP_CPU(NTTParameters4Step<T> parameters_) : parameters(parameters_)
    {
    }

    template <typename T>
    std::vector<T> NTT_4STEP_CPU<T>::mult(std::vector<T>& input1, std::vector<T>& input2)
    {
        std::vector<T> out(static_cast<size_t>(parameters.n));
        for (size_t i = 0; i < out.size(); i++)
            out[i] = OPERATOR<T>::mult(input1[i], input2[i], parameters.modulus);
        return out;
    }

    template <typename T>
    std::vector<T> NTT_4STEP_CPU<T>::intt_first_transpose(const std::vector<T>& input)
    {
        // flat[i*n2 + j] = input[i + j*n1]
        return transposed<T>(input, parameters.n2, parameters.n1);
    }

    template <typename T> std::vector<T> NTT_4STEP_CPU<T>::ntt(std::vector<T>& input)
    {
        const int n1 = parameters.n1, n2 = parameters.n2;
        const int l1 = ilog2(n1), l2 = ilog2(n2);
        const auto& q = parameters.modulus;
        std::vector<T> a = transposed<T>(input, n1, n2); // n2 rows of n1
        const auto& t1 = parameters.n1_based_root_of_unity_table;
        for (int r = 0; r < n2; r++)
            ct_inplace<T>(a.data() + size_t(r) * n1, l1, q,
                          [&](size_t, size_t i) { return t1[bitreverse(int(i), l1 - 1)]; });
        std::vector<T> b = transposed<T>(a, n2, n1); // n1 rows of n2
        for (size_t i = 0; i < b.size(); i++)
            b[i] = OPERATOR<T>::mult(b[i], parameters.W_root_of_unity_table[i], q);
        const auto& t2 = parameters.n2_based_root_of_unity_table;
        for (int r = 0; r < n1; r++)
            ct_inplace<T>(b.data() + size_t(r) * n2, l2, q,
                          [&](size_t, size_t i) { return t2[bitreverse(int(i), l2 - 1)]; });
        return transposed<T>(b, n1, n2);
    }

    template <typename T> std::vector<T> NTT_4STEP_CPU<T>::intt(std::vector<T>& input)
    {
        const int n1 = parameters.n1, n2 = parameters.n2;
        const int l1 = ilog2(n1), l2 = ilog2(n2);
        const auto& q = parameters.modulus;
        std::vector<T> a = intt_first_transpose(input); // n2 rows of n1
        const auto& t1 = parameters.n1_based_inverse_root_of_unity_table;
        for (int r = 0; r < n2; r++)
            gs_inplace<T>(a.data() + size_t(r) * n1, l1, q,
                          [&](size_t, size_t i) { return t1[bitreverse(int(i), l1 - 1)]; });
        std::vector<T> b = transposed<T>(a, n2, n1);
        for (size_t i = 0; i < b.size(); i++)
            b[i] = OPERATOR<T>::mult(b[i], parameters.W_inverse_root_of_unity_table[i], q);
        const auto& t2 = parameters.n2_based_inverse_root_of_unity_table;
        for (int r = 0; r < n1; r++)
            gs_inplace<T>(b.data() + size_t(r) * n2, l2, q,
                          [&](size_t, size_t i) { return t2[bitreverse(int(i), l2 - 1)]; });
        std::vector<T> out = transposed<T>(b, n1, n2);
        for (auto& x : out)
            x = OPERATOR<T>::mult(x, parameters.n_inv, q);
        return out;
    }

    template class NTT_4STEP_CPU<Data32>;
    template class NTT_4STEP_CPU<Data64>;
} // namespace gpuntt
