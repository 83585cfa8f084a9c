// nttparameters.cpp -- host-side parameter and twiddle-table generation.
//
// Own implementation of the reference's NTTParameters<T> / NTTParameters4Step<T>
// (src/lib/common/nttparameters.cu:22-471): same built-in prime / root pools, same table
// contents and orders, so tables built here are interchangeable with the reference's.
// Differences are speed only: powers are produced by running products (the W matrix row by
// row instead of one modular exponentiation per entry, which takes minutes at N = 2^24 in
// the reference) -- exact modular arithmetic makes the values identical.
#include <stdexcept>

#include "gpuntt/common/nttparameters.cuh"

namespace gpuntt
{
    int bitreverse(int index, int n_power)
    {
        int r = 0;
        for (int i = 0; i < n_power; i++)
        {
            r = (r << 1) | (index & 1);
            index >>= 1;
        }
        return r;
    }

    namespace
    {
        template <typename T> struct MergePool;
        template <> struct MergePool<Data32> // nttparameters.cu:86-93,102-109,121-128
        {
            static constexpr Data32 q = 469762049u, w = 900u, psi = 30u;
            static constexpr int top = 25;
        };
        template <> struct MergePool<Data64> // nttparameters.cu:94-98,111-118,130-141
        {
            static constexpr Data64 q = 576460756061519873ULL, w = 229929041166717729ULL,
                                    psi = 4517306222ULL;
            static constexpr int top = 28;
        };

        template <typename T> std::vector<T> powers(T root, size_t count, const Modulus<T>& m)
        {
            std::vector<T> t(count);
            if (count == 0)
                return t;
            t[0] = 1;
            for (size_t i = 1; i < count; i++)
                t[i] = OPERATOR<T>::mult(t[i - 1], root, m);
            return t;
        }

        template <typename T> std::vector<T> bitrev_copy(const std::vector<T>& table)
        {
            int lg = 0;
            while ((size_t(1) << lg) < table.size())
                lg++;
            std::vector<T> out(table.size());
            for (size_t i = 0; i < table.size(); i++)
                out[i] = table[bitreverse(static_cast<int>(i), lg)];
            return out;
        }

        // 4-step pools, nttparameters.cu:229-303 (logn 12..24) and shapes :305-354
        const Data32 kPrimes32[13] = {268460033, 268582913, 268664833, 268369921, 269221889,
                                      269221889, 270532609, 270532609, 270532609, 377487361,
                                      377487361, 469762049, 469762049};
        const Data32 kOmega32[13] = {36747374, 249229369, 4092529, 175218169, 10653696, 238764304,
                                     240100,   23104,     179776,  19321,     38809,    1600,
                                     169};
        const Data32 kPsi32[13] = {77090, 15787, 2023, 13237, 3264, 15452, 490,
                                   152,   424,   139,  197,   40,   13};
        const Data64 kPrimes64[13] = {576460752303415297ULL, 576460752303439873ULL,
                                      576460752304439297ULL, 576460752308273153ULL,
                                      576460752308273153ULL, 576460752315482113ULL,
                                      576460752315482113ULL, 576460752340123649ULL,
                                      576460752364240897ULL, 576460752475389953ULL,
                                      576460752597024769ULL, 576460753024843777ULL,
                                      576460753175838721ULL};
        const Data64 kOmega64[13] = {288482366111684746ULL, 37048445140799662ULL,
                                     459782973201979845ULL, 64800917766465203ULL,
                                     425015386842055933ULL, 18734847765732801ULL,
                                     119109113519742895ULL, 227584740857897520ULL,
                                     477282059544659462ULL, 570131728462077067ULL,
                                     433594414095420776ULL, 219263994987749328ULL,
                                     189790554094222112ULL};
        const Data64 kPsi64[13] = {238394956950829ULL, 54612008597396ULL, 8242615629351ULL,
                                   16141297350887ULL,  3760097055997ULL,  11571974431275ULL,
                                   328867687796ULL,    2298846063117ULL,  731868219707ULL,
                                   409596963254ULL,    189266227206ULL,   31864818375ULL,
                                   92067739764ULL};
        const int kShape[13][2] = {{32, 128},   {32, 256},   {32, 512},   {64, 512},   {128, 512},
                                   {32, 4096},  {32, 8192},  {32, 16384}, {32, 32768}, {64, 32768},
                                   {128, 32768}, {128, 65536}, {256, 65536}};

        template <typename T> void pool4(int logn, T& q, T& w, T& psi);
        template <> void pool4<Data32>(int logn, Data32& q, Data32& w, Data32& psi)
        {
            q = kPrimes32[logn - 12];
            w = kOmega32[logn - 12];
            psi = kPsi32[logn - 12];
        }
        template <> void pool4<Data64>(int logn, Data64& q, Data64& w, Data64& psi)
        {
            q = kPrimes64[logn - 12];
            w = kOmega64[logn - 12];
            psi = kPsi64[logn - 12];
        }
    } // namespace

    // ------------------------------------------------------------------ NTTParameters ----
    template <typename T> void NTTParameters<T>::build_tables()
    {
        root_of_unity = (poly_reduction == ReductionPolynomial::X_N_minus) ? omega : psi;
        inverse_root_of_unity = OPERATOR<T>::modinv(root_of_unity, modulus);
        root_of_unity_size = (poly_reduction == ReductionPolynomial::X_N_minus)
                                 ? static_cast<T>(T(1) << (logn - 1))
                                 : static_cast<T>(T(1) << logn);
        forward_root_of_unity_table = powers<T>(root_of_unity, root_of_unity_size, modulus);
        inverse_root_of_unity_table = powers<T>(inverse_root_of_unity, root_of_unity_size, modulus);
        n_inv = OPERATOR<T>::modinv(n, modulus);
    }

    template <typename T>
    NTTParameters<T>::NTTParameters(int LOGN, ReductionPolynomial poly_reduce_type)
    {
        customAssert(LOGN >= 1 && LOGN <= MergePool<T>::top, "LOGN is outside the built-in pool range.");
        logn = LOGN;
        n = static_cast<T>(T(1) << logn);
        poly_reduction = poly_reduce_type;
        modulus = Modulus<T>(MergePool<T>::q);
        const T e = static_cast<T>(T(1) << (MergePool<T>::top - logn));
        omega = OPERATOR<T>::exp(MergePool<T>::w, e, modulus);
        psi = OPERATOR<T>::exp(MergePool<T>::psi, e, modulus);
        build_tables();
    }

    template <typename T>
    NTTParameters<T>::NTTParameters(int LOGN, NTTFactors<T> ntt_factors,
                                    ReductionPolynomial poly_reduce_type)
    {
        customAssert(LOGN >= 1 && LOGN <= 28, "LOGN should be in range 1 to 28.");
        logn = LOGN;
        n = static_cast<T>(T(1) << logn);
        poly_reduction = poly_reduce_type;
        modulus = ntt_factors.modulus;
        omega = ntt_factors.omega;
        psi = ntt_factors.psi;
        build_tables();
    }

    template <typename T>
    NTTParameters<T>::NTTParameters()
        : logn(0), n(0), poly_reduction(X_N_minus), modulus(), omega(0), psi(0), n_inv(0),
          root_of_unity(0), inverse_root_of_unity(0), root_of_unity_size(0)
    {
    }

    template <typename T>
    std::vector<Root<T>> NTTParameters<T>::gpu_root_of_unity_table_generator(std::vector<T> table)
    {
        return bitrev_copy<T>(table);
    }

    // ------------------------------------------------------------- NTTParameters4Step ----
    template <typename T>
    NTTParameters4Step<T>::NTTParameters4Step(int LOGN, ReductionPolynomial poly_reduce_type)
    {
        customAssert(LOGN >= 12 && LOGN <= 24, "LOGN should be in range 12 to 24.");
        logn = LOGN;
        n = static_cast<T>(T(1) << logn);
        poly_reduction = poly_reduce_type;
        T q, w, p;
        pool4<T>(logn, q, w, p);
        modulus = Modulus<T>(q);
        omega = w;
        psi = p;
        root_of_unity = (poly_reduce_type == ReductionPolynomial::X_N_minus) ? omega : psi;
        inverse_root_of_unity = OPERATOR<T>::modinv(root_of_unity, modulus);
        root_of_unity_size = (poly_reduce_type == ReductionPolynomial::X_N_minus)
                                 ? static_cast<T>(T(1) << (logn - 1))
                                 : static_cast<T>(T(1) << logn);
        n1 = kShape[logn - 12][0];
        n2 = kShape[logn - 12][1];
        int lg1 = 0, lg2 = 0;
        while ((1 << lg1) < n1)
            lg1++;
        while ((1 << lg2) < n2)
            lg2++;

        // small tables: powers of root^(n/n1), root^(n/n2) and of their inverses
        const T r1 = OPERATOR<T>::exp(root_of_unity, static_cast<T>(n / n1), modulus);
        const T r2 = OPERATOR<T>::exp(root_of_unity, static_cast<T>(n / n2), modulus);
        n1_based_root_of_unity_table = powers<T>(r1, n1 >> 1, modulus);
        n2_based_root_of_unity_table = powers<T>(r2, n2 >> 1, modulus);
        n1_based_inverse_root_of_unity_table =
            powers<T>(OPERATOR<T>::modinv(r1, modulus), n1 >> 1, modulus);
        n2_based_inverse_root_of_unity_table =
            powers<T>(OPERATOR<T>::modinv(r2, modulus), n2 >> 1, modulus);

        // W[i*n2 + j]     = root^(bitreverse(i, lg n1) * j)
        W_root_of_unity_table.resize(static_cast<size_t>(n));
        for (int i = 0; i < n1; i++)
        {
            const T base = OPERATOR<T>::exp(root_of_unity, static_cast<T>(bitreverse(i, lg1)), modulus);
            T cur = 1;
            T* rowp = W_root_of_unity_table.data() + static_cast<size_t>(i) * n2;
            for (int j = 0; j < n2; j++)
            {
                rowp[j] = cur;
                cur = OPERATOR<T>::mult(cur, base, modulus);
            }
        }
        // W_inv[i*n2 + j] = inv_root^(bitreverse(j, lg n2) * i) = (inv_root^i)^bitreverse(j)
        W_inverse_root_of_unity_table.resize(static_cast<size_t>(n));
        std::vector<T> rowpow(static_cast<size_t>(n2));
        T base = 1; // inv_root^i
        for (int i = 0; i < n1; i++)
        {
            rowpow[0] = 1;
            for (int e = 1; e < n2; e++)
                rowpow[e] = OPERATOR<T>::mult(rowpow[e - 1], base, modulus);
            T* rowp = W_inverse_root_of_unity_table.data() + static_cast<size_t>(i) * n2;
            for (int j = 0; j < n2; j++)
                rowp[j] = rowpow[bitreverse(j, lg2)];
            base = OPERATOR<T>::mult(base, inverse_root_of_unity, modulus);
        }

        n_inv = OPERATOR<T>::modinv(n, modulus);
        n_inv_gpu = n_inv;
    }

    template <typename T>
    NTTParameters4Step<T>::NTTParameters4Step()
        : logn(0), n(0), poly_reduction(X_N_minus), modulus(), omega(0), psi(0), n_inv(0),
          n_inv_gpu(0), root_of_unity(0), inverse_root_of_unity(0), root_of_unity_size(0), n1(0), n2(0)
    {
    }

    template <typename T>
    std::vector<Root<T>>
    NTTParameters4Step<T>::gpu_root_of_unity_table_generator(std::vector<T> table)
    {
        return bitrev_copy<T>(table);
    }

    template class NTTParameters<Data32>;
    template class NTTParameters<Data64>;
    template class NTTParameters4Step<Data32>;
    template class NTTParameters4Step<Data64>;
} // namespace gpuntt
