// gpuntt/common/parameter_sets.hpp -- host-side parameter sets: prime pools, roots, N^-1 and the
// twiddle tables in natural order (public members of reference nttparameters.cuh:56-170, which
// the reference's examples read directly; the generators live in
// gpu-ntt_amd/csrc/nttparameters.cpp).
#pragma once

#include "gpuntt/common/descriptors.hpp"

namespace gpuntt
{
    // Merge-NTT parameters: built-in prime pool or caller-supplied NTTFactors.
    template <typename T> class NTTParameters
    {
      public:
        int logn;
        T n;
        ReductionPolynomial poly_reduction;
        Modulus<T> modulus;
        T omega;
        T psi;
        Ninverse<T> n_inv;
        T root_of_unity;
        T inverse_root_of_unity;
        T root_of_unity_size;
        std::vector<T> forward_root_of_unity_table; // natural order: root^0 .. root^(size-1)
        std::vector<T> inverse_root_of_unity_table;

        NTTParameters(int LOGN, ReductionPolynomial poly_reduce_type);
        NTTParameters(int LOGN, NTTFactors<T> ntt_factors, ReductionPolynomial poly_reduce_type);
        NTTParameters();

        // bit-reversed copy = the device-table order GPU_NTT / GPU_INTT expect
        std::vector<Root<T>> gpu_root_of_unity_table_generator(std::vector<T> table);

      private:
        void build_tables();
    };

    // 4-Step parameters (cyclic only, logn 12..24): per-size prime pool, n1 x n2 shape,
    // n1/n2 small tables (n/2 entries each, natural order) and the N-entry W twiddle matrix.
    template <typename T> class NTTParameters4Step
    {
      public:
        int logn;
        T n;
        ReductionPolynomial poly_reduction;
        Modulus<T> modulus;
        T omega;
        T psi;
        T n_inv;
        Ninverse<T> n_inv_gpu;
        T root_of_unity;
        T inverse_root_of_unity;
        T root_of_unity_size;

        int n1, n2;
        std::vector<T> n1_based_root_of_unity_table;
        std::vector<T> n2_based_root_of_unity_table;
        std::vector<T> W_root_of_unity_table;
        std::vector<T> n1_based_inverse_root_of_unity_table;
        std::vector<T> n2_based_inverse_root_of_unity_table;
        std::vector<T> W_inverse_root_of_unity_table;

        NTTParameters4Step(int LOGN, ReductionPolynomial poly_reduce_type);
        NTTParameters4Step();

        std::vector<Root<T>> gpu_root_of_unity_table_generator(std::vector<T> table);
    };

    // ---- extension: the tables above built on the device -----------------------------
    // The reference builds every table on the host and uploads it (nttparameters.cu:356-444; at 2^24 the W
    // matrix alone is N modular exponentiations and a 128 MiB copy).  These entry points write the same words
    // straight into device memory: one modular product per set exponent bit from the host-made squares
    // base^(2^k), OPERATOR_GPU<T>::mult throughout, so the tables equal the host-generated ones word for word.
    //
    //   GPU_GeneratePowerTable: out[k] = base^(bit_reversed ? bitreverse(k, log_count) : k), k < 2^log_count.
    //     bit_reversed = true with base = root_of_unity, log_count = log2(root_of_unity_size) is
    //     gpu_root_of_unity_table_generator(forward_root_of_unity_table) -- the table GPU_NTT takes; with the
    //     inverse root, the one GPU_INTT takes; with root^(n/n1), root^(n/n2) the 4-step n1 / n2 tables.
    //   GPU_Generate4StepW: FORWARD  W[i*n2 + j] = root^(bitreverse(i, log n1) * j)   (root = root_of_unity)
    //                       INVERSE  W[i*n2 + j] = root^(bitreverse(j, log n2) * i)   (root = inverse_root_of_unity)
    //     for the n1 x n2 shape of n_power (12 .. 24) -- W_root_of_unity_table / W_inverse_root_of_unity_table.
    template <typename T>
    void GPU_GeneratePowerTable(T* device_out, T base, Modulus<T> modulus, int log_count, bool bit_reversed,
                                stream_t stream);
    template <typename T>
    void GPU_Generate4StepW(T* device_W, T root, Modulus<T> modulus, int n_power, type ntt_type, stream_t stream);

} // namespace gpuntt
