// gpuntt/common/nttparameters.cuh -- transform descriptors and host-side parameter/table
// generators (public surface of reference src/include/gpuntt/common/nttparameters.cuh:17-170;
// generators restated in gpu-ntt_amd/csrc/nttparameters.cpp).
#pragma once

#include <vector>

#include "gpuntt/common/common.cuh"
#include "gpuntt/common/modular_arith.cuh"

namespace gpuntt
{
    int bitreverse(int index, int n_power);

    enum type { FORWARD, INVERSE };

    enum NTTLayout
    {
        PerPolynomial, // one transform per row of the (batch x N) matrix
        PerCoefficient // one transform per column
    };

    enum ReductionPolynomial
    {
        X_N_plus, // negacyclic, Z_q[X]/(X^N + 1): tables hold powers of psi (N entries)
        X_N_minus // cyclic,     Z_q[X]/(X^N - 1): tables hold powers of omega (N/2 entries)
    };

    template <typename T> struct NTTFactors
    {
        Modulus<T> modulus;
        T omega;
        T psi;
        __host__ NTTFactors(Modulus<T> q_, T omega_, T psi_) : modulus(q_), omega(omega_), psi(psi_) {}
        __host__ NTTFactors() : modulus(), omega(0), psi(0) {}
    };

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

} // namespace gpuntt
