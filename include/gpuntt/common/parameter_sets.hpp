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

} // namespace gpuntt
