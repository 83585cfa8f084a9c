// gpuntt/ntt_merge/ntt_cpu.cuh -- host reference transform shipped with the library, same
// surface as reference src/include/gpuntt/ntt_merge/ntt_cpu.cuh:13-33.
#pragma once

#include "gpuntt/common/nttparameters.cuh"

namespace gpuntt
{
    template <typename T>
    std::vector<T> schoolbook_poly_multiplication(std::vector<T> a, std::vector<T> b,
                                                  Modulus<T> modulus,
                                                  ReductionPolynomial reduction_poly);

    template <typename T> class NTTCPU
    {
      public:
        NTTParameters<T> parameters;
        NTTCPU(NTTParameters<T> parameters_);

        std::vector<T> mult(std::vector<T>& input1, std::vector<T>& input2);
        std::vector<T> ntt(std::vector<T>& input);  // natural in -> bit-reversed out
        std::vector<T> intt(std::vector<T>& input); // bit-reversed in -> natural out
    };
} // namespace gpuntt
