// gpuntt/common/cpu_transforms.hpp -- the host transforms the library ships as a self-check for
// callers (and that the reference's examples use as their oracle): radix-2 Merge transform,
// 4-step transform, pointwise product and schoolbook ring multiplication.  Class and member
// names as in reference ntt_merge/ntt_cpu.cuh:13-33 and ntt_4step/ntt_4step_cpu.cuh:13-52;
// implementations in gpu-ntt_amd/csrc/ntt_cpu.cpp.
#pragma once

#include "gpuntt/common/parameter_sets.hpp"

namespace gpuntt
{
    // ---- Merge (one radix-2 pass per stage) ------------------------------------------------

    // a * b in Z_q[X]/(X^N -+ 1) by the O(N^2) definition
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

    // ---- 4-step (n1 x n2 decomposition, cyclic) ----------------------------------------------

    template <typename T> class NTT_4STEP_CPU
    {
      public:
        NTTParameters4Step<T> parameters;
        NTT_4STEP_CPU(NTTParameters4Step<T> parameters_);

        std::vector<T> mult(std::vector<T>& input1, std::vector<T>& input2);
        std::vector<T> ntt(std::vector<T>& input);
        std::vector<T> intt(std::vector<T>& input);
        std::vector<T> intt_first_transpose(const std::vector<T>& input);
    };
} // namespace gpuntt
