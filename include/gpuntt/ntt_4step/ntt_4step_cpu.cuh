// gpuntt/ntt_4step/ntt_4step_cpu.cuh -- host 4-step transform, same surface as reference
// src/include/gpuntt/ntt_4step/ntt_4step_cpu.cuh:13-52.
#pragma once

#include "gpuntt/common/nttparameters.cuh"

namespace gpuntt
{
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
