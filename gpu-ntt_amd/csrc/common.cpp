// common.cpp -- host utilities (reference src/lib/common/common.cu:5-54).
#include <cstdio>
#include <iostream>
#include <stdexcept>

#include "gpuntt/common/common.cuh"

namespace gpuntt
{
    void customAssert(bool condition, const std::string& errorMessage)
    {
        if (!condition)
            throw std::invalid_argument(errorMessage);
    }

    void HipDevice()
    {
        hipDeviceProp_t prop;
        const int device = 0;
        GPUNTT_HIP_CHECK(hipSetDevice(device));
        GPUNTT_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        std::printf("GPU Device %d: %s (%s, %d CUs)\n\n", device, prop.name, prop.gcnArchName,
                    prop.multiProcessorCount);
    }

    template <typename T> bool check_result(T* input1, T* input2, int size)
    {
        for (int i = 0; i < size; i++)
            if (input1[i] != input2[i])
            {
                std::cout << "Error in index: " << i << " -> " << input1[i] << " - " << input2[i]
                          << " " << std::endl;
                return false;
            }
        return true;
    }

    template bool check_result<std::uint64_t>(std::uint64_t*, std::uint64_t*, int);
    template bool check_result<std::uint32_t>(std::uint32_t*, std::uint32_t*, int);
    template bool check_result<std::int64_t>(std::int64_t*, std::int64_t*, int);
    template bool check_result<std::int32_t>(std::int32_t*, std::int32_t*, int);
} // namespace gpuntt
