// host_util.cpp -- small host helpers callers of the reference expect from the library
// (customAssert, the device banner, check_result; reference src/lib/common/common.cu:5-54).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <sstream>
#include <stdexcept>
#include <string>

#include "gpuntt/common/common.cuh"

namespace gpuntt
{
    namespace
    {
        // position of the first differing element, or `count` when the ranges agree
        template <typename T> int first_difference(const T* lhs, const T* rhs, int count)
        {
            const auto hit = std::mismatch(lhs, lhs + count, rhs);
            return static_cast<int>(hit.first - lhs);
        }
    } // namespace

    template <typename T> bool check_result(T* input1, T* input2, int size)
    {
        const int at = (size > 0) ? first_difference(input1, input2, size) : size;
        if (at >= size)
            return true;
        // same report line as the reference prints for the first mismatch
        std::ostringstream line;
        line << "Error in index: " << at << " -> " << input1[at] << " - " << input2[at] << " ";
        std::puts(line.str().c_str());
        return false;
    }
    template bool check_result<std::uint32_t>(std::uint32_t*, std::uint32_t*, int);
    template bool check_result<std::int32_t>(std::int32_t*, std::int32_t*, int);
    template bool check_result<std::uint64_t>(std::uint64_t*, std::uint64_t*, int);
    template bool check_result<std::int64_t>(std::int64_t*, std::int64_t*, int);

    void HipDevice()
    {
        constexpr int ordinal = 0;
        hipDeviceProp_t info;
        GPUNTT_HIP_CHECK(hipSetDevice(ordinal));
        GPUNTT_HIP_CHECK(hipGetDeviceProperties(&info, ordinal));
        std::printf("GPU Device %d: %s (%s, %d CUs)\n\n", ordinal, info.name, info.gcnArchName,
                    info.multiProcessorCount);
    }

    void customAssert(bool condition, const std::string& errorMessage)
    {
        if (condition)
            return;
        throw std::invalid_argument(errorMessage);
    }
} // namespace gpuntt
