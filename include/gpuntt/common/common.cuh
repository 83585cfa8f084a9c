// gpuntt/common/common.cuh -- error convention and small host utilities.
//
// MI355X-native replacement for the reference header of the same include path
// (reference: src/include/gpuntt/common/common.cuh:20-56, src/lib/common/common.cu:5-54).
// The exception keeps the reference's name (CudaException) as an alias so caller code that
// catches it still compiles; the native names are HipException / GPUNTT_HIP_CHECK.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <exception>
#include <string>

namespace gpuntt
{
    using stream_t = hipStream_t; // the reference's cudaStream_t slot in every config struct

    class HipException : public std::exception
    {
      public:
        HipException(const std::string& file, int line, hipError_t error)
            : error_(error), message_("HIP Error in " + file + " at line " +
                                      std::to_string(line) + ": " +
                                      hipGetErrorString(error))
        {
        }
        const char* what() const noexcept override { return message_.c_str(); }
        hipError_t code() const noexcept { return error_; }

      private:
        hipError_t error_;
        std::string message_;
    };
    using CudaException = HipException;

#define GPUNTT_HIP_CHECK(expr)                                                 \
    do                                                                         \
    {                                                                          \
        hipError_t gpuntt_err_ = (expr);                                       \
        if (gpuntt_err_ != hipSuccess)                                         \
            throw ::gpuntt::HipException(__FILE__, __LINE__, gpuntt_err_);     \
    } while (0)
#define GPUNTT_CUDA_CHECK(expr) GPUNTT_HIP_CHECK(expr)

    // throws std::invalid_argument(errorMessage) when !condition   (reference common.cu:5-11)
    void customAssert(bool condition, const std::string& errorMessage);

    // selects device 0 and prints its name                         (reference common.cu:13-22)
    void HipDevice();
    inline void CudaDevice() { HipDevice(); }

    // exact element-wise equality, prints the first mismatch        (reference common.cu:24-42)
    template <typename T> bool check_result(T* input1, T* input2, int size);

} // namespace gpuntt
