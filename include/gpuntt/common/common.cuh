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

    // Failed HIP runtime call: carries the call site and the runtime's error text.
    class HipException : public std::exception
    {
        hipError_t code_;
        std::string text_;

        static std::string describe(const std::string& where, int line, hipError_t code)
        {
            std::string s("HIP Error in ");
            s += where;
            s += " at line ";
            s += std::to_string(line);
            s += ": ";
            s += hipGetErrorString(code);
            return s;
        }

      public:
        HipException(const std::string& file, int line, hipError_t error)
            : code_(error), text_(describe(file, line, error))
        {
        }
        hipError_t code() const noexcept { return code_; }
        const char* what() const noexcept override { return text_.c_str(); }
    };
    using CudaException = HipException; // the name caller code catches

    // every runtime call and every launch (hipGetLastError) goes through this check
    inline void throw_on_hip_error(hipError_t status, const char* file, int line)
    {
        if (status != hipSuccess)
            throw HipException(file, line, status);
    }
#define GPUNTT_HIP_CHECK(expr) ::gpuntt::throw_on_hip_error((expr), __FILE__, __LINE__)
#define GPUNTT_CUDA_CHECK(expr) GPUNTT_HIP_CHECK(expr)

    // throws std::invalid_argument(errorMessage) when !condition   (reference common.cu:5-11)
    void customAssert(bool condition, const std::string& errorMessage);

    // selects device 0 and prints its name                         (reference common.cu:13-22)
    void HipDevice();
    inline void CudaDevice() { HipDevice(); }

    // exact element-wise equality, prints the first mismatch        (reference common.cu:24-42)
    template <typename T> bool check_result(T* input1, T* input2, int size);

} // namespace gpuntt
