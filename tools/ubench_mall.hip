// ubench_mall.hip -- does the 256 MiB Infinity Cache keep a freshly written chunk so that a
// second pass over it avoids HBM?  (two-pass NTT question)
//   pass A: out[chunk] = f(in[chunk])   (streams in from HBM, writes out)
//   pass B: out[chunk] = g(out[chunk])  (in place) -- timed separately
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void passA(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out, size_t n)
{
    size_t i = blockIdx.x * size_t(256) + threadIdx.x;
    const size_t stride = size_t(gridDim.x) * 256;
    for (; i < n; i += stride) { ulonglong2 v = in[i]; v.x += 1; v.y ^= 3; out[i] = v; }
}
__global__ __launch_bounds__(256) void passB(ulonglong2* __restrict__ io, size_t n)
{
    size_t i = blockIdx.x * size_t(256) + threadIdx.x;
    const size_t stride = size_t(gridDim.x) * 256;
    for (; i < n; i += stride) { ulonglong2 v = io[i]; v.x += 7; v.y ^= 5; io[i] = v; }
}
int main()
{
    const size_t total = size_t(2) << 30; // 2 GiB in, 2 GiB out
    ulonglong2 *in, *out;
    CHECK(hipMalloc(&in, total)); CHECK(hipMalloc(&out, total));
    CHECK(hipMemset(in, 1, total)); CHECK(hipMemset(out, 2, total));
    hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    const int grid = 256 * 8;
    for (size_t mb : {8, 16, 32, 64, 96, 128, 192, 256, 512, 1024})
    {
        const size_t bytes = mb << 20, n = bytes / 16, nchunks = total / bytes;
        float ta = 0, tb = 0; int cnt = 0;
        for (int rep = 0; rep < 24; rep++)
        {
            const size_t c = rep % nchunks;
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(passA, dim3(grid), dim3(256), 0, 0, in + c * n, out + c * n, n);
            CHECK(hipEventRecord(e1));
            hipLaunchKernelGGL(passB, dim3(grid), dim3(256), 0, 0, out + c * n, n);
            CHECK(hipEventRecord(e2));
            CHECK(hipEventSynchronize(e2));
            float a, b; CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
            if (rep >= 4) { ta += a; tb += b; cnt++; }
        }
        ta /= cnt; tb /= cnt;
        printf("chunk %5zu MiB: passA %8.1f us (%6.0f GB/s r+w)   passB in-place %8.1f us (%6.0f GB/s r+w)\n", mb,
               ta * 1e3, 2.0 * bytes / (ta * 1e-3) / 1e9, tb * 1e3, 2.0 * bytes / (tb * 1e-3) / 1e9);
    }
    return 0;
}
