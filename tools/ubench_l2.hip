// ubench_l2.hip -- does a line written with a plain store stay in the XCD's L2 for a later reader?
// Every workgroup writes a 32 KiB tile, waits for the stores, and reads the tile back with loads that
// bypass the vector L1 (sc1) through a different lane mapping.  Run under
//   rocprofv3 --pmc FETCH_SIZE  /  --pmc WRITE_SIZE
// FETCH_SIZE ~ 0 for the read-back => the L2 kept the written lines (write-through, line retained);
// FETCH_SIZE ~ bytes written => stores do not leave a readable line behind.
//   variant 0: read back at once; 1: read back after streaming `pollute` MiB per XCD through the L2
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_l2.hip -o tools/ubench_l2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LOADKIND, bool WRITE = true>
__global__ __launch_bounds__(256) void wr_rd(uint64_t* buf, const uint64_t* stream, uint64_t* out, int pollute_kib)
{
    uint64_t* tile = buf + static_cast<size_t>(blockIdx.x) * 4096;
    const int t = threadIdx.x;
    if (WRITE)
    {
#pragma unroll
        for (int j = 0; j < 16; j++)
            tile[t + 256 * j] = (static_cast<uint64_t>(blockIdx.x) << 32) | static_cast<unsigned>(t + 256 * j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint64_t acc = 0;
    // optional pollution: stream other data through this XCD's L2 (plain loads)
    const uint64_t* sp = stream + static_cast<size_t>(blockIdx.x) * (static_cast<size_t>(pollute_kib) * 128);
    for (int i = t; i < pollute_kib * 128; i += 256)
        acc += sp[i];
    __syncthreads();
    // read back: thread t reads element (t * 16 + j) -- lines written by other waves of the block
    // (LOADKIND 4: no read-back at all -- do the stores themselves fetch the lines?)
#pragma unroll
    for (int j = 0; j < (LOADKIND == 4 ? 0 : 16); j++)
    {
        const uint64_t* p = tile + ((t * 16 + j * 17) & 4095);
        if (LOADKIND == 0)
            acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1
        else if (LOADKIND == 1)
            acc += __builtin_nontemporal_load(p);
        else if (LOADKIND == 2)
            acc += *reinterpret_cast<const volatile uint64_t*>(p); // plain load (may hit this CU's L1)
        else
            acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // sc0
    }
    if (acc == 0x123456789ull)
        out[0] = acc;
}

int main(int argc, char** argv)
{
    const int pollute = argc > 1 ? atoi(argv[1]) : 0; // KiB streamed per workgroup between write and read
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    uint64_t *buf, *stream, *out;
    CHECK(hipMalloc(&buf, size_t(blocks) * 32768));
    CHECK(hipMalloc(&stream, size_t(blocks) * (pollute > 0 ? pollute : 1) * 1024));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(stream, 1, size_t(blocks) * (pollute > 0 ? pollute : 1) * 1024));
    for (int r = 0; r < 5; r++)
    {
        hipLaunchKernelGGL((wr_rd<0>), dim3(blocks), dim3(256), 0, 0, buf, stream, out, pollute);
        hipLaunchKernelGGL((wr_rd<1>), dim3(blocks), dim3(256), 0, 0, buf, stream, out, pollute);
        hipLaunchKernelGGL((wr_rd<2>), dim3(blocks), dim3(256), 0, 0, buf, stream, out, pollute);
        hipLaunchKernelGGL((wr_rd<3>), dim3(blocks), dim3(256), 0, 0, buf, stream, out, pollute);
        hipLaunchKernelGGL((wr_rd<4>), dim3(blocks), dim3(256), 0, 0, buf, stream, out, pollute);
    }
    // calibration: the same read-back pattern on a buffer nobody touched in this launch (cold L2)
    uint64_t* cold;
    CHECK(hipMalloc(&cold, size_t(blocks) * 32768 * 5));
    CHECK(hipMemset(cold, 2, size_t(blocks) * 32768 * 5));
    CHECK(hipDeviceSynchronize());
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL((wr_rd<0, false>), dim3(blocks), dim3(256), 0, 0, cold + size_t(r) * blocks * 4096, stream, out, 0);
    CHECK(hipDeviceSynchronize());
    printf("done: %d workgroups x 32 KiB written and read back, pollution %d KiB per workgroup\n", blocks, pollute);
    return 0;
}
