// tools/ubench_runs.hip -- what do short write runs cost?  A two-sweep 2^24 transform (12 + 12 stages) needs a first
// sweep whose tile holds 4096 x c coefficients and writes 4096 runs of c coefficients (8c bytes), 32 KiB apart; the
// three-sweep form writes 128-byte runs.  This emulates the memory side only: every block reads its tile as one
// contiguous block and writes it as runs of c u64, for c = 2 .. 16 (and fully contiguous as the reference point), with
// the tiles of one polynomial walked by consecutive block indices (different XCDs) or by the blocks of one XCD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_runs.hip -o tools/ubench_runs && tools/ubench_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// polynomial = 2^24 u64; tile (p, t): reads in[p][t * 4096 * C ...), writes out[p][r * 4096 + t * C + e], r < 4096, e < C
template <int C, bool XCD, bool NT>
__global__ __launch_bounds__(256) void runs(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int polys)
{
    constexpr unsigned TPP = 4096 / C; // tiles per polynomial
    unsigned b = blockIdx.x;
    unsigned p, t;
    if (XCD)
    {
        // blocks x, x + 8, x + 16, ... (one XCD) walk adjacent tiles of one polynomial
        const unsigned x = b & 7, k = b >> 3;
        t = k % TPP;
        p = (k / TPP) * 8 + x;
    }
    else
    {
        t = b % TPP;
        p = b / TPP;
    }
    if (p >= (unsigned) polys)
        return;
    const uint64_t* src = in + ((size_t) p << 24) + (size_t) t * 4096 * C;
    uint64_t* dst = out + ((size_t) p << 24) + (size_t) t * C;
    for (unsigned it = 0; it < (4096 * C) / (256 * 16); it++)
    {
        uint64_t v[16];
#pragma unroll
        for (int j = 0; j < 16; j++)
            v[j] = __builtin_nontemporal_load(src + (it * 16 + j) * 256 + threadIdx.x);
#pragma unroll
        for (int j = 0; j < 16; j++)
        {
            const unsigned o = (it * 16 + j) * 256 + threadIdx.x; // element of the tile: run o / C, position o % C
            uint64_t* q = dst + (size_t) (o / C) * 4096 + (o % C);
            if (NT)
                __builtin_nontemporal_store(v[j] + 1, q);
            else
                *q = v[j] + 1;
        }
    }
}

template <int C, bool XCD, bool NT> void bench(const uint64_t* in, uint64_t* out, int polys)
{
    const unsigned grid = (unsigned) polys * (4096 / C);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL((runs<C, XCD, NT>), dim3(grid), dim3(256), 0, 0, in, out, polys);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL((runs<C, XCD, NT>), dim3(grid), dim3(256), 0, 0, in, out, polys);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = 2.0 * polys * 8.0 * (1 << 24);
    printf("run %4d B  %-22s %-5s  %8.3f ms  %7.1f GB/s (read + write)\n", C * 8,
           XCD ? "tiles walk one XCD" : "tiles across XCDs", NT ? "nt" : "plain", ms, bytes / ms * 1e-6);
}

int main()
{
    const int polys = 16;
    uint64_t *in, *out;
    CK(hipMalloc(&in, (size_t) polys << 27));
    CK(hipMalloc(&out, (size_t) polys << 27));
    CK(hipMemset(in, 1, (size_t) polys << 27));
    bench<2, false, false>(in, out, polys);
    bench<2, true, false>(in, out, polys);
    bench<4, false, false>(in, out, polys);
    bench<4, true, false>(in, out, polys);
    bench<4, true, true>(in, out, polys);
    bench<8, false, false>(in, out, polys);
    bench<8, true, false>(in, out, polys);
    bench<8, true, true>(in, out, polys);
    bench<16, false, false>(in, out, polys);
    bench<16, true, false>(in, out, polys);
    bench<16, true, true>(in, out, polys);
    bench<64, false, false>(in, out, polys);
    bench<64, true, false>(in, out, polys);
    
    return 0;
}
