// ubench_bfly32.hip -- register-only microbenchmark of the 32-bit lazy butterfly (gfx950), the twin of ubench_bfly.hip:
// SIMD cycles per radix-2 butterfly when nothing but the VALU is involved.  One round = 4 stages on 16 registers = 32
// butterflies; LIMIT = 8 family (q < 2^29: one conditional subtraction of 4q on U every third stage) and the default
// LIMIT = 4 family (q < 2^30: one of 2q every stage); Cooley-Tukey (forward) and Gentleman-Sande (inverse) forms.
//   hipcc --offload-arch=gfx950 -O3 -I gpu-ntt_amd/csrc -I include tools/ubench_bfly32.hip -o tools/ubench_bfly32
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "lazy.hpp"

using namespace gpuntt::lazy;

// 64-bit pair whose low word is `lo` and whose high word is anything (no instruction)
__device__ __forceinline__ uint64_t pair_lo(uint32_t lo)
{
    const uint32_t hi = __builtin_nondeterministic_value(lo);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// LIM: 8 / 0 (= the 4 q family); GS: Gentleman-Sande form
template <int LIM, bool GS, bool MAD = false>
__global__ __launch_bounds__(256, 8) void bfly_rounds(uint32_t* out, const Tw32* tw, uint32_t q, int iters)
{
    Mod<uint32_t, LIM> m;
    m.set(q, NormConst{});
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; j++)
        v[j] = (threadIdx.x * 16 + j) * 0x9E3779B9u % q;
    Tw32 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        t[i] = tw[threadIdx.x * 8 + i];
    int stage = 0;
    for (int it = 0; it < iters; it++)
    {
        int off = 0;
#pragma unroll
        for (int s = 0; s < 4; s++)
        {
            const int jb = 3 - s;
#pragma unroll
            for (int h = 0; h < 8; h++)
            {
                const int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                const int j1 = j0 | (1 << jb);
                const Tw32 w = t[(off + (j0 >> (jb + 1))) & 7];
                if (!GS)
                {
                    uint32_t U = v[j0];
                    if (LIM == 8 ? (s == 1) : true) // steady state: 8 q headroom = every third stage (here: one in four + the next round's)
                        U = (LIM == 8) ? m.template csub<4>(U) : m.template csub<2>(U);
                    if (MAD)
                    {
                        // U + T out of the product's own multiply-add chain: v_mul_hi_u32 + 2 v_mad_u64_u32
                        const uint32_t qh = __umulhi(v[j1], w.wp);
                        uint64_t a = mad32<false>(v[j1], w.w, pair_lo(U));
                        a = mad32<true>(qh, 0u - q, a);
                        const uint32_t nu = lo32(a);
                        uint32_t d;
                        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(d) : "v"(U), "s"(q * 2u));
                        v[j0] = nu;
                        v[j1] = d - nu;
                    }
                    else
                    {
                        const uint32_t T = m.template mul<false>(v[j1], w);
                        v[j0] = U + T;
                        v[j1] = U + m.kq(2) - T;
                    }
                }
                else
                {
                    const uint32_t U = v[j0], V = v[j1];
                    uint32_t S = U + V;
                    if (LIM == 8 ? (s & 1) : true)
                        S = (LIM == 8) ? m.template csub<4>(S) : m.template csub<2>(S);
                    v[j0] = S;
                    if (MAD)
                    {
                        const uint32_t x = U + m.kq(LIM == 8 ? 4 : 2) - V;
                        const uint32_t qh = __umulhi(x, w.wp);
                        uint64_t a = mad32z<false>(x, w.w);
                        a = mad32<true>(qh, 0u - q, a);
                        v[j1] = lo32(a);
                    }
                    else
                        v[j1] = m.template mul<false>(U + m.kq(LIM == 8 ? 4 : 2) - V, w);
                }
            }
            off += 1 << (3 - jb);
        }
        stage += 4;
    }
    uint32_t r = stage;
#pragma unroll
    for (int j = 0; j < 16; j++)
        r ^= v[j];
    if (r == 0x1234567)
        out[0] = r;
}

template <int LIM, bool GS, bool MAD = false> int run(const char* name, uint32_t* d_out, const Tw32* d_tw, uint32_t q, int blocks_per_cu, bool sustained)
{
    const int iters = 400;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++)
        hipLaunchKernelGGL((bfly_rounds<LIM, GS, MAD>), dim3(grid), dim3(256), 0, 0, d_out, d_tw, q, iters);
    CHECK(hipDeviceSynchronize());
    if (sustained)
        for (int w = 0; w < int(1500.0 / (0.1 * blocks_per_cu)); w++)
            hipLaunchKernelGGL((bfly_rounds<LIM, GS, MAD>), dim3(grid), dim3(256), 0, 0, d_out, d_tw, q, iters);
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < reps; w++)
        hipLaunchKernelGGL((bfly_rounds<LIM, GS, MAD>), dim3(grid), dim3(256), 0, 0, d_out, d_tw, q, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bf_per_simd = double(blocks_per_cu) * iters * 32.0; // per SIMD: blocks_per_cu waves
    printf("%-44s %d wave/SIMD  %8.3f ms  %6.1f cycles/butterfly/SIMD (@2.4 GHz nominal)\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / bf_per_simd);
    return 0;
}

int main(int argc, char** argv)
{
    const bool sustained = (argc > 1 && argv[1][0] == 's');
    printf("%s\n", sustained ? "sustained (1.5 s of load before each timing)" : "burst");
    const uint32_t q = 469762049u;
    uint32_t* d_out;
    Tw32* d_tw;
    CHECK(hipMalloc(&d_out, 64));
    std::vector<Tw32> h(256 * 8 + 16);
    for (size_t i = 0; i < h.size(); i++)
    {
        h[i].w = static_cast<uint32_t>((i * 0x9E3779B97F4A7C15ull + 12345) % q);
        h[i].wp = static_cast<uint32_t>((static_cast<uint64_t>(h[i].w) << 32) / q);
    }
    CHECK(hipMalloc(&d_tw, h.size() * sizeof(Tw32)));
    CHECK(hipMemcpy(d_tw, h.data(), h.size() * sizeof(Tw32), hipMemcpyHostToDevice));
    for (int occ = 4; occ <= 8; occ *= 2)
    {
        run<8, false>("LIMIT 8, forward (CT), csub 4q every 4th stage", d_out, d_tw, q, occ, sustained);
        run<8, true>("LIMIT 8, inverse (GS), csub 4q every 2nd stage", d_out, d_tw, q, occ, sustained);
        run<0, false>("LIMIT 4, forward (CT), csub 2q every stage", d_out, d_tw, q, occ, sustained);
        run<0, true>("LIMIT 4, inverse (GS), csub 2q every stage", d_out, d_tw, q, occ, sustained);
        run<8, false, true>("LIMIT 8, forward (CT), mad chain", d_out, d_tw, q, occ, sustained);
        run<8, true, true>("LIMIT 8, inverse (GS), mad chain", d_out, d_tw, q, occ, sustained);
        run<0, false, true>("LIMIT 4, forward (CT), mad chain", d_out, d_tw, q, occ, sustained);
        run<0, true, true>("LIMIT 4, inverse (GS), mad chain", d_out, d_tw, q, occ, sustained);
    }
    return 0;
}
