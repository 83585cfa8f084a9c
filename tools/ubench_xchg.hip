// ubench_xchg.hip -- what does moving a tile bit between "register index" and "lane index" cost on gfx950?
// A wave holds 1024 64-bit coefficients (64 lanes x 16 registers).  Between rounds of 32 register-only
// butterflies (tools/ubench_bfly.hip) the kernels need an exchange that brings new tile bits into the
// register index.  Variants timed here, per 4-stage round, 4 waves per SIMD:
//   none        butterflies only
//   lds         what the kernels do: 16 ds_write_b64 + 16 ds_read_b64 through a padded wave-private LDS
//               region (brings 4 new bits in at once)
//   swap32      gfx950 v_permlane32_swap_b32: lane bit 5 <-> register bit 3, 16 instructions (1 new bit)
//   swap32+16   v_permlane32_swap + v_permlane16_swap: lane bits 5, 4 <-> register bits 3, 2 (2 new bits)
//   dpp8        lane bit 3 <-> register bit 3 with v_mov_b32_dpp row_ror:8 + v_cndmask (the best form found
//               for the lane bits below 4: 3 instructions per dword)
//   hipcc --offload-arch=gfx950 -O3 -I gpu-ntt_amd/csrc -I include tools/ubench_xchg.hip -o tools/ubench_xchg
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "lazy.hpp"

using namespace gpuntt::lazy;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void swap32(uint32_t& lo_regs_hi_lanes, uint32_t& hi_regs_lo_lanes)
{
    // lanes 32..63 of the first operand <-> lanes 0..31 of the second
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo_regs_hi_lanes), "+v"(hi_regs_lo_lanes));
}
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b)
{
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <int XCHG>
__global__ __launch_bounds__(256, 4) void rounds(uint64_t* out, const Tw64* tw, uint64_t q, int iters)
{
    __shared__ uint64_t lds[4 * (1024 + 64)];
    Mod<uint64_t> m;
    m.set(q, make_norm_const(q, 60));
    uint64_t v[16];
#pragma unroll
    for (int j = 0; j < 16; j++)
        v[j] = (threadIdx.x * 16 + j) * 0x9E3779B97F4A7C15ull % q;
    Tw64 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        t[i] = tw[threadIdx.x * 8 + i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t* wl = lds + wave * (1024 + 64);
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
                const Tw64 w = t[(off + (j0 >> (jb + 1))) & 7];
                uint64_t U = v[j0];
                if (s & 1)
                    U = m.csub<8>(U);
                const uint64_t nu = m.mul_acc<false>(v[j1], w, U);
                v[j0] = nu;
                v[j1] = (U << 1) + m.kq(4) - nu;
            }
            off += 1 << (3 - jb);
        }
        if (XCHG == 1)
        {
            // 16-contiguous window -> 64-strided window through the wave's padded LDS region
            const int wbase = lane * 16 + lane; // e + (e >> 4) with e = lane * 16
#pragma unroll
            for (int j = 0; j < 16; j++)
                wl[wbase + j] = v[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int rbase = lane + (lane >> 4);
#pragma unroll
            for (int j = 0; j < 16; j++)
                v[j] = wl[rbase + 64 * j + 4 * j];
        }
        else if (XCHG == 2 || XCHG == 3)
        {
#pragma unroll
            for (int j = 0; j < 8; j++)
            {
                uint32_t a0 = lo32(v[j]), a1 = hi32(v[j]), b0 = lo32(v[j + 8]), b1 = hi32(v[j + 8]);
                swap32(a0, b0);
                swap32(a1, b1);
                v[j] = (static_cast<uint64_t>(a1) << 32) | a0;
                v[j + 8] = (static_cast<uint64_t>(b1) << 32) | b0;
            }
            if (XCHG == 3)
            {
#pragma unroll
                for (int jj = 0; jj < 8; jj++)
                {
                    const int j = (jj & 3) | ((jj >> 2) << 3); // pairs (j, j + 4)
                    uint32_t a0 = lo32(v[j]), a1 = hi32(v[j]), b0 = lo32(v[j + 4]), b1 = hi32(v[j + 4]);
                    swap16(a0, b0);
                    swap16(a1, b1);
                    v[j] = (static_cast<uint64_t>(a1) << 32) | a0;
                    v[j + 4] = (static_cast<uint64_t>(b1) << 32) | b0;
                }
            }
        }
        else if (XCHG == 4)
        {
            // lane bit 3 <-> register bit 3: every lane pulls its partner's (lane ^ 8) copy of the register it
            // gives away, then keeps / replaces by lane parity: mov_dpp + 2 cndmask per dword
            const bool upper = (lane & 8) != 0;
#pragma unroll
            for (int j = 0; j < 8; j++)
            {
                uint32_t x[2] = {lo32(v[j]), hi32(v[j])}, y[2] = {lo32(v[j + 8]), hi32(v[j + 8])};
#pragma unroll
                for (int d = 0; d < 2; d++)
                {
                    const uint32_t send = upper ? x[d] : y[d];
                    const uint32_t got = __builtin_amdgcn_update_dpp(0u, send, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
                    x[d] = upper ? got : x[d];
                    y[d] = upper ? y[d] : got;
                }
                v[j] = (static_cast<uint64_t>(x[1]) << 32) | x[0];
                v[j + 8] = (static_cast<uint64_t>(y[1]) << 32) | y[0];
            }
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int j = 0; j < 16; j++)
        r ^= v[j];
    if (r == 0x1234567)
        out[0] = r;
}

template <int XCHG> int run(const char* name, uint64_t* d_out, const Tw64* d_tw, uint64_t q, float base_ms, float* ms_out)
{
    const int iters = 200, grid = 256 * 4, reps = 10;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 300; w++) // sustained clocks
        hipLaunchKernelGGL((rounds<XCHG>), dim3(grid), dim3(256), 0, 0, d_out, d_tw, q, iters);
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < reps; w++)
        hipLaunchKernelGGL((rounds<XCHG>), dim3(grid), dim3(256), 0, 0, d_out, d_tw, q, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double cyc_round = ms * 1e-3 * 2.4e9 / (4.0 * iters); // per SIMD: 4 waves x iters rounds
    const double base_round = base_ms * 1e-3 * 2.4e9 / (4.0 * iters);
    printf("%-44s %8.3f ms  %7.1f cycles/round/SIMD  (+%.1f over butterflies only)\n", name, ms, cyc_round,
           base_ms > 0 ? cyc_round - base_round : 0.0);
    if (ms_out)
        *ms_out = ms;
    return 0;
}

int main()
{
    const uint64_t q = 576460756061519873ull;
    uint64_t* d_out;
    Tw64* d_tw;
    CHECK(hipMalloc(&d_out, 64));
    std::vector<Tw64> h(256 * 8 + 16);
    for (size_t i = 0; i < h.size(); i++)
    {
        h[i].w = (i * 0x9E3779B97F4A7C15ull + 12345) % q;
        h[i].wp = static_cast<uint64_t>((static_cast<unsigned __int128>(h[i].w) << 64) / q);
    }
    CHECK(hipMalloc(&d_tw, h.size() * sizeof(Tw64)));
    CHECK(hipMemcpy(d_tw, h.data(), h.size() * sizeof(Tw64), hipMemcpyHostToDevice));
    float base = 0;
    run<0>("butterflies only (32 per round)", d_out, d_tw, q, 0, &base);
    run<1>("+ LDS transposition, wave-private (4 bits)", d_out, d_tw, q, base, nullptr);
    run<2>("+ v_permlane32_swap (1 bit, 16 instr)", d_out, d_tw, q, base, nullptr);
    run<3>("+ permlane32 + permlane16 swap (2 bits, 32)", d_out, d_tw, q, base, nullptr);
    run<4>("+ DPP row_ror:8 + selects (1 bit, 48 instr)", d_out, d_tw, q, base, nullptr);
    return 0;
}
