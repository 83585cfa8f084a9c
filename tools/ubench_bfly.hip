// ubench_bfly.hip -- register-only microbenchmark of the 64-bit lazy butterfly (gfx950): how many
// SIMD cycles one radix-2 butterfly costs when nothing but the VALU is involved (no LDS, no memory).
// One "round" = 4 stages on 16 registers = 32 butterflies with the range corrections of a steady-state
// forward pass (one conditional subtraction of 8q on U every second stage), twiddles per lane (VGPR)
// or wave-uniform (SGPR).  Variants: the multiply written in plain C++ (round-1 formulation: mul_lo +
// add3 cross terms, separate U + T) and the multiply-add chain of lazy.hpp.
//   hipcc --offload-arch=gfx950 -O3 -I gpu-ntt_amd/csrc -I include tools/ubench_bfly.hip -o tools/ubench_bfly
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "lazy.hpp"

using namespace gpuntt::lazy;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// round-1 formulation (kept here for the A/B only)
__device__ __forceinline__ uint64_t mul_r1(const Mod<uint64_t>& m, uint64_t x, const Tw64& t)
{
    const uint32_t x0 = lo32(x), x1 = hi32(x);
    const uint32_t h1 = __umulhi(x1, lo32(t.wp)), h2 = __umulhi(x0, hi32(t.wp));
    uint64_t qh = static_cast<uint64_t>(x1) * hi32(t.wp) + h1;
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(qh), "=s"(carry) : "v"(h2), "v"(qh));
    return x * t.w + qh * m.qneg;
}

// ---- two independent products, their multiply-add chains interleaved statement by statement and pinned (asm volatile);
// chain A sends its carry-outs to vcc, chain B to a scalar pair, so neighbouring statements share no register and the
// compiler has no reason to put an s_nop between them (round 5, profiles/ubench_bfly32_r05.txt, last note)
template <bool SB> __device__ __forceinline__ uint64_t vmad32(uint32_t a, uint32_t b, uint64_t c)
{
    uint64_t d, cy;
    if constexpr (SB)
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(b), "v"(c));
    else
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
}
template <bool SB> __device__ __forceinline__ uint64_t vmad32z(uint32_t a, uint32_t b)
{
    uint64_t d, cy;
    if constexpr (SB)
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "s"(b));
    else
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "v"(b));
    return d;
}
template <bool SB> __device__ __forceinline__ uint64_t cmad32(uint32_t a, uint32_t b, uint64_t c)
{
    uint64_t d;
    if constexpr (SB)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c) : "vcc");
    else
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
}
template <bool SB> __device__ __forceinline__ uint64_t cmad32z(uint32_t a, uint32_t b)
{
    uint64_t d;
    if constexpr (SB)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "s"(b) : "vcc");
    else
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
    return d;
}
template <bool UNI>
__device__ __forceinline__ void mul_acc2(const Mod<uint64_t>& m, uint64_t xa, const Tw64& ta, uint64_t acca, uint64_t xb,
                                         const Tw64& tb, uint64_t accb, uint64_t& ra, uint64_t& rb)
{
    const uint32_t xa0 = lo32(xa), xa1 = hi32(xa), xb0 = lo32(xb), xb1 = hi32(xb);
    const uint32_t ha2 = __umulhi(xa0, hi32(ta.wp)), hb2 = __umulhi(xb0, hi32(tb.wp));
    const uint32_t ha1 = __umulhi(xa1, lo32(ta.wp)), hb1 = __umulhi(xb1, lo32(tb.wp));
    uint64_t qa = cmad32<false>(xa1, hi32(ta.wp), static_cast<uint64_t>(ha1));
    uint64_t qb = vmad32<false>(xb1, hi32(tb.wp), static_cast<uint64_t>(hb1));
    asm volatile("v_mad_u64_u32 %0, vcc, %1, 1, %2" : "=v"(qa) : "v"(ha2), "v"(qa) : "vcc");
    uint64_t cy;
    asm volatile("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(qb), "=s"(cy) : "v"(hb2), "v"(qb));
    uint64_t ca = cmad32z<UNI>(xa0, hi32(ta.w));
    uint64_t cb = vmad32z<UNI>(xb0, hi32(tb.w));
    ca = cmad32<UNI>(xa1, lo32(ta.w), ca);
    cb = vmad32<UNI>(xb1, lo32(tb.w), cb);
    ca = cmad32<true>(lo32(qa), hi32(m.qneg), ca);
    cb = vmad32<true>(lo32(qb), hi32(m.qneg), cb);
    ca = cmad32<true>(hi32(qa), lo32(m.qneg), ca);
    cb = vmad32<true>(hi32(qb), lo32(m.qneg), cb);
    uint64_t a = cmad32<UNI>(xa0, lo32(ta.w), acca);
    uint64_t b = vmad32<UNI>(xb0, lo32(tb.w), accb);
    uint32_t ah, bh;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(ah) : "v"(hi32(a)), "v"(lo32(ca)));
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(bh) : "v"(hi32(b)), "v"(lo32(cb)));
    a = (static_cast<uint64_t>(ah) << 32) | lo32(a);
    b = (static_cast<uint64_t>(bh) << 32) | lo32(b);
    ra = cmad32<true>(lo32(qa), lo32(m.qneg), a);
    rb = vmad32<true>(lo32(qb), lo32(m.qneg), b);
}

__device__ __forceinline__ uint32_t keep32(uint32_t x)
{
    asm("" : "+v"(x));
    return x;
}
// the chain in plain C++ -- the compiler selects v_mad_u64_u32 itself and knows there is no hazard behind it (no s_nop); an
// empty asm that "uses" all 64 bits of the cross-term accumulator keeps it from narrowing that chain to v_mul_lo_u32 + v_add3_u32
__device__ __forceinline__ uint64_t keep64(uint64_t x)
{
    asm("" : "+v"(x));
    return x;
}
template <bool UNI> __device__ __forceinline__ uint64_t mul_acc_c(const Mod<uint64_t>& m, uint64_t x, const Tw64& t, uint64_t acc, uint32_t one)
{
    const uint32_t x0 = lo32(x), x1 = hi32(x);
    const uint32_t h2 = __umulhi(x0, hi32(t.wp));
    uint64_t qh, carry;
    asm("v_mul_hi_u32 v126, %2, %3\n\tv_mad_u64_u32 %0, %1, %2, %4, v[126:127]"
        : "=v"(qh), "=s"(carry)
        : "v"(x1), "v"(lo32(t.wp)), "v"(hi32(t.wp)), "{v127}"(m.zero)
        : "v126");
    qh = static_cast<uint64_t>(h2) * one + qh;
    uint64_t c = static_cast<uint64_t>(x0) * hi32(t.w);
    c = static_cast<uint64_t>(x1) * lo32(t.w) + c;
    c = static_cast<uint64_t>(lo32(qh)) * hi32(m.qneg) + c;
    c = static_cast<uint64_t>(hi32(qh)) * lo32(m.qneg) + c;
    c = keep64(c);
    uint64_t a = static_cast<uint64_t>(x0) * lo32(t.w) + acc;
    uint32_t ah;
    asm("v_add_u32 %0, %1, %2" : "=v"(ah) : "v"(hi32(a)), "v"(lo32(c)));
    a = (static_cast<uint64_t>(ah) << 32) | lo32(a);
    a = static_cast<uint64_t>(lo32(qh)) * lo32(m.qneg) + a;
    return a;
}

template <int VARIANT, bool UNI, bool CSUB>
__global__ __launch_bounds__(256, 4) void bfly_rounds(uint64_t* out, const Tw64* tw, uint64_t q, int iters)
{
    Mod<uint64_t> m;
    m.set(q, make_norm_const(q, 60));
    uint64_t v[16];
    uint32_t one; // 1, opaque to the optimiser
    asm("v_mov_b32 %0, 1" : "=v"(one));
#pragma unroll
    for (int j = 0; j < 16; j++)
        v[j] = (threadIdx.x * 16 + j) * 0x9E3779B97F4A7C15ull % q;
    // 8 distinct twiddles per thread keep the per-lane variant inside the 128-VGPR budget of 4 waves per SIMD
    Tw64 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        t[i] = UNI ? tw[i + (blockIdx.x & 1)] : tw[threadIdx.x * 8 + i];
    for (int it = 0; it < iters; it++)
    {
        int off = 0;
#pragma unroll
        for (int s = 0; s < 4; s++)
        {
            const int jb = 3 - s;
            if (VARIANT == 2)
            {
#pragma unroll
                for (int h = 0; h < 8; h += 2)
                {
                    const int ja0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1)), ja1 = ja0 | (1 << jb);
                    const int hb = h + 1;
                    const int jb0 = (hb & ((1 << jb) - 1)) | ((hb >> jb) << (jb + 1)), jb1 = jb0 | (1 << jb);
                    const Tw64 wa = t[(off + (ja0 >> (jb + 1))) & 7], wb = t[(off + (jb0 >> (jb + 1))) & 7];
                    uint64_t Ua = v[ja0], Ub = v[jb0];
                    if (CSUB && (s & 1))
                    {
                        Ua = m.csub<8>(Ua);
                        Ub = m.csub<8>(Ub);
                    }
                    uint64_t na, nb;
                    mul_acc2<UNI>(m, v[ja1], wa, Ua, v[jb1], wb, Ub, na, nb);
                    v[ja0] = na;
                    v[jb0] = nb;
                    v[ja1] = (Ua << 1) + m.kq(4) - na;
                    v[jb1] = (Ub << 1) + m.kq(4) - nb;
                }
            }
            else
#pragma unroll
            for (int h = 0; h < 8; h++)
            {
                const int j0 = (h & ((1 << jb) - 1)) | ((h >> jb) << (jb + 1));
                const int j1 = j0 | (1 << jb);
                const Tw64 w = t[(off + (j0 >> (jb + 1))) & 7];
                uint64_t U = v[j0];
                if (CSUB && (s & 1))
                    U = m.csub<8>(U);
                if (VARIANT == 3)
                {
                    const uint64_t nu = mul_acc_c<UNI>(m, v[j1], w, U, one);
                    v[j0] = nu;
                    v[j1] = (U << 1) + m.kq(4) - nu;
                }
                else if (VARIANT == 0)
                {
                    const uint64_t T = mul_r1(m, v[j1], w);
                    v[j0] = U + T;
                    v[j1] = U + m.kq(4) - T;
                }
                else
                {
                    const uint64_t nu = m.mul_acc<UNI>(v[j1], w, U);
                    v[j0] = nu;
                    v[j1] = (U << 1) + m.kq(4) - nu;
                }
            }
            off += 1 << (3 - jb);
        }
        // keep the values in range for the next round without extra work in the loop: the
        // arithmetic is mod 2^64 either way, only the timing is read
    }
    uint64_t r = 0;
#pragma unroll
    for (int j = 0; j < 16; j++)
        r ^= v[j];
    if (r == 0x1234567)
        out[0] = r;
}

static bool g_sustained = false;

template <int VARIANT, bool UNI, bool CSUB>
int run(const char* name, uint64_t* d_out, const Tw64* d_tw, uint64_t q, int blocks_per_cu)
{
    const int iters = 200;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++)
        hipLaunchKernelGGL((bfly_rounds<VARIANT, UNI, CSUB>), dim3(grid), dim3(256), 0, 0, d_out, d_tw, q, iters);
    CHECK(hipDeviceSynchronize());
    // sustained = true: ~1.5 s of back-to-back launches first, so the timed launches run at the clock the
    // part holds under its power cap rather than at the boost clock of a short burst
    if (g_sustained)
    {
        for (int w = 0; w < int(1500.0 / (0.19 * blocks_per_cu)); w++)
            hipLaunchKernelGGL((bfly_rounds<VARIANT, UNI, CSUB>), dim3(grid), dim3(256), 0, 0, d_out, d_tw, q, iters);
    }
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int w = 0; w < reps; w++)
        hipLaunchKernelGGL((bfly_rounds<VARIANT, UNI, CSUB>), dim3(grid), dim3(256), 0, 0, d_out, d_tw, q, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    // per SIMD: blocks_per_cu waves, each iters * 32 butterflies
    const double bf_per_simd = double(blocks_per_cu) * iters * 32.0;
    const double cyc = ms * 1e-3 * 2.4e9 / bf_per_simd;
    printf("%-34s %d wave/SIMD  %8.3f ms  %6.1f cycles/butterfly/SIMD (@2.4 GHz nominal)\n", name, blocks_per_cu, ms, cyc);
    return 0;
}

int main(int argc, char** argv)
{
    g_sustained = (argc > 1 && argv[1][0] == 's');
    printf("%s\n", g_sustained ? "sustained (1.5 s of load before each timing)" : "burst");
    const uint64_t q = 576460756061519873ull;
    uint64_t* d_out;
    Tw64* d_tw;
    CHECK(hipMalloc(&d_out, 64));
    std::vector<Tw64> h(256 * 15 + 16);
    for (size_t i = 0; i < h.size(); i++)
    {
        h[i].w = (i * 0x9E3779B97F4A7C15ull + 12345) % q;
        h[i].wp = static_cast<uint64_t>((static_cast<unsigned __int128>(h[i].w) << 64) / q);
    }
    CHECK(hipMalloc(&d_tw, h.size() * sizeof(Tw64)));
    CHECK(hipMemcpy(d_tw, h.data(), h.size() * sizeof(Tw64), hipMemcpyHostToDevice));
    for (int occ = (g_sustained ? 4 : 1); occ <= 4; occ *= 2)
    {
        run<0, false, true>("r1 mul, lane twiddles, csub/2", d_out, d_tw, q, occ);
        run<1, false, true>("mad chain, lane twiddles, csub/2", d_out, d_tw, q, occ);
        run<2, false, true>("two chains interleaved + pinned, lane tw", d_out, d_tw, q, occ);
        run<3, false, true>("chain in C++ (compiler's own mads), lane tw", d_out, d_tw, q, occ);
        run<0, true, true>("r1 mul, scalar twiddles, csub/2", d_out, d_tw, q, occ);
        run<1, true, true>("mad chain, scalar twiddles, csub/2", d_out, d_tw, q, occ);
        run<0, false, false>("r1 mul, lane twiddles, no csub", d_out, d_tw, q, occ);
        run<1, false, false>("mad chain, lane twiddles, no csub", d_out, d_tw, q, occ);
    }
    return 0;
}
