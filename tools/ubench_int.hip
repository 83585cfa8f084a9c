// ubench_int.hip -- gfx950 integer-VALU issue-rate microbenchmark.
// Measures wave-instruction throughput of the opcodes a 64-bit modular multiply is made of.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_int.hip -o /tmp/ubench_int && /tmp/ubench_int
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <thread>
#include <atomic>
#include <chrono>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, BODY, NOUT)                                                           \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)             \
    {                                                                                       \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, \
                 a4 = a0 * 9 + 4, a5 = a0 * 11 + 5, a6 = a0 * 13 + 6, a7 = a0 * 15 + 7;   \
        uint32_t b = seed * 2654435761u + 12345u;                                          \
        uint64_t c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a4, c5 = a5, c6 = a6, c7 = a7;  \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;    \
        double db = 1.0000001;                                                             \
        uint64_t smask = __builtin_amdgcn_readfirstlane(seed) * 0x9E3779B97F4A7C15ull;     \
        uint32_t sb = __builtin_amdgcn_readfirstlane(seed * 77u + 5u);                     \
        for (int i = 0; i < ITERS; i++) { BODY BODY BODY BODY }                            \
        uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                \
        uint64_t rc = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;                               \
        double rd = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;                                 \
        if (r == 0x12345 && rc == 0x777 && rd == 1.5) out[0] = r;                          \
    }

#define MUL_LO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define MUL_HI(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c##k) : "v"(a##k), "v"(b) : "vcc");
#define MUL24(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define MULHI24(k) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define MAD24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a##k) : "v"(b));
#define ADD32(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define ADDCO(k) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a##k) : "v"(b) : "vcc");
#define ADDC(k) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##k) : "v"(b) : "vcc");
#define LSHLADD64(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c##k) : "v"(c7));
#define CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(b) : "vcc");
#define FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d##k) : "v"(db));
#define MUL64F(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##k) : "v"(db));
#define FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a##k) : "v"(b));
#define LSHR64(k) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(c##k));
#define CMP64(k) asm volatile("v_cmp_ge_u64 vcc, %0, %1" : : "v"(c##k), "v"(c7) : "vcc");
#define SUBCO(k) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a##k) : "v"(b) : "vcc");
#define MIN32(k) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define ALIGNBIT(k) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a##k) : "v"(b));
#define CMPCND(k) asm volatile("v_cmp_ge_u64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a##k) : "v"(c##k), "v"(c7), "v"(b) : "vcc");
#define CNDE64(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "s"(smask));
#define AND32(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define XOR32(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define SUB32(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define LSHL32(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a##k));
#define ADD3(k) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##k) : "v"(b));
#define MOV32(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a##k) : "v"(b));
#define MAD64S(k) asm volatile("v_mad_u64_u32 %0, %3, %1, %2, %0" : "+v"(c##k) : "v"(a##k), "s"(sb), "s"(smask));
#define MULLOS(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##k) : "s"(sb));
#define CMP32(k) asm volatile("v_cmp_ge_u32 vcc, %0, %1" : : "v"(a##k), "v"(b) : "vcc");
#define MAX32(k) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define LSHLADD32(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a##k) : "v"(b));
#define MADU32(k) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a##k) : "v"(b));

KERNEL(k_mul_lo, REP8(MUL_LO), 0)
KERNEL(k_mul_hi, REP8(MUL_HI), 0)
KERNEL(k_mad64, REP8(MAD64), 0)
KERNEL(k_mul24, REP8(MUL24), 0)
KERNEL(k_mulhi24, REP8(MULHI24), 0)
KERNEL(k_mad24, REP8(MAD24), 0)
KERNEL(k_add32, REP8(ADD32), 0)
KERNEL(k_addco, REP8(ADDCO), 0)
KERNEL(k_addc, REP8(ADDC), 0)
KERNEL(k_lshladd64, REP8(LSHLADD64), 0)
KERNEL(k_cndmask, REP8(CNDMASK), 0)
KERNEL(k_fma64, REP8(FMA64), 0)
KERNEL(k_mul64f, REP8(MUL64F), 0)
KERNEL(k_fma32, REP8(FMA32), 0)
KERNEL(k_lshr64, REP8(LSHR64), 0)
KERNEL(k_cmp64, REP8(CMP64), 0)
KERNEL(k_subco, REP8(SUBCO), 0)
KERNEL(k_min32, REP8(MIN32), 0)
KERNEL(k_alignbit, REP8(ALIGNBIT), 0)
KERNEL(k_cmpcnd, REP8(CMPCND), 0)
KERNEL(k_cnde64, REP8(CNDE64), 0)
KERNEL(k_and32, REP8(AND32), 0)
KERNEL(k_xor32, REP8(XOR32), 0)
KERNEL(k_sub32, REP8(SUB32), 0)
KERNEL(k_lshl32, REP8(LSHL32), 0)
KERNEL(k_add3, REP8(ADD3), 0)
KERNEL(k_mov32, REP8(MOV32), 0)
KERNEL(k_mad64s, REP8(MAD64S), 0)
KERNEL(k_mullos, REP8(MULLOS), 0)
KERNEL(k_cmp32, REP8(CMP32), 0)
KERNEL(k_max32, REP8(MAX32), 0)
KERNEL(k_lshladd32, REP8(LSHLADD32), 0)

typedef void (*kfn)(uint32_t*, uint32_t);

// `ubench_int power`: socket power and shader clock (rocm-smi) while ONE opcode runs back to back on every SIMD
// for ~4 s: at equal clock the power difference between opcodes is their energy per instruction
static std::string smi_sample()
{
    std::string out;
    FILE* f = popen("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Socket' | sed 's/.*: //' | tr '\\n' ' '", "r");
    if (!f)
        return "n/a";
    char buf[256];
    while (fgets(buf, sizeof(buf), f))
        out += buf;
    pclose(f);
    return out;
}

int main(int argc, char** argv)
{
    const bool power_mode = argc > 1 && std::strcmp(argv[1], "power") == 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* out;
    CHECK(hipMalloc(&out, 4096));
    struct E { const char* name; kfn fn; };
    std::vector<E> ks = {{"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mad_u64_u32", k_mad64},
                         {"v_mul_u32_u24", k_mul24}, {"v_mul_hi_u32_u24", k_mulhi24}, {"v_mad_u32_u24", k_mad24},
                         {"v_add_u32", k_add32}, {"v_add_co_u32", k_addco}, {"v_addc_co_u32", k_addc},
                         {"v_sub_co_u32", k_subco}, {"v_lshl_add_u64", k_lshladd64}, {"v_cndmask_b32", k_cndmask},
                         {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64f}, {"v_fma_f32", k_fma32},
                         {"v_lshrrev_b64", k_lshr64}, {"v_cmp_ge_u64", k_cmp64}, {"v_min_u32", k_min32},
                         {"v_alignbit_b32", k_alignbit}, {"cmp64+cndmask(2 instr)", k_cmpcnd},
                         {"v_cndmask_b32_e64 sgpr", k_cnde64}, {"v_and_b32", k_and32}, {"v_xor_b32", k_xor32},
                         {"v_sub_u32", k_sub32}, {"v_lshlrev_b32", k_lshl32}, {"v_add3_u32", k_add3},
                         {"v_mov_b32", k_mov32}, {"v_mad_u64_u32 sgpr", k_mad64s}, {"v_mul_lo_u32 sgpr", k_mullos},
                         {"v_cmp_ge_u32", k_cmp32}, {"v_max_u32", k_max32}, {"v_lshl_add_u32", k_lshladd32}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    if (power_mode)
    {
        printf("idle: %s\n", smi_sample().c_str());
        const int blocks = cus * 4;
        for (auto& k : ks)
        {
            const std::string nm = k.name;
            if (nm != "v_mul_lo_u32" && nm != "v_mul_hi_u32" && nm != "v_mad_u64_u32" && nm != "v_add_u32" &&
                nm != "v_lshl_add_u64" && nm != "v_sub_co_u32" && nm != "v_cndmask_b32_e64 sgpr" && nm != "v_mov_b32" &&
                nm != "v_fma_f64" && nm != "v_add3_u32" && nm != "v_mul_u32_u24")
                continue;
            std::atomic<bool> stop{false};
            std::string s1, s2;
            std::thread sampler([&] {
                std::this_thread::sleep_for(std::chrono::milliseconds(1500));
                s1 = smi_sample();
                s2 = smi_sample();
                stop = true;
            });
            auto t0 = std::chrono::steady_clock::now();
            long launches = 0;
            CHECK(hipEventRecord(e0));
            while (!stop)
            {
                for (int i = 0; i < 16; i++)
                    hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
                launches += 16;
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            sampler.join();
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double winstr = double(launches) * double(ITERS) * 32 * 4 * cus * 4; // wave-instructions on the chip
            printf("%-24s %6.2f G wave-instr/s   clock/power samples: %s | %s\n", k.name, winstr / sec / 1e9, s1.c_str(),
                   s2.c_str());
        }
        return 0;
    }
    for (int wpsimd : {2, 4})
    {
        printf("--- %d wave(s) per SIMD ---\n", wpsimd);
        const int blocks = cus * wpsimd; // 256 threads = 4 waves = 1 per SIMD
        for (auto& k : ks)
        {
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++)
            {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1u + rep);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double winstr_per_simd = double(ITERS) * 32 * wpsimd; // wave-instructions per SIMD
            const double cyc = best * 1e-3 * 2.4e9;                      // at nominal 2.4 GHz
            printf("%-18s %8.3f ms  %6.2f cycles/wave-instr/SIMD (@2.4GHz)  %7.1f Gop/s/CU-lane\n",
                   k.name, best, cyc / winstr_per_simd, winstr_per_simd * 4 * 64 / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
