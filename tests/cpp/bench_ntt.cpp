// bench_ntt.cpp -- native timing loop over the reference benchmark's axes (benchmark/bench_merge_ntt.cu:57-75,
// bench_4step_ntt.cu: batch = 1, log2 N = 12 .. 24, GPU_NTT_Inplace / GPU_4STEP_NTT), without the Python
// call overhead of tools/bench_batch1.py.  Calls are timed with HIP events on their stream, one JSON object per line:
//   ./bench_ntt [merge|4step] [u64|u32] [LOGN_FIRST LOGN_LAST] [BATCH] [ITERS]
// "us" = drop-in call, "plan_us" = NTTPlan / FourStepPlan::execute, "graph_us" = the plan's launches replayed from
// a hipGraph (launch-overhead floor).  GB/s: algorithmic 2 N B sizeof(T), and the reference benchmark's own
// accounting 2.5 N B sizeof(T) (bench_merge_ntt.cu:34-38) for comparison with its published-style numbers.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "gpuntt/ntt_4step/ntt_4step.cuh"
#include "gpuntt/ntt_merge/ntt.cuh"

using namespace gpuntt;

static double time_us(hipStream_t s, int iters, const std::function<void()>& call)
{
    hipEvent_t e0, e1;
    GPUNTT_HIP_CHECK(hipEventCreate(&e0));
    GPUNTT_HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 10; i++)
        call();
    GPUNTT_HIP_CHECK(hipStreamSynchronize(s));
    GPUNTT_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++)
        call();
    GPUNTT_HIP_CHECK(hipEventRecord(e1, s));
    GPUNTT_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    GPUNTT_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return 1e3 * ms / iters;
}

// the launches of `call` captured once and replayed: what the transform costs without per-launch host work
static double graph_us(hipStream_t s, int iters, const std::function<void()>& call)
{
    hipGraph_t g;
    hipGraphExec_t ge;
    GPUNTT_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 16; i++)
        call();
    GPUNTT_HIP_CHECK(hipStreamEndCapture(s, &g));
    GPUNTT_HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const double us = time_us(s, (iters + 15) / 16, [&] { GPUNTT_HIP_CHECK(hipGraphLaunch(ge, s)); }) / 16.0;
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    return us;
}

template <typename T> static void fill(T* d, size_t count, T q)
{
    std::vector<T> h(count);
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < count; i++)
    {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17; // xorshift: any residues below q do
        h[i] = static_cast<T>(x % q);
    }
    GPUNTT_HIP_CHECK(hipMemcpy(d, h.data(), count * sizeof(T), hipMemcpyHostToDevice));
}

static void report(const char* algo, int bits, int logn, int batch, double us, double plan, double graph)
{
    const double bytes = 2.0 * (double(1ull << logn)) * batch * (bits / 8);
    std::printf("{\"algo\": \"%s\", \"dtype\": \"u%d\", \"log2N\": %d, \"batch\": %d, \"us\": %.2f, \"plan_us\": %.2f, "
                "\"graph_us\": %.2f, \"alg_GBps\": %.1f, \"ref_bench_GBps\": %.1f}\n",
                algo, bits, logn, batch, us, plan, graph, bytes / (graph * 1e3), 1.25 * bytes / (graph * 1e3));
    std::fflush(stdout);
}

template <typename T> static void merge(int logn, int batch, int iters, hipStream_t s)
{
    NTTParameters<T> prm(logn, ReductionPolynomial::X_N_minus);
    const size_t n = size_t(1) << logn;
    T *data, *table;
    GPUNTT_HIP_CHECK(hipMalloc(&data, n * batch * sizeof(T)));
    GPUNTT_HIP_CHECK(hipMalloc(&table, prm.root_of_unity_size * sizeof(T)));
    fill<T>(data, n * batch, prm.modulus.value);
    const std::vector<Root<T>> host_table = prm.gpu_root_of_unity_table_generator(prm.forward_root_of_unity_table);
    GPUNTT_HIP_CHECK(hipMemcpy(table, host_table.data(), host_table.size() * sizeof(T), hipMemcpyHostToDevice));
    ntt_configuration<T> cfg{};
    cfg.n_power = logn;
    cfg.ntt_type = FORWARD;
    cfg.ntt_layout = PerPolynomial;
    cfg.reduction_poly = ReductionPolynomial::X_N_minus;
    cfg.zero_padding = false;
    cfg.stream = s;
    const double us = time_us(s, iters, [&] { GPU_NTT_Inplace(data, table, prm.modulus, cfg, batch); });
    NTTPlan<T> plan(table, &prm.modulus, 1, logn, ReductionPolynomial::X_N_minus, FORWARD, nullptr, batch, s, nullptr);
    const auto exec = [&] { plan.execute(data, data, batch, s); };
    const double pus = time_us(s, iters, exec);
    report("merge-fwd-inplace", int(sizeof(T) * 8), logn, batch, us, pus, graph_us(s, iters, exec));
    hipFree(data);
    hipFree(table);
}

template <typename T> static void fourstep(int logn, int batch, int iters, hipStream_t s)
{
    NTTParameters4Step<T> prm(logn, ReductionPolynomial::X_N_minus);
    const size_t n = size_t(1) << logn;
    T *in, *out, *t1, *t2, *w;
    GPUNTT_HIP_CHECK(hipMalloc(&in, n * batch * sizeof(T)));
    GPUNTT_HIP_CHECK(hipMalloc(&out, n * batch * sizeof(T)));
    GPUNTT_HIP_CHECK(hipMalloc(&t1, (prm.n1 >> 1) * sizeof(T)));
    GPUNTT_HIP_CHECK(hipMalloc(&t2, (prm.n2 >> 1) * sizeof(T)));
    GPUNTT_HIP_CHECK(hipMalloc(&w, n * sizeof(T)));
    fill<T>(in, n * batch, prm.modulus.value);
    const auto upload = [](T* dst, const std::vector<T>& v) {
        GPUNTT_HIP_CHECK(hipMemcpy(dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    };
    upload(t1, prm.gpu_root_of_unity_table_generator(prm.n1_based_root_of_unity_table));
    upload(t2, prm.gpu_root_of_unity_table_generator(prm.n2_based_root_of_unity_table));
    upload(w, prm.W_root_of_unity_table);
    ntt4step_configuration<T> cfg{};
    cfg.n_power = logn;
    cfg.ntt_type = FORWARD;
    cfg.stream = s;
    const double us = time_us(s, iters, [&] { GPU_4STEP_NTT(in, out, t1, t2, w, prm.modulus, cfg, batch); });
    FourStepPlan<T> plan(t1, t2, w, prm.modulus, cfg, false, batch, nullptr);
    const auto exec = [&] { plan.execute(in, out, batch, s); };
    const double pus = time_us(s, iters, exec);
    report("4step-fwd", int(sizeof(T) * 8), logn, batch, us, pus, graph_us(s, iters, exec));
    hipFree(in), hipFree(out), hipFree(t1), hipFree(t2), hipFree(w);
}

int main(int argc, char** argv)
{
    const char* algo = argc > 1 ? argv[1] : "merge";
    const bool u32 = argc > 2 && std::strcmp(argv[2], "u32") == 0;
    const int first = argc > 3 ? std::atoi(argv[3]) : 12, last = argc > 4 ? std::atoi(argv[4]) : 24;
    const int batch = argc > 5 ? std::atoi(argv[5]) : 1, iters = argc > 6 ? std::atoi(argv[6]) : 200;
    hipStream_t s;
    GPUNTT_HIP_CHECK(hipStreamCreate(&s));
    for (int logn = first; logn <= last; logn++)
    {
        if (std::strcmp(algo, "4step") == 0)
            u32 ? fourstep<Data32>(logn, batch, iters, s) : fourstep<Data64>(logn, batch, iters, s);
        else
            u32 ? merge<Data32>(logn, batch, iters, s) : merge<Data64>(logn, batch, iters, s);
    }
    GPUNTT_HIP_CHECK(hipStreamDestroy(s));
    return 0;
}
