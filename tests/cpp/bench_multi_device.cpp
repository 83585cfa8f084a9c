// bench_multi_device.cpp -- north_star's 1 / 2 / 4 / 8-GPU table from a C++ caller of the PRODUCT (no Python, no RCCL):
// one process, one host thread per device, every thread owns a resident shard of the batch and times K drop-in
// GPU_NTT_Inplace calls on a stream of its own with HIP events; the threads start together (std::barrier) and the step
// time of a device count is the MAX over its devices -- bench.py's contract (barrier, synchronize, max over ranks) with
// threads for ranks.  Polynomials are independent, so there is no data-path collective (the reference has
// no multi-device code at all).
//
//   c2  Merge u64 2^16, 1024 polynomials PER DEVICE, X^N-1   (weak scaling, BASELINE configs[1])
//   c4  Merge u32 2^14, 8192 polynomials IN TOTAL, X^N-1     (strong scaling, BASELINE configs[3])
//
//   ./bench_multi_device <c2|c4> [STEPS = 50] [WARMUP = 10] [DEVICE COUNTS = 1,2,4,8 up to what exists]
// prints ONE JSON line per device count; polynomial 0 of every shard is checked against the library's host NTTCPU<T>.
#include <algorithm>
#include <barrier>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "gpuntt/ntt_merge/ntt.cuh"

using namespace gpuntt;

namespace
{
    struct Result
    {
        double event_ms = 0.0; // K calls between two HIP events on the device's stream
        bool ok = false;
        std::string error;
    };

    template <typename T>
    void worker(int device, int logn, int polys, int steps, int warmup, const NTTParameters<T>& prm,
                const std::vector<T>& table_host, const std::vector<T>& one, const std::vector<T>& want, std::barrier<>& gate,
                Result& res)
    {
        bool arrived_start = false, arrived_end = false;
        try
        {
            GPUNTT_CUDA_CHECK(hipSetDevice(device));
            hipStream_t stream;
            GPUNTT_CUDA_CHECK(hipStreamCreate(&stream));
            const size_t n = size_t(1) << logn, words = static_cast<size_t>(polys) * n;
            T *data = nullptr, *table = nullptr;
            GPUNTT_CUDA_CHECK(hipMalloc(reinterpret_cast<void**>(&data), words * sizeof(T)));
            GPUNTT_CUDA_CHECK(hipMalloc(reinterpret_cast<void**>(&table), table_host.size() * sizeof(T)));
            GPUNTT_CUDA_CHECK(hipMemcpy(table, table_host.data(), table_host.size() * sizeof(T), hipMemcpyHostToDevice));
            // the shard: the same polynomial in every slot (the time does not depend on the values; slot 0 is checked)
            for (int p = 0; p < polys; p++)
                GPUNTT_CUDA_CHECK(hipMemcpyAsync(data + static_cast<size_t>(p) * n, one.data(), n * sizeof(T), hipMemcpyHostToDevice, stream));
            ntt_configuration<T> cfg = {.n_power = logn,
                                        .ntt_type = FORWARD,
                                        .ntt_layout = PerPolynomial,
                                        .reduction_poly = ReductionPolynomial::X_N_minus,
                                        .zero_padding = false,
                                        .stream = stream};
            GPU_NTT_Inplace(data, table, prm.modulus, cfg, polys);
            std::vector<T> got(n), last(n);
            GPUNTT_CUDA_CHECK(hipMemcpyAsync(got.data(), data, n * sizeof(T), hipMemcpyDeviceToHost, stream));
            GPUNTT_CUDA_CHECK(hipMemcpyAsync(last.data(), data + (words - n), n * sizeof(T), hipMemcpyDeviceToHost, stream));
            GPUNTT_CUDA_CHECK(hipStreamSynchronize(stream));
            res.ok = (got == want) && (last == want);
            // clock settle + warm-up (the values stay residues below q: every output is canonical)
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.15)
            {
                for (int i = 0; i < 4; i++)
                    GPU_NTT_Inplace(data, table, prm.modulus, cfg, polys);
                GPUNTT_CUDA_CHECK(hipStreamSynchronize(stream));
            }
            for (int i = 0; i < warmup; i++)
                GPU_NTT_Inplace(data, table, prm.modulus, cfg, polys);
            GPUNTT_CUDA_CHECK(hipStreamSynchronize(stream));
            hipEvent_t e0, e1;
            GPUNTT_CUDA_CHECK(hipEventCreate(&e0));
            GPUNTT_CUDA_CHECK(hipEventCreate(&e1));
            gate.arrive_and_wait(); // every device starts its K steps together
            arrived_start = true;
            GPUNTT_CUDA_CHECK(hipEventRecord(e0, stream));
            for (int i = 0; i < steps; i++)
                GPU_NTT_Inplace(data, table, prm.modulus, cfg, polys);
            GPUNTT_CUDA_CHECK(hipEventRecord(e1, stream));
            GPUNTT_CUDA_CHECK(hipStreamSynchronize(stream));
            gate.arrive_and_wait(); // ... and the wall clock stops when the last one is done
            arrived_end = true;
            float ms = 0.f;
            GPUNTT_CUDA_CHECK(hipEventElapsedTime(&ms, e0, e1));
            res.event_ms = ms;
            (void) hipEventDestroy(e0);
            (void) hipEventDestroy(e1);
            (void) hipFree(data);
            (void) hipFree(table);
            (void) hipStreamDestroy(stream);
        }
        catch (const std::exception& e)
        {
            res.ok = false;
            res.error = e.what();
            if (!arrived_start)
                gate.arrive_and_wait();
            if (!arrived_end)
                gate.arrive_and_wait();
        }
    }

    template <typename T>
    int run(const std::string& name, int logn, int polys_total_or_per_device, bool strong, int steps, int warmup,
            const std::vector<int>& counts)
    {
        NTTParameters<T> prm(logn, ReductionPolynomial::X_N_minus);
        const std::vector<T> table_host = prm.gpu_root_of_unity_table_generator(prm.forward_root_of_unity_table);
        const size_t n = size_t(1) << logn;
        std::mt19937_64 rng(2);
        std::vector<T> one(n);
        for (T& c : one)
            c = static_cast<T>(rng() % prm.modulus.value);
        NTTCPU<T> cpu(prm);
        const std::vector<T> want = cpu.ntt(one);
        int rc = 0;
        for (int g : counts)
        {
            if (strong && polys_total_or_per_device % g != 0)
                continue;
            const int polys = strong ? polys_total_or_per_device / g : polys_total_or_per_device;
            std::vector<Result> res(g);
            std::barrier<> gate(g + 1);
            std::vector<std::thread> threads;
            for (int d = 0; d < g; d++)
                threads.emplace_back(worker<T>, d, logn, polys, steps, warmup, std::cref(prm), std::cref(table_host), std::cref(one),
                                     std::cref(want), std::ref(gate), std::ref(res[d]));
            gate.arrive_and_wait();
            const auto t0 = std::chrono::steady_clock::now();
            gate.arrive_and_wait();
            const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            for (auto& t : threads)
                t.join();
            bool ok = true;
            double max_ev = 0.0;
            std::ostringstream per;
            for (int d = 0; d < g; d++)
            {
                ok = ok && res[d].ok;
                max_ev = std::max(max_ev, res[d].event_ms);
                per << (d ? ", " : "") << res[d].event_ms / steps;
            }
            const double step_ms = max_ev / steps; // MAX over the devices of the HIP-event time per call
            const double alg = 2.0 * n * sizeof(T) * polys;
            std::cout << "{\"program\": \"bench_multi_device\", \"config\": \"" << name << "\", \"dtype\": \"u" << 8 * sizeof(T)
                      << "\", \"log2N\": " << logn << ", \"n_gpus\": " << g << ", \"polys_per_gpu\": " << polys
                      << ", \"scaling\": \"" << (strong ? "strong" : "weak") << "\", \"steps\": " << steps << ", \"warmup\": " << warmup
                      << ", \"ms_per_step\": " << step_ms << ", \"wall_ms_per_step\": " << wall_ms / steps
                      << ", \"value_ntt_per_s\": " << (static_cast<double>(g) * polys / (step_ms * 1e-3))
                      << ", \"frac_of_8TBps_per_gpu\": " << (alg / (step_ms * 1e-3) / 8e12) << ", \"per_device_ms\": [" << per.str()
                      << "], \"api\": \"GPU_NTT_Inplace (drop-in)\", \"bit_exact_vs_NTTCPU\": " << (ok ? "true" : "false") << "}" << std::endl;
            if (!ok)
            {
                for (int d = 0; d < g; d++)
                    if (!res[d].error.empty())
                        std::cerr << "device " << d << ": " << res[d].error << std::endl;
                rc = 1;
            }
        }
        return rc;
    }
} // namespace

int main(int argc, char** argv)
{
    const std::string which = argc > 1 ? argv[1] : "c2";
    const int steps = argc > 2 ? std::atoi(argv[2]) : 50;
    const int warmup = argc > 3 ? std::atoi(argv[3]) : 10;
    int have = 0;
    GPUNTT_CUDA_CHECK(hipGetDeviceCount(&have));
    std::vector<int> counts;
    if (argc > 4)
    {
        std::stringstream ss(argv[4]);
        std::string tok;
        while (std::getline(ss, tok, ','))
            counts.push_back(std::atoi(tok.c_str()));
    }
    else
        counts = {1, 2, 4, 8};
    counts.erase(std::remove_if(counts.begin(), counts.end(), [&](int g) { return g < 1 || g > have; }), counts.end());
    if (have < 1 || counts.empty() || steps < 1 || (which != "c2" && which != "c4"))
    {
        std::cout << "usage: bench_multi_device <c2|c4> [STEPS] [WARMUP] [DEVICE COUNTS, e.g. 1,2,4,8]  (devices present: " << have
                  << ")" << std::endl;
        return 2;
    }
    if (which == "c2")
        return run<Data64>("c2", 16, 1024, false, steps, warmup, counts);
    return run<Data32>("c4", 14, 8192, true, steps, warmup, counts);
}
