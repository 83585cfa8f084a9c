// example_multi_device.cpp -- the batch shard of north_star's multi-GPU form from a C++ caller: one process, one host
// thread per device (hipSetDevice), every thread transforms its contiguous share of the batch with the drop-in call on
// a stream of its own and -- second round -- through an NTTPlan, no data-path collective anywhere (polynomials are
// independent: SURVEY.md 8e; the reference has no multi-device code, grep cudaSetDevice src/ -> common.cu:18 only).
// Shards are aligned to mod_count, so the local p % mod_count equals the global one.  Every polynomial of every shard is
// checked against the library's host transform NTTCPU<T>.  With fewer devices than asked for the program degrades to
// what exists (one device: the same code path with one worker) and says so.
//
//   ./example_multi_device <LOGN> <BATCH> [DEVICES = all] [MOD_COUNT = 2]
#include <cstdlib>
#include <iostream>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "gpuntt/ntt_merge/ntt.cuh"

using namespace gpuntt;
using T = Data64;

namespace
{
    std::mutex g_io;

    struct Shard
    {
        int device, first, count; // polynomials [first, first + count)
        bool ok = false;
        std::string error;
    };

    // the stack: mod_count primes of the library's own pools that support this ring (the 4-step pools hold one prime per
    // log2 N >= 12, each with 2^(log2 N + 1) | q - 1), with psi brought down to order 2N
    std::vector<NTTFactors<T>> stack_factors(int logn, int mod_count)
    {
        std::vector<NTTFactors<T>> out;
        for (int lg = (logn > 12 ? logn : 12); lg <= 24 && static_cast<int>(out.size()) < mod_count; lg++)
        {
            NTTParameters4Step<T> p(lg, ReductionPolynomial::X_N_minus);
            T psi = p.psi;
            for (int i = logn; i < lg; i++)
                psi = OPERATOR<T>::mult(psi, psi, p.modulus);
            bool dup = false;
            for (const auto& f : out)
                dup = dup || f.modulus.value == p.modulus.value;
            if (!dup)
                out.push_back(NTTFactors<T>(p.modulus, OPERATOR<T>::mult(psi, psi, p.modulus), psi));
        }
        return out;
    }

    void worker(Shard& sh, int logn, int mod_count, const std::vector<NTTParameters<T>>& prms, const std::vector<T>& table_host,
                const std::vector<T>& coeffs, const std::vector<T>& expected)
    {
        try
        {
            GPUNTT_CUDA_CHECK(hipSetDevice(sh.device));
            hipStream_t stream;
            GPUNTT_CUDA_CHECK(hipStreamCreate(&stream));
            const size_t n = size_t(1) << logn, words = static_cast<size_t>(sh.count) * n, off = static_cast<size_t>(sh.first) * n;
            T *data = nullptr, *table = nullptr;
            Modulus<T>* mods = nullptr;
            GPUNTT_CUDA_CHECK(hipMalloc(reinterpret_cast<void**>(&data), words * sizeof(T)));
            GPUNTT_CUDA_CHECK(hipMalloc(reinterpret_cast<void**>(&table), table_host.size() * sizeof(T)));
            GPUNTT_CUDA_CHECK(hipMalloc(reinterpret_cast<void**>(&mods), mod_count * sizeof(Modulus<T>)));
            std::vector<Modulus<T>> mh;
            for (const auto& p : prms)
                mh.push_back(p.modulus);
            GPUNTT_CUDA_CHECK(hipMemcpy(table, table_host.data(), table_host.size() * sizeof(T), hipMemcpyHostToDevice));
            GPUNTT_CUDA_CHECK(hipMemcpy(mods, mh.data(), mod_count * sizeof(Modulus<T>), hipMemcpyHostToDevice));
            ntt_rns_configuration<T> cfg = {.n_power = logn,
                                            .ntt_type = FORWARD,
                                            .ntt_layout = PerPolynomial,
                                            .reduction_poly = ReductionPolynomial::X_N_plus,
                                            .zero_padding = false,
                                            .stream = stream};
            bool ok = true;
            for (int round = 0; round < 2 && ok; round++)
            {
                GPUNTT_CUDA_CHECK(hipMemcpyAsync(data, coeffs.data() + off, words * sizeof(T), hipMemcpyHostToDevice, stream));
                if (round == 0)
                    GPU_NTT_Inplace(data, table, mods, cfg, sh.count, mod_count); // drop-in: the shard starts at a multiple of mod_count
                else
                {
                    NTTPlan<T> plan(table, mh.data(), mod_count, logn, ReductionPolynomial::X_N_plus, FORWARD, nullptr, sh.count,
                                    stream);
                    plan.execute(data, data, sh.count, stream);
                    GPUNTT_CUDA_CHECK(hipStreamSynchronize(stream));
                }
                std::vector<T> got(words);
                GPUNTT_CUDA_CHECK(hipMemcpyAsync(got.data(), data, words * sizeof(T), hipMemcpyDeviceToHost, stream));
                GPUNTT_CUDA_CHECK(hipStreamSynchronize(stream));
                for (size_t i = 0; i < words && ok; i++)
                    ok = got[i] == expected[off + i];
            }
            sh.ok = ok;
            (void) hipFree(data);
            (void) hipFree(table);
            (void) hipFree(mods);
            (void) hipStreamDestroy(stream);
        }
        catch (const std::exception& e)
        {
            sh.error = e.what();
        }
        std::lock_guard<std::mutex> lock(g_io);
        std::cout << "device " << sh.device << ": polynomials [" << sh.first << ", " << sh.first + sh.count << ") "
                  << (sh.ok ? "correct" : ("WRONG " + sh.error)) << std::endl;
    }
} // namespace

int main(int argc, char** argv)
{
    const int logn = argc > 1 ? std::atoi(argv[1]) : 13;
    int batch = argc > 2 ? std::atoi(argv[2]) : 64;
    int want_devices = argc > 3 ? std::atoi(argv[3]) : 0;
    const int mod_count = argc > 4 ? std::atoi(argv[4]) : 2;
    int have = 0;
    GPUNTT_CUDA_CHECK(hipGetDeviceCount(&have));
    if (have < 1 || logn < 1 || logn > 24 || mod_count < 1 || batch < mod_count)
    {
        std::cout << "usage: example_multi_device <LOGN 1..24> <BATCH >= MOD_COUNT> [DEVICES] [MOD_COUNT]" << std::endl;
        return 2;
    }
    int devices = (want_devices <= 0 || want_devices > have) ? have : want_devices;
    if (want_devices > have)
        std::cout << "asked for " << want_devices << " devices, " << have << " present: running on " << devices << std::endl;
    batch -= batch % mod_count;

    const std::vector<NTTFactors<T>> fl = stack_factors(logn, mod_count);
    if (static_cast<int>(fl.size()) < mod_count)
    {
        std::cout << "not enough pool primes for this ring" << std::endl;
        return 2;
    }
    std::vector<NTTParameters<T>> prms;
    const size_t n = size_t(1) << logn;
    std::vector<T> table_host(static_cast<size_t>(mod_count) * n, 0);
    for (int i = 0; i < mod_count; i++)
    {
        prms.emplace_back(logn, fl[i], ReductionPolynomial::X_N_plus);
        const std::vector<T> t = prms.back().gpu_root_of_unity_table_generator(prms.back().forward_root_of_unity_table);
        std::copy(t.begin(), t.end(), table_host.begin() + static_cast<size_t>(i) * n); // modulus i at i << n_power
    }
    std::mt19937_64 rng(0);
    std::vector<T> coeffs(static_cast<size_t>(batch) * n), expected(coeffs.size());
    for (int p = 0; p < batch; p++)
    {
        const NTTParameters<T>& prm = prms[p % mod_count];
        NTTCPU<T> cpu(prm);
        std::vector<T> one(n);
        for (T& c : one)
            c = rng() % prm.modulus.value;
        std::copy(one.begin(), one.end(), coeffs.begin() + static_cast<size_t>(p) * n);
        const std::vector<T> f = cpu.ntt(one);
        std::copy(f.begin(), f.end(), expected.begin() + static_cast<size_t>(p) * n);
    }

    // rank r of G owns the groups [r * groups / G, (r + 1) * groups / G) of mod_count polynomials each
    const int groups = batch / mod_count;
    if (devices > groups)
        devices = groups;
    std::vector<Shard> shards(devices);
    std::vector<std::thread> threads;
    for (int r = 0; r < devices; r++)
    {
        const int lo = static_cast<int>(static_cast<long long>(groups) * r / devices) * mod_count;
        const int hi = static_cast<int>(static_cast<long long>(groups) * (r + 1) / devices) * mod_count;
        shards[r] = Shard{r, lo, hi - lo};
    }
    for (int r = 0; r < devices; r++)
        threads.emplace_back(worker, std::ref(shards[r]), logn, mod_count, std::cref(prms), std::cref(table_host),
                             std::cref(coeffs), std::cref(expected));
    for (auto& t : threads)
        t.join();
    bool all = true;
    for (const Shard& s : shards)
        all = all && s.ok;
    if (all)
        std::cout << "All Correct on " << devices << " device(s)." << std::endl;
    return all ? 0 : 1;
}
