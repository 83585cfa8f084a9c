// ref_driver.cpp -- flat extern "C" handles over the REFERENCE's own CPU classes.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes
// the reference headers from /root/reference/src/include and is compiled by
// oracle/Makefile together with the reference sources *where they lie*
//   src/lib/common/common.cu  src/lib/common/nttparameters.cu
//   src/lib/ntt_merge/ntt_cpu.cu  src/lib/ntt_4step/ntt_4step_cpu.cu
// into oracle/_ref/libgpuntt_ref.so (git-ignored, travels with gpurun).
// The CUDA runtime headers those sources include are the genuine NVIDIA ones that
// the image already carries (triton/backends/nvidia/include); the three cuda*
// runtime functions common.cu references (CudaDevice(), never called here) stay
// unresolved and lazily bound.
//
// Used to (1) pin oracle/ntt_oracle.c, (2) generate tests/golden/, (3) serve as
// bench.py's cpu_baseline of kind "reference".
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "gpuntt/ntt_4step/ntt_4step_cpu.cuh"
#include "gpuntt/ntt_merge/ntt_cpu.cuh"

using namespace gpuntt;

namespace
{
    template <typename T> struct MergeHandle
    {
        NTTParameters<T> params;
        NTTCPU<T> cpu;
        MergeHandle(const NTTParameters<T>& p) : params(p), cpu(p) {}
    };

    template <typename T> struct FourStepHandle
    {
        NTTParameters4Step<T> params;
        NTT_4STEP_CPU<T> cpu;
        FourStepHandle(const NTTParameters4Step<T>& p) : params(p), cpu(p) {}
    };

    template <typename T>
    void* merge_create(int logn, int poly, int custom, T q, T omega, T psi)
    {
        ReductionPolynomial rp = poly ? ReductionPolynomial::X_N_minus
                                      : ReductionPolynomial::X_N_plus;
        if (custom)
        {
            NTTFactors<T> f(Modulus<T>(q), omega, psi);
            NTTParameters<T> p(logn, f, rp);
            return new MergeHandle<T>(p);
        }
        NTTParameters<T> p(logn, rp);
        return new MergeHandle<T>(p);
    }

    template <typename T> void merge_info(void* h_, uint64_t* out)
    {
        auto* h = static_cast<MergeHandle<T>*>(h_);
        const auto& p = h->params;
        out[0] = p.modulus.value;
        out[1] = p.modulus.bit;
        out[2] = p.modulus.mu;
        out[3] = p.omega;
        out[4] = p.psi;
        out[5] = p.n_inv;
        out[6] = p.root_of_unity;
        out[7] = p.inverse_root_of_unity;
        out[8] = p.root_of_unity_size;
        out[9] = p.n;
    }

    // which: 0 forward natural, 1 inverse natural, 2 forward gpu(bit-reversed), 3 inverse gpu
    template <typename T> void merge_table(void* h_, int which, T* out)
    {
        auto* h = static_cast<MergeHandle<T>*>(h_);
        auto& p = h->params;
        std::vector<T> t;
        switch (which)
        {
            case 0:
                t = p.forward_root_of_unity_table;
                break;
            case 1:
                t = p.inverse_root_of_unity_table;
                break;
            case 2:
                t = p.gpu_root_of_unity_table_generator(
                    p.forward_root_of_unity_table);
                break;
            default:
                t = p.gpu_root_of_unity_table_generator(
                    p.inverse_root_of_unity_table);
                break;
        }
        std::memcpy(out, t.data(), t.size() * sizeof(T));
    }

    template <typename T>
    void merge_run(void* h_, int inverse, const T* in, T* out, int batch)
    {
        auto* h = static_cast<MergeHandle<T>*>(h_);
        size_t n = h->params.n;
        for (int b = 0; b < batch; b++)
        {
            std::vector<T> v(in + b * n, in + (b + 1) * n);
            std::vector<T> r = inverse ? h->cpu.intt(v) : h->cpu.ntt(v);
            std::memcpy(out + b * n, r.data(), n * sizeof(T));
        }
    }

    // all-core leg of bench.py's cpu_baseline: the same NTTCPU<T>::ntt / ::intt per polynomial (it only READS its
    // parameters), the batch loop shared out over `nthreads` OpenMP threads
    template <typename T>
    void merge_run_mt(void* h_, int inverse, const T* in, T* out, int batch, int nthreads)
    {
        auto* h = static_cast<MergeHandle<T>*>(h_);
        size_t n = h->params.n;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (int b = 0; b < batch; b++)
        {
            std::vector<T> v(in + b * n, in + (b + 1) * n);
            std::vector<T> r = inverse ? h->cpu.intt(v) : h->cpu.ntt(v);
            std::memcpy(out + b * n, r.data(), n * sizeof(T));
        }
    }

    template <typename T> void* fourstep_create(int logn)
    {
        NTTParameters4Step<T> p(logn, ReductionPolynomial::X_N_minus);
        return new FourStepHandle<T>(p);
    }

    template <typename T> void fourstep_info(void* h_, uint64_t* out)
    {
        auto* h = static_cast<FourStepHandle<T>*>(h_);
        const auto& p = h->params;
        out[0] = p.modulus.value;
        out[1] = p.modulus.bit;
        out[2] = p.modulus.mu;
        out[3] = p.omega;
        out[4] = p.psi;
        out[5] = p.n_inv;
        out[6] = p.root_of_unity;
        out[7] = p.inverse_root_of_unity;
        out[8] = p.n1;
        out[9] = p.n2;
        out[10] = p.n;
    }

    // which: 0 n1 fwd, 1 n2 fwd, 2 W fwd, 3 n1 inv, 4 n2 inv, 5 W inv  (natural order);
    //        +8 -> gpu (bit-reversed) order for the small tables
    template <typename T> void fourstep_table(void* h_, int which, T* out)
    {
        auto* h = static_cast<FourStepHandle<T>*>(h_);
        auto& p = h->params;
        std::vector<T> t;
        switch (which & 7)
        {
            case 0:
                t = p.n1_based_root_of_unity_table;
                break;
            case 1:
                t = p.n2_based_root_of_unity_table;
                break;
            case 2:
                t = p.W_root_of_unity_table;
                break;
            case 3:
                t = p.n1_based_inverse_root_of_unity_table;
                break;
            case 4:
                t = p.n2_based_inverse_root_of_unity_table;
                break;
            default:
                t = p.W_inverse_root_of_unity_table;
                break;
        }
        if (which & 8)
            t = p.gpu_root_of_unity_table_generator(t);
        std::memcpy(out, t.data(), t.size() * sizeof(T));
    }

    // mode: 0 ntt, 1 intt, 2 intt_first_transpose
    template <typename T>
    void fourstep_run(void* h_, int mode, const T* in, T* out, int batch)
    {
        auto* h = static_cast<FourStepHandle<T>*>(h_);
        size_t n = h->params.n;
        for (int b = 0; b < batch; b++)
        {
            std::vector<T> v(in + b * n, in + (b + 1) * n);
            std::vector<T> r = (mode == 0)   ? h->cpu.ntt(v)
                               : (mode == 1) ? h->cpu.intt(v)
                                             : h->cpu.intt_first_transpose(v);
            std::memcpy(out + b * n, r.data(), n * sizeof(T));
        }
    }

    // NTT_4STEP_CPU<T> on CALLER-SUPPLIED tables: the public table vectors of NTTParameters4Step<T> (reference
    // src/include/gpuntt/common/nttparameters.cuh:119-170, natural order: n1/2, n2/2 and N words) of the chosen direction are
    // overwritten before the CPU class is built from the parameter set, so ::ntt / ::intt compute whatever those
    // tables say (src/lib/ntt_4step/ntt_4step_cpu.cu:33-111) -- the CPU-side meaning of a GPU_4STEP_NTT call with
    // arbitrary tables (src/lib/ntt_4step/ntt_4step.cu:1049-1058, 776-779).  q != 0 replaces the modulus, n_inv the
    // final scaling of ::intt.
    template <typename T>
    void fourstep_run_tables(int logn, int inverse, const T* n1_table, const T* n2_table, const T* W, T q, T n_inv,
                             const T* in, T* out, int batch)
    {
        NTTParameters4Step<T> p(logn, ReductionPolynomial::X_N_minus);
        if (q != 0)
            p.modulus = Modulus<T>(q);
        const size_t h1 = static_cast<size_t>(p.n1) >> 1, h2 = static_cast<size_t>(p.n2) >> 1, n = p.n;
        if (inverse)
        {
            p.n1_based_inverse_root_of_unity_table.assign(n1_table, n1_table + h1);
            p.n2_based_inverse_root_of_unity_table.assign(n2_table, n2_table + h2);
            p.W_inverse_root_of_unity_table.assign(W, W + n);
            p.n_inv = n_inv;
        }
        else
        {
            p.n1_based_root_of_unity_table.assign(n1_table, n1_table + h1);
            p.n2_based_root_of_unity_table.assign(n2_table, n2_table + h2);
            p.W_root_of_unity_table.assign(W, W + n);
        }
        NTT_4STEP_CPU<T> cpu(p);
        for (int b = 0; b < batch; b++)
        {
            std::vector<T> v(in + b * n, in + (b + 1) * n);
            std::vector<T> r = inverse ? cpu.intt(v) : cpu.ntt(v);
            std::memcpy(out + b * n, r.data(), n * sizeof(T));
        }
    }

    template <typename T>
    void schoolbook(const T* a, const T* b, T* out, int n, T q, int poly)
    {
        std::vector<T> va(a, a + n), vb(b, b + n);
        std::vector<T> r = schoolbook_poly_multiplication<T>(
            va, vb, Modulus<T>(q),
            poly ? ReductionPolynomial::X_N_minus : ReductionPolynomial::X_N_plus);
        std::memcpy(out, r.data(), n * sizeof(T));
    }
} // namespace

#define REF_EXPORTS(S, T)                                                            \
    void* ref##S##_merge_create(int logn, int poly, int custom, T q, T omega, T psi) \
    {                                                                                \
        return merge_create<T>(logn, poly, custom, q, omega, psi);                   \
    }                                                                                \
    void ref##S##_merge_destroy(void* h)                                             \
    {                                                                                \
        delete static_cast<MergeHandle<T>*>(h);                                      \
    }                                                                                \
    void ref##S##_merge_info(void* h, uint64_t* out) { merge_info<T>(h, out); }      \
    void ref##S##_merge_table(void* h, int which, T* out)                            \
    {                                                                                \
        merge_table<T>(h, which, out);                                               \
    }                                                                                \
    void ref##S##_merge_run(void* h, int inverse, const T* in, T* out, int batch)    \
    {                                                                                \
        merge_run<T>(h, inverse, in, out, batch);                                    \
    }                                                                                \
    void ref##S##_merge_run_mt(void* h, int inverse, const T* in, T* out, int batch, \
                               int nthreads)                                         \
    {                                                                                \
        merge_run_mt<T>(h, inverse, in, out, batch, nthreads);                       \
    }                                                                                \
    void ref##S##_pointwise(void* h, T* a, T* b, T* out)                             \
    {                                                                                \
        auto* mh = static_cast<MergeHandle<T>*>(h);                                  \
        size_t n = mh->params.n;                                                     \
        std::vector<T> va(a, a + n), vb(b, b + n);                                   \
        std::vector<T> r = mh->cpu.mult(va, vb);                                     \
        std::memcpy(out, r.data(), n * sizeof(T));                                   \
    }                                                                                \
    void* ref##S##_4step_create(int logn) { return fourstep_create<T>(logn); }       \
    void ref##S##_4step_destroy(void* h)                                             \
    {                                                                                \
        delete static_cast<FourStepHandle<T>*>(h);                                   \
    }                                                                                \
    void ref##S##_4step_info(void* h, uint64_t* out) { fourstep_info<T>(h, out); }   \
    void ref##S##_4step_table(void* h, int which, T* out)                            \
    {                                                                                \
        fourstep_table<T>(h, which, out);                                            \
    }                                                                                \
    void ref##S##_4step_run(void* h, int mode, const T* in, T* out, int batch)       \
    {                                                                                \
        fourstep_run<T>(h, mode, in, out, batch);                                    \
    }                                                                                \
    void ref##S##_4step_run_tables(int logn, int inverse, const T* n1_table,         \
                                   const T* n2_table, const T* W, T q, T n_inv,      \
                                   const T* in, T* out, int batch)                   \
    {                                                                                \
        fourstep_run_tables<T>(logn, inverse, n1_table, n2_table, W, q, n_inv, in,   \
                               out, batch);                                          \
    }                                                                                \
    void ref##S##_schoolbook(const T* a, const T* b, T* out, int n, T q, int poly)   \
    {                                                                                \
        schoolbook<T>(a, b, out, n, q, poly);                                        \
    }                                                                                \
    void ref##S##_modulus(T q, T* bit, T* mu)                                        \
    {                                                                                \
        Modulus<T> m(q);                                                             \
        *bit = m.bit;                                                                \
        *mu = m.mu;                                                                  \
    }                                                                                \
    T ref##S##_mult(T a, T b, T q) { return OPERATOR<T>::mult(a, b, Modulus<T>(q)); } \
    T ref##S##_exp(T a, T e, T q) { return OPERATOR<T>::exp(a, e, Modulus<T>(q)); }  \
    T ref##S##_modinv(T a, T q) { return OPERATOR<T>::modinv(a, Modulus<T>(q)); }

extern "C"
{
    REF_EXPORTS(32, Data32)
    REF_EXPORTS(64, Data64)

    int ref_bitreverse(int index, int n_power) { return bitreverse(index, n_power); }

    // The deterministic input stream of the reference's GPU merge examples
    // (example/ntt_merge/test_merge_ntt.cu:70-84): std::mt19937 gen(seed) +
    // std::uniform_int_distribution<uint64_t>(0, q-1), libstdc++ semantics.
    void ref_mt19937_uniform(uint32_t seed, uint64_t q, uint64_t count, uint64_t* out)
    {
        std::mt19937 gen(seed);
        std::uniform_int_distribution<std::uint64_t> dis(0, q - 1);
        for (uint64_t i = 0; i < count; i++)
            out[i] = dis(gen);
    }
}
