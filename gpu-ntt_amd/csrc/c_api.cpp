// c_api.cpp -- extern "C" boundary (include/gpuntt_c.h) over the C++ template API.
#include <algorithm>
#include <cstring>
#include <exception>
#include <initializer_list>
#include <stdexcept>
#include <string>
#include <vector>

#include "gpuntt/ntt_4step/ntt_4step.cuh"
#include "gpuntt/ntt_merge/ntt.cuh"
#include "gpuntt_c.h"
#include "test_hooks.h"

namespace gpuntt
{
    namespace host
    {
        // prep.hip (diagnostic): the preparation kernels' normalised reciprocal of every q[i]
        template <typename T> void debug_recip_norm(const T* q, T* out, unsigned long long count, hipStream_t stream);
        // prep.hip: test hooks (csrc/test_hooks.h)
        bool set_test_hook(const char* name, const char* value);
        void launch_log_start();
        std::string launch_log_take();
        void scratch_stats(unsigned long long out[6]);
    } // namespace host
} // namespace gpuntt

using namespace gpuntt;

namespace
{
    thread_local std::string g_last_error;

    template <typename F> int guarded(F&& f)
    {
        try
        {
            f();
            return GPUNTT_OK;
        }
        catch (const std::invalid_argument& e)
        {
            g_last_error = e.what();
            return GPUNTT_ERR_INVALID_ARGUMENT;
        }
        catch (const HipException& e)
        {
            g_last_error = e.what();
            return GPUNTT_ERR_HIP;
        }
        catch (const std::exception& e)
        {
            g_last_error = e.what();
            return GPUNTT_ERR_UNKNOWN;
        }
    }

    // device / host pointer arguments of the C entry points must not be NULL (the C++ templates,
    // like the reference's, do not check)
    inline int need(std::initializer_list<const void*> ptrs)
    {
        for (const void* p : ptrs)
            if (p == nullptr)
            {
                g_last_error = "null pointer argument";
                return GPUNTT_ERR_INVALID_ARGUMENT;
            }
        return GPUNTT_OK;
    }
#define GPUNTT_NEED(...)                                                                          \
    if (int rc_ = need({__VA_ARGS__}))                                                            \
        return rc_;

    template <typename T, typename CM> Modulus<T> to_mod(const CM& m)
    {
        Modulus<T> r;
        r.value = m.value;
        r.bit = m.bit;
        r.mu = m.mu;
        return r;
    }

    template <typename T, typename CM>
    int ntt_single(const void* in, T* out, const T* roots, CM modulus, int n_power, int layout,
                   int poly, int input_signed, void* stream, int batch)
    {
        using S = typename std::make_signed<T>::type;
        return guarded([&] {
            ntt_configuration<T> cfg = {n_power,
                                        FORWARD,
                                        static_cast<NTTLayout>(layout),
                                        static_cast<ReductionPolynomial>(poly),
                                        false,
                                        0,
                                        static_cast<hipStream_t>(stream)};
            if (input_signed)
                GPU_NTT<S>(static_cast<S*>(const_cast<void*>(in)), out, const_cast<T*>(roots),
                           to_mod<T>(modulus), cfg, batch);
            else
                GPU_NTT<T>(static_cast<T*>(const_cast<void*>(in)), out, const_cast<T*>(roots),
                           to_mod<T>(modulus), cfg, batch);
        });
    }

    template <typename T, typename CM>
    int intt_single(const T* in, void* out, const T* roots, CM modulus, int n_power, int layout,
                    int poly, T mod_inverse, int output_signed, void* stream, int batch)
    {
        using S = typename std::make_signed<T>::type;
        return guarded([&] {
            ntt_configuration<T> cfg = {n_power,
                                        INVERSE,
                                        static_cast<NTTLayout>(layout),
                                        static_cast<ReductionPolynomial>(poly),
                                        false,
                                        mod_inverse,
                                        static_cast<hipStream_t>(stream)};
            if (output_signed)
                GPU_INTT<S>(const_cast<T*>(in), static_cast<S*>(out), const_cast<T*>(roots),
                            to_mod<T>(modulus), cfg, batch);
            else
                GPU_INTT<T>(const_cast<T*>(in), static_cast<T*>(out), const_cast<T*>(roots),
                            to_mod<T>(modulus), cfg, batch);
        });
    }

    template <typename T, typename CM>
    int ntt_rns(const void* in, T* out, const T* roots, const CM* modulus, int n_power, int layout,
                int poly, int input_signed, void* stream, int batch, int mod_count)
    {
        using S = typename std::make_signed<T>::type;
        static_assert(sizeof(CM) == sizeof(Modulus<T>), "C and C++ modulus layouts differ");
        return guarded([&] {
            ntt_rns_configuration<T> cfg = {n_power,
                                            FORWARD,
                                            static_cast<NTTLayout>(layout),
                                            static_cast<ReductionPolynomial>(poly),
                                            false,
                                            nullptr,
                                            static_cast<hipStream_t>(stream)};
            auto* mods = reinterpret_cast<Modulus<T>*>(const_cast<CM*>(modulus));
            if (input_signed)
                GPU_NTT<S>(static_cast<S*>(const_cast<void*>(in)), out, const_cast<T*>(roots), mods,
                           cfg, batch, mod_count);
            else
                GPU_NTT<T>(static_cast<T*>(const_cast<void*>(in)), out, const_cast<T*>(roots), mods,
                           cfg, batch, mod_count);
        });
    }

    template <typename T, typename CM>
    int intt_rns(const T* in, void* out, const T* roots, const CM* modulus, int n_power, int layout,
                 int poly, const T* mod_inverse, int output_signed, void* stream, int batch,
                 int mod_count)
    {
        using S = typename std::make_signed<T>::type;
        return guarded([&] {
            ntt_rns_configuration<T> cfg = {n_power,
                                            INVERSE,
                                            static_cast<NTTLayout>(layout),
                                            static_cast<ReductionPolynomial>(poly),
                                            false,
                                            const_cast<T*>(mod_inverse),
                                            static_cast<hipStream_t>(stream)};
            auto* mods = reinterpret_cast<Modulus<T>*>(const_cast<CM*>(modulus));
            if (output_signed)
                GPU_INTT<S>(const_cast<T*>(in), static_cast<S*>(out), const_cast<T*>(roots), mods,
                            cfg, batch, mod_count);
            else
                GPU_INTT<T>(const_cast<T*>(in), static_cast<T*>(out), const_cast<T*>(roots), mods,
                            cfg, batch, mod_count);
        });
    }

    template <typename T, typename CM>
    int polymul_single(T* a, T* b, T* out, const T* fwd, const T* inv, CM modulus, int n_power,
                       int reduction_poly, T mod_inverse, void* stream, int batch)
    {
        return guarded([&] {
            ntt_configuration<T> cfg = {n_power, FORWARD, PerPolynomial,
                                        static_cast<ReductionPolynomial>(reduction_poly), false, mod_inverse,
                                        static_cast<hipStream_t>(stream)};
            GPU_PolyMul<T>(a, b, out, const_cast<T*>(fwd), const_cast<T*>(inv), to_mod<T>(modulus), cfg, batch);
        });
    }
    template <typename T, typename CM>
    int polymul_rns(T* a, T* b, T* out, const T* fwd, const T* inv, const CM* modulus, int n_power,
                    int reduction_poly, const T* mod_inverse, void* stream, int batch, int mod_count)
    {
        return guarded([&] {
            ntt_rns_configuration<T> cfg = {n_power, FORWARD, PerPolynomial,
                                            static_cast<ReductionPolynomial>(reduction_poly), false,
                                            const_cast<T*>(mod_inverse), static_cast<hipStream_t>(stream)};
            GPU_PolyMul<T>(a, b, out, const_cast<T*>(fwd), const_cast<T*>(inv),
                           reinterpret_cast<Modulus<T>*>(const_cast<CM*>(modulus)), cfg, batch, mod_count);
        });
    }

    template <typename T, typename CM>
    int fourstep_single(const T* in, T* out, const T* t1, const T* t2, const T* w, CM modulus,
                        int n_power, int ntt_type, T mod_inverse, void* stream, int batch)
    {
        return guarded([&] {
            ntt4step_configuration<T> cfg = {n_power, static_cast<type>(ntt_type), mod_inverse,
                                             static_cast<hipStream_t>(stream)};
            GPU_4STEP_NTT<T>(const_cast<T*>(in), out, const_cast<T*>(t1), const_cast<T*>(t2),
                             const_cast<T*>(w), to_mod<T>(modulus), cfg, batch);
        });
    }

    template <typename T, typename CM>
    int fourstep_natural(T* in, T* out, const T* t1, const T* t2, const T* w, CM modulus, int n_power,
                         int ntt_type, T mod_inverse, void* stream, int batch)
    {
        return guarded([&] {
            ntt4step_configuration<T> cfg = {n_power, static_cast<type>(ntt_type), mod_inverse,
                                             static_cast<hipStream_t>(stream)};
            GPU_4STEP_NTT_NaturalOrder<T>(in, out, const_cast<T*>(t1), const_cast<T*>(t2), const_cast<T*>(w),
                                          to_mod<T>(modulus), cfg, batch);
        });
    }

    template <typename T, typename CM>
    int fourstep_rns(const T* in, T* out, const T* t1, const T* t2, const T* w, const CM* modulus,
                     int n_power, int ntt_type, const T* mod_inverse, void* stream, int batch,
                     int mod_count)
    {
        return guarded([&] {
            ntt4step_rns_configuration<T> cfg = {n_power, static_cast<type>(ntt_type),
                                                 const_cast<T*>(mod_inverse),
                                                 static_cast<hipStream_t>(stream)};
            GPU_4STEP_NTT<T>(const_cast<T*>(in), out, const_cast<T*>(t1), const_cast<T*>(t2),
                             const_cast<T*>(w),
                             reinterpret_cast<Modulus<T>*>(const_cast<CM*>(modulus)), cfg, batch,
                             mod_count);
        });
    }

    template <typename T, typename CM>
    int ordered(bool poly_ordered, const T* in, T* out, const T* roots, const CM* modulus, int n_power,
                int ntt_type, int poly, const T* mod_inverse, void* stream, int batch, int mod_count,
                const int* order)
    {
        return guarded([&] {
            ntt_rns_configuration<T> cfg = {n_power,
                                            static_cast<type>(ntt_type),
                                            PerPolynomial,
                                            static_cast<ReductionPolynomial>(poly),
                                            false,
                                            const_cast<T*>(mod_inverse),
                                            static_cast<hipStream_t>(stream)};
            auto* mods = reinterpret_cast<Modulus<T>*>(const_cast<CM*>(modulus));
            if (poly_ordered)
                GPU_NTT_Poly_Ordered<T>(const_cast<T*>(in), out, const_cast<T*>(roots), mods, cfg, batch,
                                        mod_count, const_cast<int*>(order));
            else
                GPU_NTT_Modulus_Ordered<T>(const_cast<T*>(in), out, const_cast<T*>(roots), mods, cfg, batch,
                                           mod_count, const_cast<int*>(order));
        });
    }

    template <typename T>
    int merge_params(int logn, int poly, const T* factors, uint64_t* info, T* fwd, T* inv)
    {
        return guarded([&] {
            if (poly != GPUNTT_X_N_PLUS && poly != GPUNTT_X_N_MINUS)
                throw std::invalid_argument("Invalid reduction_poly!");
            const auto rp = static_cast<ReductionPolynomial>(poly);
            NTTParameters<T> p = factors ? NTTParameters<T>(logn,
                                                            NTTFactors<T>(Modulus<T>(factors[0]),
                                                                          factors[1], factors[2]),
                                                            rp)
                                         : NTTParameters<T>(logn, rp);
            if (info)
            {
                info[0] = p.modulus.value;
                info[1] = p.modulus.bit;
                info[2] = p.modulus.mu;
                info[3] = p.omega;
                info[4] = p.psi;
                info[5] = p.n_inv;
                info[6] = p.root_of_unity_size;
                info[7] = p.n;
            }
            if (fwd)
            {
                auto t = p.gpu_root_of_unity_table_generator(p.forward_root_of_unity_table);
                std::memcpy(fwd, t.data(), t.size() * sizeof(T));
            }
            if (inv)
            {
                auto t = p.gpu_root_of_unity_table_generator(p.inverse_root_of_unity_table);
                std::memcpy(inv, t.data(), t.size() * sizeof(T));
            }
        });
    }

    template <typename T>
    int fourstep_params(int logn, int inverse, uint64_t* info, T* t1, T* t2, T* w)
    {
        return guarded([&] {
            NTTParameters4Step<T> p(logn, ReductionPolynomial::X_N_minus);
            if (info)
            {
                info[0] = p.modulus.value;
                info[1] = p.modulus.bit;
                info[2] = p.modulus.mu;
                info[3] = p.omega;
                info[4] = p.psi;
                info[5] = p.n_inv;
                info[6] = p.n1;
                info[7] = p.n2;
                info[8] = p.n;
            }
            if (t1)
            {
                auto t = p.gpu_root_of_unity_table_generator(
                    inverse ? p.n1_based_inverse_root_of_unity_table : p.n1_based_root_of_unity_table);
                std::memcpy(t1, t.data(), t.size() * sizeof(T));
            }
            if (t2)
            {
                auto t = p.gpu_root_of_unity_table_generator(
                    inverse ? p.n2_based_inverse_root_of_unity_table : p.n2_based_root_of_unity_table);
                std::memcpy(t2, t.data(), t.size() * sizeof(T));
            }
            if (w)
            {
                const auto& t = inverse ? p.W_inverse_root_of_unity_table : p.W_root_of_unity_table;
                std::memcpy(w, t.data(), t.size() * sizeof(T));
            }
        });
    }
} // namespace

namespace
{
    // OPERATOR_GPU<T> applied elementwise (diagnostic entry point: the public device class on real hardware)
    template <typename T>
    __global__ __launch_bounds__(256) void operator_gpu_apply(int op, const T* a, const T* b, T* out, Modulus<T> m,
                                                              unsigned long long count)
    {
        using S = typename std::make_signed<T>::type;
        for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < count;
             i += static_cast<unsigned long long>(gridDim.x) * 256ull)
        {
            const T x = a[i];
            const T y = (b != nullptr) ? b[i] : T(0);
            T r = 0;
            switch (op)
            {
                case 0: r = OPERATOR_GPU<T>::add(x, y, m); break;
                case 1: r = OPERATOR_GPU<T>::sub(x, y, m); break;
                case 2: r = OPERATOR_GPU<T>::mult(x, y, m); break;
                case 3: r = OPERATOR_GPU<T>::reduce(x, m); break;
                case 4: r = OPERATOR_GPU<T>::reduce(static_cast<S>(x), m); break;
                default: r = static_cast<T>(OPERATOR_GPU<T>::centered_reduction(x, m)); break;
            }
            out[i] = r;
        }
    }
    template <typename T, typename CM>
    int operator_gpu(int op, const T* a, const T* b, T* out, const CM& cm, uint64_t count, void* stream)
    {
        return guarded([&] {
            if (op < 0 || op > 5)
                throw std::invalid_argument("Invalid operator!");
            if (count == 0)
                return;
            unsigned long long blocks = (count + 255) / 256;
            if (blocks > 65536)
                blocks = 65536;
            hipLaunchKernelGGL((operator_gpu_apply<T>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                               static_cast<hipStream_t>(stream), op, a, b, out, to_mod<T>(cm), count);
            GPUNTT_HIP_CHECK(hipGetLastError());
        });
    }

    // the public device butterflies CooleyTukeyUnit / GentlemanSandeUnit (gpuntt/ntt_merge/ntt.cuh) applied pairwise
    template <typename T>
    __global__ __launch_bounds__(256) void butterfly_unit_apply(int gs, T* u, T* v, const T* roots, Modulus<T> m,
                                                                unsigned long long count)
    {
        for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < count;
             i += static_cast<unsigned long long>(gridDim.x) * 256ull)
        {
            T U = u[i], V = v[i];
            if (gs)
                GentlemanSandeUnit<T>(U, V, roots[i], m);
            else
                CooleyTukeyUnit<T>(U, V, roots[i], m);
            u[i] = U;
            v[i] = V;
        }
    }
    template <typename T, typename CM>
    int butterfly_unit(int gs, T* u, T* v, const T* roots, const CM& cm, uint64_t count, void* stream)
    {
        return guarded([&] {
            if (count == 0)
                return;
            unsigned long long blocks = (count + 255) / 256;
            if (blocks > 65536)
                blocks = 65536;
            hipLaunchKernelGGL((butterfly_unit_apply<T>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                               static_cast<hipStream_t>(stream), gs, u, v, roots, to_mod<T>(cm), count);
            GPUNTT_HIP_CHECK(hipGetLastError());
        });
    }

    template <typename T, typename CM>
    int plan_create(gpuntt_plan** plan, const T* table, const CM* moduli, int mod_count, int n_power, int poly,
                    int ntt_type, const T* ninv, int batch_hint, void* ws, void* stream)
    {
        return guarded([&] {
            if (plan == nullptr || moduli == nullptr || mod_count <= 0)
                throw std::invalid_argument("Invalid mod_count!");
            std::vector<Modulus<T>> ms;
            for (int i = 0; i < mod_count; i++)
                ms.push_back(to_mod<T>(moduli[i]));
            auto* p = new NTTPlan<T>(table, ms.data(), mod_count, n_power, static_cast<ReductionPolynomial>(poly),
                                     static_cast<type>(ntt_type), ninv, batch_hint,
                                     static_cast<hipStream_t>(stream), ws);
            *plan = reinterpret_cast<gpuntt_plan*>(p);
        });
    }
} // namespace

extern "C"
{
    const char* gpuntt_last_error(void) { return g_last_error.c_str(); }
    int gpuntt_release_workspaces(void)
    {
        return guarded([] { GPU_NTT_ReleaseWorkspaces(); });
    }
    int gpuntt_set_option(const char* name, const char* value)
    {
        return guarded([&] {
            if (!GPU_NTT_SetOption(name, value))
                throw std::invalid_argument("Unknown option or value!");
        });
    }
    int gpuntt_version(void) { return 101; }

    // ---- test hooks (csrc/test_hooks.h; not declared by the public headers) ----------------------------------------
    int gpuntt_test_set_hook(const char* name, const char* value)
    {
        return guarded([&] {
            if (!host::set_test_hook(name, value))
                throw std::invalid_argument("Unknown option or value!");
        });
    }
    int gpuntt_test_launch_log_start(void)
    {
        return guarded([] { host::launch_log_start(); });
    }
    int gpuntt_test_launch_log_take(char* buf, int capacity)
    {
        const std::string s = host::launch_log_take();
        if (buf != nullptr && capacity > 0)
        {
            const size_t n = std::min(s.size(), static_cast<size_t>(capacity - 1));
            std::memcpy(buf, s.data(), n);
            buf[n] = '\0';
        }
        return static_cast<int>(s.size()) + 1;
    }

    int gpuntt_test_scratch_stats(unsigned long long out[6])
    {
        return guarded([&] { host::scratch_stats(out); });
    }

    int gpuntt_modulus_u32(uint32_t q, gpuntt_modulus32* out)
    {
        return guarded([&] {
            if (!out || q < 2)
                throw std::invalid_argument("Invalid modulus!");
            Modulus<Data32> m(q);
            out->value = m.value;
            out->bit = m.bit;
            out->mu = m.mu;
        });
    }
    int gpuntt_modulus_u64(uint64_t q, gpuntt_modulus64* out)
    {
        return guarded([&] {
            if (!out || q < 2)
                throw std::invalid_argument("Invalid modulus!");
            Modulus<Data64> m(q);
            out->value = m.value;
            out->bit = m.bit;
            out->mu = m.mu;
        });
    }

#define GPUNTT_C_API(S, T, CM)                                                                    \
    int gpuntt_ntt_##S(const void* in, T* out, const T* roots, CM modulus, int n_power,           \
                       int ntt_layout, int reduction_poly, int input_signed, void* stream,        \
                       int batch_size)                                                            \
    {                                                                                             \
        GPUNTT_NEED(in, out, roots)                                                                      \
        return ntt_single<T>(in, out, roots, modulus, n_power, ntt_layout, reduction_poly,        \
                             input_signed, stream, batch_size);                                   \
    }                                                                                             \
    int gpuntt_intt_##S(const T* in, void* out, const T* inverse_roots, CM modulus, int n_power,  \
                        int ntt_layout, int reduction_poly, T mod_inverse, int output_signed,     \
                        void* stream, int batch_size)                                             \
    {                                                                                             \
        GPUNTT_NEED(in, out, inverse_roots)                                                              \
        return intt_single<T>(in, out, inverse_roots, modulus, n_power, ntt_layout,               \
                              reduction_poly, mod_inverse, output_signed, stream, batch_size);    \
    }                                                                                             \
    int gpuntt_ntt_rns_##S(const void* in, T* out, const T* roots, const CM* modulus,             \
                           int n_power, int ntt_layout, int reduction_poly, int input_signed,     \
                           void* stream, int batch_size, int mod_count)                           \
    {                                                                                             \
        GPUNTT_NEED(in, out, roots, modulus)                                                             \
        return ntt_rns<T>(in, out, roots, modulus, n_power, ntt_layout, reduction_poly,           \
                          input_signed, stream, batch_size, mod_count);                           \
    }                                                                                             \
    int gpuntt_intt_rns_##S(const T* in, void* out, const T* inverse_roots, const CM* modulus,    \
                            int n_power, int ntt_layout, int reduction_poly,                      \
                            const T* mod_inverse, int output_signed, void* stream,                \
                            int batch_size, int mod_count)                                        \
    {                                                                                             \
        GPUNTT_NEED(in, out, inverse_roots, modulus, mod_inverse)                                        \
        return intt_rns<T>(in, out, inverse_roots, modulus, n_power, ntt_layout, reduction_poly,  \
                           mod_inverse, output_signed, stream, batch_size, mod_count);            \
    }                                                                                             \
    int gpuntt_ntt_modulus_ordered_##S(const T* in, T* out, const T* roots, const CM* modulus,    \
                                       int n_power, int ntt_type, int reduction_poly,             \
                                       const T* mod_inverse, void* stream, int batch_size,        \
                                       int mod_count, const int* order)                           \
    {                                                                                             \
        GPUNTT_NEED(in, out, roots, modulus, order)                                                      \
        return ordered<T>(false, in, out, roots, modulus, n_power, ntt_type, reduction_poly,      \
                          mod_inverse, stream, batch_size, mod_count, order);                     \
    }                                                                                             \
    int gpuntt_ntt_poly_ordered_##S(const T* in, T* out, const T* roots, const CM* modulus,       \
                                    int n_power, int ntt_type, int reduction_poly,                \
                                    const T* mod_inverse, void* stream, int batch_size,           \
                                    int mod_count, const int* order)                              \
    {                                                                                             \
        GPUNTT_NEED(in, out, roots, modulus, order)                                                      \
        return ordered<T>(true, in, out, roots, modulus, n_power, ntt_type, reduction_poly,       \
                          mod_inverse, stream, batch_size, mod_count, order);                     \
    }                                                                                             \
    int gpuntt_4step_##S(const T* in, T* out, const T* n1_table, const T* n2_table,               \
                         const T* w_table, CM modulus, int n_power, int ntt_type, T mod_inverse,  \
                         void* stream, int batch_size)                                            \
    {                                                                                             \
        GPUNTT_NEED(in, out, n1_table, n2_table, w_table)                                                \
        return fourstep_single<T>(in, out, n1_table, n2_table, w_table, modulus, n_power,         \
                                  ntt_type, mod_inverse, stream, batch_size);                     \
    }                                                                                             \
    int gpuntt_polymul_##S(T* a, T* b, T* out, const T* forward_table, const T* inverse_table,     \
                           CM modulus, int n_power, int reduction_poly, T mod_inverse, void* stream, \
                           int batch_size)                                                          \
    {                                                                                             \
        GPUNTT_NEED(a, b, out, forward_table, inverse_table)                                             \
        return polymul_single<T>(a, b, out, forward_table, inverse_table, modulus, n_power,        \
                                 reduction_poly, mod_inverse, stream, batch_size);                \
    }                                                                                             \
    int gpuntt_polymul_rns_##S(T* a, T* b, T* out, const T* forward_table, const T* inverse_table, \
                               const CM* modulus, int n_power, int reduction_poly,                 \
                               const T* mod_inverse, void* stream, int batch_size, int mod_count)  \
    {                                                                                             \
        GPUNTT_NEED(a, b, out, forward_table, inverse_table, modulus, mod_inverse)                       \
        return polymul_rns<T>(a, b, out, forward_table, inverse_table, modulus, n_power,           \
                              reduction_poly, mod_inverse, stream, batch_size, mod_count);        \
    }                                                                                             \
    int gpuntt_4step_natural_##S(T* in_scratch, T* out, const T* n1_table, const T* n2_table,      \
                                 const T* w_table, CM modulus, int n_power, int ntt_type,          \
                                 T mod_inverse, void* stream, int batch_size)                      \
    {                                                                                             \
        GPUNTT_NEED(in_scratch, out, n1_table, n2_table, w_table)                                        \
        return fourstep_natural<T>(in_scratch, out, n1_table, n2_table, w_table, modulus, n_power, \
                                   ntt_type, mod_inverse, stream, batch_size);                    \
    }                                                                                             \
    int gpuntt_4step_rns_##S(const T* in, T* out, const T* n1_table, const T* n2_table,           \
                             const T* w_table, const CM* modulus, int n_power, int ntt_type,      \
                             const T* mod_inverse, void* stream, int batch_size, int mod_count)   \
    {                                                                                             \
        GPUNTT_NEED(in, out, n1_table, n2_table, w_table, modulus)                                       \
        return fourstep_rns<T>(in, out, n1_table, n2_table, w_table, modulus, n_power, ntt_type,  \
                               mod_inverse, stream, batch_size, mod_count);                       \
    }                                                                                             \
    int gpuntt_transpose_##S(const T* in, T* out, int row, int col, int n_power, int batch_size)  \
    {                                                                                             \
        GPUNTT_NEED(in, out)                                                                             \
        return guarded(                                                                           \
            [&] { GPU_Transpose<T>(const_cast<T*>(in), out, row, col, n_power, batch_size); });   \
    }                                                                                             \
    int gpuntt_merge_params_##S(int logn, int reduction_poly, const T* factors_host,              \
                                uint64_t* info_host, T* forward_table_host,                       \
                                T* inverse_table_host)                                            \
    {                                                                                             \
        return merge_params<T>(logn, reduction_poly, factors_host, info_host,                     \
                               forward_table_host, inverse_table_host);                           \
    }                                                                                             \
    int gpuntt_4step_params_##S(int logn, int inverse, uint64_t* info_host, T* n1_table_host,     \
                                T* n2_table_host, T* w_table_host)                                \
    {                                                                                             \
        return fourstep_params<T>(logn, inverse, info_host, n1_table_host, n2_table_host,         \
                                  w_table_host);                                                  \
    }                                                                                             \
    int gpuntt_plan_workspace_bytes_##S(int n_power, int mod_count, uint64_t* bytes_host)         \
    {                                                                                             \
        GPUNTT_NEED(bytes_host)                                                                   \
        return guarded([&] { *bytes_host = NTTPlan<T>::workspace_bytes(n_power, mod_count); });    \
    }                                                                                             \
    int gpuntt_plan_create_##S(gpuntt_plan** plan_host, const T* table, const CM* moduli_host,    \
                               int mod_count, int n_power, int reduction_poly, int ntt_type,      \
                               const T* mod_inverse_host, int batch_hint, void* workspace_device, \
                               void* stream)                                                      \
    {                                                                                             \
        GPUNTT_NEED(plan_host, table, moduli_host)                                                \
        return plan_create<T>(plan_host, table, moduli_host, mod_count, n_power, reduction_poly,  \
                              ntt_type, mod_inverse_host, batch_hint, workspace_device, stream);  \
    }                                                                                             \
    int gpuntt_plan_execute_##S(const gpuntt_plan* plan, const void* in, void* out,               \
                                int batch_size, int io_signed, void* stream)                      \
    {                                                                                             \
        GPUNTT_NEED(plan, in, out)                                                                \
        return guarded([&] {                                                                      \
            reinterpret_cast<const NTTPlan<T>*>(plan)->execute(in, out, batch_size,               \
                                                               static_cast<hipStream_t>(stream),  \
                                                               io_signed != 0);                   \
        });                                                                                       \
    }                                                                                             \
    int gpuntt_plan_fast_path_##S(const gpuntt_plan* plan)                                        \
    {                                                                                             \
        GPUNTT_NEED(plan)                                                                         \
        return reinterpret_cast<const NTTPlan<T>*>(plan)->fast_path() ? 1 : 0;                    \
    }                                                                                             \
    int gpuntt_plan_destroy_##S(gpuntt_plan* plan)                                                \
    {                                                                                             \
        return guarded([&] { delete reinterpret_cast<NTTPlan<T>*>(plan); });                      \
    }                                                                                             \
    int gpuntt_4step_plan_workspace_bytes_##S(int n_power, uint64_t* bytes_host)                  \
    {                                                                                             \
        GPUNTT_NEED(bytes_host)                                                                   \
        return guarded([&] { *bytes_host = FourStepPlan<T>::workspace_bytes(n_power); });          \
    }                                                                                             \
    int gpuntt_4step_plan_create_##S(gpuntt_4step_plan** plan_host, const T* n1_table,            \
                                     const T* n2_table, const T* w_table, CM modulus, int n_power, \
                                     int ntt_type, T mod_inverse, int natural_order,              \
                                     int batch_hint, void* workspace_device, void* stream)        \
    {                                                                                             \
        GPUNTT_NEED(plan_host, n1_table, n2_table, w_table)                                       \
        return guarded([&] {                                                                      \
            ntt4step_configuration<T> cfg = {n_power, static_cast<type>(ntt_type), mod_inverse,   \
                                             static_cast<hipStream_t>(stream)};                   \
            *plan_host = reinterpret_cast<gpuntt_4step_plan*>(new FourStepPlan<T>(                \
                const_cast<T*>(n1_table), const_cast<T*>(n2_table), const_cast<T*>(w_table),      \
                to_mod<T>(modulus), cfg, natural_order != 0, batch_hint, workspace_device));      \
        });                                                                                       \
    }                                                                                             \
    int gpuntt_4step_plan_execute_##S(const gpuntt_4step_plan* plan, T* in, T* out,               \
                                      int batch_size, void* stream)                               \
    {                                                                                             \
        GPUNTT_NEED(plan, in, out)                                                                \
        return guarded([&] {                                                                      \
            reinterpret_cast<const FourStepPlan<T>*>(plan)->execute(in, out, batch_size,          \
                                                                    static_cast<hipStream_t>(stream)); \
        });                                                                                       \
    }                                                                                             \
    int gpuntt_4step_plan_fast_path_##S(const gpuntt_4step_plan* plan)                            \
    {                                                                                             \
        GPUNTT_NEED(plan)                                                                         \
        return reinterpret_cast<const FourStepPlan<T>*>(plan)->fast_path() ? 1 : 0;               \
    }                                                                                             \
    int gpuntt_4step_plan_destroy_##S(gpuntt_4step_plan* plan)                                    \
    {                                                                                             \
        return guarded([&] { delete reinterpret_cast<FourStepPlan<T>*>(plan); });                 \
    }                                                                                             \
    int gpuntt_generate_power_table_##S(T* out, T base, CM modulus, int log_count,                \
                                        int bit_reversed, void* stream)                           \
    {                                                                                             \
        GPUNTT_NEED(out)                                                                          \
        return guarded([&] {                                                                      \
            GPU_GeneratePowerTable<T>(out, base, to_mod<T>(modulus), log_count, bit_reversed != 0, \
                                      static_cast<hipStream_t>(stream));                          \
        });                                                                                       \
    }                                                                                             \
    int gpuntt_generate_4step_w_##S(T* out, T root, CM modulus, int n_power, int ntt_type,        \
                                    void* stream)                                                 \
    {                                                                                             \
        GPUNTT_NEED(out)                                                                          \
        return guarded([&] {                                                                      \
            GPU_Generate4StepW<T>(out, root, to_mod<T>(modulus), n_power,                         \
                                  static_cast<type>(ntt_type), static_cast<hipStream_t>(stream)); \
        });                                                                                       \
    }                                                                                             \
    int gpuntt_operator_gpu_##S(int op, const T* a, const T* b, T* out, CM modulus,               \
                                uint64_t count, void* stream)                                     \
    {                                                                                             \
        GPUNTT_NEED(a, out)                                                                       \
        return operator_gpu<T>(op, a, b, out, modulus, count, stream);                            \
    }                                                                                             \
    int gpuntt_debug_recip_norm_##S(const T* q, T* out, uint64_t count, void* stream)              \
    {                                                                                             \
        GPUNTT_NEED(q, out)                                                                       \
        return guarded([&] {                                                                      \
            gpuntt::host::debug_recip_norm<T>(q, out, count, static_cast<hipStream_t>(stream));   \
        });                                                                                       \
    }                                                                                             \
    int gpuntt_butterfly_unit_##S(int gentleman_sande, T* u, T* v, const T* roots, CM modulus,    \
                                  uint64_t count, void* stream)                                   \
    {                                                                                             \
        GPUNTT_NEED(u, v)                                                                         \
        GPUNTT_NEED(roots, roots)                                                                 \
        return butterfly_unit<T>(gentleman_sande, u, v, roots, modulus, count, stream);           \
    }

    GPUNTT_C_API(u32, uint32_t, gpuntt_modulus32)
    GPUNTT_C_API(u64, uint64_t, gpuntt_modulus64)
#undef GPUNTT_C_API
}
