// merge_ntt.hip -- host entry points of the Merge NTT (gpuntt::GPU_NTT / GPU_INTT /
// *_Inplace, single-modulus and RNS overloads) for MI355X.
//
// Replaces reference src/lib/ntt_merge/ntt.cu:2076-3097 (hosts) and :4948-5082 (explicit
// instantiations = the exported symbol set).  Behaviour kept: argument checks and exception
// types, asynchronous launches on cfg.stream with a hipGetLastError() check after each,
// in == out allowed, first pass reads `device_in` and later passes run in place on
// `device_out`, cfg.ntt_type / cfg.zero_padding ignored.
#include <cstdlib>
#include <cstring>

#include "gpuntt/ntt_merge/ntt.cuh"
#include "launch.hpp"
#include "lazy_launch.hpp"

namespace gpuntt
{
    namespace
    {
        template <typename TU>
        kern::PassArgs<TU> base_args(const void* in, TU* out, const TU* roots, int n_power,
                                     ReductionPolynomial poly, int batch_size)
        {
            kern::PassArgs<TU> a{};
            a.in = in;
            a.out = out;
            a.roots = roots;
            a.mods = nullptr;
            a.ninv_arr = nullptr;
            a.ninv = 0;
            a.w_table = nullptr;
            a.total = static_cast<unsigned long long>(batch_size < 0 ? 0 : batch_size) << n_power;
            a.n = n_power;
            a.poly_shift = n_power;
            a.root_shift = n_power;
            a.mod_count = 1;
            a.p_lo = 0;
            a.n2_log = 0;
            a.flags = (poly == ReductionPolynomial::X_N_plus) ? kern::F_NEGACYCLIC : 0u;
            return a;
        }

        inline void check_layout_and_range(NTTLayout layout, int n_power)
        {
            switch (layout)
            {
                case PerPolynomial:
                    if (n_power <= 0 || n_power >= 29)
                        throw std::invalid_argument("Invalid n_power range!");
                    break;
                case PerCoefficient:
                    if (n_power <= 0 || n_power >= 10)
                        throw std::invalid_argument("Invalid n_power range!");
                    // column-wise layout (reference ntt.cu:1554-2074) is scheduled after the
                    // PerPolynomial path
                    throw std::invalid_argument("PerCoefficient ntt_layout is not implemented yet!");
                default:
                    throw std::invalid_argument("Invalid ntt_layout!");
            }
        }

        // ---- fast 64-bit path (lazy residues + prepared Shoup twiddles) -----------------
        // Used for Data64 calls (single modulus and RNS).  Moduli with >= 4 bits of headroom
        // (bit <= 60) run the lazy butterflies; blocks whose modulus has bit 61/62 switch to
        // exact Barrett butterflies inside the same kernel.  Small jobs, rings above 2^24 and
        // RNS stacks of rings below one tile stay on the generic kernels.
        // GPUNTT_PATH=generic | fast overrides the size heuristic (testing / A-B timing);
        // moduli without the headroom always take the generic kernels.
        inline int forced_path()
        {
            static const int mode = [] {
                const char* e = std::getenv("GPUNTT_PATH");
                if (e == nullptr)
                    return 0;
                if (std::strcmp(e, "generic") == 0)
                    return 1;
                if (std::strcmp(e, "fast") == 0)
                    return 2;
                return 0;
            }();
            return mode;
        }

        inline bool lazy_eligible(int n_power, int batch_size, int mod_count)
        {
            if (n_power > host::LAZY_MAX_N_POWER)
                return false;
            if (mod_count > 1 && n_power < kern::TL)
                return false; // a tile would mix moduli
            if ((static_cast<unsigned long long>(mod_count) << n_power) > (1ull << 26))
                return false; // prepared table would exceed 1 GiB
            if (forced_path() == 1)
                return false;
            if (forced_path() == 2)
                return true;
            // the per-call twiddle preparation touches N entries: not worth it for tiny jobs
            return (static_cast<unsigned long long>(batch_size) << n_power) >= (1ull << 15) &&
                   batch_size >= 2;
        }

        // mods == nullptr: single modulus `m`; else device array of mod_count moduli (+ optional
        // device array of n^-1 values whose Shoup pairs are prepared alongside the twiddles)
        inline kern::LazyArgs lazy_args(const void* in, Data64* out, const Data64* roots,
                                        const Modulus<Data64>& m, const Modulus<Data64>* mods,
                                        int mod_count, const Data64* ninv_dev, int n_power,
                                        ReductionPolynomial poly, int batch_size, hipStream_t stream)
        {
            const bool neg = (poly == ReductionPolynomial::X_N_plus);
            const bool perm = (n_power >= kern::TL);
            const size_t entries = (static_cast<size_t>(mod_count) << n_power) + mod_count;
            auto* ws = static_cast<lazy::Tw64*>(host::lazy_workspace(stream, sizeof(lazy::Tw64) * entries));
            lazy::Tw64* ws_ninv = ws + (static_cast<size_t>(mod_count) << n_power);
            host::launch_prep(roots, ws, mods, m.value, mod_count, n_power, neg, perm, ninv_dev,
                              ninv_dev ? ws_ninv : nullptr, stream);
            kern::LazyArgs a{};
            a.in = in;
            a.out = out;
            a.tw = ws;
            a.mods = mods;
            a.q = m.value;
            a.q_bit = m.bit;
            a.q_mu = m.mu;
            a.ninv_arr = ninv_dev ? ws_ninv : nullptr;
            a.ninv = lazy::Tw64{0, 0};
            a.total = static_cast<unsigned long long>(batch_size) << n_power;
            a.n = n_power;
            a.poly_shift = n_power;
            a.mod_count = mod_count;
            a.p_lo = 0;
            a.flags = perm ? kern::F_PERM_LOW : 0u;
            return a;
        }

        template <typename TU> inline void set_multi(kern::PassArgs<TU>& a)
        {
            if (a.mods != nullptr && a.mod_count > 1 && a.poly_shift < kern::TL)
                a.flags |= kern::F_MULTI;
        }
    } // namespace

    // ---------------------------------------------------------------- single modulus ----
    template <typename T>
    __host__ void GPU_NTT(T* device_in, typename std::make_unsigned<T>::type* device_out,
                          Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                          Modulus<typename std::make_unsigned<T>::type> modulus,
                          ntt_configuration<typename std::make_unsigned<T>::type> cfg,
                          int batch_size)
    {
        using TU = typename std::make_unsigned<T>::type;
        check_layout_and_range(cfg.ntt_layout, cfg.n_power);
        const unsigned in_flags = std::is_signed<T>::value ? kern::F_SIGNED_IN : 0u;
        if constexpr (std::is_same<TU, Data64>::value)
        {
            if (batch_size > 0 && modulus.value >= 3 && lazy_eligible(cfg.n_power, batch_size, 1))
            {
                kern::LazyArgs la =
                    lazy_args(device_in, device_out, root_of_unity_table, modulus, nullptr, 1, nullptr,
                              cfg.n_power, cfg.reduction_poly, batch_size, cfg.stream);
                host::run_transform_lazy<false>(la, in_flags, 0u, cfg.stream);
                return;
            }
        }
        kern::PassArgs<TU> a = base_args<TU>(device_in, device_out, root_of_unity_table, cfg.n_power,
                                             cfg.reduction_poly, batch_size);
        a.mod = modulus;
        host::run_transform<TU, false>(a, in_flags, 0u, cfg.stream);
    }

    template <typename T>
    __host__ void GPU_INTT(typename std::make_unsigned<T>::type* device_in, T* device_out,
                           Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                           Modulus<typename std::make_unsigned<T>::type> modulus,
                           ntt_configuration<typename std::make_unsigned<T>::type> cfg,
                           int batch_size)
    {
        using TU = typename std::make_unsigned<T>::type;
        check_layout_and_range(cfg.ntt_layout, cfg.n_power);
        const unsigned out_flags =
            kern::F_SCALE | (std::is_signed<T>::value ? kern::F_CENTERED : 0u);
        if constexpr (std::is_same<TU, Data64>::value)
        {
            if (batch_size > 0 && modulus.value >= 3 && lazy_eligible(cfg.n_power, batch_size, 1) &&
                cfg.mod_inverse < modulus.value)
            {
                kern::LazyArgs la =
                    lazy_args(device_in, reinterpret_cast<Data64*>(device_out), root_of_unity_table,
                              modulus, nullptr, 1, nullptr, cfg.n_power, cfg.reduction_poly, batch_size,
                              cfg.stream);
                la.ninv = lazy::Tw64{cfg.mod_inverse, host::shoup_host(cfg.mod_inverse, modulus.value)};
                host::run_transform_lazy<true>(la, 0u, out_flags, cfg.stream);
                return;
            }
        }
        kern::PassArgs<TU> a =
            base_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table,
                          cfg.n_power, cfg.reduction_poly, batch_size);
        a.mod = modulus;
        a.ninv = cfg.mod_inverse;
        host::run_transform<TU, true>(a, 0u, out_flags, cfg.stream);
    }

    template <typename T>
    __host__ void GPU_NTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                  Modulus<T> modulus, ntt_configuration<T> cfg, int batch_size)
    {
        GPU_NTT<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size);
    }

    template <typename T>
    __host__ void GPU_INTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                   Modulus<T> modulus, ntt_configuration<T> cfg, int batch_size)
    {
        GPU_INTT<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size);
    }

    // --------------------------------------------------------------------------- RNS ----
    template <typename T>
    __host__ void GPU_NTT(T* device_in, typename std::make_unsigned<T>::type* device_out,
                          Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                          Modulus<typename std::make_unsigned<T>::type>* modulus,
                          ntt_rns_configuration<typename std::make_unsigned<T>::type> cfg,
                          int batch_size, int mod_count)
    {
        using TU = typename std::make_unsigned<T>::type;
        check_layout_and_range(cfg.ntt_layout, cfg.n_power);
        if (mod_count <= 0 || modulus == nullptr)
            throw std::invalid_argument("Invalid mod_count!");
        const unsigned in_flags = std::is_signed<T>::value ? kern::F_SIGNED_IN : 0u;
        if constexpr (std::is_same<TU, Data64>::value)
        {
            if (batch_size > 0 && lazy_eligible(cfg.n_power, batch_size, mod_count))
            {
                kern::LazyArgs la =
                    lazy_args(device_in, device_out, root_of_unity_table, Modulus<Data64>(), modulus,
                              mod_count, nullptr, cfg.n_power, cfg.reduction_poly, batch_size, cfg.stream);
                host::run_transform_lazy<false>(la, in_flags, 0u, cfg.stream);
                return;
            }
        }
        kern::PassArgs<TU> a = base_args<TU>(device_in, device_out, root_of_unity_table, cfg.n_power,
                                             cfg.reduction_poly, batch_size);
        a.mods = modulus;
        a.mod_count = mod_count;
        set_multi(a);
        host::run_transform<TU, false>(a, in_flags, 0u, cfg.stream);
    }

    template <typename T>
    __host__ void GPU_INTT(typename std::make_unsigned<T>::type* device_in, T* device_out,
                           Root<typename std::make_unsigned<T>::type>* root_of_unity_table,
                           Modulus<typename std::make_unsigned<T>::type>* modulus,
                           ntt_rns_configuration<typename std::make_unsigned<T>::type> cfg,
                           int batch_size, int mod_count)
    {
        using TU = typename std::make_unsigned<T>::type;
        check_layout_and_range(cfg.ntt_layout, cfg.n_power);
        if (mod_count <= 0 || modulus == nullptr)
            throw std::invalid_argument("Invalid mod_count!");
        const unsigned out_flags =
            kern::F_SCALE | (std::is_signed<T>::value ? kern::F_CENTERED : 0u);
        if constexpr (std::is_same<TU, Data64>::value)
        {
            if (batch_size > 0 && cfg.mod_inverse != nullptr &&
                lazy_eligible(cfg.n_power, batch_size, mod_count))
            {
                kern::LazyArgs la = lazy_args(device_in, reinterpret_cast<Data64*>(device_out),
                                              root_of_unity_table, Modulus<Data64>(), modulus, mod_count,
                                              cfg.mod_inverse, cfg.n_power, cfg.reduction_poly,
                                              batch_size, cfg.stream);
                host::run_transform_lazy<true>(la, 0u, out_flags, cfg.stream);
                return;
            }
        }
        kern::PassArgs<TU> a =
            base_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table,
                          cfg.n_power, cfg.reduction_poly, batch_size);
        a.mods = modulus;
        a.mod_count = mod_count;
        a.ninv_arr = cfg.mod_inverse;
        set_multi(a);
        host::run_transform<TU, true>(a, 0u, out_flags, cfg.stream);
    }

    template <typename T>
    __host__ void GPU_NTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                  Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                  int batch_size, int mod_count)
    {
        GPU_NTT<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size,
                   mod_count);
    }

    template <typename T>
    __host__ void GPU_INTT_Inplace(T* device_inout, Root<T>* root_of_unity_table,
                                   Modulus<T>* modulus, ntt_rns_configuration<T> cfg,
                                   int batch_size, int mod_count)
    {
        GPU_INTT<T>(device_inout, device_inout, root_of_unity_table, modulus, cfg, batch_size,
                    mod_count);
    }

    // ------------------------------------------------- exported symbol set ----
#define GPUNTT_INSTANTIATE(T, TU)                                                              \
    template __host__ void GPU_NTT<T>(T*, TU*, Root<TU>*, Modulus<TU>, ntt_configuration<TU>,  \
                                      int);                                                    \
    template __host__ void GPU_INTT<T>(TU*, T*, Root<TU>*, Modulus<TU>, ntt_configuration<TU>, \
                                       int);                                                   \
    template __host__ void GPU_NTT<T>(T*, TU*, Root<TU>*, Modulus<TU>*,                        \
                                      ntt_rns_configuration<TU>, int, int);                    \
    template __host__ void GPU_INTT<T>(TU*, T*, Root<TU>*, Modulus<TU>*,                       \
                                       ntt_rns_configuration<TU>, int, int);

    GPUNTT_INSTANTIATE(Data32, Data32)
    GPUNTT_INSTANTIATE(Data64, Data64)
    GPUNTT_INSTANTIATE(Data32s, Data32)
    GPUNTT_INSTANTIATE(Data64s, Data64)
#undef GPUNTT_INSTANTIATE

#define GPUNTT_INSTANTIATE_INPLACE(T)                                                          \
    template __host__ void GPU_NTT_Inplace<T>(T*, Root<T>*, Modulus<T>, ntt_configuration<T>,  \
                                              int);                                            \
    template __host__ void GPU_INTT_Inplace<T>(T*, Root<T>*, Modulus<T>, ntt_configuration<T>, \
                                               int);                                           \
    template __host__ void GPU_NTT_Inplace<T>(T*, Root<T>*, Modulus<T>*,                       \
                                              ntt_rns_configuration<T>, int, int);             \
    template __host__ void GPU_INTT_Inplace<T>(T*, Root<T>*, Modulus<T>*,                      \
                                               ntt_rns_configuration<T>, int, int);

    GPUNTT_INSTANTIATE_INPLACE(Data32)
    GPUNTT_INSTANTIATE_INPLACE(Data64)
#undef GPUNTT_INSTANTIATE_INPLACE

} // namespace gpuntt
