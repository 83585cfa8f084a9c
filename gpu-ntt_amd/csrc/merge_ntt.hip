// merge_ntt.hip -- host entry points of the Merge NTT (gpuntt::GPU_NTT / GPU_INTT /
// *_Inplace, single-modulus and RNS overloads) for MI355X.
//
// Replaces reference src/lib/ntt_merge/ntt.cu:2076-3097 (hosts) and :4948-5082 (explicit
// instantiations = the exported symbol set).  Behaviour kept: argument checks and exception
// types, asynchronous launches on cfg.stream with a hipGetLastError() check after each,
// in == out allowed, first pass reads `device_in` and later passes run in place on
// `device_out`, cfg.ntt_type / cfg.zero_padding ignored.
#include "gpuntt/ntt_merge/ntt.cuh"
#include "launch.hpp"

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
        kern::PassArgs<TU> a = base_args<TU>(device_in, device_out, root_of_unity_table, cfg.n_power,
                                             cfg.reduction_poly, batch_size);
        a.mod = modulus;
        const unsigned in_flags = std::is_signed<T>::value ? kern::F_SIGNED_IN : 0u;
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
        kern::PassArgs<TU> a =
            base_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table,
                          cfg.n_power, cfg.reduction_poly, batch_size);
        a.mod = modulus;
        a.ninv = cfg.mod_inverse;
        const unsigned out_flags =
            kern::F_SCALE | (std::is_signed<T>::value ? kern::F_CENTERED : 0u);
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
        kern::PassArgs<TU> a = base_args<TU>(device_in, device_out, root_of_unity_table, cfg.n_power,
                                             cfg.reduction_poly, batch_size);
        a.mods = modulus;
        a.mod_count = mod_count;
        set_multi(a);
        const unsigned in_flags = std::is_signed<T>::value ? kern::F_SIGNED_IN : 0u;
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
        kern::PassArgs<TU> a =
            base_args<TU>(device_in, reinterpret_cast<TU*>(device_out), root_of_unity_table,
                          cfg.n_power, cfg.reduction_poly, batch_size);
        a.mods = modulus;
        a.mod_count = mod_count;
        a.ninv_arr = cfg.mod_inverse;
        set_multi(a);
        const unsigned out_flags =
            kern::F_SCALE | (std::is_signed<T>::value ? kern::F_CENTERED : 0u);
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
