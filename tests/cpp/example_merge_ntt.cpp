// example_merge_ntt.cpp -- a C++ caller of the drop-in API (gpuntt/ntt_merge/ntt.cuh), checking the GPU against
// the library's host transform NTTCPU<T>, exact equality:
//   1. forward, in place, through the RNS overload with one device-side modulus (the call shape of the reference's
//      gpu_merge_ntt_examples, example/ntt_merge/test_merge_ntt.cu:138-144)
//   2. inverse, out of place, through the single-modulus overload, back to the input
//   3. GPU_PolyMul (extension) against NTTCPU ntt / mult / intt
//   4. the argument check keeps the reference's exception type and text
//
//   ./example_merge_ntt <LOGN> <BATCH> [u32]
#include <cstdlib>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "gpuntt/ntt_merge/ntt.cuh"

namespace
{
    // device array that frees itself
    template <typename E> class DeviceArray
    {
      public:
        explicit DeviceArray(size_t count) : count_(count)
        {
            GPUNTT_CUDA_CHECK(hipMalloc(reinterpret_cast<void**>(&ptr_), count * sizeof(E)));
        }
        explicit DeviceArray(const std::vector<E>& host) : DeviceArray(host.size()) { upload(host); }
        DeviceArray(const DeviceArray&) = delete;
        DeviceArray& operator=(const DeviceArray&) = delete;
        ~DeviceArray() { (void) hipFree(ptr_); }
        E* get() const { return ptr_; }
        void upload(const std::vector<E>& host, size_t count = 0)
        {
            GPUNTT_CUDA_CHECK(hipMemcpy(ptr_, host.data(), (count ? count : host.size()) * sizeof(E),
                                        hipMemcpyHostToDevice));
        }
        std::vector<E> download(size_t count = 0) const
        {
            std::vector<E> host(count ? count : count_);
            GPUNTT_CUDA_CHECK(hipMemcpy(host.data(), ptr_, host.size() * sizeof(E), hipMemcpyDeviceToHost));
            return host;
        }

      private:
        E* ptr_ = nullptr;
        size_t count_;
    };

    template <typename T> bool same(const std::vector<T>& got, const std::vector<T>& want, size_t offset, size_t len)
    {
        return gpuntt::check_result(const_cast<T*>(got.data()) + offset, const_cast<T*>(want.data()) + offset,
                                    static_cast<int>(len));
    }
} // namespace

template <typename T> int run(int logn, int batch)
{
    using namespace gpuntt;
    NTTParameters<T> prm(logn, ReductionPolynomial::X_N_minus);
    NTTCPU<T> cpu(prm);
    const size_t n = prm.n;

    // batch polynomials back to back; the CPU transform of each is the expected GPU output
    std::mt19937 rng(0);
    std::uniform_int_distribution<std::uint64_t> below_q(0, prm.modulus.value - 1);
    std::vector<T> coeffs(batch * n), expected(batch * n);
    for (T& c : coeffs)
        c = static_cast<T>(below_q(rng));
    for (int p = 0; p < batch; p++)
    {
        std::vector<T> one(coeffs.begin() + p * n, coeffs.begin() + (p + 1) * n);
        const std::vector<T> f = cpu.ntt(one);
        std::copy(f.begin(), f.end(), expected.begin() + p * n);
    }

    DeviceArray<T> data(coeffs), other(coeffs.size());
    DeviceArray<Root<T>> fwd_table(prm.gpu_root_of_unity_table_generator(prm.forward_root_of_unity_table));
    DeviceArray<Root<T>> inv_table(prm.gpu_root_of_unity_table_generator(prm.inverse_root_of_unity_table));
    DeviceArray<Modulus<T>> modulus_on_device(std::vector<Modulus<T>>{prm.modulus});

    // 1. forward, in place, RNS overload with mod_count = 1
    ntt_rns_configuration<T> fwd = {.n_power = logn,
                                    .ntt_type = FORWARD,
                                    .ntt_layout = PerPolynomial,
                                    .reduction_poly = ReductionPolynomial::X_N_minus,
                                    .zero_padding = false,
                                    .stream = 0};
    GPU_NTT_Inplace(data.get(), fwd_table.get(), modulus_on_device.get(), fwd, batch, 1);
    bool forward_ok = true;
    {
        const std::vector<T> got = data.download();
        for (int p = 0; p < batch && forward_ok; p++)
            forward_ok = same(got, expected, p * n, n);
    }
    if (forward_ok)
        std::cout << "All Correct for PerPolynomial NTT." << std::endl;

    // 2. inverse, out of place, single-modulus overload
    ntt_configuration<T> inv = {.n_power = logn,
                                .ntt_type = INVERSE,
                                .ntt_layout = PerPolynomial,
                                .reduction_poly = ReductionPolynomial::X_N_minus,
                                .zero_padding = false,
                                .mod_inverse = prm.n_inv,
                                .stream = 0};
    GPU_INTT(data.get(), other.get(), inv_table.get(), prm.modulus, inv, batch);
    bool inverse_ok = true;
    {
        const std::vector<T> got = other.download();
        for (int p = 0; p < batch && inverse_ok; p++)
            inverse_ok = same(got, coeffs, p * n, n);
    }
    if (inverse_ok)
        std::cout << "All Correct for PerPolynomial INTT." << std::endl;

    // 3. ring product of polynomial 0 with a fresh operand: INTT(NTT(a) . NTT(b))
    std::vector<T> a(coeffs.begin(), coeffs.begin() + n), b(n);
    for (T& c : b)
        c = static_cast<T>(below_q(rng));
    std::vector<T> fa = cpu.ntt(a), fb = cpu.ntt(b);
    std::vector<T> fc = cpu.mult(fa, fb);
    const std::vector<T> product = cpu.intt(fc);
    data.upload(a, n);
    other.upload(b, n);
    GPU_PolyMul(data.get(), other.get(), other.get(), fwd_table.get(), inv_table.get(), prm.modulus, inv, 1);
    const bool product_ok = same(other.download(n), product, 0, n);
    if (product_ok)
        std::cout << "All Correct for GPU_PolyMul." << std::endl;

    // 4. a ring size outside the supported range is refused the way the reference refuses it
    bool refused = false;
    try
    {
        fwd.n_power = 29;
        GPU_NTT_Inplace(data.get(), fwd_table.get(), modulus_on_device.get(), fwd, batch, 1);
    }
    catch (const std::invalid_argument& e)
    {
        refused = (std::string(e.what()) == "Invalid n_power range!");
    }
    return (forward_ok && inverse_ok && product_ok && refused) ? EXIT_SUCCESS : EXIT_FAILURE;
}

int main(int argc, char* argv[])
{
    gpuntt::CudaDevice();
    const int logn = (argc >= 3) ? std::atoi(argv[1]) : 12;
    const int batch = (argc >= 3) ? std::atoi(argv[2]) : 1;
    const bool u32 = (argc >= 4) && std::string(argv[3]) == "u32";
    return u32 ? run<Data32>(logn, batch) : run<Data64>(logn, batch);
}
