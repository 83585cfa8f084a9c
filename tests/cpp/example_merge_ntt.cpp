// example_merge_ntt.cpp -- what a user of the reference's C++ API writes, compiled unchanged in
// shape against this library: GPU result vs the library's host NTTCPU, exact equality.
// Same flow as the reference's gpu_merge_ntt_examples / gpu_merge_intt_examples
// (example/ntt_merge/test_merge_ntt.cu:46-182, test_merge_intt.cu:46-203) with hip* memory calls.
//
//   ./example_merge_ntt <LOGN> <BATCH> [u32]
#include <cstdlib>
#include <iostream>
#include <random>
#include <vector>

#include "gpuntt/ntt_merge/ntt.cuh"

using namespace std;
using namespace gpuntt;

template <typename TestDataType> int run(int LOGN, int BATCH)
{
    NTTParameters<TestDataType> parameters(LOGN, ReductionPolynomial::X_N_minus);
    NTTCPU<TestDataType> generator(parameters);

    std::mt19937 gen(0);
    std::uniform_int_distribution<std::uint64_t> dis(0, parameters.modulus.value - 1);
    vector<vector<TestDataType>> input1(BATCH);
    for (int j = 0; j < BATCH; j++)
        for (int i = 0; i < static_cast<int>(parameters.n); i++)
            input1[j].push_back(static_cast<TestDataType>(dis(gen)));

    vector<vector<TestDataType>> ntt_result(BATCH);
    for (int i = 0; i < BATCH; i++)
        ntt_result[i] = generator.ntt(input1[i]);

    TestDataType* InOut_Datas;
    GPUNTT_CUDA_CHECK(hipMalloc(&InOut_Datas, BATCH * parameters.n * sizeof(TestDataType)));
    for (int j = 0; j < BATCH; j++)
        GPUNTT_CUDA_CHECK(hipMemcpy(InOut_Datas + (parameters.n * j), input1[j].data(),
                                    parameters.n * sizeof(TestDataType), hipMemcpyHostToDevice));

    Root<TestDataType>* Forward_Omega_Table_Device;
    GPUNTT_CUDA_CHECK(hipMalloc(&Forward_Omega_Table_Device,
                                parameters.root_of_unity_size * sizeof(Root<TestDataType>)));
    vector<Root<TestDataType>> forward_omega_table =
        parameters.gpu_root_of_unity_table_generator(parameters.forward_root_of_unity_table);
    GPUNTT_CUDA_CHECK(hipMemcpy(Forward_Omega_Table_Device, forward_omega_table.data(),
                                parameters.root_of_unity_size * sizeof(Root<TestDataType>),
                                hipMemcpyHostToDevice));

    Root<TestDataType>* Inverse_Omega_Table_Device;
    GPUNTT_CUDA_CHECK(hipMalloc(&Inverse_Omega_Table_Device,
                                parameters.root_of_unity_size * sizeof(Root<TestDataType>)));
    vector<Root<TestDataType>> inverse_omega_table =
        parameters.gpu_root_of_unity_table_generator(parameters.inverse_root_of_unity_table);
    GPUNTT_CUDA_CHECK(hipMemcpy(Inverse_Omega_Table_Device, inverse_omega_table.data(),
                                parameters.root_of_unity_size * sizeof(Root<TestDataType>),
                                hipMemcpyHostToDevice));

    // RNS overload with mod_count = 1, exactly like the reference example
    Modulus<TestDataType>* test_modulus;
    GPUNTT_CUDA_CHECK(hipMalloc(&test_modulus, sizeof(Modulus<TestDataType>)));
    Modulus<TestDataType> test_modulus_[1] = {parameters.modulus};
    GPUNTT_CUDA_CHECK(hipMemcpy(test_modulus, test_modulus_, sizeof(Modulus<TestDataType>),
                                hipMemcpyHostToDevice));

    ntt_rns_configuration<TestDataType> cfg_ntt = {.n_power = LOGN,
                                                   .ntt_type = FORWARD,
                                                   .ntt_layout = PerPolynomial,
                                                   .reduction_poly = ReductionPolynomial::X_N_minus,
                                                   .zero_padding = false,
                                                   .stream = 0};
    GPU_NTT_Inplace(InOut_Datas, Forward_Omega_Table_Device, test_modulus, cfg_ntt, BATCH, 1);

    vector<TestDataType> Output_Host(BATCH * parameters.n);
    GPUNTT_CUDA_CHECK(hipMemcpy(Output_Host.data(), InOut_Datas,
                                BATCH * parameters.n * sizeof(TestDataType), hipMemcpyDeviceToHost));
    bool check = true;
    for (int i = 0; i < BATCH && check; i++)
        check = check_result(Output_Host.data() + (i * parameters.n), ntt_result[i].data(),
                             static_cast<int>(parameters.n));
    if (check)
        cout << "All Correct for PerPolynomial NTT." << endl;

    // single-modulus overload, out of place, inverse back to the input
    TestDataType* Out_Datas;
    GPUNTT_CUDA_CHECK(hipMalloc(&Out_Datas, BATCH * parameters.n * sizeof(TestDataType)));
    ntt_configuration<TestDataType> cfg_intt = {.n_power = LOGN,
                                                .ntt_type = INVERSE,
                                                .ntt_layout = PerPolynomial,
                                                .reduction_poly = ReductionPolynomial::X_N_minus,
                                                .zero_padding = false,
                                                .mod_inverse = parameters.n_inv,
                                                .stream = 0};
    GPU_INTT(InOut_Datas, Out_Datas, Inverse_Omega_Table_Device, parameters.modulus, cfg_intt, BATCH);
    GPUNTT_CUDA_CHECK(hipMemcpy(Output_Host.data(), Out_Datas,
                                BATCH * parameters.n * sizeof(TestDataType), hipMemcpyDeviceToHost));
    bool check2 = true;
    for (int i = 0; i < BATCH && check2; i++)
        check2 = check_result(Output_Host.data() + (i * parameters.n), input1[i].data(),
                              static_cast<int>(parameters.n));
    if (check2)
        cout << "All Correct for PerPolynomial INTT." << endl;

    // extension: ring product INTT(NTT(a) . NTT(b)) on the GPU against the CPU composition the
    // reference's example checks (NTTCPU ntt / mult / intt, test_cpu_merge_ntt.cu:69-101)
    bool check3 = true;
    {
        vector<TestDataType> b_host(parameters.n);
        for (auto& x : b_host)
            x = dis(gen);
        vector<TestDataType> fa = generator.ntt(input1[0]), fb = generator.ntt(b_host);
        vector<TestDataType> fc = generator.mult(fa, fb);
        vector<TestDataType> want = generator.intt(fc);
        GPUNTT_CUDA_CHECK(hipMemcpy(InOut_Datas, input1[0].data(), parameters.n * sizeof(TestDataType),
                                    hipMemcpyHostToDevice));
        GPUNTT_CUDA_CHECK(hipMemcpy(Out_Datas, b_host.data(), parameters.n * sizeof(TestDataType),
                                    hipMemcpyHostToDevice));
        GPU_PolyMul(InOut_Datas, Out_Datas, Out_Datas, Forward_Omega_Table_Device, Inverse_Omega_Table_Device,
                    parameters.modulus, cfg_intt, 1);
        GPUNTT_CUDA_CHECK(hipMemcpy(Output_Host.data(), Out_Datas, parameters.n * sizeof(TestDataType),
                                    hipMemcpyDeviceToHost));
        check3 = check_result(Output_Host.data(), want.data(), static_cast<int>(parameters.n));
        if (check3)
            cout << "All Correct for GPU_PolyMul." << endl;
    }

    // argument checking keeps the reference's exception type and text
    bool threw = false;
    try
    {
        cfg_ntt.n_power = 29;
        GPU_NTT_Inplace(InOut_Datas, Forward_Omega_Table_Device, test_modulus, cfg_ntt, BATCH, 1);
    }
    catch (const std::invalid_argument& e)
    {
        threw = (std::string(e.what()) == "Invalid n_power range!");
    }

    GPUNTT_CUDA_CHECK(hipFree(InOut_Datas));
    GPUNTT_CUDA_CHECK(hipFree(Out_Datas));
    GPUNTT_CUDA_CHECK(hipFree(Forward_Omega_Table_Device));
    GPUNTT_CUDA_CHECK(hipFree(Inverse_Omega_Table_Device));
    GPUNTT_CUDA_CHECK(hipFree(test_modulus));
    return (check && check2 && check3 && threw) ? EXIT_SUCCESS : EXIT_FAILURE;
}

int main(int argc, char* argv[])
{
    CudaDevice();
    const int LOGN = (argc >= 3) ? atoi(argv[1]) : 12;
    const int BATCH = (argc >= 3) ? atoi(argv[2]) : 1;
    const bool u32 = (argc >= 4) && std::string(argv[3]) == "u32";
    return u32 ? run<Data32>(LOGN, BATCH) : run<Data64>(LOGN, BATCH);
}
