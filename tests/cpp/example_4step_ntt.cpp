// example_4step_ntt.cpp -- reference flow of gpu_4step_ntt_examples / gpu_4step_intt_examples
// (example/ntt_4step/test_4step_ntt.cu:147-178, test_4step_intt.cu:81-179) against this library:
//   forward : GPU_Transpose -> GPU_4STEP_NTT(FORWARD) -> GPU_Transpose == NTT_4STEP_CPU::ntt
//   inverse : intt_first_transpose (host) -> GPU_4STEP_NTT(INVERSE) -> GPU_Transpose == input
//   ./example_4step_ntt <LOGN> <BATCH>
#include <cstdlib>
#include <iostream>
#include <random>
#include <vector>

#include "gpuntt/ntt_4step/ntt_4step.cuh"

using namespace std;
using namespace gpuntt;
typedef Data64 TestDataType;

template <typename T> T* to_device(const vector<T>& v)
{
    T* d;
    GPUNTT_CUDA_CHECK(hipMalloc(&d, v.size() * sizeof(T)));
    GPUNTT_CUDA_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char* argv[])
{
    CudaDevice();
    const int LOGN = (argc >= 3) ? atoi(argv[1]) : 12;
    const int BATCH = (argc >= 3) ? atoi(argv[2]) : 1;

    NTTParameters4Step<TestDataType> parameters(LOGN, ReductionPolynomial::X_N_minus);
    NTT_4STEP_CPU<TestDataType> generator(parameters);

    std::mt19937 gen(1);
    std::uniform_int_distribution<std::uint64_t> dis(0, parameters.modulus.value - 1);
    vector<TestDataType> flat_in(size_t(BATCH) * parameters.n), flat_expected;
    for (auto& x : flat_in)
        x = dis(gen);
    for (int b = 0; b < BATCH; b++)
    {
        vector<TestDataType> one(flat_in.begin() + size_t(b) * parameters.n,
                                 flat_in.begin() + size_t(b + 1) * parameters.n);
        vector<TestDataType> r = generator.ntt(one);
        flat_expected.insert(flat_expected.end(), r.begin(), r.end());
    }

    TestDataType* Input_Datas = to_device(flat_in);
    TestDataType* Output_Datas;
    GPUNTT_CUDA_CHECK(hipMalloc(&Output_Datas, flat_in.size() * sizeof(TestDataType)));

    Root<TestDataType>* t1 = to_device(
        parameters.gpu_root_of_unity_table_generator(parameters.n1_based_root_of_unity_table));
    Root<TestDataType>* t2 = to_device(
        parameters.gpu_root_of_unity_table_generator(parameters.n2_based_root_of_unity_table));
    Root<TestDataType>* W = to_device(parameters.W_root_of_unity_table);
    Root<TestDataType>* it1 = to_device(parameters.gpu_root_of_unity_table_generator(
        parameters.n1_based_inverse_root_of_unity_table));
    Root<TestDataType>* it2 = to_device(parameters.gpu_root_of_unity_table_generator(
        parameters.n2_based_inverse_root_of_unity_table));
    Root<TestDataType>* iW = to_device(parameters.W_inverse_root_of_unity_table);

    Modulus<TestDataType>* test_modulus = to_device(vector<Modulus<TestDataType>>{parameters.modulus});
    Ninverse<TestDataType>* test_ninverse = to_device(vector<Ninverse<TestDataType>>{parameters.n_inv});

    ntt4step_rns_configuration<TestDataType> cfg_ntt = {
        .n_power = LOGN, .ntt_type = FORWARD, .mod_inverse = test_ninverse, .stream = 0};

    GPU_Transpose(Input_Datas, Output_Datas, parameters.n1, parameters.n2, parameters.logn, BATCH);
    GPU_4STEP_NTT(Output_Datas, Input_Datas, t1, t2, W, test_modulus, cfg_ntt, BATCH, 1);
    GPU_Transpose(Input_Datas, Output_Datas, parameters.n1, parameters.n2, parameters.logn, BATCH);

    vector<TestDataType> Output_Host(flat_in.size());
    GPUNTT_CUDA_CHECK(hipMemcpy(Output_Host.data(), Output_Datas,
                                Output_Host.size() * sizeof(TestDataType), hipMemcpyDeviceToHost));
    bool check = check_result(Output_Host.data(), flat_expected.data(), static_cast<int>(Output_Host.size()));
    if (check)
        cout << "All Correct." << endl;

    // inverse of the forward result must give the input back
    vector<TestDataType> intt_in;
    for (int b = 0; b < BATCH; b++)
    {
        vector<TestDataType> one(flat_expected.begin() + size_t(b) * parameters.n,
                                 flat_expected.begin() + size_t(b + 1) * parameters.n);
        vector<TestDataType> tr = generator.intt_first_transpose(one); // INTT TRANSPOSE IN CPU
        intt_in.insert(intt_in.end(), tr.begin(), tr.end());
    }
    GPUNTT_CUDA_CHECK(hipMemcpy(Input_Datas, intt_in.data(), intt_in.size() * sizeof(TestDataType),
                                hipMemcpyHostToDevice));
    ntt4step_configuration<TestDataType> cfg_intt = {
        .n_power = LOGN, .ntt_type = INVERSE, .mod_inverse = parameters.n_inv, .stream = 0};
    GPU_4STEP_NTT(Input_Datas, Output_Datas, it1, it2, iW, parameters.modulus, cfg_intt, BATCH);
    GPU_Transpose(Output_Datas, Input_Datas, parameters.n1, parameters.n2, parameters.logn, BATCH);
    GPUNTT_CUDA_CHECK(hipMemcpy(Output_Host.data(), Input_Datas,
                                Output_Host.size() * sizeof(TestDataType), hipMemcpyDeviceToHost));
    bool check2 = check_result(Output_Host.data(), flat_in.data(), static_cast<int>(Output_Host.size()));
    if (check2)
        cout << "All Correct (inverse)." << endl;

    // extension: the same two pipelines as single calls (GPU_4STEP_NTT_NaturalOrder)
    GPUNTT_CUDA_CHECK(hipMemcpy(Input_Datas, flat_in.data(), flat_in.size() * sizeof(TestDataType),
                                hipMemcpyHostToDevice));
    ntt4step_configuration<TestDataType> cfg_nat = {
        .n_power = LOGN, .ntt_type = FORWARD, .mod_inverse = 0, .stream = 0};
    GPU_4STEP_NTT_NaturalOrder(Input_Datas, Output_Datas, t1, t2, W, parameters.modulus, cfg_nat, BATCH);
    GPUNTT_CUDA_CHECK(hipMemcpy(Output_Host.data(), Output_Datas,
                                Output_Host.size() * sizeof(TestDataType), hipMemcpyDeviceToHost));
    bool check3 = check_result(Output_Host.data(), flat_expected.data(), static_cast<int>(Output_Host.size()));
    GPU_4STEP_NTT_NaturalOrder(Output_Datas, Input_Datas, it1, it2, iW, parameters.modulus, cfg_intt, BATCH);
    GPUNTT_CUDA_CHECK(hipMemcpy(Output_Host.data(), Input_Datas,
                                Output_Host.size() * sizeof(TestDataType), hipMemcpyDeviceToHost));
    bool check4 = check_result(Output_Host.data(), flat_in.data(), static_cast<int>(Output_Host.size()));
    if (check3 && check4)
        cout << "All Correct (natural order, one call)." << endl;
    return (check && check2 && check3 && check4) ? EXIT_SUCCESS : EXIT_FAILURE;
}
