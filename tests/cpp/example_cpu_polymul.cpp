// example_cpu_polymul.cpp -- the reference's host-only examples (cpu_merge_ntt_examples,
// cpu_4step_ntt_examples; example/ntt_merge/test_cpu_merge_ntt.cu:28-101,
// example/ntt_4step/test_cpu_4step_ntt.cu:40-79): INTT(NTT(a) .* NTT(b)) == schoolbook(a*b).
// Needs no GPU.
#include <cstdlib>
#include <iostream>
#include <random>
#include <vector>

#include "gpuntt/ntt_4step/ntt_4step_cpu.cuh"
#include "gpuntt/ntt_merge/ntt_cpu.cuh"

using namespace std;
using namespace gpuntt;
typedef Data64 TestDataType;

int main(int argc, char* argv[])
{
    const int LOGN = (argc >= 2) ? atoi(argv[1]) : 12;
    bool ok = true;
    for (ReductionPolynomial rp : {ReductionPolynomial::X_N_minus, ReductionPolynomial::X_N_plus})
    {
        NTTFactors<TestDataType> factor(Modulus<TestDataType>(576460752303415297ULL),
                                        288482366111684746ULL, 238394956950829ULL);
        NTTParameters<TestDataType> parameters(LOGN <= 12 ? LOGN : 12, factor, rp);
        NTTCPU<TestDataType> generator(parameters);
        std::mt19937 gen(7);
        std::uniform_int_distribution<TestDataType> dis(0, parameters.modulus.value - 1);
        vector<TestDataType> a, b;
        for (int i = 0; i < static_cast<int>(parameters.n); i++)
        {
            a.push_back(dis(gen));
            b.push_back(dis(gen));
        }
        vector<TestDataType> na = generator.ntt(a), nb = generator.ntt(b);
        vector<TestDataType> prod = generator.mult(na, nb);
        vector<TestDataType> got = generator.intt(prod);
        vector<TestDataType> want = schoolbook_poly_multiplication<TestDataType>(a, b, parameters.modulus, rp);
        ok = ok && check_result(got.data(), want.data(), static_cast<int>(parameters.n));
    }
    {
        NTTParameters4Step<TestDataType> parameters(12, ReductionPolynomial::X_N_minus);
        NTT_4STEP_CPU<TestDataType> generator(parameters);
        std::mt19937 gen(8);
        std::uniform_int_distribution<TestDataType> dis(0, parameters.modulus.value - 1);
        vector<TestDataType> a, b;
        for (int i = 0; i < static_cast<int>(parameters.n); i++)
        {
            a.push_back(dis(gen));
            b.push_back(dis(gen));
        }
        vector<TestDataType> na = generator.ntt(a), nb = generator.ntt(b);
        vector<TestDataType> prod = generator.mult(na, nb);
        vector<TestDataType> got = generator.intt(prod);
        vector<TestDataType> want = schoolbook_poly_multiplication<TestDataType>(
            a, b, parameters.modulus, ReductionPolynomial::X_N_minus);
        ok = ok && check_result(got.data(), want.data(), static_cast<int>(parameters.n));
    }
    if (ok)
        cout << "All Correct." << endl;
    return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
