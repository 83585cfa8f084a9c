// gpuntt/common/descriptors.hpp -- the small vocabulary every entry point shares: transform
// direction, data layout, reduction polynomial, the (q, omega, psi) triple of a caller-chosen
// prime, and the bit-reversal helper.  Names and enumerator order follow the reference
// (src/include/gpuntt/common/nttparameters.cuh:17-54) because caller code spells them out.
#pragma once

#include <vector>

#include "gpuntt/common/common.cuh"
#include "gpuntt/common/modular_arith.cuh"

namespace gpuntt
{
    int bitreverse(int index, int n_power);

    enum type { FORWARD, INVERSE };

    enum NTTLayout
    {
        PerPolynomial, // one transform per row of the (batch x N) matrix
        PerCoefficient // one transform per column
    };

    enum ReductionPolynomial
    {
        X_N_plus, // negacyclic, Z_q[X]/(X^N + 1): tables hold powers of psi (N entries)
        X_N_minus // cyclic,     Z_q[X]/(X^N - 1): tables hold powers of omega (N/2 entries)
    };

    template <typename T> struct NTTFactors
    {
        Modulus<T> modulus;
        T omega;
        T psi;
        __host__ NTTFactors(Modulus<T> q_, T omega_, T psi_) : modulus(q_), omega(omega_), psi(psi_) {}
        __host__ NTTFactors() : modulus(), omega(0), psi(0) {}
    };

} // namespace gpuntt
