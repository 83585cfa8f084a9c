// gpuntt/ntt_merge/ntt_cpu.cuh -- include path kept for drop-in callers
// (reference src/include/gpuntt/ntt_merge/ntt_cpu.cuh): NTTCPU<T> and
// schoolbook_poly_multiplication<T> are declared in gpuntt/common/cpu_transforms.hpp.
#pragma once

#include "gpuntt/common/cpu_transforms.hpp"
