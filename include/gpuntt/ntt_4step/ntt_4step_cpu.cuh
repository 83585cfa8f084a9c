// gpuntt/ntt_4step/ntt_4step_cpu.cuh -- include path kept for drop-in callers
// (reference src/include/gpuntt/ntt_4step/ntt_4step_cpu.cuh): NTT_4STEP_CPU<T> is declared in
// gpuntt/common/cpu_transforms.hpp.
#pragma once

#include "gpuntt/common/cpu_transforms.hpp"
