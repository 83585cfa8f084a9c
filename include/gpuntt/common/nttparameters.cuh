// gpuntt/common/nttparameters.cuh -- include path kept for drop-in callers
// (reference src/include/gpuntt/common/nttparameters.cuh); the declarations live in
// descriptors.hpp (enums, NTTFactors, bitreverse) and parameter_sets.hpp (NTTParameters,
// NTTParameters4Step).
#pragma once

#include "gpuntt/common/descriptors.hpp"
#include "gpuntt/common/parameter_sets.hpp"
