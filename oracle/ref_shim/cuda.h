// oracle/ref_shim/cuda.h -- TEST INFRASTRUCTURE ONLY: the reference TU includes <cuda.h>.
#pragma once
#include <hip/hip_runtime.h>
