// oracle/ref_shim/cuda_runtime.h -- TEST INFRASTRUCTURE ONLY: the reference TU includes <cuda_runtime.h>.
#pragma once
#include <hip/hip_runtime.h>
