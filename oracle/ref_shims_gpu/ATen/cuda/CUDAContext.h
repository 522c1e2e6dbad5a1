// TEST-ONLY shim (oracle/_ref GPU build): stands in for <ATen/cuda/CUDAContext.h> so that the reference's
// CUDA sources compile with hipcc without libtorch; see sg_cuda_on_hip.h.
#pragma once
#include "sg_cuda_on_hip.h"
