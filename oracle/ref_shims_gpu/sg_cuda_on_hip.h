// TEST-ONLY shim (oracle/_ref GPU build): lets the reference's own CUDA kernels
// (/root/reference/softgroup/ops/src/cuda.cu and the six *.cu files it includes) compile
// unmodified with hipcc for gfx950.  Only the handful of runtime names those files use are
// mapped; nothing here is part of the product (softgroup_amd/ never includes or loads it).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

typedef hipStream_t cudaStream_t;
typedef hipError_t cudaError_t;
#define cudaSuccess hipSuccess
#define cudaMalloc hipMalloc
#define cudaMemcpy hipMemcpy
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString
#define cudaDeviceSynchronize hipDeviceSynchronize
#define AT_CUDA_CHECK(expr)                                                               \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "oracle/_ref gpu: %s\n", hipGetErrorString(e_));                    \
      abort();                                                                            \
    }                                                                                     \
  } while (0)

// at::Tensor only appears in declarations of the .cpp launchers (never defined or called in
// this build): an incomplete type is enough.
namespace at {
class Tensor;
namespace cuda {
// the reference asks torch for "the current stream"; the test harness sets it explicitly
inline hipStream_t &sg_ref_stream() {
  static hipStream_t s = nullptr;
  return s;
}
inline hipStream_t getCurrentCUDAStream() { return sg_ref_stream(); }
}  // namespace cuda
}  // namespace at
