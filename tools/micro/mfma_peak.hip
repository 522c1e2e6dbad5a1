// Developer microbenchmark: what v_mfma_f32_32x32x2_f32 sustains on this chip (no memory traffic):
// waves per SIMD x independent accumulator chains per wave.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) mfma_kernel(float *out, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int CHAINS>
static void run(int wgs_per_cu, int iters) {
  float *out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  mfma_kernel<CHAINS><<<grid, 256>>>(out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_kernel<CHAINS><<<grid, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = double(grid) * 4 * iters * 8 * CHAINS * 4096.0;
  printf("chains %d  waves/SIMD %d  iters %d: %.3f ms  %.1f TFLOP/s\n", CHAINS, wgs_per_cu, iters, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int it : {200, 2000}) {
    run<1>(1, it);
    run<1>(2, it);
    run<1>(4, it);
    run<2>(1, it);
    run<2>(2, it);
    run<4>(1, it);
    run<4>(2, it);
  }
  return 0;
}
