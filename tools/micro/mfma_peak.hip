// Developer microbenchmark: fp32 MFMA issue rate and s_memtime tick rate on this GPU.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/micro/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k(int iters, float *out, unsigned long long *ticks) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][5];
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
void run(int wgs_per_cu, int iters) {
  const int grid = 256 * wgs_per_cu;
  float *out; unsigned long long *ticks;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&ticks, grid * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS><<<grid, 256>>>(10, out, ticks);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS><<<grid, 256>>>(iters, out, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, ticks, 64, hipMemcpyDeviceToHost);
  const double mfma = double(grid) * 4 * iters * 8 * CHAINS;
  const double flops = mfma * 2 * 32 * 32 * 2;
  printf("chains %d wgs/cu %d: %.3f ms  %.1f TFLOP/s  | ticks/kernel %llu -> tick rate %.3f GHz | MFMA-cycles/SIMD %.0f -> implied clock %.3f GHz at 64 cyc/MFMA\n",
         CHAINS, wgs_per_cu, ms, flops / ms / 1e9, h[0], h[0] / (ms * 1e6),
         double(wgs_per_cu) * iters * 8 * CHAINS * 64, double(wgs_per_cu) * iters * 8 * CHAINS * 64 / (ms * 1e6));
  hipFree(out); hipFree(ticks);
}

int main() {
  run<1>(1, 20000); run<2>(1, 10000); run<1>(4, 5000); run<2>(4, 2500); run<1>(5, 4000);
  return 0;
}
