// Developer microbenchmark: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the two
// access patterns that matter here, on known byte counts (working set 1 GiB >> the 256 MiB
// Infinity Cache, every byte touched exactly once):
//   stream_kernel   coalesced 16 B per lane (the guide's calibrated case: FETCH_SIZE reads 1/2)
//   gather_kernel   the sparse-conv operand pattern: a wave owns 32 random rows of 256 B
//                   (64 fp32 channels); lane (h, i) reads row i in 16-B pieces at
//                   s*64 + h*32 (+16) for s = 0..3 -- gather_conv_persistent_kernel's A loads
//   write_kernel    coalesced 16 B per lane stores (WRITE_SIZE)
// Build / run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/micro/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal -- /tmp/fetch_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cal -- /tmp/fetch_calib
// then: factor = known bytes (printed below) / (counter value * 1024).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256) stream_kernel(const float4 *__restrict__ in, long long n4,
                                                    float *out) {
  float s = 0.f;
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < n4; t += gridDim.x * 256LL) {
    const float4 v = in[t];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}

__global__ void __launch_bounds__(256) gather_kernel(const float *__restrict__ in,
                                                    const int *__restrict__ row_of, long long n_rows,
                                                    float *out) {
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const long long wave = (blockIdx.x * 256LL + threadIdx.x) >> 6, n_waves = (gridDim.x * 256LL) >> 6;
  float s = 0.f;
  for (long long g = wave; g * 32 < n_rows; g += n_waves) {
    const long long r = g * 32 + i;
    if (r >= n_rows) continue;
    const float *row = in + static_cast<long long>(row_of[r]) * 64;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const float4 a = *reinterpret_cast<const float4 *>(row + sl * 16 + h * 8);
      const float4 b = *reinterpret_cast<const float4 *>(row + sl * 16 + h * 8 + 4);
      s += a.x + a.w + b.y + b.z;
    }
  }
  if (s == 123.456f) out[0] = s;
}

__global__ void __launch_bounds__(256) write_kernel(float4 *out, long long n4) {
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < n4; t += gridDim.x * 256LL)
    out[t] = make_float4(1.f, 2.f, 3.f, static_cast<float>(t & 7));
}

int main() {
  const long long bytes = 1LL << 30, n_rows = bytes / 256;
  float *buf, *out;
  int *row_of;
  hipMalloc(&buf, bytes);
  hipMalloc(&out, 256);
  hipMalloc(&row_of, n_rows * 4);
  hipMemset(buf, 0, bytes);
  std::vector<int> perm(n_rows);
  for (long long r = 0; r < n_rows; ++r) perm[r] = static_cast<int>(r);
  unsigned long long x = 88172645463325252ULL;
  for (long long r = n_rows - 1; r > 0; --r) {       // Fisher-Yates, xorshift
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const long long j = static_cast<long long>(x % static_cast<unsigned long long>(r + 1));
    const int t = perm[r]; perm[r] = perm[j]; perm[j] = t;
  }
  hipMemcpy(row_of, perm.data(), n_rows * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    stream_kernel<<<4096, 256>>>(reinterpret_cast<const float4 *>(buf), bytes / 16, out);
    gather_kernel<<<4096, 256>>>(buf, row_of, n_rows, out);
    write_kernel<<<4096, 256>>>(reinterpret_cast<float4 *>(buf), bytes / 16);
    hipDeviceSynchronize();
  }
  printf("known bytes per launch: stream_kernel read %lld, gather_kernel read %lld (+ %lld of row ids), "
         "write_kernel written %lld\n", bytes, bytes, n_rows * 4, bytes);
  return 0;
}
