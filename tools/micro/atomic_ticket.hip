// Micro-benchmark: cost of drawing unit tickets from a few shared counters, the way the
// persistent conv kernel's hand-out does.  G workgroups, each draws `draws` tickets from counter
// [blockIdx % n_counters], with `work` ticks of busy time between draws (a unit of work).
// Reports the round-trip latency of a draw as seen by the drawing lane.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/atomic_ticket.hip -o tools/micro/atomic_ticket
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

template <int SCOPE>
__global__ void draw_kernel(unsigned *counters, int n_counters, int stride, int draws, int work,
                            unsigned long long *lat) {
  unsigned *c = counters + (blockIdx.x % n_counters) * stride;
  unsigned long long sum = 0, mx = 0;
  for (int i = 0; i < draws; ++i) {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = __builtin_readcyclecounter();
      unsigned v;
      if (SCOPE == 0) v = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" : "+v"(v));
      const unsigned long long t1 = __builtin_readcyclecounter();
      sum += t1 - t0;
      mx = t1 - t0 > mx ? t1 - t0 : mx;
    }
    const unsigned long long w0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - w0 < static_cast<unsigned long long>(work)) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    lat[2 * blockIdx.x] = sum / draws;
    lat[2 * blockIdx.x + 1] = mx;
  }
}

int main() {
  const int G = 1536;
  unsigned *counters;
  unsigned long long *lat;
  hipMalloc(&counters, 64 * 128);
  hipMalloc(&lat, G * 16);
  std::vector<unsigned long long> h(2 * G);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int scope = 0; scope < 2; ++scope)
    for (int nc : {1, 8, 64})
      for (int stride : {1, 32})
        for (int work : {2000, 10000, 30000}) {
          hipMemset(counters, 0, 64 * 128);
          hipEventRecord(e0);
          if (scope == 0) draw_kernel<0><<<G, 256>>>(counters, nc, stride, 4, work, lat);
          else draw_kernel<1><<<G, 256>>>(counters, nc, stride, 4, work, lat);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          hipMemcpy(h.data(), lat, G * 16, hipMemcpyDeviceToHost);
          std::vector<unsigned long long> mean, mx;
          for (int i = 0; i < G; ++i) { mean.push_back(h[2 * i]); mx.push_back(h[2 * i + 1]); }
          std::sort(mean.begin(), mean.end());
          std::sort(mx.begin(), mx.end());
          printf("scope %s counters %2d stride %3d B work %5d ticks: draw latency mean p50 %6llu p90 %6llu  max p50 %6llu p99 %6llu ticks; kernel %.1f us\n",
                 scope ? "workgroup" : "agent    ", nc, stride * 4, work, mean[G / 2], mean[G * 9 / 10], mx[G / 2],
                 mx[G * 99 / 100], ms * 1e3);
        }
  return 0;
}
