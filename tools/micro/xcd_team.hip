// Developer microbenchmark: a team of workgroups on ONE XCD (formed at run time from the hardware
// XCC ids) meeting at barriers through the XCD's L2, against the same barrier with device-scope (sc1)
// traffic over all launched workgroups.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int l2_read(int *p) {
  int zero = 0;
  asm volatile("" : "+v"(zero));   // fetch_or(p, 0) would be folded into an sc0 atomic LOAD
  return __hip_atomic_fetch_or(p, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define LD_L2(p) l2_read(p)
#define LD_DEV(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// sync: [0] bar [1] fail [2,3] meet [8..] diagnostics
template <int MODE>   // 0: team on one XCD, L2 atomics;  1: all workgroups, device scope
__global__ void __launch_bounds__(512) team_kernel(int *sync, int *data, int rounds, long long *cycles) {
  __shared__ int lds_team, lds_rank, lds_flag;
  int *bar = sync, *fail = sync + 1;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned raw = xcc;
    xcc &= 7u;
    unsigned long long *meet = reinterpret_cast<unsigned long long *>(sync + 2);
    const unsigned long long old = __hip_atomic_fetch_add(meet, 1ull << (8 * xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long cur = 0;
    unsigned spins = 0;
    bool ok = true;
    while (true) {
      cur = __hip_atomic_load(meet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned arrived = 0;
      for (int x = 0; x < 8; ++x) arrived += static_cast<unsigned>(cur >> (8 * x)) & 0xffu;
      if (arrived == gridDim.x) break;
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 20)) { ok = false; break; }
    }
    int best = 0, best_n = -1;
    for (int x = 0; x < 8; ++x) {
      const int c = static_cast<int>(cur >> (8 * x)) & 0xff;
      if (c > best_n) { best_n = c; best = x; }
    }
    if (blockIdx.x == 0) {
      sync[8] = ok; sync[9] = best; sync[10] = best_n; sync[11] = raw;
      for (int x = 0; x < 8; ++x) sync[16 + x] = static_cast<int>(cur >> (8 * x)) & 0xff;
    }
    lds_rank = MODE != 1 ? static_cast<int>(old >> (8 * xcc)) & 0xff : blockIdx.x;
    lds_team = !ok ? -1 : MODE != 1 ? (static_cast<int>(xcc) == best ? best_n : 0) : gridDim.x;
  }
  __syncthreads();
  if (lds_team <= 0) return;
  const int G = lds_team, b = lds_rank;
  long long t0 = clock64();
  int done = 0;
  for (int r = 0; r < rounds; ++r) {
    // every workgroup writes a word, barrier, reads its neighbour's word and checks it
    if (threadIdx.x == 0) {
      if (MODE == 0) __hip_atomic_store(&data[b], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 2) (void)__hip_atomic_exchange(&data[b], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(&data[b], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 2 && b == 0) {
        (void)__hip_atomic_exchange(&data[512 + 2 * r], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        (void)__hip_atomic_exchange(&data[513 + 2 * r], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      int ok = 1;
      if (MODE != 1) __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int target = (r + 1) * G;
      unsigned spins = 0;
      while ((MODE != 1 ? LD_L2(bar) : LD_DEV(bar)) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 20)) { ok = 0; break; }
      }
      if (ok) {
        const int nb = (b + 1) % G;
        const int v = MODE != 1 ? LD_L2(&data[nb]) : LD_DEV(&data[nb]);
        if (v < r + 1) { ok = 0; sync[12] = r; sync[13] = v; }
      }
      lds_flag = ok;
    }
    __syncthreads();
    if (!lds_flag) { if (threadIdx.x == 0) *fail = 1; break; }
    ++done;
  }
  if (threadIdx.x == 0 && b == 0) { cycles[0] = clock64() - t0; cycles[1] = done; cycles[2] = G; }
}

template <int MODE>
static void run(int wgs, int rounds) {
  int *sync, *data;
  long long *cyc;
  hipMalloc(&sync, 4096); hipMalloc(&data, 65536); hipMalloc(&cyc, 64);
  hipMemset(sync, 0, 4096); hipMemset(data, 0, 65536); hipMemset(cyc, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  team_kernel<MODE><<<wgs, 512>>>(sync, data, rounds, cyc);
  hipEventRecord(e1);
  hipError_t err = hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  int h[32]; long long hc[3];
  hipMemcpy(h, sync, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  printf("mode %d wgs %d: err %d meet_ok %d best_xcd %d team %lld rounds done %lld/%d fail %d  %.3f ms  -> %.2f us per barrier round | per-XCD",
         MODE, wgs, (int)err, h[8], h[9], hc[2], hc[1], rounds, h[1], ms, ms * 1e3 / (hc[1] ? hc[1] : 1));
  for (int x = 0; x < 8; ++x) printf(" %d", h[16 + x]);
  printf(" raw_xcc_reg 0x%x stale(r=%d v=%d)\n", h[11], h[12], h[13]);
  hipFree(sync); hipFree(data); hipFree(cyc);
}

int main() {
  run<0>(248, 2000);
  run<2>(248, 2000);
  run<2>(248, 3);
  run<0>(128, 2000);
  run<1>(32, 2000);
  run<1>(248, 2000);
  return 0;
}
