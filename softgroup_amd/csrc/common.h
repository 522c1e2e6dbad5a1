// common.h -- shared helpers for libsoftgroup_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/softgroup_hip.h"

namespace sg {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);
int32_t *pinned_words();   // 64 pinned int32 of the calling host thread (core.hip), or null

inline hipStream_t as_stream(sg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SG_ERR_LAUNCH;
  }
  return SG_OK;
}

#define SG_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      sg::set_error(__VA_ARGS__);  \
      return SG_ERR_ARG;           \
    }                              \
  } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Workspace {
  char *base;
  size_t size, off;
  Workspace(void *p, size_t n) : base(static_cast<char *>(p)), size(n), off(0) {}
  template <typename T>
  T *take(size_t count) {
    size_t bytes = align_up(count * sizeof(T));
    if (off + bytes > size) return nullptr;
    T *r = reinterpret_cast<T *>(base + off);
    off += bytes;
    return r;
  }
};

inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return static_cast<int>(g);
}

// ---- device helpers -----------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  return x ^ (x >> 33);
}

// exclusive prefix count of set bits below this lane in a wave ballot
__device__ __forceinline__ int mask_prefix(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}
// inclusive scan across the wave
__device__ __forceinline__ int wave_incl_scan(int v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (l >= o) v += t;
  }
  return v;
}

// conv arithmetic requested by the calling thread (spconv_conv.hip); -1 = none
extern thread_local int t_conv_arith;

}  // namespace sg
