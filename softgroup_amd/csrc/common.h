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

// Wave-wide integer reductions / scan on the DPP path (row_shr within rows of 16 lanes, row_bcast:15 /
// row_bcast:31 across them): six plain VALU operations, ~100 cycles.  The __shfl_xor / __shfl_up forms
// these replace compile to ds_bpermute -- a trip through the LDS crossbar per step, ~600 cycles per
// reduction -- which was most of the per-instance time of the single-wave panoptic walk.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_or_zero(int v) {      // lanes without a source read 0
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, false);
}
// CONTRACT: all 64 lanes active (EXEC full) at the call -- an inactive source lane contributes 0 to the
// scan and an inactive lane 63 leaves wave_sum's readlane with a stale register.  Every caller in this
// library calls them from wave-uniform control flow; divergent code must use wave_max's shuffle form (or
// ballot + popcount).  row_bcast:15 / :31 are GFX9-family DPP controls (this library targets gfx950 only).
// inclusive scan across the wave
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += dpp_or_zero<0x111>(v);                 // row_shr:1
  v += dpp_or_zero<0x112>(v);                 // row_shr:2
  v += dpp_or_zero<0x114>(v);                 // row_shr:4
  v += dpp_or_zero<0x118>(v);                 // row_shr:8
  v += dpp_or_zero<0x142, 0xa>(v);            // row_bcast:15 into rows 1 and 3
  v += dpp_or_zero<0x143, 0xc>(v);            // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ int wave_sum(int v) {
  return __builtin_amdgcn_readlane(wave_incl_scan(v), 63);
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// Several buffers filled by ONE launch (each hipMemsetAsync is a graph node of its own: ~5 us of GPU
// time and a launch gap; an index build used to issue ten of them).  Regions are 4-byte aligned, sizes
// multiples of 4; `byte` is replicated like memset's value.
constexpr int kFillMax = 16;
struct FillList {
  uint32_t *p[kFillMax];
  size_t words[kFillMax];
  uint32_t pattern[kFillMax];
  int n = 0;
  void add(void *ptr, size_t bytes, int byte) {
    if (bytes == 0 || ptr == nullptr || n >= kFillMax) return;      // (callers static_assert their region count)
    p[n] = static_cast<uint32_t *>(ptr);
    words[n] = bytes / 4;
    pattern[n] = 0x01010101u * static_cast<uint32_t>(byte & 0xff);
    ++n;
  }
};
static __global__ void __launch_bounds__(256) fill_many_kernel(FillList f) {
  const int r = blockIdx.y;
  uint32_t *p = f.p[r];
  const size_t w = f.words[r];
  const uint32_t v = f.pattern[r];
  // 16-byte stores over the aligned middle, scalar head / tail
  const size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) / 4 % 4;
  const size_t h = head < w ? head : w;
  const size_t quads = (w - h) / 4;
  uint4 *q = reinterpret_cast<uint4 *>(p + h);
  const uint4 vv = make_uint4(v, v, v, v);
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < quads; i += gridDim.x * 256ull) q[i] = vv;
  if (blockIdx.x == 0) {
    if (threadIdx.x < h) p[threadIdx.x] = v;
    const size_t t0 = h + quads * 4;
    if (threadIdx.x < w - t0) p[t0 + threadIdx.x] = v;
  }
}
inline void fill_many(const FillList &f, hipStream_t stream) {
  if (f.n == 0) return;
  size_t mx = 0;
  for (int i = 0; i < f.n; ++i) mx = f.words[i] > mx ? f.words[i] : mx;
  const int gx = grid_for(static_cast<int64_t>(mx / 4 + 1), 256, 1024);
  fill_many_kernel<<<dim3(gx, f.n), 256, 0, stream>>>(f);
}

// conv arithmetic requested by the calling thread (spconv_conv.hip); -1 = none
extern thread_local int t_conv_arith;
// multi-layer conv launches (spconv_conv.hip, "conv chain"): between _begin and _end on the calling thread
// sg_spconv_gather_conv_f32 records the layers the chain kernel takes instead of launching them
void conv_chain_begin(hipStream_t stream);
bool conv_chain_recording();
int conv_chain_end();       // launches what was recorded and closes the chain
void conv_chain_abort();    // closes the chain, dropping what was recorded (error paths)
int conv_chain_concat(const float *a, const float *b, int64_t rows, int ca, int cb, const float *scale,
                      const float *shift, float *out, float *out_act);
int conv_chain_bn_relu(const float *x, const float *scale, const float *shift, int64_t rows, int c, float *out);
int conv_chain_check_abort(const char *who);   // SG_ERR_LAUNCH once after a barrier of an earlier chain timed out
// sg_spconv_gather_conv_f32 with bf16 rows on either side (spconv_conv.hip; the executor's arithmetic 3)
int conv_gather_rows(const void *in, int num_in_rows, const int32_t *nbr, int M_out, int K, int Cin, int Cout,
                     const float *w_k8, const float *post_scale, const float *post_shift, const void *residual,
                     const float *act_scale, const float *act_shift, void *out_act, const int32_t *order,
                     const uint32_t *tile_mask, const int32_t *nbr_tiles, void *out, void *ws, size_t ws_bytes,
                     sg_stream_t stream, int in16, int out16, int res16);
// sg_spconv_wgrad writing the parameter's layout [Cout][K][Cin] when out_oki (spconv_train.hip; the training tape)
int spconv_wgrad_layout(const void *in, int in_bf16, const void *g_out, int g_bf16, const int32_t *nbr_t, int M_out,
                        int K, int Cin, int Cout, float *dw, int out_oki, void *ws, size_t ws_bytes,
                        sg_stream_t stream);
// one launch for many sg_spconv_pack_weight calls (spconv_conv.hip; the training tape packs a step's weights twice)
struct PackJob {
  const float *w;
  float *out;
  int cout, kvol, cin, mode;
};
int spconv_pack_weights(const PackJob *jobs, int n, sg_stream_t stream);
// octree ball query whose count pass parks short lists for the fill pass (octree.hip; sg_scan_grouping_pp)
size_t octree_stash_bytes(int n);
int octree_ballquery_count_stash(const float *points, const float *boxes, const int32_t *pt_inds,
                                 const int32_t *pt_start_len, int n, float radius, int32_t *start_len, int32_t *stash,
                                 hipStream_t stream);
int octree_ballquery_fill_stash(const float *points, const float *boxes, const int32_t *pt_inds,
                                const int32_t *pt_start_len, int n, float radius, const int32_t *start_len,
                                const int32_t *stash, int32_t *idx, hipStream_t stream);
// per-(device, stream) runtime state of the conv launches / the executors' index builds (sg_stream_release)
void conv_release_stream(int dev, hipStream_t stream);
void unet_release_stream(int dev, hipStream_t stream);
void scan_release_stream(int dev, hipStream_t stream);
void bfs_release_stream(int dev, hipStream_t stream);
// deferred join of the giant clusters' replay (bfs.hip; sg_scan_grouping_pp overlaps it with the next classes)
constexpr int kBfsGiantMin = 16384;      // clusters above this many points are "giant" (bfs.hip: kBigMin)
struct BfsDefer {
  bool deferred = false;
  hipStream_t side = nullptr;
};
void bfs_emit_defer(BfsDefer *d);                        // request for the thread's next sg_bfs_cluster_emit
int bfs_emit_join(const BfsDefer &d, hipStream_t stream);   // `stream` waits for everything queued on d.side
int bfs_label_max_kept(const void *ws);                  // largest kept cluster of the thread's last labelling on ws (-1: unknown)
int bfs_count_kept_members(const void *ws, size_t ws_bytes, int n, int64_t n_edges, const int32_t *l2p, int n_pts,
                           const float *thr_dev, int32_t *out, hipStream_t stream);

}  // namespace sg
