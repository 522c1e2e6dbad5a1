// unet_exec.hip -- native executor of the sparse U-Net (inference): one C call runs
//   [input SubMConv3d] -> UBlock (recursive: residual blocks, down conv, inner UBlock, inverse conv,
//   skip concat, tail blocks) -> [output BatchNorm1d + ReLU]
// i.e. softgroup/model/softgroup.py:60-65 (backbone) and :93-95 (tiny U-Net) with the modules of
// softgroup/model/blocks.py:44-143, on the rulebook / plan / conv entry points of this library.
// The Python module path (softgroup_amd/spconv + model/blocks.py) launches the same kernels in
// the same order; it stays the path for training.  Here nothing but kernel launches happens
// between two layers: no interpreter, no allocator calls (a caller-provided arena, stack
// discipline per level) and ONE host sync per forward.  Everything that depends only on voxel
// coordinates -- the gather tables and tile plans of ALL levels -- is built up front by the
// whole-pyramid index build (sg_spconv_pyramid_rows / _build, spconv_rulebook.hip): ~30 launches
// and one read-back of the level row counts for the entire U-Net, on an internal index stream;
// the caller's stream waits for it once and then runs nothing but convolutions.
// Runtime state (index stream, events, pinned read-back words) is kept per (device, caller
// stream): concurrent scans on different streams do not share any of it.
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include <stdlib.h>

#include "common.h"
#include "radix_sort.h"
#include "unet_common.h"

namespace sg {

// ---------------------------------------------------------------------------------------------
// Internal row order (round 5; SG_UNET_MORTON=1, off by default).  The rows of an API tensor are in first-seen order of a randomly
// ordered point cloud, i.e. spatially unsorted: the 27-neighbour gather of a 32-row tile then
// touches ~150 unrelated 128-B lines, no two tiles on an XCD share any, and every re-read of a
// row misses that XCD's 4 MB L2.  The executor therefore works on a PERMUTED copy of the level-0
// voxels, sorted by (batch, Morton code of the coordinates): the whole-pyramid index build numbers
// the sites of every coarser level by their smallest level-0 descendant, and the descendants of a
// coarse cell are contiguous in Morton order -- so every level of the U-Net comes out in Morton
// order from this ONE sort.  Behind the API nothing changes: the input features are gathered into
// the internal order where they are padded anyway, the output rows are scattered back.
// key = top 24 bits of (batch | z-order interleave of the coordinates >> s): 3 radix passes; rows
// that share a key (the same 2^s-cell) keep their first-seen order (stable sort).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) batch_max_kernel(const int32_t *__restrict__ indices, int M,
                                                       int32_t *__restrict__ bmax) {
  int m = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) m = max(m, indices[4 * i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(bmax, m);
}
__device__ __forceinline__ uint64_t spread3(uint32_t v) {      // 21 bits -> every third bit
  uint64_t x = v & 0x1fffffu;
  x = (x | (x << 32)) & 0x1f00000000ffffull;
  x = (x | (x << 16)) & 0x1f0000ff0000ffull;
  x = (x | (x << 8)) & 0x100f00f00f00f00full;
  x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}
__global__ void __launch_bounds__(256) morton_key_kernel(const int32_t *__restrict__ indices, int M,
                                                        int dim_bits, const int32_t *__restrict__ bmax,
                                                        uint32_t *__restrict__ key, int32_t *__restrict__ val) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int4 c = reinterpret_cast<const int4 *>(indices)[i];
  const int bb = 32 - __clz(*bmax);                 // bits of the batch index
  const uint64_t code = (static_cast<uint64_t>(static_cast<uint32_t>(c.x)) << (3 * dim_bits)) |
                        (spread3(c.y) << 2) | (spread3(c.z) << 1) | spread3(c.w);
  const int total = bb + 3 * dim_bits;
  key[i] = static_cast<uint32_t>(total > 24 ? code >> (total - 24) : code);
  val[i] = i;
}
__global__ void __launch_bounds__(256) permute_indices_kernel(const int32_t *__restrict__ indices,
                                                             const int32_t *__restrict__ perm, int M,
                                                             int32_t *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < M) reinterpret_cast<int4 *>(out)[i] = reinterpret_cast<const int4 *>(indices)[perm[i]];
}
// out[perm ? dst-major gather : identity]: rows of `in` [*, c4 float4] gathered (out[j] = in[perm[j]]) or
// scattered (out[perm[j]] = in[j])
template <bool SCATTER>
__global__ void __launch_bounds__(256) permute_rows_kernel(const float4 *__restrict__ in,
                                                          const int32_t *__restrict__ perm, int64_t rows,
                                                          int c4, float4 *__restrict__ out) {
  const int64_t total = rows * c4;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t r = t / c4;
    const int c = static_cast<int>(t - r * c4);
    const int64_t o = perm[r];
    if (SCATTER) out[o * c4 + c] = in[t];
    else out[t] = in[o * c4 + c];
  }
}

// out = [a | b] row-wise; optionally also out_act = relu(out * scale + shift) (the BatchNorm1d +
// ReLU in front of the first tail block)
__global__ void __launch_bounds__(256) concat2_kernel(const float4 *__restrict__ a,
                                                     const float4 *__restrict__ b, int64_t rows,
                                                     int ca4, int cb4,
                                                     const float4 *__restrict__ scale,
                                                     const float4 *__restrict__ shift,
                                                     float4 *__restrict__ out,
                                                     float4 *__restrict__ out_act) {
  const int c4 = ca4 + cb4;
  const int64_t total = rows * c4;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t r = t / c4;
    const int c = static_cast<int>(t - r * c4);
    const float4 v = c < ca4 ? a[r * ca4 + c] : b[r * cb4 + (c - ca4)];
    out[t] = v;
    if (out_act) {
      const float4 s = scale[c], h = shift[c];
      out_act[t] = make_float4(fmaxf(fmaf(v.x, s.x, h.x), 0.f), fmaxf(fmaf(v.y, s.y, h.y), 0.f),
                               fmaxf(fmaf(v.z, s.z, h.z), 0.f), fmaxf(fmaf(v.w, s.w, h.w), 0.f));
    }
  }
}

// the same over bf16 rows (arithmetic 3), 4 channels = 8 bytes per thread; the activation in fp32, rounded to
// nearest even when stored
__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__global__ void __launch_bounds__(256) concat2_bf16_kernel(const uint2 *__restrict__ a, const uint2 *__restrict__ b,
                                                          int64_t rows, int ca4, int cb4,
                                                          const float4 *__restrict__ scale,
                                                          const float4 *__restrict__ shift, uint2 *__restrict__ out,
                                                          uint2 *__restrict__ out_act) {
  const int c4 = ca4 + cb4;
  const int64_t total = rows * c4;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t r = t / c4;
    const int c = static_cast<int>(t - r * c4);
    const uint2 v = c < ca4 ? a[r * ca4 + c] : b[r * cb4 + (c - ca4)];
    out[t] = v;
    if (out_act) {
      const float4 s = scale[c], h = shift[c];
      const float x0 = fmaxf(fmaf(bf16_to_f32(v.x & 0xffffu), s.x, h.x), 0.f), x1 = fmaxf(fmaf(bf16_to_f32(v.x >> 16), s.y, h.y), 0.f);
      const float x2 = fmaxf(fmaf(bf16_to_f32(v.y & 0xffffu), s.z, h.z), 0.f), x3 = fmaxf(fmaf(bf16_to_f32(v.y >> 16), s.w, h.w), 0.f);
      out_act[t] = make_uint2(f32_to_bf16(x0) | (f32_to_bf16(x1) << 16), f32_to_bf16(x2) | (f32_to_bf16(x3) << 16));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// index build shared with the training executor (unet_common.h)
// ---------------------------------------------------------------------------------------------
// Trivial plans of the 1x1 convs on the identity branches (kernel volume 1, neighbour = the row
// itself): order[l][i] = i for i < rows, -1 up to whole 32-row tiles -- with K = 1 the same buffer is
// the gather table, the plan's row order and its per-tile gather blocks; mask[l][t] = 1 is every
// tile's offset mask (+ the plan's histogram behind them).  blockIdx.y = level.
struct IdentSegs {
  int32_t *order[SG_PYRAMID_MAX_LEVELS];
  uint32_t *mask[SG_PYRAMID_MAX_LEVELS];
  int rows[SG_PYRAMID_MAX_LEVELS];
  int n;
};
__global__ void __launch_bounds__(256) ident_plan_kernel(IdentSegs s) {
  const int l = blockIdx.y;
  const int rows = s.rows[l], tiles = (rows + 31) / 32, padded = tiles * 32;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < padded; i += gridDim.x * 256) s.order[l][i] = i < rows ? i : -1;
  // every tile has the one offset; histogram behind the masks: hist[1] = tiles
  for (int i = blockIdx.x * 256 + threadIdx.x; i < tiles + SG_PLAN_HIST_WORDS; i += gridDim.x * 256)
    s.mask[l][i] = i < tiles ? 1u : (i == tiles + 1 ? static_cast<uint32_t>(tiles) : 0u);
}

// ---- runtime state per (device, caller's stream): callers that run scans concurrently on
//      several streams (one host thread each) get an index stream of their own and never wait for
//      each other; calls on the same stream are serialised by the state's mutex
static bool index_stream_on() {      // SG_UNET_INDEX_STREAM=1: the index build on a stream of its own (see unet_build_index)
  static const bool on = getenv("SG_UNET_INDEX_STREAM") && atoi(getenv("SG_UNET_INDEX_STREAM")) != 0;
  return on;
}
struct StreamState {
  std::mutex mu;
  int32_t *host_rows = nullptr;     // pinned [SG_PYRAMID_MAX_LEVELS]
  int32_t *dev_rows = nullptr;      // device
  hipStream_t istream = nullptr;
  hipEvent_t ev_start = nullptr, ev_index = nullptr;
  bool ready = false;
};
static std::mutex table_mu;
static std::map<std::pair<int, hipStream_t>, StreamState *> table;

// sg_stream_release: the caller's stream is idle and about to be destroyed
void unet_release_stream(int dev, hipStream_t stream) {
  StreamState *st = nullptr;
  {
    std::lock_guard<std::mutex> g(table_mu);
    auto it = table.find(std::make_pair(dev, stream));
    if (it == table.end()) return;
    st = it->second;
    table.erase(it);
  }
  {
    std::lock_guard<std::mutex> g(st->mu);      // (no forward of that stream is inside the build)
    if (st->ready) {
      if (st->istream != nullptr) {
        hipStreamSynchronize(st->istream);
        hipStreamDestroy(st->istream);
      }
      hipEventDestroy(st->ev_start);
      hipEventDestroy(st->ev_index);
      hipHostFree(st->host_rows);
      hipFree(st->dev_rows);
    }
  }
  delete st;
}

int unet_build_index(const char *who, int L, const int32_t *indices, int num_rows,
                     const int32_t *spatial_shape_host, void *arena, size_t arena_bytes, sg_stream_t stream,
                     LevelIdx *li, size_t *used, std::unique_lock<std::mutex> *lock) {
  int dev = 0;
  SG_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0, "%s: no current device", who);
  StreamState *stp = nullptr;
  {
    std::lock_guard<std::mutex> g(table_mu);
    StreamState *&slot = table[std::make_pair(dev, as_stream(stream))];
    if (slot == nullptr) slot = new StreamState();     // until sg_stream_release (else: the process)
    stp = slot;
  }
  StreamState &st = *stp;
  std::unique_lock<std::mutex> guard(st.mu);
  if (!st.ready) {
    SG_REQUIRE(hipHostMalloc(reinterpret_cast<void **>(&st.host_rows), SG_PYRAMID_MAX_LEVELS * 4) == hipSuccess,
               "%s: pinned allocation failed", who);
    SG_REQUIRE(hipMalloc(reinterpret_cast<void **>(&st.dev_rows), SG_PYRAMID_MAX_LEVELS * 4) == hipSuccess,
               "%s: device allocation failed", who);
    SG_REQUIRE(!index_stream_on() || hipStreamCreateWithFlags(&st.istream, hipStreamNonBlocking) == hipSuccess,
               "%s: stream creation failed", who);
    SG_REQUIRE(hipEventCreateWithFlags(&st.ev_start, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&st.ev_index, hipEventDisableTiming) == hipSuccess,
               "%s: event creation failed", who);
    st.ready = true;
  }
  // The index build runs on the CALLER's stream (since the end of round 6).  Until then it had a stream of its own per
  // caller stream, so that the host's wait for the row counts did not wait for whatever the caller had queued; with the
  // scan as one C call nothing is queued there, and one stream less per scan worker measured 2.42-2.57 against 2.66-2.81
  // ms/scan with five scans in flight (five interleaved pairs), 3.20-3.48 against 3.38-3.91 in the 20-step region and
  // 4.49 / 4.57 against 4.61 / 4.61 ms for one scan (profiles/r06_index_stream.txt).  SG_UNET_INDEX_STREAM=1: as before.
  const bool own_istream = index_stream_on();
  hipStream_t istream = own_istream ? st.istream : as_stream(stream);
  sg_stream_t is = reinterpret_cast<sg_stream_t>(istream);
  // the index stream starts where the caller's stream is now: the coordinates are ready, and the
  // previous forward's convolutions no longer read the tables about to be overwritten
  if (own_istream && (hipEventRecord(st.ev_start, as_stream(stream)) != hipSuccess ||
                      hipStreamWaitEvent(istream, st.ev_start, 0) != hipSuccess)) {
    set_error("%s: event record/wait failed", who);
    return SG_ERR_LAUNCH;
  }
  // ---- index part of the arena (bump, never recycled inside a forward)
  Arena ix(arena, arena_bytes);
#define SG_IALLOC(var, T, count)                                                       \
  T *var = ix.take<T>(count);                                                          \
  if (var == nullptr) {                                                                \
    set_error("%s: arena too small (%zu bytes) for the index tables", who, arena_bytes); \
    return SG_ERR_WORKSPACE;                                                           \
  }
  // ---- rows of every level: one pass + one read-back (the only host sync of the forward; it
  //      waits for the index stream only -- whatever the caller's stream has queued keeps running)
  const size_t pws_bytes = sg_spconv_pyramid_workspace_bytes(num_rows, L);
  SG_IALLOC(pws, char, pws_bytes);
  SG_TRY(sg_spconv_pyramid_rows(indices, num_rows, spatial_shape_host, L, st.dev_rows, pws, pws_bytes, is));
  if (hipMemcpyAsync(st.host_rows, st.dev_rows, sizeof(int32_t) * L, hipMemcpyDeviceToHost, istream) != hipSuccess ||
      hipStreamSynchronize(istream) != hipSuccess) {
    set_error("%s: reading the level row counts failed", who);
    return SG_ERR_LAUNCH;
  }
  SG_REQUIRE(st.host_rows[0] == num_rows, "%s: duplicate voxel coordinates in the input "
             "(%d distinct of %d rows)", who, st.host_rows[0], num_rows);
  // ---- tables and plans of all levels
  sg_pyramid_level pl[SG_PYRAMID_MAX_LEVELS];
  for (int l = 0; l < L; ++l) {
    const int rows = st.host_rows[l];
    const int rows2 = l + 1 < L ? st.host_rows[l + 1] : 0;
    const size_t r = static_cast<size_t>(rows ? rows : 1), r2 = static_cast<size_t>(rows2 ? rows2 : 1);
    const size_t t = (r + 31) / 32, t2 = (r2 + 31) / 32;
    sg_pyramid_level &P = pl[l];
    P = sg_pyramid_level();
    P.rows = rows;
    SG_IALLOC(indices_l, int32_t, r * 4);
    SG_IALLOC(nbr, int32_t, r * 27);
    SG_IALLOC(so, int32_t, t * 32);
    SG_IALLOC(sm, uint32_t, t + SG_PLAN_HIST_WORDS);
    SG_IALLOC(sn, int32_t, t * 32 * 27);
    P.indices = indices_l; P.nbr = nbr;
    P.subm = sg_plan_ptrs{so, sm, sn};
    LevelIdx &I = li[l];
    I.rows = rows;
    for (int a = 0; a < 3; ++a) I.shape[a] = spatial_shape_host[a] >> l;
    I.subm.nbr = nbr; I.subm.order = so; I.subm.tile_mask = sm; I.subm.nbr_tiles = sn;
    I.subm.rows = rows; I.subm.kvol = 27;
    if (l + 1 < L) {
      SG_IALLOC(in2out, int32_t, r);
      SG_IALLOC(child, int32_t, r2 * 8);
      SG_IALLOC(inv, int32_t, r * 8);
      SG_IALLOC(dord, int32_t, t2 * 32);
      SG_IALLOC(dm, uint32_t, t2 + SG_PLAN_HIST_WORDS);
      SG_IALLOC(dn, int32_t, t2 * 32 * 8);
      SG_IALLOC(uo, int32_t, t * 32);
      SG_IALLOC(um, uint32_t, t + SG_PLAN_HIST_WORDS);
      SG_IALLOC(un, int32_t, t * 32 * 8);
      P.in2out = in2out; P.child = child; P.inv = inv;
      P.down = sg_plan_ptrs{dord, dm, dn};
      P.up = sg_plan_ptrs{uo, um, un};
      I.down.nbr = child; I.down.order = dord; I.down.tile_mask = dm; I.down.nbr_tiles = dn;
      I.down.rows = rows2; I.down.kvol = 8;
      I.up.nbr = inv; I.up.order = uo; I.up.tile_mask = um; I.up.nbr_tiles = un;
      I.up.rows = rows; I.up.kvol = 8;
    }
  }
  {
    const size_t nb = sg_spconv_pyramid_build_workspace_bytes(pl, L);
    SG_IALLOC(ws2, char, nb);
    SG_TRY(sg_spconv_pyramid_build(indices, num_rows, spatial_shape_host, L, pl, pws, pws_bytes, ws2, nb, is));
  }
  if (L > 1) {      // trivial plans of the 1x1 convs on the tail's identity branches
    IdentSegs segs;
    segs.n = L - 1;
    for (int l = 0; l + 1 < L; ++l) {
      const size_t tiles = (static_cast<size_t>(li[l].rows) + 31) / 32;
      SG_IALLOC(ord, int32_t, tiles ? tiles * 32 : 32);
      SG_IALLOC(tm, uint32_t, tiles + SG_PLAN_HIST_WORDS);
      segs.order[l] = ord;
      segs.mask[l] = tm;
      segs.rows[l] = li[l].rows;
      Plan &P = li[l].ident;
      P.nbr = ord; P.order = ord; P.nbr_tiles = ord; P.tile_mask = tm;
      P.rows = li[l].rows; P.kvol = 1;
    }
    ident_plan_kernel<<<dim3(grid_for(num_rows, 256), L - 1), 256, 0, istream>>>(segs);
  }
#undef SG_IALLOC
  if (own_istream && (hipEventRecord(st.ev_index, istream) != hipSuccess ||
                      hipStreamWaitEvent(as_stream(stream), st.ev_index, 0) != hipSuccess)) {
    set_error("%s: event record/wait failed", who);
    return SG_ERR_LAUNCH;
  }
  *used = ix.off;
  *lock = std::move(guard);
  return SG_OK;
}

// feature rows zero-padded to `cpad` channels (the input conv on the persistent kernel: Cin % 16 == 0)
// (`perm` != null: row r of the result is row perm[r] of `in` -- the executor's internal row order)
__global__ void __launch_bounds__(256) pad_channels_kernel(const float *__restrict__ in, int64_t rows, int cin,
                                                          int cpad, const int32_t *__restrict__ perm,
                                                          float *__restrict__ out) {
  const int64_t total = rows * cpad;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t r = t / cpad;
    const int c = static_cast<int>(t - r * cpad);
    const int64_t src = perm ? perm[r] : r;
    out[t] = c < cin ? in[src * cin + c] : 0.f;
  }
}

#define SG_ALLOC(var, T, count)                                                        \
  T *var = ar.take<T>(count);                                                          \
  if (var == nullptr) {                                                                \
    set_error("sg_unet_forward: arena too small (%zu bytes, need more than %zu)", ar.cap, ar.off); \
    return SG_ERR_WORKSPACE;                                                           \
  }

// called by sg_unet_forward on the calling thread when the index build has been enqueued and the first
// convolution is about to be (sg_scan_forward: the backbone token, scan_forward.hip); cleared by the call
thread_local void (*t_unet_conv_hook)(void *) = nullptr;
thread_local void *t_unet_conv_ctx = nullptr;

struct Exec {
  const sg_unet_desc *d;
  Arena ar;             // features and conv scratch: caller's stream only
  sg_stream_t stream;   // caller's stream (convolutions, elementwise)
  const LevelIdx *idx;  // per level: rows, tables and plans (complete before the first conv runs)
  const int32_t *perm = nullptr;   // internal row j = API row perm[j] (null: API order); level 0 input only

  Exec(const sg_unet_desc *desc, void *arena, size_t bytes, sg_stream_t s, const LevelIdx *li)
      : d(desc), ar(arena, bytes), stream(s), idx(li) {}

  // Conv chain (spconv_conv.hip): the levels from `l` down run as ONE multi-layer launch when level l has
  // at most SG_CONV_CHAIN_ROWS rows (default 6144: below ~1 900 units per layer the single launches use the
  // chain's decomposition anyway) and every level from l down has a multiple of 32 channels (32-channel
  // items, line-wise gather).  Opened in front of the strided conv INTO level l, closed behind level l's
  // last conv; with chains switched off the same layers are launched one by one, same decomposition.
  // arithmetic 3: the activations between the layers are bf16 rows (half the floats of a buffer); the features
  // handed in and the U-Net's output stay fp32
  bool r16 = false;
  size_t fsz(size_t n) const { return r16 ? (n + 1) / 2 : n; }
  bool chain_open = false;
  bool chain_from(int l) const {
    static const int max_rows = getenv("SG_CONV_CHAIN_ROWS") ? atoi(getenv("SG_CONV_CHAIN_ROWS")) : 6144;
    if (r16 || chain_open || idx[l].rows > max_rows || idx[l].rows <= 0) return false;
    for (int k = l; k < d->n_levels; ++k)
      if (d->levels[k].planes % 32 != 0) return false;
    return true;
  }
  struct ChainScope {      // closes the chain on every way out (an error path drops what was recorded)
    Exec &e;
    bool on;
    ChainScope(Exec &ex, bool open) : e(ex), on(open) {
      if (on) {
        conv_chain_begin(as_stream(e.stream));
        e.chain_open = true;
      }
    }
    int close() {
      if (!on) return SG_OK;
      on = false;
      e.chain_open = false;
      return conv_chain_end();
    }
    ~ChainScope() {
      if (on) {
        conv_chain_abort();
        e.chain_open = false;
      }
    }
  };

  // BatchNorm1d + ReLU of a consumer, applied by the producer: (scale, shift) and where the
  // activated copy goes
  struct Act {
    const float *scale = nullptr, *shift = nullptr;
    float *out = nullptr;
  };

  // in_f32 / out_f32: this layer's input / output rows are fp32 although the executor runs on bf16 rows (the
  // input conv reads the API's features, the U-Net's last conv writes the API's output)
  int conv(const float *in, int in_rows, const Plan &p, int cin, int cout, const float *w,
           const float *post_s, const float *post_b, const float *residual, const Act &act, float *out,
           bool in_f32 = false, bool out_f32 = false) {
    if (p.rows == 0) return SG_OK;
    const size_t m = ar.mark();
    const size_t nb = sg_spconv_conv_workspace_bytes(p.rows, cout);
    void *ws = nullptr;
    if (nb > 256) {
      ws = ar.take<char>(nb);   // optional: without it the conv simply does not split offsets
    }
    const int rc = !r16 ? sg_spconv_gather_conv_f32(in, in_rows, p.nbr, p.rows, p.kvol, cin, cout, w, post_s,
                                                    post_b, residual, act.scale, act.shift, act.out, p.order,
                                                    p.tile_mask, p.nbr_tiles, out, ws, ws ? nb : 0, stream)
                        : conv_gather_rows(in, in_rows, p.nbr, p.rows, p.kvol, cin, cout, w, post_s, post_b, residual,
                                           act.scale, act.shift, act.out, p.order, p.tile_mask, p.nbr_tiles, out, ws,
                                           ws ? nb : 0, stream, in_f32 ? 0 : 1, out_f32 ? 0 : 1, 1);
    ar.release(m);
    return rc;
  }

  // ResidualBlock (blocks.py:44-79): x + SubM(ReLU(BN(SubM(ReLU(BN(x)))))), 1x1 conv on the
  // identity branch when the channel count changes.  `xa` = relu(bn1(x)), made by whoever produced
  // x; `next` = the activation the consumer of this block's output wants (second output of conv2).
  int block(const sg_unet_block &b, const float *x, const float *xa, int rows, const Plan &subm,
            const Plan &ident, const float *post_s, const float *post_b, const Act &next, float *out,
            bool out_f32 = false) {
    const size_t m = ar.mark();
    const float *shortcut = x;
    if (b.w_i != nullptr) {
      SG_ALLOC(sc, float, fsz(static_cast<size_t>(rows) * b.cout));
      SG_TRY(conv(x, rows, ident, b.cin, b.cout, b.w_i, nullptr, nullptr, nullptr, Act(), sc));
      shortcut = sc;
    }
    SG_ALLOC(h, float, fsz(static_cast<size_t>(rows) * b.cout));
    SG_TRY(conv(xa, rows, subm, b.cin, b.cout, b.w1, b.bn2_scale, b.bn2_shift, nullptr, Act(), h));
    SG_TRY(conv(h, rows, subm, b.cout, b.cout, b.w2, post_s, post_b, shortcut, next, out, false, out_f32));
    ar.release(m);
    return SG_OK;
  }

  // UBlock (blocks.py:82-143).  `x` [rows, planes] -> `out` [rows, planes]; `xa` = relu(bn(x)) for
  // the first block's BatchNorm if the producer of x made it (else it is computed here); post =
  // BatchNorm+ReLU applied in place by the level's last conv (output_layer for the outermost
  // level, the parent's deconv BatchNorm for an inner one).
  int level(int l, const float *x, const float *xa, const float *pre_in, int pre_cin,
            const float *post_s, const float *post_b, float *out) {
    const sg_unet_level &L = d->levels[l];
    const int c = L.planes;
    const bool deeper = l + 1 < d->n_levels;
    const size_t m0 = ar.mark();
    const LevelIdx &I = idx[l];
    const int rows = I.rows;
    const Plan &subm = I.subm, &ident = I.ident;
    const size_t feat = fsz(static_cast<size_t>(rows ? rows : 1) * c);      // floats of one [rows, c] buffer
    // optional input conv (outermost level only): SubMConv3d(in, planes) on the same rulebook
    if (pre_in != nullptr) {
      SG_ALLOC(x0, float, feat);
      SG_ALLOC(x0a, float, feat);
      const Act a0{L.blocks[0].bn1_scale, L.blocks[0].bn1_shift, x0a};
      // weights packed for more input channels than the features have (a multiple of 16: the
      // persistent MFMA kernel instead of the general one): convolve a zero-padded copy
      const int cpk = d->input_cin_packed > pre_cin ? d->input_cin_packed : pre_cin;
      if ((cpk != pre_cin || perm != nullptr) && rows) {       // (the copy that also brings the rows into internal order)
        SG_ALLOC(xp, float, static_cast<size_t>(rows) * cpk);
        pad_channels_kernel<<<grid_for(static_cast<int64_t>(rows) * cpk, 256), 256, 0, as_stream(stream)>>>(
            pre_in, rows, pre_cin, cpk, perm, xp);
        pre_in = xp;
      }
      SG_TRY(conv(pre_in, rows, subm, cpk, c, d->input_w, nullptr, nullptr, nullptr, a0, x0, true, false));
      x = x0;
      xa = x0a;
    } else if (xa == nullptr) {
      SG_ALLOC(x0a, float, feat);
      if (conv_chain_recording())
        SG_TRY(conv_chain_bn_relu(x, L.blocks[0].bn1_scale, L.blocks[0].bn1_shift, rows, c, x0a));
      else
        SG_TRY(sg_bn_relu_f32(x, L.blocks[0].bn1_scale, L.blocks[0].bn1_shift, rows, c, 1, x0a, stream));
      xa = x0a;
    }
    // blocks: every conv2 also emits the activation its consumer wants
    const float *cur = x, *cur_a = xa;
    for (int i = 0; i < L.n_blocks; ++i) {
      const bool last = i == L.n_blocks - 1;
      float *dst = out;
      Act next;
      if (!(last && !deeper)) {
        SG_ALLOC(t, float, feat);
        SG_ALLOC(ta, float, feat);
        dst = t;
        next.out = ta;
        next.scale = last ? L.down_bn_scale : L.blocks[i + 1].bn1_scale;
        next.shift = last ? L.down_bn_shift : L.blocks[i + 1].bn1_shift;
      }
      const bool fin = last && !deeper;
      SG_TRY(block(L.blocks[i], cur, cur_a, rows, subm, ident, fin ? post_s : nullptr,
                   fin ? post_b : nullptr, next, dst, fin && l == 0));
      cur = dst;
      cur_a = next.out;
    }
    if (deeper) {
      const int c2 = d->levels[l + 1].planes;
      const int rows2 = idx[l + 1].rows;
      const Plan &down = I.down, &up = I.up;
      // ---- (BN -> ReLU done by the last block) -> SparseConv3d(c, c2, k2 s2); its second output
      //      feeds the first BatchNorm of the inner level
      const sg_unet_level &L2 = d->levels[l + 1];
      const size_t feat2 = fsz(static_cast<size_t>(rows2 ? rows2 : 1) * c2);
      SG_ALLOC(y, float, feat2);
      SG_ALLOC(ya, float, feat2);
      const Act ay{L2.blocks[0].bn1_scale, L2.blocks[0].bn1_shift, ya};
      SG_ALLOC(z, float, feat2);
      {
        ChainScope chain(*this, chain_from(l + 1));      // the strided conv and everything below it: one launch
        SG_TRY(conv(cur_a, rows, down, c, c2, L.down_w, nullptr, nullptr, nullptr, ay, y));
        // ---- inner UBlock; its last conv applies this level's deconv BatchNorm + ReLU in place
        SG_TRY(level(l + 1, y, ya, nullptr, 0, L.up_bn_scale, L.up_bn_shift, z));
        SG_TRY(chain.close());
      }
      // ---- SparseInverseConv3d(c2, c): gather table = parent row per fine voxel (plan `up`, built
      //      on the index stream before the descent), then the skip concat (blocks.py:135-139)
      //      with the first tail block's BatchNorm + ReLU as a second output
      SG_ALLOC(cat, float, 2 * feat);
      SG_ALLOC(cata, float, 2 * feat);
      {
        const size_t m = ar.mark();
        SG_ALLOC(upf, float, feat);
        SG_TRY(conv(z, rows2, up, c2, c, L.up_w, nullptr, nullptr, nullptr, Act(), upf));
        if (rows && r16)
          concat2_bf16_kernel<<<grid_for(static_cast<int64_t>(rows) * (2 * c / 4), 256), 256, 0, as_stream(stream)>>>(
              reinterpret_cast<const uint2 *>(cur), reinterpret_cast<const uint2 *>(upf), rows, c / 4, c / 4,
              reinterpret_cast<const float4 *>(L.tail[0].bn1_scale), reinterpret_cast<const float4 *>(L.tail[0].bn1_shift),
              reinterpret_cast<uint2 *>(cat), reinterpret_cast<uint2 *>(cata));
        else if (rows && conv_chain_recording())
          SG_TRY(conv_chain_concat(cur, upf, rows, c, c, L.tail[0].bn1_scale, L.tail[0].bn1_shift, cat, cata));
        else if (rows)
          concat2_kernel<<<grid_for(static_cast<int64_t>(rows) * (2 * c / 4), 256), 256, 0, as_stream(stream)>>>(
              reinterpret_cast<const float4 *>(cur), reinterpret_cast<const float4 *>(upf), rows, c / 4,
              c / 4, reinterpret_cast<const float4 *>(L.tail[0].bn1_scale),
              reinterpret_cast<const float4 *>(L.tail[0].bn1_shift), reinterpret_cast<float4 *>(cat),
              reinterpret_cast<float4 *>(cata));
        ar.release(m);
      }
      // ---- tail blocks: (2c -> c), (c -> c)
      cur = cat;
      cur_a = cata;
      for (int i = 0; i < L.n_blocks; ++i) {
        const bool last = i == L.n_blocks - 1;
        float *dst = out;
        Act next;
        if (!last) {
          SG_ALLOC(t, float, feat);
          SG_ALLOC(ta, float, feat);
          dst = t;
          next = Act{L.tail[i + 1].bn1_scale, L.tail[i + 1].bn1_shift, ta};
        }
        SG_TRY(block(L.tail[i], cur, cur_a, rows, subm, ident, last ? post_s : nullptr,
                     last ? post_b : nullptr, next, dst, last && l == 0));
        cur = dst;
        cur_a = next.out;
      }
    }
    ar.release(m0);
    return check_launch("sg_unet_forward");
  }
};

// index part of the arena (the pyramid's hash workspace, every level's tables and plans, the plan
// build scratch): priced with all `num_rows` voxels on every level (levels can only shrink)
static size_t level_index_bytes(size_t rows, size_t rows_next, bool deeper) {
  const size_t t = (rows + 31) / 32, t2 = (rows_next + 31) / 32;
  size_t b = align_up(rows * 4 * 4) + align_up(rows * 27 * 4) +                       // indices, nbr
             align_up(t * 32 * 4) + align_up((t + SG_PLAN_HIST_WORDS) * 4) + align_up(t * 32 * 27 * 4);  // subm plan
  if (deeper)
    b += align_up(rows * 4) + align_up(rows_next * 8 * 4) + align_up(rows * 8 * 4) +   // in2out, child, inv
         align_up(t2 * 32 * 4) + align_up((t2 + SG_PLAN_HIST_WORDS) * 4) + align_up(t2 * 32 * 8 * 4) +   // down plan
         align_up(t * 32 * 4) + align_up((t + SG_PLAN_HIST_WORDS) * 4) + align_up(t * 32 * 8 * 4);       // up plan
  return b + 4096;
}
size_t unet_index_bytes(int L, int num_rows) {
  const size_t rows = static_cast<size_t>(num_rows > 0 ? num_rows : 1);
  size_t total = (1 << 20) + sg_spconv_pyramid_workspace_bytes(num_rows, L) +
                 (L + 1) * (align_up(rows * 4) + align_up(rows / 8 + 256) + 4096);   // trivial plans of the 1x1 convs
  sg_pyramid_level bound[SG_PYRAMID_MAX_LEVELS];
  for (int l = 0; l < L && l < SG_PYRAMID_MAX_LEVELS; ++l) {
    total += level_index_bytes(rows, rows, l + 1 < L);
    bound[l].rows = num_rows;
  }
  total += sg_spconv_pyramid_build_workspace_bytes(bound, L < SG_PYRAMID_MAX_LEVELS ? L : SG_PYRAMID_MAX_LEVELS);
  return align_up(total, 4096);
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_unet_arena_bytes(const sg_unet_desc *d, int num_rows) {
  // feature part: at most ~12 live buffers of 2*planes floats per level (stack discipline)
  size_t total = unet_index_bytes(d->n_levels, num_rows) + (1 << 20);
  const size_t rows = static_cast<size_t>(num_rows > 0 ? num_rows : 1);
  // internal row order: keys + row ids + sort scratch, permuted coordinates, permuted input and output rows
  total += 2 * align_up(rows * 4) + radix_sort_workspace_bytes(static_cast<int64_t>(rows)) + align_up(rows * 16) +
           2 * align_up(rows * static_cast<size_t>(d->levels[0].planes) * 4) + 4096;
  for (int l = 0; l < d->n_levels; ++l)
    total += rows * 12 * 2 * static_cast<size_t>(d->levels[l].planes) * 4 + (64 << 10);
  return total;
}

int sg_unet_forward(const sg_unet_desc *d, const float *feats, const int32_t *indices, int num_rows,
                    const int32_t *spatial_shape_host, float *out, void *arena, size_t arena_bytes,
                    sg_stream_t stream) {
  SG_REQUIRE(d != nullptr && d->n_levels >= 1 && d->levels != nullptr, "sg_unet_forward: bad descriptor");
  SG_REQUIRE(d->n_levels <= SG_PYRAMID_MAX_LEVELS, "sg_unet_forward: at most %d levels", SG_PYRAMID_MAX_LEVELS);
  SG_REQUIRE(num_rows >= 0, "sg_unet_forward: bad num_rows");
  for (int l = 0; l < d->n_levels; ++l)
    SG_REQUIRE(d->levels[l].planes % 4 == 0 && d->levels[l].n_blocks >= 1,
               "sg_unet_forward: level %d: planes must be a multiple of 4", l);
  SG_REQUIRE(d->arithmetic == 0 || d->arithmetic == 2 || d->arithmetic == 3, "sg_unet_forward: arithmetic must be 0, 2 or 3");
  if (num_rows == 0) return SG_OK;
  SG_TRY(conv_chain_check_abort("sg_unet_forward"));
  const int L = d->n_levels;
  struct ArithScope {      // this call's convolutions, on this thread only
    int keep;
    explicit ArithScope(int a) : keep(t_conv_arith) { if (a > 0) t_conv_arith = a; }
    ~ArithScope() { t_conv_arith = keep; }
  } arith_scope(d->arithmetic == 3 ? 2 : d->arithmetic);
  // bf16 rows between the layers (arithmetic 3) need an input conv (the API's fp32 features enter through it) and
  // 32-channel items on every level; anything else runs as arithmetic 2 (bf16 operands, fp32 rows)
  bool rows16 = d->arithmetic == 3 && d->input_w != nullptr;
  for (int l = 0; l < d->n_levels; ++l) rows16 = rows16 && d->levels[l].planes % 32 == 0;
  // ---- internal row order (see morton_key_kernel): SG_UNET_MORTON=1 turns it on (A/B knob; off by
  //      default: measured neutral under the default tile plan and not enough to pay for the spatially
  //      local plans' extra items, profiles/r05_conv_locality.txt), SG_UNET_MORTON_MIN = smallest input
  static const int morton_env = getenv("SG_UNET_MORTON") ? atoi(getenv("SG_UNET_MORTON")) : 0;
  static const int morton_min = getenv("SG_UNET_MORTON_MIN") ? atoi(getenv("SG_UNET_MORTON_MIN")) : 16384;
  const bool pre = d->input_w != nullptr;
  const int c0 = d->levels[0].planes;
  const bool morton = morton_env != 0 && num_rows >= morton_min;
  Arena head(arena, arena_bytes);
  const int32_t *perm = nullptr;
  const int32_t *idx_in = indices;
  const float *feats_in = feats;
  float *out_int = out;
  if (morton) {
    hipStream_t st = as_stream(stream);
    const size_t M = static_cast<size_t>(num_rows);
    uint32_t *key = head.take<uint32_t>(M);
    int32_t *val = head.take<int32_t>(M);
    const size_t rs_bytes = radix_sort_workspace_bytes(static_cast<int64_t>(M));
    void *rs_ws = head.take<char>(rs_bytes);
    int32_t *idx_p = head.take<int32_t>(M * 4);
    int32_t *bmax = head.take<int32_t>(64);
    float *out_p = head.take<float>(M * c0);
    float *feats_p = pre ? out_p : head.take<float>(M * c0);     // (with an input conv the padding copy permutes)
    if (feats_p == nullptr || out_p == nullptr) {
      set_error("sg_unet_forward: arena too small (%zu bytes) for the internal row order", arena_bytes);
      return SG_ERR_WORKSPACE;
    }
    int dim_bits = 1;
    for (int a = 0; a < 3; ++a)
      while ((1 << dim_bits) < spatial_shape_host[a]) ++dim_bits;
    SG_REQUIRE(dim_bits <= 20, "sg_unet_forward: spatial extent above 2^20");
    hipMemsetAsync(bmax, 0, 4, st);
    batch_max_kernel<<<grid_for(num_rows, 256, 256), 256, 0, st>>>(indices, num_rows, bmax);
    morton_key_kernel<<<(num_rows + 255) / 256, 256, 0, st>>>(indices, num_rows, dim_bits, bmax, key, val);
    uint32_t *ks;
    int32_t *vs;
    SG_TRY(radix_sort_pairs(key, val, static_cast<int64_t>(M), 24, rs_ws, rs_bytes, st, &ks, &vs));
    permute_indices_kernel<<<(num_rows + 255) / 256, 256, 0, st>>>(indices, vs, num_rows, idx_p);
    perm = vs;
    idx_in = idx_p;
    out_int = out_p;
    if (!pre) {
      permute_rows_kernel<false><<<grid_for(static_cast<int64_t>(M) * (c0 / 4), 256), 256, 0, st>>>(
          reinterpret_cast<const float4 *>(feats), vs, num_rows, c0 / 4, reinterpret_cast<float4 *>(feats_p));
      feats_in = feats_p;
    }
    SG_TRY(check_launch("sg_unet_forward(row order)"));
  }
  const size_t head_bytes = align_up(head.off, 4096);
  char *rest = static_cast<char *>(arena) + head_bytes;
  const size_t rest_bytes = arena_bytes > head_bytes ? arena_bytes - head_bytes : 0;
  // ---- gather tables and tile plans of all levels (index stream; this stream waits for them once)
  LevelIdx li[SG_PYRAMID_MAX_LEVELS];
  std::unique_lock<std::mutex> guard;
  size_t index_bytes = 0;
  SG_TRY(unet_build_index("sg_unet_forward", L, idx_in, num_rows, spatial_shape_host, rest, rest_bytes, stream,
                          li, &index_bytes, &guard));
  // ---- the convolutions: feature part of the arena
  const size_t used = align_up(index_bytes, 4096);
  if (rest_bytes <= used) {
    set_error("sg_unet_forward: arena too small (%zu bytes)", arena_bytes);
    return SG_ERR_WORKSPACE;
  }
  if (t_unet_conv_hook != nullptr) {
    void (*hook)(void *) = t_unet_conv_hook;
    t_unet_conv_hook = nullptr;
    hook(t_unet_conv_ctx);
  }
  Exec ex(d, rest + used, rest_bytes - used, stream, li);
  ex.perm = pre ? perm : nullptr;
  ex.r16 = rows16;
  int rc;
  {
    // a U-Net that is small from level 0 on (the tiny U-Net over the proposals' voxels): the whole forward
    Exec::ChainScope chain(ex, !pre && ex.chain_from(0));
    rc = ex.level(0, pre ? nullptr : feats_in, nullptr, pre ? feats : nullptr, d->input_cin,
                  d->out_bn_scale, d->out_bn_shift, out_int);
    if (rc == SG_OK) rc = chain.close();
  }
  if (rc == SG_OK && morton) {      // back to the API's row order
    permute_rows_kernel<true><<<grid_for(static_cast<int64_t>(num_rows) * (c0 / 4), 256), 256, 0, as_stream(stream)>>>(
        reinterpret_cast<const float4 *>(out_int), perm, num_rows, c0 / 4, reinterpret_cast<float4 *>(out));
    rc = check_launch("sg_unet_forward(row order)");
  }
  return rc;
}

}  // extern "C"
