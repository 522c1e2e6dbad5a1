// scan_forward.hip -- ONE C call per scan: SoftGroup.forward_test (softgroup/model/softgroup.py:299-361)
// for the plain SoftGroup configuration, chained on the caller's stream:
//   voxel feature pooling (:305, ops.voxelization of [feats | coords_float])
//   -> backbone (:307-309; sg_unet_forward) -> devoxelize + point-wise heads + arg-max (:363-378, :320;
//   sg_pointwise_heads) -> softmax of the semantic scores (:415) -> grouping head + proposal
//   voxelisation (:411-480, :655-709; sg_scan_grouping) -> tiny U-Net (:671-675; sg_unet_forward) ->
//   mask_linear on the proposal points, global average pool, cls_linear / iou_score_linear (:676-686)
//   -> instance extraction + RLE text (:537-604; sg_scan_instances); the dense per-point results of
//   get_point_wise_results / get_gt_instances (:641-653) leave in one packed device-to-host copy.
// The reference's test loop runs one scan per process at a time (tools/test.py:145-150) and every stage
// above is a Python call there; rounds 4-5 of this library had the stages as C calls with Python in
// between, so that several scans in flight meant several Python threads sharing one interpreter lock.
// Here the host thread makes ONE call and holds no lock while the scan runs.
//
// Device memory: the caller's arena, carved in call order:
//   [ persistent part: int32 voxel coordinates, pooled voxel features, backbone output, point tensors,
//     the packed dense results ] [ scratch: the backbone's executor arena, reused afterwards by the
//     grouping sub-arena, the refinement tensors + the tiny U-Net's executor arena, the instance sub-arena ]
// Everything runs on the caller's stream except the dense results' device-to-host copy, which goes to a
// side stream owned by this file (per (device, caller stream)) and is started right before the ordered
// BFS emission -- one workgroup per cluster, the rest of the chip idle -- instead of right after the
// point-wise heads, where its 131 072-workgroup blit kept the one-workgroup class-selection scan waiting
// for a slot for 0.2 ms (profiles/r05_kernel_stats.csv: select_scan_kernel min 20.6 / max 230 us).
#include <stdlib.h>
#include <string.h>
#include <condition_variable>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"
#include "heads.h"

#define SG_TRY_(expr)             \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != SG_OK) return rc_; \
  } while (0)

namespace sg {

// hook of sg_scan_grouping (scan_exec.hip): called once, right before the ordered emission
extern thread_local void (*t_scan_emit_hook)(void *);
extern thread_local void (*t_unet_conv_hook)(void *);
extern thread_local void *t_unet_conv_ctx;
extern thread_local void *t_scan_emit_ctx;

// ---- voxel feature pooling over [a | b] (softgroup.py:302-305: torch.cat + ops.voxelization): the
//      arithmetic of voxelize_fp_kernel (seg_ops.hip), column p taken from a (p < ca) or b
__global__ void __launch_bounds__(256) voxelize_cat_kernel(const float *__restrict__ a, int ca,
                                                          const float *__restrict__ b, int cb,
                                                          const int32_t *__restrict__ rules, int M, int max_active,
                                                          float *__restrict__ out) {
  const int C = ca + cb;
  const int64_t total = static_cast<int64_t>(M) * C;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int row = static_cast<int>(t / C), p = static_cast<int>(t - static_cast<int64_t>(row) * C);
    const float *src = p < ca ? a + p : b + (p - ca);
    const int pitch = p < ca ? ca : cb;
    const int32_t *r = rules + static_cast<int64_t>(row) * (max_active + 1);
    const int cnt = r[0];
    const float m = cnt > 0 ? __fdiv_rn(1.0f, static_cast<float>(cnt)) : 1.0f;
    float acc = 0.0f;
    int i = 1;
    for (; i + 3 <= cnt; i += 4) {
      const int a0 = r[i], a1 = r[i + 1], a2 = r[i + 2], a3 = r[i + 3];
      const float f0 = src[static_cast<int64_t>(a0) * pitch], f1 = src[static_cast<int64_t>(a1) * pitch],
                  f2 = src[static_cast<int64_t>(a2) * pitch], f3 = src[static_cast<int64_t>(a3) * pitch];
      acc = __fadd_rn(acc, __fmul_rn(m, f0));
      acc = __fadd_rn(acc, __fmul_rn(m, f1));
      acc = __fadd_rn(acc, __fmul_rn(m, f2));
      acc = __fadd_rn(acc, __fmul_rn(m, f3));
    }
    for (; i <= cnt; ++i) acc = __fadd_rn(acc, __fmul_rn(m, src[static_cast<int64_t>(r[i]) * pitch]));
    out[t] = acc;
  }
}

__global__ void __launch_bounds__(256) narrow_i64_kernel(const int64_t *__restrict__ in, int64_t n,
                                                        int32_t *__restrict__ out) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) out[i] = static_cast<int32_t>(in[i]);
}

// ---- row softmax with the arithmetic of torch's softmax_warp_forward (aten/src/ATen/native/cuda/
//      PersistentSoftmax.cuh) for rows of <= 32 columns: one 32-lane group per row there, one thread per
//      row here -- the maximum is order-independent, exp(x - max) is the same expf, and the sum is
//      replayed as the 32-lane xor butterfly (offsets 16, 8, 4, 2, 1; absent columns contribute
//      exp(-inf) = 0), whose result is the same in every lane because fp32 addition commutes.
template <int COLS_MAX>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float *__restrict__ x, int64_t rows, int cols,
                                                          float *__restrict__ out) {
  const int64_t r = blockIdx.x * 256LL + threadIdx.x;
  if (r >= rows) return;
  float e[32];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < COLS_MAX; ++j) {
    e[j] = j < cols ? x[r * cols + j] : -INFINITY;
    mx = fmaxf(mx, e[j]);        // (NaN rows: torch's Max functor is `a < b ? b : a`; scores are finite here)
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) e[j] = j < COLS_MAX ? expf(e[j] - mx) : 0.f;
  float s[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) s[j] = e[j];
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    float t[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) t[j] = __fadd_rn(s[j], s[j ^ off]);
#pragma unroll
    for (int j = 0; j < 32; ++j) s[j] = t[j];
  }
  const float sum = s[0];
#pragma unroll
  for (int j = 0; j < COLS_MAX; ++j)
    if (j < cols) out[r * cols + j] = __fdiv_rn(e[j], sum);
}

// ---- out[r, :] = mlp(feats[idx[r], :])
template <int C>
__global__ void __launch_bounds__(256) mlp_rows_kernel(const float *__restrict__ feats, const int32_t *__restrict__ idx,
                                                      int64_t rows, Mlp2 m, float *__restrict__ out) {
  const int64_t r = blockIdx.x * 256LL + threadIdx.x;
  if (r >= rows) return;
  const int64_t row = idx ? static_cast<int64_t>(idx[r]) : r;
  float x[C];
  const float4 *src = reinterpret_cast<const float4 *>(feats + row * C);
#pragma unroll
  for (int q = 0; q < C / 4; ++q) {
    const float4 v = src[q];
    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
  }
  float y[32];
  mlp2<C, 32>(x, m, y);
#pragma unroll
  for (int o = 0; o < 32; ++o)
    if (o < m.out) out[r * m.out + o] = y[o];
}

__global__ void __launch_bounds__(256) linear_rows_kernel(const float *__restrict__ x, int64_t rows,
                                                         const float *__restrict__ w, const float *__restrict__ b,
                                                         int n_out, int n_in, float *__restrict__ out) {
  const int64_t total = rows * n_out;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t r = t / n_out;
    const int j = static_cast<int>(t - r * n_out);
    float a = b ? b[j] : 0.f;
    for (int c = 0; c < n_in; ++c) a = fmaf(x[r * n_in + c], w[static_cast<int64_t>(j) * n_in + c], a);
    out[t] = a;
  }
}

// ---- gt_instances (softgroup.py:641-653): sem * 1000 + inst + 1 with sem = max(label - shift + 1, 0),
//      0 where the instance label is negative (ignore)
__global__ void __launch_bounds__(256) gt_instances_kernel(const int64_t *__restrict__ sem, const int64_t *__restrict__ inst,
                                                          int n, int64_t shift, int64_t *__restrict__ out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int64_t s = sem[i] - shift + 1;
    if (s < 0) s = 0;
    const int64_t k = inst[i] + 1;
    out[i] = k < 0 ? 0 : s * 1000 + k;
  }
}

// ---- the dense results packed into one block (256-byte aligned segments): one launch, 16-byte moves
//      where source, destination and size allow, bytes otherwise
struct PackList {
  const char *src[SG_SCAN_DENSE_MAX];
  size_t dst[SG_SCAN_DENSE_MAX], bytes[SG_SCAN_DENSE_MAX];
  int n;
};
__global__ void __launch_bounds__(256) pack_segments_kernel(PackList p, char *__restrict__ out) {
  const int s = blockIdx.y;
  const char *src = p.src[s];
  char *dst = out + p.dst[s];
  const size_t nb = p.bytes[s];
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {        // (dst offsets are multiples of 256)
    const size_t q = nb / 16;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < q; i += gridDim.x * 256ull) d4[i] = s4[i];
    if (blockIdx.x == 0)
      for (size_t i = q * 16 + threadIdx.x; i < nb; i += 256) dst[i] = src[i];
  } else {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < nb; i += gridDim.x * 256ull) dst[i] = src[i];
  }
}

// ---- device -> pinned host copy of the dense results as a SMALL kernel (SG_SCAN_D2H_WGS workgroups; 0 = hipMemcpyAsync).
// The runtime's copy of 12 MB is a shader blit with one workgroup on every CU (__amd_rocclr_copyBuffer, 219 us =
// the PCIe rate); whatever runs next to it is stretched to its end (bfs_seed_kernel 5 -> 130 us, select_scan_kernel
// 20 -> 229 us in round 5's traces): its stalled system-memory stores sit in every CU's memory queue.  A few workgroups
// move the same bytes at the same PCIe rate and leave the other CUs' queues alone.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) d2h_copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull)
    __builtin_nontemporal_store(src[i], dst + i);
}

// ---- per (device, caller stream): the side stream of the dense results' copy and its two events
struct ScanStream {
  hipStream_t copy = nullptr;
  hipEvent_t packed = nullptr, copied = nullptr, backbone_done = nullptr;
};
// Backbone token: with several scans in flight on one
// device (one host thread and stream each) only ONE of them is inside its backbone -- the part of a scan
// whose kernels fill the whole chip and gain nothing from running next to another backbone -- while the
// others' grouping / refinement stages (small grids, host read-backs) run in its shadow.  Scans that
// start together no longer move through their phases in lock step.
// Form 2 (SG_SCAN_TOKEN=2) orders the backbones on the GPU instead of on the host: the convolutions of a scan
// wait (hipStreamWaitEvent) for the event behind the previous scan's backbone, whichever stream that ran on; the
// mutex is held only while the backbone is being enqueued, nobody blocks on the device.
static std::mutex g_backbone_mu[16];
static hipEvent_t g_backbone_last[16];        // event behind the latest backbone enqueued on the device (under the mutex)
// Form 4 (SG_SCAN_TOKEN=4): like form 1 with SG_SCAN_TOKEN_PERMITS (default 2) backbones at a time -- a counting
// semaphore instead of the mutex.
struct BackboneSem {
  std::mutex m;
  std::condition_variable cv;
  int in_use = 0;
};
static BackboneSem g_backbone_sem[16];
static int backbone_permits() {
  static const int p = getenv("SG_SCAN_TOKEN_PERMITS") ? atoi(getenv("SG_SCAN_TOKEN_PERMITS")) : 2;
  return p < 1 ? 1 : p;
}
struct BackboneToken {
  int dev;
  hipStream_t stream;
  std::unique_lock<std::mutex> lock;
  bool taken = false;
  bool sem = false;      // form 4: a permit of g_backbone_sem instead of the mutex
};
static void backbone_token_take(void *ctx) {      // (sg_unet_forward's pre-conv hook)
  BackboneToken *t = static_cast<BackboneToken *>(ctx);
  if (t->sem) {
    BackboneSem &s = g_backbone_sem[t->dev];
    std::unique_lock<std::mutex> g(s.m);
    s.cv.wait(g, [&] { return s.in_use < backbone_permits(); });
    ++s.in_use;
    t->taken = true;
    return;
  }
  t->lock = std::unique_lock<std::mutex>(g_backbone_mu[t->dev]);
  t->taken = true;
  if (g_backbone_last[t->dev] != nullptr) hipStreamWaitEvent(t->stream, g_backbone_last[t->dev], 0);
}
static void backbone_token_give(BackboneToken *t) {
  if (!t->taken) return;
  if (t->sem) {
    BackboneSem &s = g_backbone_sem[t->dev];
    {
      std::lock_guard<std::mutex> g(s.m);
      --s.in_use;
    }
    s.cv.notify_one();
  } else {
    t->lock.unlock();
  }
  t->taken = false;
}
static std::mutex g_scan_mu;
static std::map<std::pair<int, hipStream_t>, ScanStream> g_scan_streams;

static ScanStream *scan_stream(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> g(g_scan_mu);
  ScanStream &s = g_scan_streams[{dev, stream}];
  if (s.copy == nullptr) {
    if (hipStreamCreateWithFlags(&s.copy, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s.packed, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.backbone_done, hipEventDisableTiming) != hipSuccess) {
      s = ScanStream();
      return nullptr;
    }
  }
  return &s;
}

void scan_release_stream(int dev, hipStream_t stream) {
  std::lock_guard<std::mutex> g(g_scan_mu);
  auto it = g_scan_streams.find({dev, stream});
  if (it == g_scan_streams.end()) return;
  if (it->second.copy) {
    hipStreamSynchronize(it->second.copy);
    hipStreamDestroy(it->second.copy);
    hipEventDestroy(it->second.packed);
    hipEventDestroy(it->second.copied);
    if (dev >= 0 && dev < 16) {
      std::lock_guard<std::mutex> b(g_backbone_mu[dev]);
      if (g_backbone_last[dev] == it->second.backbone_done) g_backbone_last[dev] = nullptr;
    }
    hipEventDestroy(it->second.backbone_done);
  }
  g_scan_streams.erase(it);
}

struct DenseCopy {       // what the emission hook needs to start the copy
  ScanStream *ss;
  hipStream_t main;
  const void *dev_block;
  void *host_block;
  size_t bytes;
  bool issued, failed;
};
static void start_dense_copy(void *ctx) {
  DenseCopy *d = static_cast<DenseCopy *>(ctx);
  if (d->issued || d->bytes == 0) return;
  d->issued = true;
  // the copy waits for the point where the caller's stream is NOW (the packed block is long complete)
  static const int d2h_wgs = getenv("SG_SCAN_D2H_WGS") ? atoi(getenv("SG_SCAN_D2H_WGS")) : 0;
  // (SG_SCAN_COPY_STREAM=0, developer A/B: the copy on the caller's stream, in line, instead of the side stream)
  static const bool side = !(getenv("SG_SCAN_COPY_STREAM") && atoi(getenv("SG_SCAN_COPY_STREAM")) == 0);
  hipStream_t cs = side ? d->ss->copy : d->main;
  if (side && (hipEventRecord(d->ss->packed, d->main) != hipSuccess ||
               hipStreamWaitEvent(d->ss->copy, d->ss->packed, 0) != hipSuccess)) {
    d->failed = true;
    return;
  }
  if (d2h_wgs > 0 && d->bytes % 16 == 0) {
    d2h_copy_kernel<<<d2h_wgs, 256, 0, cs>>>(static_cast<const u32x4 *>(d->dev_block),
                                             static_cast<u32x4 *>(d->host_block), d->bytes / 16);
    if (hipGetLastError() != hipSuccess) d->failed = true;
  } else if (hipMemcpyAsync(d->host_block, d->dev_block, d->bytes, hipMemcpyDeviceToHost, cs) != hipSuccess) {
    d->failed = true;
  }
  if (hipEventRecord(d->ss->copied, cs) != hipSuccess) d->failed = true;
}

static Mlp2 as_mlp(const sg_mlp2 &m) { return Mlp2{m.w1, m.b1, m.bn_scale, m.bn_shift, m.w2, m.b2, m.out}; }

// bump allocator that keeps counting past the end (what a failed call needed so far)
struct Carve {
  char *base;
  size_t cap, off;
  Carve(void *p, size_t n) : base(static_cast<char *>(p)), cap(n), off(0) {}
  template <typename T>
  T *take(size_t count) {
    const size_t bytes = align_up((count ? count : 1) * sizeof(T));
    const size_t at = off;
    off += bytes;
    return off > cap ? nullptr : reinterpret_cast<T *>(base + at);
  }
  size_t at(const void *p) const { return static_cast<size_t>(static_cast<const char *>(p) - base); }
};

}  // namespace sg

using namespace sg;

extern "C" {

int sg_softmax_rows(const float *x, int64_t rows, int cols, float *out, sg_stream_t stream) {
  SG_REQUIRE(rows >= 0 && cols >= 1 && cols <= 32 && x && out, "sg_softmax_rows: rows of 1..32 columns (got %d)", cols);
  if (rows == 0) return SG_OK;
  const int grid = static_cast<int>((rows + 255) / 256);
  if (cols <= 4) softmax_rows_kernel<4><<<grid, 256, 0, as_stream(stream)>>>(x, rows, cols, out);
  else if (cols <= 8) softmax_rows_kernel<8><<<grid, 256, 0, as_stream(stream)>>>(x, rows, cols, out);
  else if (cols <= 16) softmax_rows_kernel<16><<<grid, 256, 0, as_stream(stream)>>>(x, rows, cols, out);
  else softmax_rows_kernel<32><<<grid, 256, 0, as_stream(stream)>>>(x, rows, cols, out);
  return check_launch("sg_softmax_rows");
}

int sg_mlp_rows(const float *feats, const int32_t *idx, int64_t rows, int channels, const sg_mlp2 *mlp, float *out,
                sg_stream_t stream) {
  SG_REQUIRE(rows >= 0 && mlp && feats && out, "sg_mlp_rows: bad arguments");
  SG_REQUIRE(channels == 16 || channels == 32, "sg_mlp_rows: channels must be 16 or 32 (got %d)", channels);
  SG_REQUIRE(mlp->out >= 1 && mlp->out <= 32, "sg_mlp_rows: head width %d out of range", mlp->out);
  if (rows == 0) return SG_OK;
  const int grid = static_cast<int>((rows + 255) / 256);
  if (channels == 32) mlp_rows_kernel<32><<<grid, 256, 0, as_stream(stream)>>>(feats, idx, rows, as_mlp(*mlp), out);
  else mlp_rows_kernel<16><<<grid, 256, 0, as_stream(stream)>>>(feats, idx, rows, as_mlp(*mlp), out);
  return check_launch("sg_mlp_rows");
}

int sg_linear_rows(const float *x, int64_t rows, const sg_linear *lin, float *out, sg_stream_t stream) {
  SG_REQUIRE(rows >= 0 && lin && lin->w && lin->out >= 1 && lin->in >= 1 && x && out, "sg_linear_rows: bad arguments");
  if (rows == 0) return SG_OK;
  linear_rows_kernel<<<grid_for(rows * lin->out, 256), 256, 0, as_stream(stream)>>>(x, rows, lin->w, lin->b, lin->out,
                                                                                  lin->in, out);
  return check_launch("sg_linear_rows");
}

// persistent part + the largest scratch user (the backbone's executor arena) + room for the later stages
size_t sg_scan_arena_bytes(const sg_scan_desc *d, int n_points, int n_voxels) {
  if (d == nullptr || d->backbone == nullptr) return 0;
  const size_t N = static_cast<size_t>(n_points > 0 ? n_points : 1), M = static_cast<size_t>(n_voxels > 0 ? n_voxels : 1);
  const size_t C = static_cast<size_t>(d->channels), ns = static_cast<size_t>(d->semantic_classes);
  size_t p = align_up(M * 16) + align_up(M * 8 * 4) + align_up(M * C * 4) + align_up(N * C * 4) + 2 * align_up(N * ns * 4) +
             align_up(N * 12) + 2 * align_up(N * 8) + align_up(N * 96) + (1 << 20);
  size_t bb = sg_unet_arena_bytes(d->backbone, n_voxels) / 4;      // (what UNetExecutor tries first, too)
  if (bb < (64u << 20)) bb = 64u << 20;
  const size_t later = (160u << 20) + N * 256;      // grouping + refinement + instances on an ordinary scene
  return p + (bb > later ? bb : later) + (8 << 20);
}

int sg_scan_forward(const sg_scan_desc *d, const sg_scan_input *in, void *arena, size_t arena_bytes, void *host_dense,
                    size_t host_dense_bytes, void *host_inst, size_t host_inst_bytes, sg_scan_result *res,
                    sg_stream_t stream_) {
  static const char *kWhat = "sg_scan_forward";
  SG_REQUIRE(d && in && res && d->backbone, "sg_scan_forward: null descriptor");
  SG_REQUIRE(!d->want_instances || d->tiny, "sg_scan_forward: instances need the tiny U-Net's descriptor");
  SG_REQUIRE(d->channels == 16 || d->channels == 32, "sg_scan_forward: backbone channels must be 16 or 32");
  SG_REQUIRE(d->semantic.out == d->semantic_classes && d->semantic_classes >= 1 && d->semantic_classes <= 32,
             "sg_scan_forward: 1..32 semantic classes");
  SG_REQUIRE(in->n_points >= 0 && in->n_voxels >= 0 && in->max_active >= 0 && in->batch_size >= 1 && in->feat_dim >= 1,
             "sg_scan_forward: bad input sizes");
  SG_REQUIRE(in->n_dense >= 0 && in->n_dense <= SG_SCAN_DENSE_MAX, "sg_scan_forward: at most %d dense items",
             SG_SCAN_DENSE_MAX);
  memset(res, 0, sizeof(*res));
  hipStream_t stream = as_stream(stream_);
  const int N = in->n_points, M = in->n_voxels, C = d->channels, ns = d->semantic_classes;
  const int cin = in->feat_dim + (d->with_coords ? 3 : 0);
  SG_REQUIRE(d->backbone->input_w == nullptr ? cin == C : cin == d->backbone->input_cin,
             "sg_scan_forward: %d feature columns, the backbone takes %d", cin,
             d->backbone->input_w ? d->backbone->input_cin : C);
  if (N == 0 || M == 0) return SG_OK;
  ScanStream *ss = scan_stream(stream);
  SG_REQUIRE(ss != nullptr, "sg_scan_forward: side stream / event creation failed");
  Carve ar(arena, arena_bytes);
#define SG_CARVE(var, T, count)                                                                          \
  T *var = ar.take<T>(count);                                                                            \
  if (var == nullptr) {                                                                                  \
    res->arena_needed = ar.off + ar.off / 4 + (64u << 20);                                               \
    set_error("%s: arena too small (%zu bytes, need about %zu)", kWhat, ar.cap, res->arena_needed);      \
    return SG_ERR_WORKSPACE;                                                                             \
  }

  // ---- persistent part
  const int32_t *vc32 = static_cast<const int32_t *>(in->voxel_coords);
  if (in->voxel_coords_is_int64) {
    SG_CARVE(vc, int32_t, static_cast<size_t>(M) * 4);
    narrow_i64_kernel<<<grid_for(static_cast<int64_t>(M) * 4, 256), 256, 0, stream>>>(
        static_cast<const int64_t *>(in->voxel_coords), static_cast<int64_t>(M) * 4, vc);
    vc32 = vc;
  }
  SG_CARVE(vfeat, float, static_cast<size_t>(M) * cin);
  SG_CARVE(bb_out, float, static_cast<size_t>(M) * C);
  SG_CARVE(out_feats, float, static_cast<size_t>(N) * C);
  SG_CARVE(sem, float, static_cast<size_t>(N) * ns);
  SG_CARVE(prob, float, static_cast<size_t>(N) * ns);
  SG_CARVE(off, float, static_cast<size_t>(N) * 3);
  SG_CARVE(preds, int64_t, N);
  // dense results: layout of the host block
  PackList pl;
  pl.n = 0;
  size_t dense_total = 0;
  bool want_gt = false;
  for (int i = 0; i < in->n_dense; ++i) {
    const sg_scan_dense_item &it = in->dense[i];
    size_t nb = 0;
    switch (it.kind) {
      case 0: nb = it.bytes; SG_REQUIRE(it.ptr != nullptr || nb == 0, "sg_scan_forward: dense item %d has no source", i); break;
      case 1: nb = static_cast<size_t>(N) * 8; break;
      case 2: nb = static_cast<size_t>(N) * 12; break;
      case 3:
        nb = static_cast<size_t>(N) * 8;
        want_gt = true;
        SG_REQUIRE(in->semantic_labels && in->instance_labels, "sg_scan_forward: gt_instances need the label arrays");
        break;
      case 4: nb = static_cast<size_t>(N) * ns * 4; break;
      default: SG_REQUIRE(false, "sg_scan_forward: dense item %d: unknown kind %d", i, it.kind);
    }
    res->dense_offset[i] = dense_total;
    pl.dst[i] = dense_total;
    pl.bytes[i] = nb;
    dense_total += align_up(nb);
  }
  pl.n = in->n_dense;
  res->dense_bytes = dense_total;
  res->host_dense_needed = dense_total;
  if (dense_total > 0 && (host_dense == nullptr || host_dense_bytes < dense_total)) {
    set_error("%s: host block for the dense results too small (%zu bytes, need %zu)", kWhat, host_dense_bytes, dense_total);
    return SG_ERR_WORKSPACE;
  }
  int64_t *gt = nullptr;
  if (want_gt) {
    SG_CARVE(gt_, int64_t, N);
    gt = gt_;
  }
  SG_CARVE(dense_dev, char, dense_total);
  for (int i = 0; i < in->n_dense; ++i) {
    switch (in->dense[i].kind) {
      case 0: pl.src[i] = static_cast<const char *>(in->dense[i].ptr); break;
      case 1: pl.src[i] = reinterpret_cast<const char *>(preds); break;
      case 2: pl.src[i] = reinterpret_cast<const char *>(off); break;
      case 3: pl.src[i] = reinterpret_cast<const char *>(gt); break;
      default: pl.src[i] = reinterpret_cast<const char *>(sem); break;
    }
  }
  res->voxel_feats_in = ar.at(vfeat);
  res->backbone_out = ar.at(bb_out);
  res->output_feats = ar.at(out_feats);
  res->semantic_scores = ar.at(sem);
  res->semantic_prob = ar.at(prob);
  res->pt_offsets = ar.at(off);
  res->semantic_preds = ar.at(preds);

  // ---- 1. voxel feature pooling, backbone, point-wise heads, softmax
  // (default: form 4, two backbones at a time -- 2.66-2.83 ms/scan, median 2.70, against 2.74-3.04, median 2.81,
  //  without a token over six interleaved runs of the bench's default region; 3.28-3.54 against 3.33-3.49 at 20 steps;
  //  three permits: no gain; one permit, form 1: 2.70-2.85 but outliers of 4.1 / 4.6 at 20 steps.
  //  profiles/r06_token_matrix.txt.  SG_SCAN_TOKEN=0: off)
  static const int token_env = getenv("SG_SCAN_TOKEN") ? atoi(getenv("SG_SCAN_TOKEN")) : 4;
  int token_mode = 0;
  int dev = 0;
  if (token_env != 0 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) {
    std::lock_guard<std::mutex> g(g_scan_mu);
    int n = 0;
    for (const auto &kv : g_scan_streams) n += kv.first.first == dev;
    if (n > 1) token_mode = token_env;      // (a single caller stream: nothing to order)
  }
  const bool use_token = token_mode == 3;       // (3: host-held from before the index build, the first form measured)
  std::unique_lock<std::mutex> token;
  if (use_token) token = std::unique_lock<std::mutex>(g_backbone_mu[dev]);
  BackboneToken bt{dev, stream, {}, false, token_mode == 4};
  struct TokenHookScope {      // (the hook is cleared by sg_unet_forward when it runs; here for the error paths before that)
    BackboneToken &t;
    ~TokenHookScope() {
      t_unet_conv_hook = nullptr;
      t_unet_conv_ctx = nullptr;
      if (t.sem) backbone_token_give(&t);      // (an error return between the hook and the hand-back below)
    }
  } token_hook_scope{bt};
  if (token_mode == 1 || token_mode == 2 || token_mode == 4) {
    t_unet_conv_hook = backbone_token_take;
    t_unet_conv_ctx = &bt;
  }
  voxelize_cat_kernel<<<grid_for(static_cast<int64_t>(M) * cin, 256), 256, 0, stream>>>(
      in->feats, in->feat_dim, d->with_coords ? in->coords_float : nullptr, d->with_coords ? 3 : 0, in->p2v_map, M,
      in->max_active, vfeat);
  SG_TRY_(check_launch(kWhat));
  {
    // the executor takes what it needs from the rest of the arena (sg_unet_arena_bytes prices every level
    // with all M voxels -- an order of magnitude above what a scene uses: only asked for after a failure)
    const size_t room = arena_bytes > ar.off ? arena_bytes - ar.off : 0;
    const int rc = sg_unet_forward(d->backbone, vfeat, vc32, M, in->spatial_shape, bb_out,
                                   static_cast<char *>(arena) + ar.off, room, stream_);
    if (rc == SG_ERR_WORKSPACE) res->arena_needed = ar.off + sg_unet_arena_bytes(d->backbone, M) + (64u << 20);
    SG_TRY_(rc);
    // (stream order: whatever is carved from here next is written after the backbone has run)
  }
  SG_TRY_(sg_pointwise_heads(bb_out, in->v2p_map, in->v2p_is_int64, N, C, &d->semantic, &d->offset, out_feats, sem, off,
                             preds, stream_));
  if (want_gt)
    gt_instances_kernel<<<grid_for(N, 256, 1024), 256, 0, stream>>>(in->semantic_labels, in->instance_labels, N,
                                                                  static_cast<int64_t>(d->semantic_classes - d->instance_classes), gt);
  if (pl.n > 0) {
    size_t mx = 0;
    for (int i = 0; i < pl.n; ++i) mx = pl.bytes[i] > mx ? pl.bytes[i] : mx;
    pack_segments_kernel<<<dim3(grid_for(static_cast<int64_t>(mx / 16 + 1), 256, 512), pl.n), 256, 0, stream>>>(pl, dense_dev);
  }
  SG_TRY_(check_launch(kWhat));
  if (bt.taken) {       // behind this point of the stream the next scan's convolutions may run
    if (hipEventRecord(ss->backbone_done, stream) == hipSuccess) {
      if (!bt.sem) g_backbone_last[dev] = ss->backbone_done;
      if (token_mode == 1 || token_mode == 4) hipEventSynchronize(ss->backbone_done);      // forms 1, 4: the host holds the token until then
    }
    backbone_token_give(&bt);
  }
  DenseCopy dc{ss, stream, dense_dev, host_dense, dense_total, false, false};
  struct DenseGuard {      // no return leaves a copy into the caller's host block in flight
    DenseCopy &d;
    ~DenseGuard() {
      if (d.issued && !d.failed && d.bytes > 0) hipEventSynchronize(d.ss->copied);
    }
  } dense_guard{dc};
  res->stage = 1;
  auto finish_dense = [&]() -> int {       // the copy has been started (now at the latest) and has landed
    start_dense_copy(&dc);
    if (dc.failed || (dc.bytes > 0 && hipEventSynchronize(ss->copied) != hipSuccess)) {
      set_error("%s: copy of the dense results failed", kWhat);
      return SG_ERR_LAUNCH;
    }
    return SG_OK;
  };
  if (!d->want_instances) {
    res->arena_used = ar.off;
    SG_TRY_(finish_dense());
    if (hipStreamSynchronize(stream) != hipSuccess) return SG_ERR_LAUNCH;
    return SG_OK;
  }
  SG_TRY_(sg_softmax_rows(sem, N, ns, prob, stream_));
  if (use_token) {      // the next scan's backbone may start when this one's has left the GPU
    if (hipEventRecord(ss->backbone_done, stream) != hipSuccess || hipEventSynchronize(ss->backbone_done) != hipSuccess) {
      set_error("%s: waiting for the backbone failed", kWhat);
      return SG_ERR_LAUNCH;
    }
    token.unlock();
  }

  // ---- 2. grouping head + proposal voxelisation (its sub-arena starts at the scratch mark)
  sg_grouping_cfg gc = d->grouping;
  gc.n_points = N;
  gc.n_sem_classes = ns;
  gc.batch_size = in->batch_size;
  gc.feat_channels = C;
  res->grouping_base = ar.off;
  {
    struct HookScope {
      HookScope(void (*f)(void *), void *c) { t_scan_emit_hook = f; t_scan_emit_ctx = c; }
      ~HookScope() { t_scan_emit_hook = nullptr; t_scan_emit_ctx = nullptr; }
    } hook(start_dense_copy, &dc);
    const size_t room = arena_bytes > ar.off ? arena_bytes - ar.off : 0;
    int rc;
    if (d->with_pyramid || d->with_octree) {      // SoftGroup++: the per-class loop inside its own C call
      sg_grouping_pp_cfg pc;
      memset(&pc, 0, sizeof(pc));
      pc.base = gc;
      pc.with_pyramid = d->with_pyramid;
      pc.with_octree = d->with_octree;
      pc.lvl_fusion = 0;
      pc.radius = d->pp_radius;
      pc.base_size = d->pp_base_size;
      rc = sg_scan_grouping_pp(&pc, prob, off, in->coords_float, in->batch_idxs, out_feats,
                               static_cast<char *>(arena) + ar.off, room, &res->grouping, stream_);
    } else {
      rc = sg_scan_grouping(&gc, prob, off, in->coords_float, in->batch_idxs, out_feats,
                            static_cast<char *>(arena) + ar.off, room, &res->grouping, stream_);
    }
    if (rc == SG_ERR_WORKSPACE) res->arena_needed = ar.off + res->grouping.arena_needed + (64u << 20);
    if (rc != SG_OK) return rc;      // (the guard waits for a copy the hook may have started)
  }
  res->stage = 2;
  const sg_grouping_result &g = res->grouping;
  if (g.sum_npoint == 0) {       // nothing selected / no proposal: the dense results are the scan's output
    res->arena_used = ar.off;
    SG_TRY_(finish_dense());
    if (hipStreamSynchronize(stream) != hipSuccess) return SG_ERR_LAUNCH;
    return SG_OK;
  }
  char *gbase = static_cast<char *>(arena) + res->grouping_base;
  ar.off += align_up(g.arena_used);
  const int nP = g.n_proposals, Mv = g.n_voxels;
  const int64_t S = g.sum_npoint;
  const int32_t *pairs = reinterpret_cast<const int32_t *>(gbase + g.proposals_idx);
  const int32_t *pvc = reinterpret_cast<const int32_t *>(gbase + g.voxel_coords);
  const int32_t *pvoff = reinterpret_cast<const int32_t *>(gbase + g.voxel_offsets);
  const float *pvfeat = reinterpret_cast<const float *>(gbase + g.voxel_feats);
  const int32_t *p2v = reinterpret_cast<const int32_t *>(gbase + g.point_to_voxel);

  // ---- 3. refinement: tiny U-Net, mask head on the proposal points, pooled class / IoU heads
  const int ncol = d->instance_classes + 1;
  SG_REQUIRE(d->mask.out == ncol && d->cls.out == ncol && d->iou.out == ncol && d->cls.in == C && d->iou.in == C,
             "sg_scan_forward: refinement heads must have instance_classes + 1 = %d outputs", ncol);
  SG_CARVE(tiny_out, float, static_cast<size_t>(Mv) * C);
  SG_CARVE(mask_scores, float, static_cast<size_t>(S) * ncol);
  SG_CARVE(pooled, float, static_cast<size_t>(nP) * C);
  SG_CARVE(cls, float, static_cast<size_t>(nP) * ncol);
  SG_CARVE(cls_prob, float, static_cast<size_t>(nP) * ncol);
  SG_CARVE(iou, float, static_cast<size_t>(nP) * ncol);
  {
    const size_t room = arena_bytes > ar.off ? arena_bytes - ar.off : 0;
    const int32_t shape[3] = {d->grouping.voxel_shape, d->grouping.voxel_shape, d->grouping.voxel_shape};
    const int rc = sg_unet_forward(d->tiny, pvfeat, pvc, Mv, shape, tiny_out, static_cast<char *>(arena) + ar.off, room,
                                   stream_);
    if (rc == SG_ERR_WORKSPACE) res->arena_needed = ar.off + sg_unet_arena_bytes(d->tiny, Mv) + (64u << 20);
    if (rc != SG_OK) return rc;
  }
  int rc = sg_mlp_rows(tiny_out, p2v, S, C, &d->mask, mask_scores, stream_);
  if (rc == SG_OK) rc = sg_global_avg_pool_fp(tiny_out, pvoff, nP, C, pooled, stream_);
  if (rc == SG_OK) rc = sg_linear_rows(pooled, nP, &d->cls, cls, stream_);
  if (rc == SG_OK) rc = sg_linear_rows(pooled, nP, &d->iou, iou, stream_);
  if (rc == SG_OK) rc = sg_softmax_rows(cls, nP, ncol, cls_prob, stream_);
  if (rc != SG_OK) return rc;
  res->tiny_out = ar.at(tiny_out);
  res->mask_scores = ar.at(mask_scores);
  res->cls_scores = ar.at(cls);
  res->cls_prob = ar.at(cls_prob);
  res->iou_scores = ar.at(iou);
  res->stage = 3;

  // ---- 4. instances: kept table, bit rows, runs, RLE text -> host_inst
  sg_instances_cfg ic;
  ic.n_proposals = nP;
  ic.n_classes = d->instance_classes;
  ic.score_stride = ncol;
  ic.sum_npoint = S;
  ic.n_points = N;
  ic.cls_score_thr = d->cls_score_thr;
  ic.mask_score_thr = d->mask_score_thr;
  ic.min_npoint = d->min_npoint;
  res->instances_base = ar.off;
  {
    const size_t room = arena_bytes > ar.off ? arena_bytes - ar.off : 0;
    rc = sg_scan_instances(&ic, pairs, mask_scores, cls_prob, iou, static_cast<char *>(arena) + ar.off, room, host_inst,
                           host_inst_bytes, &res->instances, stream_);
    res->host_inst_needed = res->instances.host_needed;
    if (rc == SG_ERR_WORKSPACE && res->instances.arena_needed > room)
      res->arena_needed = ar.off + res->instances.arena_needed + (16u << 20);
    if (rc != SG_OK) return rc;
  }
  res->stage = 4;
  res->arena_used = ar.off + res->instances.arena_used;
  SG_TRY_(finish_dense());
  return SG_OK;      // (sg_scan_instances has synchronised the caller's stream)
#undef SG_CARVE
}

}  // extern "C"
