// unet_exec.hip -- native executor of the sparse U-Net (inference): one C call runs
//   [input SubMConv3d] -> UBlock (recursive: residual blocks, down conv, inner UBlock, inverse conv,
//   skip concat, tail blocks) -> [output BatchNorm1d + ReLU]
// i.e. softgroup/model/softgroup.py:60-65 (backbone) and :93-95 (tiny U-Net) with the modules of
// softgroup/model/blocks.py:44-143, on the rulebook / plan / conv entry points of this library.
// The Python module path (softgroup_amd/spconv + model/blocks.py) launches the same kernels in
// the same order; it stays the path for training.  Here nothing but kernel launches happens
// between two layers: no interpreter, no allocator calls (a caller-provided arena, stack
// discipline per level) and ONE host sync per forward: the row counts of all levels come from one
// pass over the finest coordinates (sg_spconv_level_rows) and are read back right at the start.
// Two streams: everything that depends only on voxel coordinates (rulebooks, plans) runs on an
// internal index stream and races ahead of the convolutions on the caller's stream, which wait
// per level on an event; index tables are never recycled inside a forward, so the only
// cross-stream hazards are the read-after-write ones the events cover.
// Runtime state (index stream, events, pinned read-back words) is kept per device; calls on one
// device are serialised by a mutex (one forward owns the device's index stream at a time).
#include <mutex>
#include <vector>

#include "common.h"

namespace sg {

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

__global__ void __launch_bounds__(256) iota_kernel(int32_t *out, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = i;
}

// bump allocator with stack discipline
struct Arena {
  char *base;
  size_t cap, off, peak;
  Arena(void *p, size_t n) : base(static_cast<char *>(p)), cap(n), off(0), peak(0) {}
  template <typename T>
  T *take(size_t count) {
    const size_t bytes = align_up(count * sizeof(T));
    if (off + bytes > cap) return nullptr;
    T *r = reinterpret_cast<T *>(base + off);
    off += bytes;
    if (off > peak) peak = off;
    return r;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

struct Plan {
  const int32_t *nbr = nullptr;
  int32_t *order = nullptr;
  uint32_t *tile_mask = nullptr;
  int32_t *nbr_tiles = nullptr;
  int rows = 0, kvol = 0;
};

#define SG_TRY(expr)              \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != SG_OK) return rc_; \
  } while (0)
#define SG_ALLOC(var, T, count)                                                        \
  T *var = ar.take<T>(count);                                                          \
  if (var == nullptr) {                                                                \
    set_error("sg_unet_forward: arena too small (%zu bytes, need more than %zu)", ar.cap, ar.off); \
    return SG_ERR_WORKSPACE;                                                           \
  }

struct Exec {
  const sg_unet_desc *d;
  Arena ar;             // features and conv scratch: caller's stream only
  Arena ix;             // index tables and their scratch: index stream only (bump, no recycling)
  sg_stream_t stream;   // caller's stream (convolutions, elementwise)
  sg_stream_t istream;  // index stream
  const int32_t *level_rows;   // host: rows of every level (known before anything is enqueued)
  hipEvent_t *events;
  int n_events, next_event;

  Exec(const sg_unet_desc *desc, void *arena, size_t bytes, size_t index_bytes, sg_stream_t s,
       sg_stream_t is)
      : d(desc), ar(static_cast<char *>(arena) + index_bytes, bytes - index_bytes),
        ix(arena, index_bytes), stream(s), istream(is), level_rows(nullptr), events(nullptr),
        n_events(0), next_event(0) {}

  // the caller's stream may not run past this point before the index stream got here
  int index_ready() {
    if (next_event >= n_events) {
      set_error("sg_unet_forward: out of events");
      return SG_ERR_LAUNCH;
    }
    hipEvent_t e = events[next_event++];
    if (hipEventRecord(e, as_stream(istream)) != hipSuccess ||
        hipStreamWaitEvent(as_stream(stream), e, 0) != hipSuccess) {
      set_error("sg_unet_forward: event record/wait failed");
      return SG_ERR_LAUNCH;
    }
    return SG_OK;
  }

#define SG_IALLOC(var, T, count)                                                       \
  T *var = ix.take<T>(count);                                                          \
  if (var == nullptr) {                                                                \
    set_error("sg_unet_forward: index arena too small (%zu bytes)", ix.cap);           \
    return SG_ERR_WORKSPACE;                                                           \
  }

  int make_plan(const int32_t *nbr, int rows, int kvol, Plan &p) {
    p.nbr = nbr; p.rows = rows; p.kvol = kvol;
    const size_t nt = (static_cast<size_t>(rows) + 31) / 32;
    SG_IALLOC(order, int32_t, nt * 32);
    SG_IALLOC(tmask, uint32_t, nt ? nt : 1);
    SG_IALLOC(ntiles, int32_t, nt * 32 * kvol);
    p.order = order; p.tile_mask = tmask; p.nbr_tiles = ntiles;
    if (rows == 0) return SG_OK;
    const size_t m = ix.mark();
    const size_t nb = sg_spconv_plan_workspace_bytes(rows);
    SG_IALLOC(ws, char, nb);
    SG_TRY(sg_spconv_plan(nbr, rows, kvol, order, tmask, ntiles, ws, nb, istream));
    ix.release(m);     // scratch only: index-stream order keeps it alive until the plan kernels are done
    return SG_OK;
  }

  // BatchNorm1d + ReLU of a consumer, applied by the producer: (scale, shift) and where the
  // activated copy goes
  struct Act {
    const float *scale = nullptr, *shift = nullptr;
    float *out = nullptr;
  };

  int conv(const float *in, int in_rows, const Plan &p, int cin, int cout, const float *w,
           const float *post_s, const float *post_b, const float *residual, const Act &act, float *out) {
    if (p.rows == 0) return SG_OK;
    const size_t m = ar.mark();
    const size_t nb = sg_spconv_conv_workspace_bytes(p.rows, cout);
    void *ws = nullptr;
    if (nb > 256) {
      ws = ar.take<char>(nb);   // optional: without it the conv simply does not split offsets
    }
    const int rc = sg_spconv_gather_conv_f32(in, in_rows, p.nbr, p.rows, p.kvol, cin, cout, w, post_s,
                                             post_b, residual, act.scale, act.shift, act.out, p.order,
                                             p.tile_mask, p.nbr_tiles, out, ws, ws ? nb : 0, stream);
    ar.release(m);
    return rc;
  }

  // ResidualBlock (blocks.py:44-79): x + SubM(ReLU(BN(SubM(ReLU(BN(x)))))), 1x1 conv on the
  // identity branch when the channel count changes.  `xa` = relu(bn1(x)), made by whoever produced
  // x; `next` = the activation the consumer of this block's output wants (second output of conv2).
  int block(const sg_unet_block &b, const float *x, const float *xa, int rows, const Plan &subm,
            const Plan &ident, const float *post_s, const float *post_b, const Act &next, float *out) {
    const size_t m = ar.mark();
    const float *shortcut = x;
    if (b.w_i != nullptr) {
      SG_ALLOC(sc, float, static_cast<size_t>(rows) * b.cout);
      SG_TRY(conv(x, rows, ident, b.cin, b.cout, b.w_i, nullptr, nullptr, nullptr, Act(), sc));
      shortcut = sc;
    }
    SG_ALLOC(h, float, static_cast<size_t>(rows) * b.cout);
    SG_TRY(conv(xa, rows, subm, b.cin, b.cout, b.w1, b.bn2_scale, b.bn2_shift, nullptr, Act(), h));
    SG_TRY(conv(h, rows, subm, b.cout, b.cout, b.w2, post_s, post_b, shortcut, next, out));
    ar.release(m);
    return SG_OK;
  }

  // UBlock (blocks.py:82-143).  `x` [rows, planes] -> `out` [rows, planes]; `xa` = relu(bn(x)) for
  // the first block's BatchNorm if the producer of x made it (else it is computed here); post =
  // BatchNorm+ReLU applied in place by the level's last conv (output_layer for the outermost
  // level, the parent's deconv BatchNorm for an inner one).
  int level(int l, const float *x, const float *xa, const int32_t *indices, int rows,
            const int32_t shape[3], const float *pre_in, int pre_cin, const float *post_s,
            const float *post_b, float *out) {
    const sg_unet_level &L = d->levels[l];
    const int c = L.planes;
    const bool deeper = l + 1 < d->n_levels;
    const size_t m0 = ar.mark();
    // ---- index stream: SubM rulebook + plan of this level (indice_key 'subm<l>', shared by all
    //      its blocks), identity table for the 1x1 convs of the tail
    SG_IALLOC(nbr, int32_t, static_cast<size_t>(rows ? rows : 1) * 27);
    if (rows) {
      const size_t m = ix.mark();
      const size_t nb = sg_spconv_hash_workspace_bytes(rows);
      SG_IALLOC(ws, char, nb);
      SG_TRY(sg_spconv_subm_rulebook(indices, rows, shape, nbr, ws, nb, istream));
      ix.release(m);
    }
    Plan subm;
    SG_TRY(make_plan(nbr, rows, 27, subm));
    Plan ident;
    if (deeper && rows) {
      SG_IALLOC(iota, int32_t, rows);
      iota_kernel<<<grid_for(rows, 256), 256, 0, as_stream(istream)>>>(iota, rows);
      ident.nbr = iota; ident.rows = rows; ident.kvol = 1;
    }
    SG_TRY(index_ready());
    const size_t feat = static_cast<size_t>(rows ? rows : 1) * c;
    // optional input conv (outermost level only): SubMConv3d(in, planes) on the same rulebook
    if (pre_in != nullptr) {
      SG_ALLOC(x0, float, feat);
      SG_ALLOC(x0a, float, feat);
      const Act a0{L.blocks[0].bn1_scale, L.blocks[0].bn1_shift, x0a};
      SG_TRY(conv(pre_in, rows, subm, pre_cin, c, d->input_w, nullptr, nullptr, nullptr, a0, x0));
      x = x0;
      xa = x0a;
    } else if (xa == nullptr) {
      SG_ALLOC(x0a, float, feat);
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
                   fin ? post_b : nullptr, next, dst));
      cur = dst;
      cur_a = next.out;
    }
    if (deeper) {
      const int c2 = d->levels[l + 1].planes;
      // ---- index stream: strided-conv pairs (the number of coarse voxels is already known on the
      //      host: level_rows), their plan, and the inverse table + plan the way back up will need
      SG_IALLOC(in2out, int32_t, rows ? rows : 1);
      SG_IALLOC(meta, int32_t, 64);
      const size_t nbh = sg_spconv_hash_workspace_bytes(rows);
      SG_IALLOC(hws, char, nbh);          // coordinate hash: built by down_build, read by down_fill
      const int rows2 = rows ? level_rows[l + 1] : 0;
      if (rows) SG_TRY(sg_spconv_down_build(indices, rows, shape, in2out, meta, hws, nbh, istream));
      SG_IALLOC(idx2, int32_t, static_cast<size_t>(rows2 ? rows2 : 1) * 4);
      SG_IALLOC(child, int32_t, static_cast<size_t>(rows2 ? rows2 : 1) * 8);
      if (rows) SG_TRY(sg_spconv_down_fill(indices, rows, in2out, rows2, idx2, child, hws, nbh, istream));
      Plan down;
      SG_TRY(make_plan(child, rows2, 8, down));
      SG_IALLOC(inv, int32_t, static_cast<size_t>(rows ? rows : 1) * 8);
      if (rows) SG_TRY(sg_spconv_inverse_rulebook(indices, in2out, rows, inv, istream));
      Plan up;
      SG_TRY(make_plan(inv, rows, 8, up));
      SG_TRY(index_ready());
      const int32_t shape2[3] = {shape[0] / 2, shape[1] / 2, shape[2] / 2};
      // ---- (BN -> ReLU done by the last block) -> SparseConv3d(c, c2, k2 s2); its second output
      //      feeds the first BatchNorm of the inner level
      const sg_unet_level &L2 = d->levels[l + 1];
      const size_t feat2 = static_cast<size_t>(rows2 ? rows2 : 1) * c2;
      SG_ALLOC(y, float, feat2);
      SG_ALLOC(ya, float, feat2);
      const Act ay{L2.blocks[0].bn1_scale, L2.blocks[0].bn1_shift, ya};
      SG_TRY(conv(cur_a, rows, down, c, c2, L.down_w, nullptr, nullptr, nullptr, ay, y));
      // ---- inner UBlock; its last conv applies this level's deconv BatchNorm + ReLU in place
      SG_ALLOC(z, float, feat2);
      SG_TRY(level(l + 1, y, ya, idx2, rows2, shape2, nullptr, 0, L.up_bn_scale, L.up_bn_shift, z));
      // ---- SparseInverseConv3d(c2, c): gather table = parent row per fine voxel (plan `up`, built
      //      on the index stream before the descent), then the skip concat (blocks.py:135-139)
      //      with the first tail block's BatchNorm + ReLU as a second output
      SG_ALLOC(cat, float, 2 * feat);
      SG_ALLOC(cata, float, 2 * feat);
      {
        const size_t m = ar.mark();
        SG_ALLOC(upf, float, feat);
        SG_TRY(conv(z, rows2, up, c2, c, L.up_w, nullptr, nullptr, nullptr, Act(), upf));
        if (rows)
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
                     last ? post_b : nullptr, next, dst));
        cur = dst;
        cur_a = next.out;
      }
    }
    ar.release(m0);
    return check_launch("sg_unet_forward");
  }
};

}  // namespace sg

using namespace sg;

extern "C" {

// index part of the arena: tables are never recycled inside a forward, so every level is priced
// with all `num_rows` voxels (they can only shrink): gather tables 27 + 8 + 8 ints per row, their
// plan copies (27 + 8 + 8), rows / orders / maps (~12), hash tables, plan scratch
static size_t unet_index_bytes(const sg_unet_desc *d, int num_rows) {
  const size_t rows = static_cast<size_t>(num_rows > 0 ? num_rows : 1);
  size_t total = 1 << 20;
  for (int l = 0; l < d->n_levels; ++l)
    total += rows * (2 * 43 + 16) * 4 + 2 * sg_spconv_hash_workspace_bytes(num_rows) + (256 << 10);
  return align_up(total + sg_spconv_plan_workspace_bytes(num_rows) +
                  sg_spconv_level_rows_workspace_bytes(num_rows, d->n_levels), 4096);
}

size_t sg_unet_arena_bytes(const sg_unet_desc *d, int num_rows) {
  // feature part: at most ~12 live buffers of 2*planes floats per level (stack discipline)
  size_t total = unet_index_bytes(d, num_rows) + (1 << 20);
  const size_t rows = static_cast<size_t>(num_rows > 0 ? num_rows : 1);
  for (int l = 0; l < d->n_levels; ++l)
    total += rows * 12 * 2 * static_cast<size_t>(d->levels[l].planes) * 4 + (64 << 10);
  return total;
}

int sg_unet_forward(const sg_unet_desc *d, const float *feats, const int32_t *indices, int num_rows,
                    const int32_t *spatial_shape_host, float *out, void *arena, size_t arena_bytes,
                    sg_stream_t stream) {
  SG_REQUIRE(d != nullptr && d->n_levels >= 1 && d->levels != nullptr, "sg_unet_forward: bad descriptor");
  SG_REQUIRE(num_rows >= 0, "sg_unet_forward: bad num_rows");
  for (int l = 0; l < d->n_levels; ++l)
    SG_REQUIRE(d->levels[l].planes % 4 == 0 && d->levels[l].n_blocks >= 1,
               "sg_unet_forward: level %d: planes must be a multiple of 4", l);
  if (num_rows == 0) return SG_OK;
  const size_t index_bytes = unet_index_bytes(d, num_rows);
  SG_REQUIRE(arena != nullptr && arena_bytes > index_bytes,
             "sg_unet_forward: arena of %zu bytes, need sg_unet_arena_bytes()", arena_bytes);
  // ---- per-device runtime state
  constexpr int kEvents = 64, kMaxDev = 64, kMaxLevels = 16;
  struct DeviceState {
    std::mutex mu;
    int32_t *host_rows = nullptr;     // pinned [kMaxLevels]
    int32_t *dev_rows = nullptr;      // device [kMaxLevels]
    hipStream_t istream = nullptr;
    hipEvent_t events[kEvents];
    bool ready = false;
  };
  static DeviceState states[kMaxDev];
  SG_REQUIRE(d->n_levels <= kMaxLevels, "sg_unet_forward: at most %d levels", kMaxLevels);
  int dev = 0;
  SG_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDev,
             "sg_unet_forward: no current device");
  DeviceState &st = states[dev];
  std::lock_guard<std::mutex> guard(st.mu);
  if (!st.ready) {
    SG_REQUIRE(hipHostMalloc(reinterpret_cast<void **>(&st.host_rows), kMaxLevels * 4) == hipSuccess,
               "sg_unet_forward: pinned allocation failed");
    SG_REQUIRE(hipMalloc(reinterpret_cast<void **>(&st.dev_rows), kMaxLevels * 4) == hipSuccess,
               "sg_unet_forward: device allocation failed");
    SG_REQUIRE(hipStreamCreateWithFlags(&st.istream, hipStreamNonBlocking) == hipSuccess,
               "sg_unet_forward: stream creation failed");
    for (int i = 0; i < kEvents; ++i)
      SG_REQUIRE(hipEventCreateWithFlags(&st.events[i], hipEventDisableTiming) == hipSuccess,
                 "sg_unet_forward: event creation failed");
    st.ready = true;
  }
  hipStream_t istream = st.istream;
  hipEvent_t *events = st.events;
  Exec ex(d, arena, arena_bytes, index_bytes, stream, reinterpret_cast<sg_stream_t>(istream));
  ex.events = events;
  ex.n_events = kEvents;
  // the index stream starts where the caller's stream is now: the coordinates are ready, and the
  // previous forward's convolutions no longer read the tables about to be overwritten
  if (hipEventRecord(events[0], as_stream(stream)) != hipSuccess ||
      hipStreamWaitEvent(istream, events[0], 0) != hipSuccess) {
    set_error("sg_unet_forward: event record/wait failed");
    return SG_ERR_LAUNCH;
  }
  ex.next_event = 1;
  // ---- rows of every level, one read-back (the only host sync of the forward; it waits for the
  //      index stream only -- whatever the caller's stream still has queued keeps running)
  int32_t level_rows[kMaxLevels];
  level_rows[0] = num_rows;
  if (d->n_levels > 1) {
    const size_t m = ex.ix.mark();
    const size_t nb = sg_spconv_level_rows_workspace_bytes(num_rows, d->n_levels);
    char *ws = ex.ix.take<char>(nb);
    SG_REQUIRE(ws != nullptr, "sg_unet_forward: index arena too small (%zu bytes)", ex.ix.cap);
    SG_TRY(sg_spconv_level_rows(indices, num_rows, spatial_shape_host, d->n_levels, st.dev_rows, ws, nb,
                                reinterpret_cast<sg_stream_t>(istream)));
    if (hipMemcpyAsync(st.host_rows, st.dev_rows, sizeof(int32_t) * d->n_levels, hipMemcpyDeviceToHost,
                       istream) != hipSuccess ||
        hipStreamSynchronize(istream) != hipSuccess) {
      set_error("sg_unet_forward: reading the level row counts failed");
      return SG_ERR_LAUNCH;
    }
    for (int l = 1; l < d->n_levels; ++l) level_rows[l] = st.host_rows[l];
    ex.ix.release(m);
  }
  ex.level_rows = level_rows;
  const bool pre = d->input_w != nullptr;
  return ex.level(0, pre ? nullptr : feats, nullptr, indices, num_rows, spatial_shape_host,
                  pre ? feats : nullptr, d->input_cin, d->out_bn_scale, d->out_bn_shift, out);
}

}  // extern "C"
