// unet_exec.hip -- native executor of the sparse U-Net (inference): one C call runs
//   [input SubMConv3d] -> UBlock (recursive: residual blocks, down conv, inner UBlock, inverse conv,
//   skip concat, tail blocks) -> [output BatchNorm1d + ReLU]
// i.e. softgroup/model/softgroup.py:60-65 (backbone) and :93-95 (tiny U-Net) with the modules of
// softgroup/model/blocks.py:44-143, on the rulebook / plan / conv entry points of this library.
// The Python module path (softgroup_amd/spconv + model/blocks.py) launches the same kernels in
// the same order; it stays the path for training.  Here nothing but kernel launches happens
// between two layers: no interpreter, no allocator calls (a caller-provided arena, stack
// discipline per level), one host sync per down-sampling (the number of coarse voxels).
#include <vector>

#include "common.h"

namespace sg {

__global__ void __launch_bounds__(256) concat2_kernel(const float4 *__restrict__ a,
                                                     const float4 *__restrict__ b, int64_t rows,
                                                     int ca4, int cb4, float4 *__restrict__ out) {
  const int c4 = ca4 + cb4;
  const int64_t total = rows * c4;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t r = t / c4;
    const int c = static_cast<int>(t - r * c4);
    out[t] = c < ca4 ? a[r * ca4 + c] : b[r * cb4 + (c - ca4)];
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
  Arena ar;
  sg_stream_t stream;
  int32_t *host_meta;   // pinned, 2 ints

  Exec(const sg_unet_desc *desc, void *arena, size_t bytes, sg_stream_t s)
      : d(desc), ar(arena, bytes), stream(s), host_meta(nullptr) {}

  int make_plan(const int32_t *nbr, int rows, int kvol, Plan &p) {
    p.nbr = nbr; p.rows = rows; p.kvol = kvol;
    const size_t nt = (static_cast<size_t>(rows) + 31) / 32;
    SG_ALLOC(order, int32_t, nt * 32);
    SG_ALLOC(tmask, uint32_t, nt ? nt : 1);
    SG_ALLOC(ntiles, int32_t, nt * 32 * kvol);
    p.order = order; p.tile_mask = tmask; p.nbr_tiles = ntiles;
    if (rows == 0) return SG_OK;
    const size_t m = ar.mark();
    const size_t nb = sg_spconv_plan_workspace_bytes(rows);
    SG_ALLOC(ws, char, nb);
    SG_TRY(sg_spconv_plan(nbr, rows, kvol, order, tmask, ntiles, ws, nb, stream));
    ar.release(m);     // stream order keeps the scratch alive until the plan kernels are done
    return SG_OK;
  }

  int conv(const float *in, int in_rows, const Plan &p, int cin, int cout, const float *w,
           const float *post_s, const float *post_b, const float *residual, float *out) {
    if (p.rows == 0) return SG_OK;
    const size_t m = ar.mark();
    const size_t nb = sg_spconv_conv_workspace_bytes(p.rows, cout);
    void *ws = nullptr;
    if (nb > 256) {
      ws = ar.take<char>(nb);   // optional: without it the conv simply does not split offsets
    }
    const int rc = sg_spconv_gather_conv_f32(in, in_rows, p.nbr, p.rows, p.kvol, cin, cout, w, post_s,
                                             post_b, residual, p.order, p.tile_mask, p.nbr_tiles, out,
                                             ws, ws ? nb : 0, stream);
    ar.release(m);
    return rc;
  }

  // ResidualBlock (blocks.py:44-79): x + SubM(ReLU(BN(SubM(ReLU(BN(x)))))), 1x1 conv on the
  // identity branch when the channel count changes
  int block(const sg_unet_block &b, const float *x, int rows, const Plan &subm, const Plan &ident,
            const float *post_s, const float *post_b, float *out) {
    const size_t m = ar.mark();
    const float *shortcut = x;
    if (b.w_i != nullptr) {
      SG_ALLOC(sc, float, static_cast<size_t>(rows) * b.cout);
      SG_TRY(conv(x, rows, ident, b.cin, b.cout, b.w_i, nullptr, nullptr, nullptr, sc));
      shortcut = sc;
    }
    SG_ALLOC(a, float, static_cast<size_t>(rows) * b.cin);
    SG_TRY(sg_bn_relu_f32(x, b.bn1_scale, b.bn1_shift, rows, b.cin, 1, a, stream));
    SG_ALLOC(h, float, static_cast<size_t>(rows) * b.cout);
    SG_TRY(conv(a, rows, subm, b.cin, b.cout, b.w1, b.bn2_scale, b.bn2_shift, nullptr, h));
    SG_TRY(conv(h, rows, subm, b.cout, b.cout, b.w2, post_s, post_b, shortcut, out));
    ar.release(m);
    return SG_OK;
  }

  // UBlock (blocks.py:82-143).  `x` [rows, planes] -> `out` [rows, planes]; post = BatchNorm+ReLU
  // folded into the level's last conv (only the outermost level has one).
  int level(int l, const float *x, const int32_t *indices, int rows, const int32_t shape[3],
            const float *pre_in, int pre_cin, const float *post_s, const float *post_b, float *out) {
    const sg_unet_level &L = d->levels[l];
    const int c = L.planes;
    const bool deeper = l + 1 < d->n_levels;
    const size_t m0 = ar.mark();
    // SubM rulebook + plan of this level (indice_key 'subm<l>': shared by all its blocks)
    SG_ALLOC(nbr, int32_t, static_cast<size_t>(rows) * 27);
    if (rows) {
      const size_t m = ar.mark();
      const size_t nb = sg_spconv_hash_workspace_bytes(rows);
      SG_ALLOC(ws, char, nb);
      SG_TRY(sg_spconv_subm_rulebook(indices, rows, shape, nbr, ws, nb, stream));
      ar.release(m);
    }
    Plan subm;
    SG_TRY(make_plan(nbr, rows, 27, subm));
    // identity table for the 1x1 convs of the tail (natural order, no plan)
    Plan ident;
    if (deeper && rows) {
      SG_ALLOC(iota, int32_t, rows);
      iota_kernel<<<grid_for(rows, 256), 256, 0, as_stream(stream)>>>(iota, rows);
      ident.nbr = iota; ident.rows = rows; ident.kvol = 1;
    }
    // optional input conv (outermost level only): SubMConv3d(in, planes) on the same rulebook
    if (pre_in != nullptr) {
      SG_ALLOC(x0, float, static_cast<size_t>(rows) * c);
      SG_TRY(conv(pre_in, rows, subm, pre_cin, c, d->input_w, nullptr, nullptr, nullptr, x0));
      x = x0;
    }
    // blocks
    const float *cur = x;
    for (int i = 0; i < L.n_blocks; ++i) {
      const bool last = !deeper && i == L.n_blocks - 1;
      float *dst = out;
      if (!last) {
        SG_ALLOC(t, float, static_cast<size_t>(rows) * c);
        dst = t;
      }
      SG_TRY(block(L.blocks[i], cur, rows, subm, ident, last ? post_s : nullptr, last ? post_b : nullptr, dst));
      cur = dst;
    }
    if (deeper) {
      const int c2 = d->levels[l + 1].planes;
      // ---- strided conv rulebook (needs the number of coarse voxels on the host)
      SG_ALLOC(in2out, int32_t, rows ? rows : 1);
      SG_ALLOC(meta, int32_t, 64);
      const size_t nbh = sg_spconv_hash_workspace_bytes(rows);
      SG_ALLOC(hws, char, nbh);
      int rows2 = 0;
      if (rows) {
        SG_TRY(sg_spconv_down_build(indices, rows, shape, in2out, meta, hws, nbh, stream));
        if (hipMemcpyAsync(host_meta, meta, sizeof(int32_t), hipMemcpyDeviceToHost, as_stream(stream)) != hipSuccess ||
            hipStreamSynchronize(as_stream(stream)) != hipSuccess) {
          set_error("sg_unet_forward: reading the coarse voxel count failed");
          return SG_ERR_LAUNCH;
        }
        rows2 = host_meta[0];
      }
      SG_ALLOC(idx2, int32_t, static_cast<size_t>(rows2 ? rows2 : 1) * 4);
      SG_ALLOC(child, int32_t, static_cast<size_t>(rows2 ? rows2 : 1) * 8);
      if (rows) SG_TRY(sg_spconv_down_fill(indices, rows, in2out, rows2, idx2, child, hws, nbh, stream));
      Plan down;
      SG_TRY(make_plan(child, rows2, 8, down));
      const int32_t shape2[3] = {shape[0] / 2, shape[1] / 2, shape[2] / 2};
      // ---- BN -> ReLU -> SparseConv3d(c, c2, k2 s2)
      SG_ALLOC(y, float, static_cast<size_t>(rows2 ? rows2 : 1) * c2);
      {
        const size_t m = ar.mark();
        SG_ALLOC(a, float, static_cast<size_t>(rows ? rows : 1) * c);
        SG_TRY(sg_bn_relu_f32(cur, L.down_bn_scale, L.down_bn_shift, rows, c, 1, a, stream));
        SG_TRY(conv(a, rows, down, c, c2, L.down_w, nullptr, nullptr, nullptr, y));
        ar.release(m);
      }
      // ---- inner UBlock
      SG_ALLOC(z, float, static_cast<size_t>(rows2 ? rows2 : 1) * c2);
      SG_TRY(level(l + 1, y, idx2, rows2, shape2, nullptr, 0, nullptr, nullptr, z));
      // ---- BN -> ReLU -> SparseInverseConv3d(c2, c): gather table = parent row per fine voxel
      SG_ALLOC(inv, int32_t, static_cast<size_t>(rows ? rows : 1) * 8);
      if (rows) SG_TRY(sg_spconv_inverse_rulebook(indices, in2out, rows, inv, stream));
      Plan up;
      SG_TRY(make_plan(inv, rows, 8, up));
      SG_ALLOC(cat, float, static_cast<size_t>(rows ? rows : 1) * 2 * c);
      {
        const size_t m = ar.mark();
        SG_ALLOC(a, float, static_cast<size_t>(rows2 ? rows2 : 1) * c2);
        SG_TRY(sg_bn_relu_f32(z, L.up_bn_scale, L.up_bn_shift, rows2, c2, 1, a, stream));
        SG_ALLOC(upf, float, static_cast<size_t>(rows ? rows : 1) * c);
        SG_TRY(conv(a, rows2, up, c2, c, L.up_w, nullptr, nullptr, nullptr, upf));
        // ---- skip concat (blocks.py:135-139)
        if (rows)
          concat2_kernel<<<grid_for(static_cast<int64_t>(rows) * (2 * c / 4), 256), 256, 0, as_stream(stream)>>>(
              reinterpret_cast<const float4 *>(cur), reinterpret_cast<const float4 *>(upf), rows, c / 4,
              c / 4, reinterpret_cast<float4 *>(cat));
        ar.release(m);
      }
      // ---- tail blocks: (2c -> c), (c -> c)
      cur = cat;
      for (int i = 0; i < L.n_blocks; ++i) {
        const bool last = i == L.n_blocks - 1;
        float *dst = out;
        if (!last) {
          SG_ALLOC(t, float, static_cast<size_t>(rows ? rows : 1) * c);
          dst = t;
        }
        SG_TRY(block(L.tail[i], cur, rows, subm, ident, last ? post_s : nullptr, last ? post_b : nullptr, dst));
        cur = dst;
      }
    }
    ar.release(m0);
    return check_launch("sg_unet_forward");
  }
};

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_unet_arena_bytes(const sg_unet_desc *d, int num_rows) {
  // every level is priced as if it kept all `num_rows` voxels (they can only shrink): tables
  // (27 + 27 + 3*8 + 2*8 + ~8 ints per row) + at most ~12 feature buffers of 2*planes floats
  size_t total = 1 << 20;
  const size_t rows = static_cast<size_t>(num_rows > 0 ? num_rows : 1);
  for (int l = 0; l < d->n_levels; ++l) {
    const size_t c = static_cast<size_t>(d->levels[l].planes);
    total += rows * (96 * 4 + 12 * 2 * c * 4) + sg_spconv_plan_workspace_bytes(num_rows) +
             sg_spconv_hash_workspace_bytes(num_rows) + (64 << 10);
  }
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
  static int32_t *host_meta = nullptr;
  if (host_meta == nullptr)
    SG_REQUIRE(hipHostMalloc(reinterpret_cast<void **>(&host_meta), 64) == hipSuccess,
               "sg_unet_forward: pinned allocation failed");
  Exec ex(d, arena, arena_bytes, stream);
  ex.host_meta = host_meta;
  const bool pre = d->input_w != nullptr;
  return ex.level(0, pre ? nullptr : feats, indices, num_rows, spatial_shape_host, pre ? feats : nullptr,
                  d->input_cin, d->out_bn_scale, d->out_bn_shift, out);
}

}  // extern "C"
