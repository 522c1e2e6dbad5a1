// heads.hip -- the point-wise heads of SoftGroup.forward_backbone as ONE kernel:
//   output_feats = voxel_feats[v2p_map]                    (devoxelize, softgroup/model/softgroup.py:374)
//   semantic_scores = semantic_linear(output_feats)         (:375; MLP = Linear, BatchNorm1d, ReLU, Linear,
//   pt_offsets      = offset_linear(output_feats)           (:376;  softgroup/model/blocks.py:9-27)
//   semantic_preds  = semantic_scores.max(1)[1]             (:320)
// The reference (and rounds 1-4 here) run this as a row gather, four hipBLASLt GEMMs, two BatchNorm
// kernels, two ReLUs, bias copies and a max-reduction: 15 launches, 0.21 ms per 150 000-point scan for
// 0.8 GFLOP -- every intermediate [N, 32] tensor makes a round trip through HBM.  Here a thread owns a
// point: its 32 input channels and the hidden layer live in registers, the weights (11 KB for both
// heads) are wave-uniform scalar loads, and every result is written once.
//   HBM bytes per point: C*4 gathered + C*4 (output_feats) + (n_sem + 3) * 4 + 8 written.
// Arithmetic: fp32 FMA chains in ascending channel order (deterministic); eval-mode BatchNorm as one
// fma with the folded (scale, shift) the conv epilogues use.  Sums differ from a GEMM library's by
// rounding order only (tolerance 1e-4, tests/test_ops_gpu.py::test_pointwise_heads_equal_the_modules).
// (Tried: two points per thread, every scalar-loaded weight feeding two FMAs: 207 VGPRs, 81 us instead of
// 58 us per 150 000 points -- the second wave per SIMD is worth more than the halved weight loads.)
#include "common.h"
#include "heads.h"

namespace sg {

template <int C, int SEM_MAX, typename IdxT>
__global__ void __launch_bounds__(256) pointwise_heads_kernel(const float *__restrict__ vox, const IdxT *__restrict__ v2p,
                                                             int n, Mlp2 sem, Mlp2 off, float *__restrict__ out_feats,
                                                             float *__restrict__ sem_scores, float *__restrict__ pt_offsets,
                                                             int64_t *__restrict__ sem_preds) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int64_t row = v2p ? static_cast<int64_t>(v2p[p]) : p;
  float x[C];
  const float4 *src = reinterpret_cast<const float4 *>(vox + row * C);
#pragma unroll
  for (int q = 0; q < C / 4; ++q) {
    const float4 v = src[q];
    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
  }
  if (out_feats) {
    float4 *dst = reinterpret_cast<float4 *>(out_feats + static_cast<int64_t>(p) * C);
#pragma unroll
    for (int q = 0; q < C / 4; ++q) dst[q] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
  }
  {
    float y[SEM_MAX];
    mlp2<C, SEM_MAX>(x, sem, y);
    float best = y[0];
    int arg = 0;
#pragma unroll
    for (int o = 0; o < SEM_MAX; ++o) {
      if (o < sem.out) {
        sem_scores[static_cast<int64_t>(p) * sem.out + o] = y[o];
        if (o > 0 && (y[o] > best || (y[o] != y[o] && best == best))) {      // first maximum; NaN wins like torch.max
          best = y[o];
          arg = o;
        }
      }
    }
    if (sem_preds) sem_preds[p] = arg;
  }
  {
    float y[4];
    mlp2<C, 4>(x, off, y);
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (o < off.out) pt_offsets[static_cast<int64_t>(p) * off.out + o] = y[o];
  }
}

template <int C, typename IdxT>
static void launch_heads(const float *vox, const void *v2p, int n, const Mlp2 &s, const Mlp2 &o, float *out_feats,
                         float *sem_scores, float *pt_offsets, int64_t *sem_preds, hipStream_t st) {
  pointwise_heads_kernel<C, 32, IdxT><<<(n + 255) / 256, 256, 0, st>>>(vox, static_cast<const IdxT *>(v2p), n, s, o,
                                                                       out_feats, sem_scores, pt_offsets, sem_preds);
}

}  // namespace sg

using namespace sg;

extern "C" int sg_pointwise_heads(const float *voxel_feats, const void *v2p_map, int v2p_is_int64, int n_points,
                                  int channels, const sg_mlp2 *semantic, const sg_mlp2 *offset, float *output_feats,
                                  float *semantic_scores, float *pt_offsets, int64_t *semantic_preds,
                                  sg_stream_t stream) {
  SG_REQUIRE(n_points >= 0 && semantic && offset && semantic_scores && pt_offsets, "sg_pointwise_heads: bad arguments");
  SG_REQUIRE(channels == 16 || channels == 32, "sg_pointwise_heads: channels must be 16 or 32 (got %d)", channels);
  SG_REQUIRE(semantic->out >= 1 && semantic->out <= 32 && offset->out >= 1 && offset->out <= 4,
             "sg_pointwise_heads: head widths out of range (semantic %d, offset %d)", semantic->out, offset->out);
  if (n_points == 0) return SG_OK;
  const Mlp2 s{semantic->w1, semantic->b1, semantic->bn_scale, semantic->bn_shift, semantic->w2, semantic->b2, semantic->out};
  const Mlp2 o{offset->w1, offset->b1, offset->bn_scale, offset->bn_shift, offset->w2, offset->b2, offset->out};
  hipStream_t st = as_stream(stream);
  if (channels == 32) {
    if (v2p_is_int64) launch_heads<32, int64_t>(voxel_feats, v2p_map, n_points, s, o, output_feats, semantic_scores, pt_offsets, semantic_preds, st);
    else launch_heads<32, int32_t>(voxel_feats, v2p_map, n_points, s, o, output_feats, semantic_scores, pt_offsets, semantic_preds, st);
  } else {
    if (v2p_is_int64) launch_heads<16, int64_t>(voxel_feats, v2p_map, n_points, s, o, output_feats, semantic_scores, pt_offsets, semantic_preds, st);
    else launch_heads<16, int32_t>(voxel_feats, v2p_map, n_points, s, o, output_feats, semantic_scores, pt_offsets, semantic_preds, st);
  }
  return check_launch("sg_pointwise_heads");
}
