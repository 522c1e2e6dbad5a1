// seg_ops.hip -- streaming / segment kernels of the grouping head (HBM-bound, integer + fp32).
//
//   voxel feature pooling fwd/bwd      (reference: voxelize/voxelize.cu:10-62)
//   segment mean / min / max           (reference: sec_mean/sec_mean.cu:13-93)
//   ROI global average pool fwd/bwd    (reference: roipool/roipool.cu:12-71)
//   proposal/GT mask IoU + mask label  (reference: cal_iou_and_masklabel.cu:9-164)
//   eval-BN+ReLU, row gather           (torch glue in softgroup.py:65,374,677)
//
// MI355X mapping: one thread per output element with the channel index fastest, so a
// wave touches whole 128-B row segments; order-sensitive fp32 sums keep the reference's
// sequential order per output element (bit-exact), order-free reductions (min/max, integer
// histograms) use the whole workgroup.  Compiled with -ffp-contract=off.
#include "common.h"

namespace sg {

// ------------------------------------------------------------------ voxelize fp / bp
// out[row,p] = sum_i (m * feats[r[i],p]) -- separate multiply and add (voxelize.cu:21)
__global__ void __launch_bounds__(256) voxelize_fp_kernel(const float *__restrict__ feats,
                                                         const int32_t *__restrict__ rules,
                                                         int M, int max_active, int C, int average,
                                                         float *__restrict__ out) {
  const int64_t total = static_cast<int64_t>(M) * C;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int row = static_cast<int>(t / C), p = static_cast<int>(t - static_cast<int64_t>(row) * C);
    const int32_t *r = rules + static_cast<int64_t>(row) * (max_active + 1);
    const int cnt = r[0];
    const float m = (average && cnt > 0) ? __fdiv_rn(1.0f, static_cast<float>(cnt)) : 1.0f;
    // (rule -> feature row are dependent loads: four points' chains in flight, the sum in order)
    float acc = 0.0f;
    int i = 1;
    for (; i + 3 <= cnt; i += 4) {
      const int a0 = r[i], a1 = r[i + 1], a2 = r[i + 2], a3 = r[i + 3];
      const float f0 = feats[static_cast<int64_t>(a0) * C + p], f1 = feats[static_cast<int64_t>(a1) * C + p],
                  f2 = feats[static_cast<int64_t>(a2) * C + p], f3 = feats[static_cast<int64_t>(a3) * C + p];
      acc = __fadd_rn(acc, __fmul_rn(m, f0));
      acc = __fadd_rn(acc, __fmul_rn(m, f1));
      acc = __fadd_rn(acc, __fmul_rn(m, f2));
      acc = __fadd_rn(acc, __fmul_rn(m, f3));
    }
    for (; i <= cnt; ++i)
      acc = __fadd_rn(acc, __fmul_rn(m, feats[static_cast<int64_t>(r[i]) * C + p]));
    out[t] = acc;
  }
}

// d_feats[r[i],p] += m * d_out[row,p]; every point sits in exactly one rule row, so there is a
// single writer per element (the reference's atomicAdd never contends, voxelize.cu:50).
__global__ void __launch_bounds__(256) voxelize_bp_kernel(const float *__restrict__ d_out,
                                                         const int32_t *__restrict__ rules,
                                                         int M, int max_active, int C, int average,
                                                         float *__restrict__ d_feats) {
  const int64_t total = static_cast<int64_t>(M) * C;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int row = static_cast<int>(t / C), p = static_cast<int>(t - static_cast<int64_t>(row) * C);
    const int32_t *r = rules + static_cast<int64_t>(row) * (max_active + 1);
    const int cnt = r[0];
    const float m = (average && cnt > 0) ? __fdiv_rn(1.0f, static_cast<float>(cnt)) : 1.0f;
    const float g = __fmul_rn(m, d_out[t]);
    for (int i = 1; i <= cnt; ++i) {
      float *dst = d_feats + static_cast<int64_t>(r[i]) * C + p;
      atomicAdd(dst, g);
    }
  }
}

// ------------------------------------------------------------------ segment ops
// Order-sensitive sums: one thread per (segment, channel), rows visited in order; loads are
// independent of the accumulate chain so the compiler keeps several in flight.
enum SegOp { kSecMean = 0, kAvgPool = 1 };

template <int OP>
__global__ void __launch_bounds__(256) seg_sum_kernel(const float *__restrict__ inp,
                                                     const int32_t *__restrict__ offsets, int nP,
                                                     int C, float *__restrict__ out) {
  const int64_t total = static_cast<int64_t>(nP) * C;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int p = static_cast<int>(t / C), c = static_cast<int>(t - static_cast<int64_t>(p) * C);
    const int s = offsets[p], e = offsets[p + 1];
    const float cnt = static_cast<float>(e - s);
    float acc = 0.0f;
    const float *src = inp + static_cast<int64_t>(s) * C + c;
    int i = s;
    for (; i + 8 <= e; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[static_cast<int64_t>(u) * C];
      src += 8LL * C;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        acc = __fadd_rn(acc, OP == kSecMean ? __fdiv_rn(v[u], cnt) : v[u]);  // sec_mean.cu:24
    }
    for (; i < e; ++i, src += C) acc = __fadd_rn(acc, OP == kSecMean ? __fdiv_rn(*src, cnt) : *src);
    out[t] = OP == kSecMean ? acc : __fdiv_rn(acc, cnt);  // roipool.cu:29: sum then one divide
  }
}

// min / max are order-free: one workgroup per segment, lanes stride over rows*C elements.
template <bool IS_MAX>
__global__ void __launch_bounds__(256) seg_minmax_kernel(const float *__restrict__ inp,
                                                        const int32_t *__restrict__ offsets,
                                                        int nP, int C, float *__restrict__ out) {
  extern __shared__ float red[];  // [256/C' lanes-per-channel groups] -> we reduce via LDS [256]
  const float init = IS_MAX ? -INFINITY : INFINITY;  // float(+-1e50) == +-inf (sec_mean.cu:48,76)
  for (int p = blockIdx.x; p < nP; p += gridDim.x) {
    const int s = offsets[p], e = offsets[p + 1];
    for (int c0 = 0; c0 < C; c0 += 256) {  // C <= 256 in one pass; larger C loops
      const int cw = min(C - c0, 256);
      // thread t handles channel c0 + (t % cw), rows s + t / cw, step 256 / cw rows
      const int rows_per_iter = 256 / cw;
      float v = init;
      if (threadIdx.x < rows_per_iter * cw) {
        const int c = c0 + threadIdx.x % cw;
#pragma unroll 8
        for (int i = s + threadIdx.x / cw; i < e; i += rows_per_iter) {      // (a giant segment: ~2 k trips)
          float x = inp[static_cast<int64_t>(i) * C + c];
          // strict compare keeps the reference's NaN behaviour (NaN never replaces the running value)
          if (IS_MAX ? (x > v) : (x < v)) v = x;
        }
      }
      red[threadIdx.x] = v;
      __syncthreads();
      if (threadIdx.x < cw) {
        float r = red[threadIdx.x];
        for (int g = 1; g < rows_per_iter; ++g) {
          float x = red[threadIdx.x + g * cw];
          if (IS_MAX ? (x > r) : (x < r)) r = x;
        }
        out[static_cast<int64_t>(p) * C + c0 + threadIdx.x] = r;
      }
      __syncthreads();
    }
  }
}

// d_feats[i,c] += d_out[p,c] / n_p  (roipool.cu:55-57); single writer per element.
__global__ void __launch_bounds__(256) avg_pool_bp_kernel(float *__restrict__ d_feats,
                                                         const int32_t *__restrict__ offsets,
                                                         const float *__restrict__ d_out, int nP,
                                                         int C) {
  for (int p = blockIdx.y; p < nP; p += gridDim.y) {
    const int s = offsets[p], e = offsets[p + 1];
    const float cnt = static_cast<float>(e - s);
    const int64_t total = static_cast<int64_t>(e - s) * C;
    for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
      const int c = static_cast<int>(t % C);
      float *dst = d_feats + static_cast<int64_t>(s) * C + t;
      *dst = __fadd_rn(*dst, __fdiv_rn(d_out[static_cast<int64_t>(p) * C + c], cnt));
    }
  }
}

// ------------------------------------------------------------------ mask IoU / label
// One workgroup per proposal.  The intersection with every GT instance is a histogram of the
// proposal's instance labels, built with LDS atomics (integer -> exact); the reference rescans
// the proposal once per instance.  Histogram bins beyond the LDS budget spill to a second pass.
constexpr int kIouBins = 8192;  // 32 KB of LDS

template <bool ON_PRED>
__global__ void __launch_bounds__(256) mask_iou_kernel(const int32_t *__restrict__ proposals_idx,
                                                      const int32_t *__restrict__ proposals_offset,
                                                      const int64_t *__restrict__ instance_labels,
                                                      const int32_t *__restrict__ instance_pointnum,
                                                      const float *__restrict__ mask_scores_sigmoid,
                                                      int nInstance, int nProposal,
                                                      float *__restrict__ iou) {
  __shared__ int hist[kIouBins];
  __shared__ int ptotal_s;
  for (int p = blockIdx.x; p < nProposal; p += gridDim.x) {
    const int s = proposals_offset[p], e = proposals_offset[p + 1];
    for (int g0 = 0; g0 < nInstance; g0 += kIouBins) {
      const int gw = min(nInstance - g0, kIouBins);
      for (int g = threadIdx.x; g < gw; g += 256) hist[g] = 0;
      if (threadIdx.x == 0) ptotal_s = 0;
      __syncthreads();
      int my_total = 0;
      for (int i = s + threadIdx.x; i < e; i += 256) {
        if (ON_PRED) {
          if (!(mask_scores_sigmoid[i] > 0.5f)) continue;  // cal_iou_and_masklabel.cu:47,56
          ++my_total;
        }
        const int lab = static_cast<int>(instance_labels[proposals_idx[i]]) - g0;
        if (lab >= 0 && lab < gw) atomicAdd(&hist[lab], 1);
      }
      if (ON_PRED) {
        my_total = wave_sum(my_total);
        if ((threadIdx.x & 63) == 0 && my_total) atomicAdd(&ptotal_s, my_total);
      }
      __syncthreads();
      const int ptotal = ON_PRED ? ptotal_s : (e - s);
      for (int g = threadIdx.x; g < gw; g += 256) {
        const int inter = hist[g];
        const int itotal = instance_pointnum[g0 + g];
        // (float)inter / ((float)total + 1e-5): the literal is a double (cu:29-31, 63-65)
        const double den = static_cast<double>(static_cast<float>(ptotal + itotal - inter)) + 1e-5;
        iou[static_cast<int64_t>(p) * nInstance + g0 + g] =
            static_cast<float>(static_cast<double>(static_cast<float>(inter)) / den);
      }
      __syncthreads();
    }
  }
}

// per proposal: first maximal IoU over non-ignored instances (strict >, starts at 0), then
// label the proposal's points 1/0 if max_iou >= thr (cal_iou_and_masklabel.cu:70-104).
__global__ void __launch_bounds__(256) mask_label_kernel(const int32_t *__restrict__ proposals_idx,
                                                        const int32_t *__restrict__ proposals_offset,
                                                        const int64_t *__restrict__ instance_labels,
                                                        const int64_t *__restrict__ instance_cls,
                                                        const float *__restrict__ proposals_iou,
                                                        int nInstance, int nProposal, float iou_thr,
                                                        float *__restrict__ mask_label) {
  __shared__ float s_val[256];
  __shared__ int s_idx[256];
  for (int p = blockIdx.x; p < nProposal; p += gridDim.x) {
    const int s = proposals_offset[p], e = proposals_offset[p + 1];
    float best = 0.0f;
    int best_i = 0x7fffffff;
    for (int g = threadIdx.x; g < nInstance; g += 256) {
      const float v = proposals_iou[static_cast<int64_t>(p) * nInstance + g];
      if (v > best && instance_cls[g] != -100) { best = v; best_i = g; }  // first max per thread
    }
    s_val[threadIdx.x] = best;
    s_idx[threadIdx.x] = best_i;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) {
        const float v = s_val[threadIdx.x + o];
        const int i = s_idx[threadIdx.x + o];
        if (v > s_val[threadIdx.x] || (v == s_val[threadIdx.x] && i < s_idx[threadIdx.x])) {
          s_val[threadIdx.x] = v;
          s_idx[threadIdx.x] = i;
        }
      }
      __syncthreads();
    }
    const float max_iou = s_val[0];
    const int max_ind = s_idx[0] == 0x7fffffff ? 0 : s_idx[0];
    __syncthreads();
    if (max_iou >= iou_thr) {
      for (int i = s + threadIdx.x; i < e; i += 256)
        mask_label[i] = static_cast<int>(instance_labels[proposals_idx[i]]) == max_ind ? 1.0f : 0.0f;
    }
  }
}

// ------------------------------------------------------------------ glue
__global__ void __launch_bounds__(256) bn_relu_kernel(const float4 *__restrict__ x,
                                                     const float *__restrict__ scale,
                                                     const float *__restrict__ shift,
                                                     int64_t total4, int C4, int relu,
                                                     float4 *__restrict__ out) {
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total4; t += gridDim.x * 256LL) {
    const int c = static_cast<int>(t % C4) * 4;
    float4 v = x[t];
    v.x = fmaf(v.x, scale[c], shift[c]);
    v.y = fmaf(v.y, scale[c + 1], shift[c + 1]);
    v.z = fmaf(v.z, scale[c + 2], shift[c + 2]);
    v.w = fmaf(v.w, scale[c + 3], shift[c + 3]);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    out[t] = v;
  }
}
__global__ void __launch_bounds__(256) bn_relu_scalar_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift,
                                                            int64_t total, int C, int relu,
                                                            float *__restrict__ out) {
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int c = static_cast<int>(t % C);
    float v = fmaf(x[t], scale[c], shift[c]);
    out[t] = relu ? fmaxf(v, 0.f) : v;
  }
}

template <typename IdxT, typename VecT>
__global__ void __launch_bounds__(256) gather_rows_kernel(const VecT *__restrict__ in,
                                                         const IdxT *__restrict__ index,
                                                         int64_t n_rows, int CV,
                                                         VecT *__restrict__ out) {
  const int64_t total = n_rows * CV;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t r = t / CV;
    const int c = static_cast<int>(t - r * CV);
    out[t] = in[static_cast<int64_t>(index[r]) * CV + c];
  }
}

}  // namespace sg

using namespace sg;

template <typename IdxT>
static int gather_rows_impl(const float *in, const IdxT *index, int64_t n, int C, float *out,
                            sg_stream_t stream, const char *name) {
  SG_REQUIRE(n >= 0 && C > 0, "%s: bad sizes", name);
  if (n == 0) return SG_OK;
  if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    gather_rows_kernel<IdxT, float4><<<grid_for(n * (C / 4), 256), 256, 0, as_stream(stream)>>>(
        reinterpret_cast<const float4 *>(in), index, n, C / 4, reinterpret_cast<float4 *>(out));
  } else {
    gather_rows_kernel<IdxT, float><<<grid_for(n * C, 256), 256, 0, as_stream(stream)>>>(
        in, index, n, C, out);
  }
  return check_launch(name);
}

extern "C" {

int sg_voxelize_fp(const float *feats, const int32_t *rules, int M, int max_active, int C,
                   int average, float *out, sg_stream_t stream) {
  SG_REQUIRE(M >= 0 && max_active >= 0 && C > 0, "sg_voxelize_fp: bad sizes");
  if (M == 0) return SG_OK;
  voxelize_fp_kernel<<<grid_for(static_cast<int64_t>(M) * C, 256), 256, 0, as_stream(stream)>>>(
      feats, rules, M, max_active, C, average, out);
  return check_launch("sg_voxelize_fp");
}

int sg_voxelize_bp(const float *d_out, const int32_t *rules, int M, int max_active, int C,
                   int average, float *d_feats, sg_stream_t stream) {
  SG_REQUIRE(M >= 0 && max_active >= 0 && C > 0, "sg_voxelize_bp: bad sizes");
  if (M == 0) return SG_OK;
  voxelize_bp_kernel<<<grid_for(static_cast<int64_t>(M) * C, 256), 256, 0, as_stream(stream)>>>(
      d_out, rules, M, max_active, C, average, d_feats);
  return check_launch("sg_voxelize_bp");
}

int sg_sec_mean(const float *inp, const int32_t *offsets, int nP, int C, float *out,
                sg_stream_t stream) {
  SG_REQUIRE(nP >= 0 && C > 0, "sg_sec_mean: bad sizes");
  if (nP == 0) return SG_OK;
  seg_sum_kernel<kSecMean><<<grid_for(static_cast<int64_t>(nP) * C, 256), 256, 0,
                             as_stream(stream)>>>(inp, offsets, nP, C, out);
  return check_launch("sg_sec_mean");
}

int sg_global_avg_pool_fp(const float *feats, const int32_t *offsets, int nP, int C, float *out,
                          sg_stream_t stream) {
  SG_REQUIRE(nP >= 0 && C > 0, "sg_global_avg_pool_fp: bad sizes");
  if (nP == 0) return SG_OK;
  seg_sum_kernel<kAvgPool><<<grid_for(static_cast<int64_t>(nP) * C, 256), 256, 0,
                             as_stream(stream)>>>(feats, offsets, nP, C, out);
  return check_launch("sg_global_avg_pool_fp");
}

int sg_sec_min(const float *inp, const int32_t *offsets, int nP, int C, float *out,
               sg_stream_t stream) {
  SG_REQUIRE(nP >= 0 && C > 0, "sg_sec_min: bad sizes");
  if (nP == 0) return SG_OK;
  seg_minmax_kernel<false><<<min(nP, 8192), 256, 256 * sizeof(float), as_stream(stream)>>>(
      inp, offsets, nP, C, out);
  return check_launch("sg_sec_min");
}

int sg_sec_max(const float *inp, const int32_t *offsets, int nP, int C, float *out,
               sg_stream_t stream) {
  SG_REQUIRE(nP >= 0 && C > 0, "sg_sec_max: bad sizes");
  if (nP == 0) return SG_OK;
  seg_minmax_kernel<true><<<min(nP, 8192), 256, 256 * sizeof(float), as_stream(stream)>>>(
      inp, offsets, nP, C, out);
  return check_launch("sg_sec_max");
}

int sg_global_avg_pool_bp(float *d_feats, const int32_t *offsets, const float *d_out, int nP, int C,
                          sg_stream_t stream) {
  SG_REQUIRE(nP >= 0 && C > 0, "sg_global_avg_pool_bp: bad sizes");
  if (nP == 0) return SG_OK;
  dim3 grid(8, min(nP, 4096));
  avg_pool_bp_kernel<<<grid, 256, 0, as_stream(stream)>>>(d_feats, offsets, d_out, nP, C);
  return check_launch("sg_global_avg_pool_bp");
}

int sg_get_mask_iou_on_cluster(const int32_t *proposals_idx, const int32_t *proposals_offset,
                               const int64_t *instance_labels, const int32_t *instance_pointnum,
                               int nInstance, int nProposal, float *iou, sg_stream_t stream) {
  SG_REQUIRE(nInstance >= 0 && nProposal >= 0, "sg_get_mask_iou_on_cluster: bad sizes");
  if (nInstance == 0 || nProposal == 0) return SG_OK;
  mask_iou_kernel<false><<<min(nProposal, 8192), 256, 0, as_stream(stream)>>>(
      proposals_idx, proposals_offset, instance_labels, instance_pointnum, nullptr, nInstance,
      nProposal, iou);
  return check_launch("sg_get_mask_iou_on_cluster");
}

int sg_get_mask_iou_on_pred(const int32_t *proposals_idx, const int32_t *proposals_offset,
                            const int64_t *instance_labels, const int32_t *instance_pointnum,
                            const float *mask_scores_sigmoid, int nInstance, int nProposal,
                            float *iou, sg_stream_t stream) {
  SG_REQUIRE(nInstance >= 0 && nProposal >= 0, "sg_get_mask_iou_on_pred: bad sizes");
  if (nInstance == 0 || nProposal == 0) return SG_OK;
  mask_iou_kernel<true><<<min(nProposal, 8192), 256, 0, as_stream(stream)>>>(
      proposals_idx, proposals_offset, instance_labels, instance_pointnum, mask_scores_sigmoid,
      nInstance, nProposal, iou);
  return check_launch("sg_get_mask_iou_on_pred");
}

int sg_get_mask_label(const int32_t *proposals_idx, const int32_t *proposals_offset,
                      const int64_t *instance_labels, const int64_t *instance_cls,
                      const float *proposals_iou, int nInstance, int nProposal, float iou_thr,
                      float *mask_label, sg_stream_t stream) {
  SG_REQUIRE(nInstance >= 0 && nProposal >= 0, "sg_get_mask_label: bad sizes");
  if (nProposal == 0) return SG_OK;
  mask_label_kernel<<<min(nProposal, 8192), 256, 0, as_stream(stream)>>>(
      proposals_idx, proposals_offset, instance_labels, instance_cls, proposals_iou, nInstance,
      nProposal, iou_thr, mask_label);
  return check_launch("sg_get_mask_label");
}

int sg_bn_relu_f32(const float *x, const float *scale, const float *shift, int64_t M, int C,
                   int relu, float *out, sg_stream_t stream) {
  SG_REQUIRE(M >= 0 && C > 0, "sg_bn_relu_f32: bad sizes");
  if (M == 0) return SG_OK;
  if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    int64_t total4 = M * C / 4;
    bn_relu_kernel<<<grid_for(total4, 256), 256, 0, as_stream(stream)>>>(
        reinterpret_cast<const float4 *>(x), scale, shift, total4, C / 4, relu,
        reinterpret_cast<float4 *>(out));
  } else {
    bn_relu_scalar_kernel<<<grid_for(M * C, 256), 256, 0, as_stream(stream)>>>(x, scale, shift,
                                                                              M * C, C, relu, out);
  }
  return check_launch("sg_bn_relu_f32");
}

int sg_gather_rows_f32(const float *in, const int32_t *index, int64_t n, int C, float *out,
                       sg_stream_t stream) {
  return gather_rows_impl<int32_t>(in, index, n, C, out, stream, "sg_gather_rows_f32");
}
int sg_gather_rows_i64idx_f32(const float *in, const int64_t *index, int64_t n, int C, float *out,
                              sg_stream_t stream) {
  return gather_rows_impl<int64_t>(in, index, n, C, out, stream, "sg_gather_rows_i64idx_f32");
}

}  // extern "C"
