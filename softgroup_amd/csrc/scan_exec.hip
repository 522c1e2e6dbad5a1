// scan_exec.hip -- native host driver of the grouping head and of the result extraction: what
// SoftGroup.forward_grouping + clusters_voxelization (softgroup/model/softgroup.py:411-480,655-709)
// and get_instances (:537-604) do between the network's dense heads, as TWO C calls
//   sg_scan_grouping   softmaxed semantic scores + offsets -> proposals -> proposal voxels + features
//   sg_scan_instances  instance-head scores -> kept instances -> RLE text in pinned host memory
// instead of ~100 torch kernels, ~50 copy / fill nodes and the interpreter in between (round 3:
// 105 torch/rocprim kernels and 62 copy/fill nodes per scan, profiles/r03_kernel_top.txt).  The
// kernels behind the reference's operator surface are the ones of this library (ball query, BFS
// clustering, voxel index, voxel pooling, instance runs, RLE text); what is new here are the fused
// replacements of the torch glue (class selection + compaction, proposal scale / shift / voxel
// coordinates, instance keep table) and the sequencing: device memory comes from ONE caller-provided
// arena (bump, never recycled inside a call), sizes that depend on the data are read back through
// the calling thread's pinned words (no pageable copies, no interpreter lock held while waiting).
// Every float operation of the glue is written as the separate IEEE operation torch performs
// (file compiled with -ffp-contract=off): results are bit-identical to the module path.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "scan.h"

namespace sg {

// ------------------------------------------------------------------------------------------------
// class selection (softgroup.py:433-441 for all classes at once): segment s = class seg_class[s];
// point i belongs to it iff scores[i, class] > score_thr; segments with fewer than min_npoint
// points are skipped.  Output order = class-major, point-ascending (torch.nonzero of the [n_seg, N]
// mask), by a block count -> scan -> emit pass.
// ------------------------------------------------------------------------------------------------
constexpr int kSelBlock = 256;
constexpr int kMaxSeg = 32;

__global__ void __launch_bounds__(kSelBlock) select_count_kernel(const float *__restrict__ scores, int n,
                                                                int n_cls, const int32_t *__restrict__ seg_class,
                                                                int n_seg, float thr, int n_blocks,
                                                                int32_t *__restrict__ blk_cnt) {
  __shared__ int part[4][kMaxSeg];
  const int i = blockIdx.x * kSelBlock + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int s = 0; s < n_seg; ++s) {
    const bool on = i < n && scores[static_cast<int64_t>(i) * n_cls + seg_class[s]] > thr;
    const int c = __popcll(__ballot(on));
    if (lane == 0) part[wave][s] = c;
  }
  __syncthreads();
  if (threadIdx.x < n_seg)
    blk_cnt[threadIdx.x * n_blocks + blockIdx.x] =
        part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// one workgroup: segment totals, drop the small segments, exclusive scan of blk_cnt in (segment,
// block) order -> blk_off; meta[0] = selected points, meta[1 + s] = points of segment s (0 = skipped)
__global__ void __launch_bounds__(1024) select_scan_kernel(const int32_t *__restrict__ blk_cnt, int n_seg,
                                                          int n_blocks, int min_npoint,
                                                          int32_t *__restrict__ blk_off,
                                                          int32_t *__restrict__ meta) {
  // (eight segments of a thread's block column are loaded together and scanned from registers: segment by segment
  //  the kernel was 2 x n_seg rounds of a memory round trip and two barriers, 22 us for 18 segments)
  constexpr int kG = 8;
  __shared__ int seg_tot[kMaxSeg], seg_base[kMaxSeg + 1], carry_s[kMaxSeg];
  __shared__ int wtot[kG][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // ---- segment totals
  for (int s0 = 0; s0 < n_seg; s0 += kG) {
    int sum[kG];
#pragma unroll
    for (int j = 0; j < kG; ++j) sum[j] = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {
      const int b = b0 + threadIdx.x;
      int v[kG];
#pragma unroll
      for (int j = 0; j < kG; ++j) v[j] = (s0 + j < n_seg && b < n_blocks) ? blk_cnt[(s0 + j) * n_blocks + b] : 0;
#pragma unroll
      for (int j = 0; j < kG; ++j) sum[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < kG; ++j) {
      const int t = wave_sum(sum[j]);
      if (lane == 0) wtot[j][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < kG && s0 + threadIdx.x < n_seg) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wtot[threadIdx.x][w];
      seg_tot[s0 + threadIdx.x] = t >= min_npoint ? t : 0;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int s = 0; s < n_seg; ++s) {
      seg_base[s] = acc;
      carry_s[s] = acc;
      acc += seg_tot[s];
      meta[1 + s] = seg_tot[s];
    }
    seg_base[n_seg] = acc;
    meta[0] = acc;
  }
  __syncthreads();
  // ---- exclusive scan of every live segment's block counts, in (segment, block) order
  for (int s0 = 0; s0 < n_seg; s0 += kG) {
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {
      const int b = b0 + threadIdx.x;
      int v[kG], incl[kG];
#pragma unroll
      for (int j = 0; j < kG; ++j)
        v[j] = (s0 + j < n_seg && b < n_blocks && seg_tot[s0 + j] > 0) ? blk_cnt[(s0 + j) * n_blocks + b] : 0;
#pragma unroll
      for (int j = 0; j < kG; ++j) {
        incl[j] = wave_incl_scan(v[j]);
        if (lane == 63) wtot[j][wave] = incl[j];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kG; ++j) {
        if (s0 + j < n_seg) {      // (uniform)
          int before = 0, all = 0;
          for (int w = 0; w < 16; ++w) {
            const int x = wtot[j][w];
            before += w < wave ? x : 0;
            all += x;
          }
          if (b < n_blocks)
            blk_off[(s0 + j) * n_blocks + b] = seg_tot[s0 + j] > 0 ? carry_s[s0 + j] + before + incl[j] - v[j] : -1;
        }
      }
      __syncthreads();
      if (threadIdx.x < kG && s0 + threadIdx.x < n_seg) {      // (the segment's running offset for its next 1024 blocks)
        int all = 0;
        for (int w = 0; w < 16; ++w) all += wtot[threadIdx.x][w];
        carry_s[s0 + threadIdx.x] += all;
      }
      __syncthreads();
    }
  }
}

// emit: obj (scene point), seg, shifted coordinates (coords + offsets: softgroup.py:447) and the
// ball query's batch key seg * batch_size + batch (points of different classes / scenes never meet)
__global__ void __launch_bounds__(kSelBlock) select_emit_kernel(
    const float *__restrict__ scores, int n, int n_cls, const int32_t *__restrict__ seg_class, int n_seg,
    float thr, int n_blocks, const int32_t *__restrict__ blk_off, const float *__restrict__ coords,
    const float *__restrict__ offsets, const int32_t *__restrict__ batch_idxs, int batch_size,
    int32_t *__restrict__ obj, int32_t *__restrict__ seg_of, float *__restrict__ pts, int32_t *__restrict__ key) {
  __shared__ int wcnt[4];
  const int i = blockIdx.x * kSelBlock + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  float c[3] = {0.f, 0.f, 0.f};
  int bi = 0;
  if (i < n) {
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = __fadd_rn(coords[3LL * i + a], offsets[3LL * i + a]);
    bi = batch_idxs[i];
  }
  for (int s = 0; s < n_seg; ++s) {
    const int base = blk_off[s * n_blocks + blockIdx.x];
    if (base < 0) continue;                       // skipped segment (uniform)
    const bool on = i < n && scores[static_cast<int64_t>(i) * n_cls + seg_class[s]] > thr;
    const uint64_t m = __ballot(on);
    if ((threadIdx.x & 63) == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wcnt[w];
    if (on) {
      const int pos = base + before + mask_prefix(m);
      obj[pos] = i;
      seg_of[pos] = s;
      pts[3LL * pos] = c[0];
      pts[3LL * pos + 1] = c[1];
      pts[3LL * pos + 2] = c[2];
      key[pos] = s * batch_size + bi;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// proposals: local point index -> scene point index (softgroup.py:466), and per proposal the
// bounding box of its points, the voxel scale and the scaled lower corner (softgroup.py:680-690):
//   cscale = min(1 / max_d((hi - lo) * (1 / ss)) - 0.01, scale)      lo_s = lo * cscale
// written as the operations torch launches: `/ ss` with a host scalar is a multiplication by the
// fp32 reciprocal, `1 / x` is reciprocal(x) * 1, `- 0.01` adds -0.01f, clamp(max) is a min that
// keeps NaN.  One workgroup per proposal (min / max are order-free).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) proposal_map_kernel(int32_t *__restrict__ pairs, int64_t S,
                                                          const int32_t *__restrict__ obj) {
  for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < S; e += gridDim.x * 256LL)
    pairs[2 * e + 1] = obj[pairs[2 * e + 1]];
}

__global__ void __launch_bounds__(256) proposal_box_kernel(const int32_t *__restrict__ pairs,
                                                          const int32_t *__restrict__ offsets, int n_prop,
                                                          const float *__restrict__ coords, float inv_ss,
                                                          float scale_max, float *__restrict__ cscale,
                                                          float *__restrict__ lo_s) {
  __shared__ float rmin[3][256], rmax[3][256];
  for (int p = blockIdx.x; p < n_prop; p += gridDim.x) {
    const int s = offsets[p], e = offsets[p + 1];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll 4
    for (int i = s + threadIdx.x; i < e; i += 256) {      // (a giant proposal: 200+ trips of two dependent loads)
      const int pt = pairs[2LL * i + 1];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float x = coords[3LL * pt + a];
        if (x < mn[a]) mn[a] = x;      // strict compares: NaN never replaces (sec_mean.cu:48,76)
        if (x > mx[a]) mx[a] = x;
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      rmin[a][threadIdx.x] = mn[a];
      rmax[a][threadIdx.x] = mx[a];
    }
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (threadIdx.x < w) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float x = rmin[a][threadIdx.x + w], y = rmax[a][threadIdx.x + w];
          if (x < rmin[a][threadIdx.x]) rmin[a][threadIdx.x] = x;
          if (y > rmax[a][threadIdx.x]) rmax[a][threadIdx.x] = y;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      float ext = -INFINITY;      // torch.max over the 3 extents (NaN propagates)
      bool nan = false;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float d = __fmul_rn(__fsub_rn(rmax[a][0], rmin[a][0]), inv_ss);
        nan |= d != d;
        if (d > ext) ext = d;
      }
      if (nan) ext = NAN;
      float cs = __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, ext), 1.0f), -0.01f);
      cs = (cs != cs) ? cs : fminf(cs, scale_max);
      cscale[p] = cs;
#pragma unroll
      for (int a = 0; a < 3; ++a) lo_s[3 * p + a] = __fmul_rn(rmin[a][0], cs);
    }
    __syncthreads();
  }
}

// voxel coordinates of every (proposal, point) pair: trunc(coords * cscale - lo_s) (softgroup.py:689,
// 696-700), int64 [S,4] = (proposal, x, y, z) for the voxel index build; out-of-range coordinates
// (the reference asserts on them) raise bad[0]
__global__ void __launch_bounds__(256) proposal_voxel_coords_kernel(
    const int32_t *__restrict__ pairs, int64_t S, const float *__restrict__ coords,
    const float *__restrict__ cscale, const float *__restrict__ lo_s, int ss, int64_t *__restrict__ vox,
    int32_t *__restrict__ bad) {
  for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < S; e += gridDim.x * 256LL) {
    const int p = pairs[2 * e], pt = pairs[2 * e + 1];
    const float cs = cscale[p];
    int64_t *o = vox + 4 * e;
    o[0] = p;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = __fsub_rn(__fmul_rn(coords[3LL * pt + a], cs), lo_s[3 * p + a]);
      ok &= v >= 0.f && v < static_cast<float>(ss);
      o[1 + a] = static_cast<int64_t>(v);
    }
    if (!ok) atomicOr(bad, 1);
  }
}

// voxel features: mean of the proposal points' backbone features (softgroup.py:706); same
// arithmetic as voxelize_fp_kernel (seg_ops.hip), the rows taken through the pair list
__global__ void __launch_bounds__(256) proposal_voxel_feats_kernel(const float *__restrict__ feats,
                                                                  const int32_t *__restrict__ pairs,
                                                                  const int32_t *__restrict__ rules, int M,
                                                                  int max_active, int C,
                                                                  float *__restrict__ out) {
  const int64_t total = static_cast<int64_t>(M) * C;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int row = static_cast<int>(t / C), c = static_cast<int>(t - static_cast<int64_t>(row) * C);
    const int32_t *r = rules + static_cast<int64_t>(row) * (max_active + 1);
    const int cnt = r[0];
    const float m = cnt > 0 ? __fdiv_rn(1.0f, static_cast<float>(cnt)) : 1.0f;
    // (rule -> pair -> feature row is three dependent loads per point: sixteen points' chains are in
    //  flight at a time -- a voxel of a giant proposal holds hundreds of points and its 32 channel threads
    //  walk them alone: 330 -> ~100 us on the KITTI sweep --, the sum keeps the reference's order)
    float acc = 0.0f;
    int i = 1;
    for (; i + 15 <= cnt; i += 16) {
      int a[16], q[16];
      float f[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = r[i + j];
#pragma unroll
      for (int j = 0; j < 16; ++j) q[j] = pairs[2LL * a[j] + 1];
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = feats[static_cast<int64_t>(q[j]) * C + c];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc = __fadd_rn(acc, __fmul_rn(m, f[j]));
    }
    for (; i + 3 <= cnt; i += 4) {
      const int a0 = r[i], a1 = r[i + 1], a2 = r[i + 2], a3 = r[i + 3];
      const int p0 = pairs[2LL * a0 + 1], p1 = pairs[2LL * a1 + 1], p2 = pairs[2LL * a2 + 1],
                p3 = pairs[2LL * a3 + 1];
      const float f0 = feats[static_cast<int64_t>(p0) * C + c], f1 = feats[static_cast<int64_t>(p1) * C + c],
                  f2 = feats[static_cast<int64_t>(p2) * C + c], f3 = feats[static_cast<int64_t>(p3) * C + c];
      acc = __fadd_rn(acc, __fmul_rn(m, f0));
      acc = __fadd_rn(acc, __fmul_rn(m, f1));
      acc = __fadd_rn(acc, __fmul_rn(m, f2));
      acc = __fadd_rn(acc, __fmul_rn(m, f3));
    }
    for (; i <= cnt; ++i)
      acc = __fadd_rn(acc, __fmul_rn(m, feats[static_cast<int64_t>(pairs[2LL * r[i] + 1]) * C + c]));
    out[t] = acc;
  }
}

// int64 [M,4] voxel coordinates -> int32 (the SparseConvTensor contract) and the first voxel of
// every proposal (voxels are numbered first-seen over pairs grouped by proposal: contiguous)
__global__ void __launch_bounds__(256) proposal_voxel_pack_kernel(const int64_t *__restrict__ vc, int M,
                                                                 int n_prop, int32_t *__restrict__ out,
                                                                 int32_t *__restrict__ vox_off) {
  for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
    const int p = static_cast<int>(vc[4LL * m]);
#pragma unroll
    for (int a = 0; a < 4; ++a) out[4LL * m + a] = static_cast<int32_t>(vc[4LL * m + a]);
    if (m == 0 || static_cast<int>(vc[4LL * (m - 1)]) != p) vox_off[p] = m;
    if (m == M - 1) vox_off[n_prop] = M;
  }
}

// ------------------------------------------------------------------------------------------------
// kept instances (softgroup.py:573-588): for class i (major) and proposal p, keep iff
// cls_prob[p, i] > cls_thr and npoint[p, i] >= min_npoint; survivors numbered in that order.
// One workgroup; head[0] = n_kept, head[1] = sum of the kept npoint (run capacity).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) instance_keep_kernel(const float *__restrict__ cls_prob,
                                                           const float *__restrict__ iou, int stride,
                                                           const int32_t *__restrict__ npoint, int n_prop,
                                                           int nc, float cls_thr, int min_npoint,
                                                           int32_t *__restrict__ inst_of,
                                                           int32_t *__restrict__ kept_cls,
                                                           float *__restrict__ kept_score,
                                                           int32_t *__restrict__ head) {
  // (1024 threads per round: with 256 the 61 725 (class, proposal) pairs of an STPLS3D scan were 241
  // rounds of two barriers each, 154 us for one workgroup)
  __shared__ int wsum[16];
  __shared__ int cap_s;
  if (threadIdx.x == 0) cap_s = 0;
  __syncthreads();
  const int total = nc * n_prop;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int carry = 0, cap = 0;
  for (int base = 0; base < total; base += 1024) {
    const int t = base + threadIdx.x;
    int keep = 0, np = 0, i = 0, p = 0;
    if (t < total) {
      i = t / n_prop;
      p = t - i * n_prop;
      np = npoint[p * nc + i];
      keep = (cls_prob[p * stride + i] > cls_thr && np >= min_npoint) ? 1 : 0;
    }
    int incl = wave_incl_scan(keep);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int x = wsum[w];
      incl += w < wave ? x : 0;
      tot += x;
    }
    __syncthreads();
    if (t < total) {
      const int k = carry + incl - keep;
      inst_of[t] = keep ? k : -1;
      if (keep) {
        kept_cls[k] = i + 1;
        const float q = iou[p * stride + i];
        const float qc = (q != q) ? q : fminf(fmaxf(q, 0.f), 1.f);      // torch.clamp(0, 1)
        kept_score[k] = __fmul_rn(cls_prob[p * stride + i], qc);
        cap += np;
      }
    }
    carry += tot;
  }
  cap = wave_sum(cap);
  if ((threadIdx.x & 63) == 0) atomicAdd(&cap_s, cap);
  __syncthreads();
  if (threadIdx.x == 0) {
    head[0] = carry;
    head[1] = cap_s;
  }
}

// ------------------------------------------------------------------------------------------------
// SoftGroup++ grouping, one class at a time (softgroup.py:443-463): the glue between the operators.
// ------------------------------------------------------------------------------------------------
// pyramid_map's inputs for the points of one class (softgroup.py:491-498): level voxel coordinates
// (batch, trunc(coords / (base_size * level))) as int64 rows -- torch divides a CUDA tensor by a host
// scalar as a multiplication with the scalar's fp32 reciprocal, `inv` -- and the gathered coordinates /
// offsets that the level's voxel pooling averages
__global__ void __launch_bounds__(256) pp_level_inputs_kernel(const int32_t *__restrict__ obj, int n,
                                                             const float *__restrict__ coords,
                                                             const float *__restrict__ offsets,
                                                             const int32_t *__restrict__ batch_idxs, float inv,
                                                             int64_t *__restrict__ vox, float *__restrict__ c3,
                                                             float *__restrict__ o3) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int p = obj[i];
  vox[4LL * i] = batch_idxs[p];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float c = coords[3LL * p + a];
    c3[3LL * i + a] = c;
    o3[3LL * i + a] = offsets[3LL * p + a];
    vox[4LL * i + 1 + a] = static_cast<int64_t>(__fmul_rn(c, inv));      // .long(): toward zero
  }
}
// query points of a level: pooled coordinates + pooled offsets (softgroup.py:447 on the level's voxels),
// batch index = first column of the level's voxel coordinates
__global__ void __launch_bounds__(256) pp_level_points_kernel(const float *__restrict__ cl, const float *__restrict__ ol,
                                                             const int64_t *__restrict__ vc, int m,
                                                             float *__restrict__ q, int32_t *__restrict__ qb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
#pragma unroll
  for (int a = 0; a < 3; ++a) q[3LL * i + a] = __fadd_rn(cl[3LL * i + a], ol[3LL * i + a]);
  qb[i] = static_cast<int32_t>(vc[4LL * i]);
}
// a class's proposals appended to the scan's: proposal ids shifted by the proposals before, the class's
// local point index -> scene point, offsets shifted by the rows before (softgroup.py:455-463)
__global__ void __launch_bounds__(256) pp_append_kernel(const int32_t *__restrict__ pairs, int rows,
                                                       const int32_t *__restrict__ offs, int n_prop,
                                                       const int32_t *__restrict__ obj, int prop_base, int row_base,
                                                       int32_t *__restrict__ out_pairs, int32_t *__restrict__ out_offs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows) {
    out_pairs[2LL * (row_base + i)] = pairs[2LL * i] + prop_base;
    out_pairs[2LL * (row_base + i) + 1] = obj[pairs[2LL * i + 1]];
  }
  if (i < n_prop) out_offs[prop_base + 1 + i] = offs[i + 1] + row_base;
  if (i == 0 && prop_base == 0) out_offs[0] = 0;
}

// bump allocator over the caller's arena that keeps counting past the end, so that a failed call
// can say how much it needed up to the point where it stopped
struct ScanArena {
  char *base;
  size_t cap, off;
  bool ok = true;
  ScanArena(void *p, size_t n) : base(static_cast<char *>(p)), cap(n), off(0) {}
  template <typename T>
  T *take(size_t count) {
    const size_t bytes = align_up((count ? count : 1) * sizeof(T));
    const size_t at = off;
    off += bytes;
    if (off > cap) {
      ok = false;
      return nullptr;
    }
    return reinterpret_cast<T *>(base + at);
  }
  size_t at(const void *p) const { return static_cast<size_t>(static_cast<const char *>(p) - base); }
};

// SG_READBACK_KERNEL=1 (developer knob): the words reach the calling thread's pinned buffer through a one-wave
// kernel that stores them there (the buffer is mapped into the device's address space) instead of a copy command
__global__ void __launch_bounds__(64) read_back_kernel(const int32_t *__restrict__ dev, int words, int32_t *host) {
  if (static_cast<int>(threadIdx.x) < words)
    __hip_atomic_store(host + threadIdx.x, dev[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static int read_back(int32_t *host, const int32_t *dev, int words, hipStream_t stream, const char *what) {
  static const bool by_kernel = getenv("SG_READBACK_KERNEL") && atoi(getenv("SG_READBACK_KERNEL")) != 0;
  if (by_kernel && words <= 64) {
    read_back_kernel<<<1, 64, 0, stream>>>(dev, words, host);
    if (hipStreamSynchronize(stream) != hipSuccess) {
      set_error("%s: device -> host read-back failed", what);
      return SG_ERR_LAUNCH;
    }
    return SG_OK;
  }
  if (hipMemcpyAsync(host, dev, sizeof(int32_t) * words, hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess) {
    set_error("%s: device -> host read-back failed", what);
    return SG_ERR_LAUNCH;
  }
  return SG_OK;
}

#define SG_TRY(expr)              \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != SG_OK) return rc_; \
  } while (0)
#define SG_TAKE(var, T, count)                                                                    \
  T *var = ar.take<T>(count);                                                                     \
  if (var == nullptr) {                                                                           \
    res->arena_needed = ar.off + ar.off / 2 + (8 << 20);                                          \
    set_error("%s: arena too small (%zu bytes, need about %zu)", kWhat, ar.cap, res->arena_needed); \
    return SG_ERR_WORKSPACE;                                                                      \
  }

// sg_scan_forward (scan_forward.hip) starts the dense results' device-to-host copy from here: called once,
// on the calling thread, right before the ordered emission (one workgroup per cluster: the chip is idle)
thread_local void (*t_scan_emit_hook)(void *) = nullptr;
thread_local void *t_scan_emit_ctx = nullptr;

}  // namespace sg

using namespace sg;

// proposal voxelisation (softgroup.py:655-709, rand_quantize = False) behind both grouping drivers: boxes,
// voxel coordinates, voxel index, pooled features; one read-back (voxel count, max points per voxel)
static int voxelise_proposals(const sg_grouping_cfg *cfg, const int32_t *pairs, const int32_t *poff, int n_prop, int S,
                              const float *coords_float, const float *point_feats, ScanArena &ar, int32_t *meta,
                              int32_t *host, sg_grouping_result *res, sg_stream_t stream_, const char *kWhat) {
  hipStream_t stream = as_stream(stream_);
  const int C = cfg->feat_channels;
  SG_TAKE(cscale, float, n_prop);
  SG_TAKE(lo_s, float, 3 * static_cast<size_t>(n_prop));
  SG_TAKE(vox, int64_t, 4 * static_cast<size_t>(S));
  SG_TAKE(bad, int32_t, 64);
  hipMemsetAsync(bad, 0, sizeof(int32_t), stream);
  const float inv_ss = 1.0f / static_cast<float>(cfg->voxel_shape);
  proposal_box_kernel<<<n_prop < 1024 ? n_prop : 1024, 256, 0, stream>>>(pairs, poff, n_prop, coords_float, inv_ss,
                                                                        cfg->voxel_scale, cscale, lo_s);
  proposal_voxel_coords_kernel<<<grid_for(S, 256), 256, 0, stream>>>(pairs, S, coords_float, cscale, lo_s,
                                                                    cfg->voxel_shape, vox, bad);
  const size_t vx_bytes = sg_voxelize_idx_workspace_bytes(S);
  SG_TAKE(vx_ws, char, vx_bytes);
  SG_TAKE(inp_map, int32_t, S);
  SG_TRY(sg_voxelize_idx_build(vox, S, 4, 4, inp_map, meta + 40, vx_ws, vx_bytes, stream_));
  // meta[40] = voxels, meta[41] = max points per voxel, bad[0] = a coordinate left the grid
  hipMemcpyAsync(meta + 42, bad, sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
  SG_TRY(read_back(host, meta + 40, 3, stream, kWhat));
  const int M = host[0], mA = host[1];
  SG_REQUIRE(host[2] == 0, "%s: a proposal voxel coordinate fell outside [0, %d) "
             "(the reference asserts here, softgroup.py:698)", kWhat, cfg->voxel_shape);
  res->n_voxels = M;
  res->max_active = mA;
  SG_TAKE(vc64, int64_t, 4 * static_cast<size_t>(M));
  SG_TAKE(rules, int32_t, static_cast<size_t>(M) * (mA + 1));
  SG_TRY(sg_voxelize_idx_fill(vox, S, 4, 4, inp_map, M, mA, vc64, rules, vx_ws, vx_bytes, stream_));
  SG_TAKE(vc32, int32_t, 4 * static_cast<size_t>(M));
  SG_TAKE(vox_off, int32_t, static_cast<size_t>(n_prop) + 1);
  SG_TAKE(vfeat, float, static_cast<size_t>(M) * C);
  proposal_voxel_pack_kernel<<<grid_for(M, 256), 256, 0, stream>>>(vc64, M, n_prop, vc32, vox_off);
  proposal_voxel_feats_kernel<<<grid_for(static_cast<int64_t>(M) * C, 256), 256, 0, stream>>>(
      point_feats, pairs, rules, M, mA, C, vfeat);
  res->voxel_coords = ar.at(vc32);
  res->voxel_offsets = ar.at(vox_off);
  res->voxel_feats = ar.at(vfeat);
  res->point_to_voxel = ar.at(inp_map);
  res->arena_used = ar.off;
  return check_launch(kWhat);
}

extern "C" {

int sg_scan_grouping(const sg_grouping_cfg *cfg, const float *scores, const float *pt_offsets,
                     const float *coords_float, const int32_t *batch_idxs, const float *point_feats,
                     void *arena, size_t arena_bytes, sg_grouping_result *res, sg_stream_t stream_) {
  static const char *kWhat = "sg_scan_grouping";
  SG_REQUIRE(cfg != nullptr && res != nullptr, "sg_scan_grouping: null descriptor");
  SG_REQUIRE(cfg->n_points >= 0 && cfg->n_sem_classes >= 1 && cfg->n_seg >= 0 && cfg->n_seg <= kMaxSeg &&
                 cfg->batch_size >= 1 && cfg->feat_channels >= 1 && cfg->voxel_shape >= 0,
             "sg_scan_grouping: bad configuration (n_seg must be <= %d)", kMaxSeg);
  memset(res, 0, sizeof(*res));
  hipStream_t stream = as_stream(stream_);
  int32_t *host = pinned_words();
  SG_REQUIRE(host != nullptr, "sg_scan_grouping: no pinned host words");
  ScanArena ar(arena, arena_bytes);
  const int n = cfg->n_points, n_seg = cfg->n_seg;
  if (n == 0 || n_seg == 0) return SG_OK;

  // ---- 1. class selection
  const int n_blocks = (n + kSelBlock - 1) / kSelBlock;
  SG_TAKE(meta, int32_t, 64);
  SG_TAKE(blk_cnt, int32_t, static_cast<size_t>(n_seg) * n_blocks);
  SG_TAKE(blk_off, int32_t, static_cast<size_t>(n_seg) * n_blocks);
  select_count_kernel<<<n_blocks, kSelBlock, 0, stream>>>(scores, n, cfg->n_sem_classes, cfg->seg_class, n_seg,
                                                         cfg->score_thr, n_blocks, blk_cnt);
  select_scan_kernel<<<1, 1024, 0, stream>>>(blk_cnt, n_seg, n_blocks, cfg->min_npoint, blk_off, meta);
  SG_TRY(check_launch(kWhat));
  SG_TRY(read_back(host, meta, 1, stream, kWhat));
  const int n_sel = host[0];
  res->n_selected = n_sel;
  if (n_sel == 0) return SG_OK;
  SG_TAKE(obj, int32_t, n_sel);
  SG_TAKE(seg_of, int32_t, n_sel);
  SG_TAKE(pts, float, 3 * static_cast<size_t>(n_sel));
  SG_TAKE(key, int32_t, n_sel);
  select_emit_kernel<<<n_blocks, kSelBlock, 0, stream>>>(scores, n, cfg->n_sem_classes, cfg->seg_class, n_seg,
                                                        cfg->score_thr, n_blocks, blk_off, coords_float,
                                                        pt_offsets, batch_idxs, cfg->batch_size, obj, seg_of,
                                                        pts, key);

  // ---- 2. ball query (functions.py:237-275): grid, count, scan, fill
  const size_t bq_bytes = sg_ballquery_workspace_bytes(n_sel);
  SG_TAKE(bq_ws, char, bq_bytes);
  SG_TAKE(start_len, int32_t, 2 * static_cast<size_t>(n_sel));
  const size_t sc_bytes = sg_scan_workspace_bytes(n_sel);
  SG_TAKE(sc_ws, char, sc_bytes);
  hipMemsetAsync(start_len, 0, sizeof(int32_t) * 2 * n_sel, stream);
  SG_TRY(sg_ballquery_build_grid(pts, key, n_sel, cfg->radius, bq_ws, bq_bytes, stream_));
  SG_TRY(sg_ballquery_count(pts, key, n_sel, cfg->radius, start_len, nullptr, bq_ws, bq_bytes, stream_));
  SG_TRY(sg_exclusive_scan_startlen(start_len, n_sel, meta + 32, sc_ws, sc_bytes, stream_));
  SG_TRY(read_back(host, meta + 32, 1, stream, kWhat));
  const int n_active = host[0];
  res->n_neighbours = n_active;
  SG_TAKE(bq_idx, int32_t, n_active);
  SG_TRY(sg_ballquery_fill(pts, key, n_sel, cfg->radius, start_len, bq_idx, bq_ws, bq_bytes, stream_));

  // ---- 3. clustering of all classes in one launch set (functions.py:278-308 per class)
  const size_t bfs_bytes = sg_bfs_workspace_bytes(n_sel, n_active);
  SG_TAKE(bfs_ws, char, bfs_bytes);
  int32_t n_prop = 0, S = 0;
  SG_TRY(sg_bfs_cluster_label(bq_idx, start_len, n_sel, n_active, SG_LISTS_SORTED | SG_LISTS_RADIUS, seg_of,
                              cfg->seg_thr, n_seg, &n_prop, &S, bfs_ws, bfs_bytes, stream_));
  res->n_proposals = n_prop;
  res->sum_npoint = S;
  if (S == 0) return SG_OK;
  SG_TAKE(pairs, int32_t, 2 * static_cast<size_t>(S));
  SG_TAKE(poff, int32_t, static_cast<size_t>(n_prop) + 1);
  hipMemsetAsync(poff, 0, sizeof(int32_t) * (n_prop + 1), stream);
  if (t_scan_emit_hook != nullptr) t_scan_emit_hook(t_scan_emit_ctx);
  SG_TRY(sg_bfs_cluster_emit(bq_idx, start_len, n_sel, n_active, seg_of, cfg->seg_thr, n_prop, S, pairs, poff,
                             bfs_ws, bfs_bytes, stream_));
  proposal_map_kernel<<<grid_for(S, 256), 256, 0, stream>>>(pairs, S, obj);
  res->proposals_idx = ar.at(pairs);
  res->proposals_offset = ar.at(poff);
  if (cfg->voxel_shape == 0) {      // proposals only (the training step voxelises with rand_quantize)
    res->arena_used = ar.off;
    return check_launch(kWhat);
  }

  // ---- 4. proposal voxelisation (softgroup.py:655-709, rand_quantize = False)
  return voxelise_proposals(cfg, pairs, poff, n_prop, S, coords_float, point_feats, ar, meta, host, res, stream_, kWhat);
}

int sg_scan_grouping_pp(const sg_grouping_pp_cfg *pcfg, const float *scores, const float *pt_offsets,
                        const float *coords_float, const int32_t *batch_idxs, const float *point_feats,
                        void *arena, size_t arena_bytes, sg_grouping_result *res, sg_stream_t stream_) {
  static const char *kWhat = "sg_scan_grouping_pp";
  SG_REQUIRE(pcfg != nullptr && res != nullptr, "sg_scan_grouping_pp: null descriptor");
  const sg_grouping_cfg *cfg = &pcfg->base;
  SG_REQUIRE(cfg->n_points >= 0 && cfg->n_sem_classes >= 1 && cfg->n_seg >= 0 && cfg->n_seg <= kMaxSeg &&
                 cfg->batch_size >= 1 && cfg->feat_channels >= 1 && cfg->voxel_shape >= 0,
             "sg_scan_grouping_pp: bad configuration (n_seg must be <= %d)", kMaxSeg);
  SG_REQUIRE(pcfg->radius > 0 && (!pcfg->with_pyramid || pcfg->base_size > 0),
             "sg_scan_grouping_pp: radius and pyramid_base_size must be positive");
  memset(res, 0, sizeof(*res));
  hipStream_t stream = as_stream(stream_);
  int32_t *host = pinned_words();
  SG_REQUIRE(host != nullptr, "sg_scan_grouping_pp: no pinned host words");
  ScanArena ar(arena, arena_bytes);
  const int n = cfg->n_points, n_seg = cfg->n_seg;
  if (n == 0 || n_seg == 0) return SG_OK;

  // ---- class selection of ALL classes (class-major, points ascending: what the reference's per-class
  //      nonzero yields), ONE read-back of the per-class counts (classes below min_npoint read 0)
  const int n_blocks = (n + kSelBlock - 1) / kSelBlock;
  SG_TAKE(meta, int32_t, 128);
  SG_TAKE(blk_cnt, int32_t, static_cast<size_t>(n_seg) * n_blocks);
  SG_TAKE(blk_off, int32_t, static_cast<size_t>(n_seg) * n_blocks);
  select_count_kernel<<<n_blocks, kSelBlock, 0, stream>>>(scores, n, cfg->n_sem_classes, cfg->seg_class, n_seg,
                                                         cfg->score_thr, n_blocks, blk_cnt);
  select_scan_kernel<<<1, 1024, 0, stream>>>(blk_cnt, n_seg, n_blocks, cfg->min_npoint, blk_off, meta);
  SG_TRY(check_launch(kWhat));
  SG_TRY(read_back(host, meta, 1 + n_seg, stream, kWhat));
  const int n_sel = host[0];
  int count[kMaxSeg];
  for (int s = 0; s < n_seg; ++s) count[s] = host[1 + s];
  res->n_selected = n_sel;
  if (n_sel == 0) return SG_OK;
  SG_TAKE(obj, int32_t, n_sel);
  SG_TAKE(seg_of, int32_t, n_sel);
  SG_TAKE(pts, float, 3 * static_cast<size_t>(n_sel));
  SG_TAKE(key, int32_t, n_sel);
  select_emit_kernel<<<n_blocks, kSelBlock, 0, stream>>>(scores, n, cfg->n_sem_classes, cfg->seg_class, n_seg,
                                                        cfg->score_thr, n_blocks, blk_off, coords_float,
                                                        pt_offsets, batch_idxs, cfg->batch_size, obj, seg_of,
                                                        pts, key);
  // the scan's proposals: a point belongs to at most one cluster of its class
  SG_TAKE(pairs, int32_t, 2 * static_cast<size_t>(n_sel));
  SG_TAKE(poff, int32_t, static_cast<size_t>(n_sel) + 2);
  int n_prop = 0, S = 0;
  int64_t n_nbr = 0;
  const size_t loop_mark0 = ar.off;      // per-class temporaries: the same region for every class (stream order)
  size_t loop_mark = loop_mark0;
  bool hook_pending = t_scan_emit_hook != nullptr;
  // A class with a giant cluster (> 16 384 points: milliseconds of multi-workgroup replay on the emission's side
  // stream) does not hold up the classes behind it: its emission hands the side stream over instead of joining
  // (bfs_emit_defer), the rest of the class -- inverse map, append at its known place -- is queued there, its
  // temporaries stay where they are, and the caller's stream goes on with the next class; one join before the
  // proposals are used.  The rows the inverse map will yield (the append offset of the classes behind) are
  // counted from the labelling before the emission.  One class per call (there is one side stream).
  const bool defer_on = !(getenv("SG_PP_DEFER") && atoi(getenv("SG_PP_DEFER")) == 0);      // developer A/B knob, read per call
  BfsDefer def;
  int def_rows = -1;
  bool def_mapped = false;
  struct DeferGuard {      // (an early return must not leave the side stream working on a released arena)
    BfsDefer &d;
    ~DeferGuard() {
      if (d.deferred && d.side != nullptr) hipStreamSynchronize(d.side);
    }
  } defer_guard{def};
  int s0 = 0;
  for (int s = 0; s < n_seg; s0 += count[s], ++s) {
    const int nc_pts = count[s];
    if (nc_pts == 0) continue;
    ar.off = loop_mark;
    // ---- level and radius of the class (softgroup.py:448-452, get_level :485-489)
    int level = 1;
    if (pcfg->with_pyramid) level = nc_pts > 1000000 ? 3 : nc_pts > 100000 ? 2 : 1;
    const float radius = static_cast<float>(pcfg->with_pyramid ? pcfg->radius * level : pcfg->radius);
    const bool mapped = pcfg->with_pyramid && (level > 1 || !pcfg->lvl_fusion);
    const float *q = pts + 3 * static_cast<size_t>(s0);
    const int32_t *qb = key + s0;      // (one class: the batch key separates scenes only)
    int n_q = nc_pts, n_lvl = 0;
    const int32_t *l2p = nullptr;
    if (mapped) {      // pyramid_map (:491-498): level voxels, pooled coordinates and offsets
      const float inv = 1.0f / static_cast<float>(pcfg->base_size * level);
      SG_TAKE(vox, int64_t, 4 * static_cast<size_t>(nc_pts));
      SG_TAKE(c3, float, 3 * static_cast<size_t>(nc_pts));
      SG_TAKE(o3, float, 3 * static_cast<size_t>(nc_pts));
      pp_level_inputs_kernel<<<grid_for(nc_pts, 256, 1 << 22), 256, 0, stream>>>(obj + s0, nc_pts, coords_float, pt_offsets,
                                                                            batch_idxs, inv, vox, c3, o3);
      const size_t vx_bytes = sg_voxelize_idx_workspace_bytes(nc_pts);
      SG_TAKE(vx_ws, char, vx_bytes);
      SG_TAKE(l2p_, int32_t, nc_pts);
      SG_TRY(sg_voxelize_idx_build(vox, nc_pts, 4, 4, l2p_, meta + 40, vx_ws, vx_bytes, stream_));
      SG_TRY(read_back(host, meta + 40, 2, stream, kWhat));
      const int M = host[0], mA = host[1];
      SG_TAKE(vc64, int64_t, 4 * static_cast<size_t>(M));
      SG_TAKE(rules, int32_t, static_cast<size_t>(M) * (mA + 1));
      SG_TRY(sg_voxelize_idx_fill(vox, nc_pts, 4, 4, l2p_, M, mA, vc64, rules, vx_ws, vx_bytes, stream_));
      SG_TAKE(cl, float, 3 * static_cast<size_t>(M));
      SG_TAKE(ol, float, 3 * static_cast<size_t>(M));
      SG_TAKE(ql, float, 3 * static_cast<size_t>(M));
      SG_TAKE(qbl, int32_t, M);
      SG_TRY(sg_voxelize_fp(c3, rules, M, mA, 3, 1, cl, stream_));
      SG_TRY(sg_voxelize_fp(o3, rules, M, mA, 3, 1, ol, stream_));
      pp_level_points_kernel<<<grid_for(M, 256, 1 << 22), 256, 0, stream>>>(cl, ol, vc64, M, ql, qbl);
      q = ql;
      qb = qbl;
      n_q = n_lvl = M;
      l2p = l2p_;
    }
    // ---- neighbour lists (functions.py:7-44): octree walk or hashed grid; count, scan, ONE read-back, fill
    SG_TAKE(start_len, int32_t, 2 * static_cast<size_t>(n_q));
    const size_t sc_bytes = sg_scan_workspace_bytes(n_q);
    SG_TAKE(sc_ws, char, sc_bytes);
    hipMemsetAsync(start_len, 0, sizeof(int32_t) * 2 * n_q, stream);
    int32_t *bq_idx = nullptr;
    int n_active = 0, flags = 0;
    if (pcfg->with_octree) {
      SG_TAKE(boxes, float, (1 + 8 + 64 + 512) * 6);
      SG_TAKE(pt_inds, int32_t, n_q);
      SG_TAKE(pt_sl, int32_t, 512 * 2);
      const size_t ob = sg_octree_build_workspace_bytes(n_q);
      SG_TAKE(o_ws, char, ob);
      SG_TAKE(stash, int32_t, octree_stash_bytes(n_q) / sizeof(int32_t));
      SG_TRY(sg_octree_build(q, n_q, boxes, pt_inds, pt_sl, o_ws, ob, stream_));
      // (the count pass parks lists of <= 64 neighbours; the fill pass copies those and walks only the longer ones)
      SG_TRY(octree_ballquery_count_stash(q, boxes, pt_inds, pt_sl, n_q, radius, start_len, stash, stream));
      SG_TRY(sg_exclusive_scan_startlen(start_len, n_q, meta + 56, sc_ws, sc_bytes, stream_));
      SG_TRY(read_back(host, meta + 56, 1, stream, kWhat));
      n_active = host[0];
      SG_TAKE(idx_, int32_t, n_active);
      SG_TRY(octree_ballquery_fill_stash(q, boxes, pt_inds, pt_sl, n_q, radius, start_len, stash, idx_, stream));
      bq_idx = idx_;
      flags = SG_LISTS_RADIUS;
    } else {
      const size_t bq_bytes = sg_ballquery_workspace_bytes(n_q);
      SG_TAKE(bq_ws, char, bq_bytes);
      SG_TRY(sg_ballquery_build_grid(q, qb, n_q, radius, bq_ws, bq_bytes, stream_));
      SG_TRY(sg_ballquery_count(q, qb, n_q, radius, start_len, nullptr, bq_ws, bq_bytes, stream_));
      SG_TRY(sg_exclusive_scan_startlen(start_len, n_q, meta + 56, sc_ws, sc_bytes, stream_));
      SG_TRY(read_back(host, meta + 56, 1, stream, kWhat));
      n_active = host[0];
      SG_TAKE(idx_, int32_t, n_active);
      SG_TRY(sg_ballquery_fill(q, qb, n_q, radius, start_len, idx_, bq_ws, bq_bytes, stream_));
      bq_idx = idx_;
      flags = SG_LISTS_SORTED | SG_LISTS_RADIUS;
    }
    n_nbr += n_active;
    // ---- clusters of the class (functions.py:278-308; threshold of the class = seg_thr[s])
    const size_t bfs_bytes = sg_bfs_workspace_bytes(n_q, n_active);
    SG_TAKE(bfs_ws, char, bfs_bytes);
    int32_t nc = 0, sp = 0;
    SG_TRY(sg_bfs_cluster_label(bq_idx, start_len, n_q, n_active, flags, nullptr, cfg->seg_thr + s, 1, &nc, &sp, bfs_ws,
                                bfs_bytes, stream_));
    if (sp == 0) continue;
    bool later = false;
    for (int t = s + 1; t < n_seg; ++t) later = later || count[t] > 0;
    const bool want_defer = defer_on && def_rows < 0 && later && bfs_label_max_kept(bfs_ws) > kBfsGiantMin;
    int rows = sp;
    if (want_defer && mapped) {
      SG_TRY(bfs_count_kept_members(bfs_ws, bfs_bytes, n_q, n_active, l2p, nc_pts, cfg->seg_thr + s, meta + 72, stream));
      SG_TRY(read_back(host, meta + 72, 1, stream, kWhat));
      rows = host[0];
    }
    SG_TAKE(cidx, int32_t, 2 * static_cast<size_t>(sp));
    SG_TAKE(coff, int32_t, static_cast<size_t>(nc) + 1);
    hipMemsetAsync(coff, 0, sizeof(int32_t) * (nc + 1), stream);
    if (hook_pending) {
      t_scan_emit_hook(t_scan_emit_ctx);
      hook_pending = false;
    }
    if (want_defer) bfs_emit_defer(&def);
    SG_TRY(sg_bfs_cluster_emit(bq_idx, start_len, n_q, n_active, nullptr, cfg->seg_thr + s, nc, sp, cidx, coff, bfs_ws,
                               bfs_bytes, stream_));
    const bool deferred = want_defer && def.deferred;      // (no side stream: the call joined as usual)
    hipStream_t tail = deferred ? def.side : stream;
    const int32_t *src_idx = cidx, *src_off = coff;
    if (mapped) {      // pyramid_inverse_map (:500-507): proposals over level voxels -> over the class's points
      SG_TAKE(oidx, int32_t, 2 * static_cast<size_t>(nc_pts));
      SG_TAKE(ooff, int32_t, static_cast<size_t>(nc) + 1);
      const size_t ib = sg_pyramid_inverse_map_workspace_bytes(nc_pts, n_lvl, nc);
      SG_TAKE(i_ws, char, ib);
      SG_TRY(sg_pyramid_inverse_map(cidx, sp, nc, l2p, nc_pts, n_lvl, oidx, ooff, deferred ? meta + 80 : meta + 48, i_ws,
                                    ib, reinterpret_cast<sg_stream_t>(tail)));
      if (!deferred) {
        SG_TRY(read_back(host, meta + 48, 1, stream, kWhat));
        rows = host[0];
      }
      src_idx = oidx;
      src_off = ooff;
    }
    if (deferred) {
      def_rows = rows;
      def_mapped = mapped;
      loop_mark = ar.off;      // the class's buffers stay until the join
    }
    if (rows == 0) continue;
    SG_REQUIRE(S + rows <= n_sel, "sg_scan_grouping_pp: more proposal rows than selected points");
    pp_append_kernel<<<grid_for(rows > nc ? rows : nc, 256, 1 << 22), 256, 0, tail>>>(src_idx, rows, src_off, nc, obj + s0,
                                                                                  n_prop, S, pairs, poff);
    n_prop += nc;
    S += rows;
  }
  res->deferred_classes = def_rows >= 0 ? 1 : 0;
  if (def.deferred) {
    SG_TRY(bfs_emit_join(def, stream));
    def.deferred = false;
    if (def_mapped) {      // the count taken from the labelling against the inverse map's own
      SG_TRY(read_back(host, meta + 80, 1, stream, kWhat));
      SG_REQUIRE(host[0] == def_rows, "sg_scan_grouping_pp: the deferred class's rows were predicted as %d, the inverse map made %d",
                 def_rows, host[0]);
    }
  }
  ar.off = loop_mark0;
  res->n_neighbours = static_cast<int>(n_nbr < 0x7fffffff ? n_nbr : 0x7fffffff);
  res->n_proposals = n_prop;
  res->sum_npoint = S;
  if (S == 0) return check_launch(kWhat);
  res->proposals_idx = ar.at(pairs);
  res->proposals_offset = ar.at(poff);
  if (cfg->voxel_shape == 0) {
    res->arena_used = ar.off;
    return check_launch(kWhat);
  }
  return voxelise_proposals(cfg, pairs, poff, n_prop, S, coords_float, point_feats, ar, meta, host, res, stream_, kWhat);
}

// ---- results -------------------------------------------------------------------------------------
int sg_scan_instances(const sg_instances_cfg *cfg, const int32_t *proposals_idx, const float *mask_scores,
                      const float *cls_prob, const float *iou_scores, void *arena, size_t arena_bytes,
                      void *host_out, size_t host_bytes, sg_instances_result *res, sg_stream_t stream_) {
  static const char *kWhat = "sg_scan_instances";
  SG_REQUIRE(cfg != nullptr && res != nullptr, "sg_scan_instances: null descriptor");
  SG_REQUIRE(cfg->n_classes >= 1 && cfg->score_stride >= cfg->n_classes && cfg->n_proposals >= 0 &&
                 cfg->sum_npoint >= 0 && cfg->n_points >= 0,
             "sg_scan_instances: bad configuration");
  memset(res, 0, sizeof(*res));
  hipStream_t stream = as_stream(stream_);
  int32_t *host = pinned_words();
  SG_REQUIRE(host != nullptr, "sg_scan_instances: no pinned host words");
  ScanArena ar(arena, arena_bytes);
  const int nP = cfg->n_proposals, nc = cfg->n_classes, stride = cfg->score_stride;
  const int64_t S = cfg->sum_npoint;
  if (nP == 0 || S == 0) return SG_OK;
  SG_TAKE(npoint, int32_t, static_cast<size_t>(nP) * nc);
  SG_TAKE(inst_of, int32_t, static_cast<size_t>(nP) * nc);
  SG_TAKE(kept_cls, int32_t, static_cast<size_t>(nP) * nc);
  SG_TAKE(kept_score, float, static_cast<size_t>(nP) * nc);
  SG_TAKE(head, int32_t, 64);
  SG_TRY(sg_instance_npoint(proposals_idx, mask_scores, S, stride, nc, cfg->mask_score_thr, nP, npoint, stream_));
  instance_keep_kernel<<<1, 1024, 0, stream>>>(cls_prob, iou_scores, stride, npoint, nP, nc, cfg->cls_score_thr,
                                             cfg->min_npoint, inst_of, kept_cls, kept_score, head);
  SG_TRY(check_launch(kWhat));
  SG_TRY(read_back(host, head, 2, stream, kWhat));
  const int n_kept = host[0];
  const int64_t cap = host[1] > 0 ? host[1] : 1;
  res->n_kept = n_kept;
  if (n_kept == 0) return SG_OK;
  SG_TAKE(starts, int32_t, cap);
  SG_TAKE(ends, int32_t, cap);
  SG_TAKE(bounds, int64_t, static_cast<size_t>(n_kept) + 1);
  const size_t rws_bytes = sg_instance_runs_workspace_bytes(n_kept, cfg->n_points);
  SG_TAKE(rws, char, rws_bytes);
  SG_TRY(sg_instance_runs(proposals_idx, mask_scores, S, stride, nc, cfg->mask_score_thr, inst_of, nP, n_kept,
                          cfg->n_points, starts, ends, bounds, cap, rws, rws_bytes, stream_));
  const int64_t tcap = sg_rle_format_device_text_bytes(cap, cfg->n_points);
  SG_TAKE(text, uint8_t, static_cast<size_t>(tcap));
  SG_TAKE(text_off, int64_t, static_cast<size_t>(n_kept) + 1);
  const size_t fws_bytes = sg_rle_format_device_workspace_bytes(cap);
  SG_TAKE(fws, char, fws_bytes);
  SG_TRY(sg_rle_format_device(starts, ends, bounds, n_kept, cap, cfg->n_points, text, tcap, text_off, fws,
                              fws_bytes, stream_));
  // ---- host image: [text_off int64 x (n_kept+1)] [class int32 x n_kept] [score f32 x n_kept] [text]
  const size_t off_cls = sizeof(int64_t) * (static_cast<size_t>(n_kept) + 1);
  const size_t off_score = off_cls + sizeof(int32_t) * static_cast<size_t>(n_kept);
  const size_t off_text = align_up(off_score + sizeof(float) * static_cast<size_t>(n_kept), 8);
  res->host_needed = off_text + static_cast<size_t>(tcap);
  if (host_out == nullptr || host_bytes < off_text + 16) {
    set_error("sg_scan_instances: host buffer too small (%zu bytes, need up to %zu)", host_bytes, res->host_needed);
    return SG_ERR_WORKSPACE;
  }
  char *h = static_cast<char *>(host_out);
  if (hipMemcpyAsync(h, text_off, off_cls, hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipMemcpyAsync(h + off_cls, kept_cls, sizeof(int32_t) * n_kept, hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipMemcpyAsync(h + off_score, kept_score, sizeof(float) * n_kept, hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess) {
    set_error("sg_scan_instances: result copy failed");
    return SG_ERR_LAUNCH;
  }
  const int64_t text_bytes = reinterpret_cast<const int64_t *>(h)[n_kept];
  res->host_needed = off_text + static_cast<size_t>(text_bytes);
  if (host_bytes < res->host_needed) {
    set_error("sg_scan_instances: host buffer too small (%zu bytes, need %zu)", host_bytes, res->host_needed);
    return SG_ERR_WORKSPACE;
  }
  if (text_bytes > 0 &&
      (hipMemcpyAsync(h + off_text, text, static_cast<size_t>(text_bytes), hipMemcpyDeviceToHost, stream) != hipSuccess ||
       hipStreamSynchronize(stream) != hipSuccess)) {
    set_error("sg_scan_instances: text copy failed");
    return SG_ERR_LAUNCH;
  }
  res->bits = ar.at(rws);          // first block of sg_instance_runs' workspace: the bit rows
  res->label_id = ar.at(kept_cls);
  res->off_class = off_cls;
  res->off_score = off_score;
  res->off_text = off_text;
  res->text_bytes = static_cast<size_t>(text_bytes);
  res->arena_used = ar.off;
  return SG_OK;
}

}  // extern "C"
