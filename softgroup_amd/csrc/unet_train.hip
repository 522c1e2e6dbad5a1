// unet_train.hip -- native TRAINING executor of the sparse U-Net: sg_unet_train_forward /
// sg_unet_train_backward (include/softgroup_hip.h).  What the DDP training step spends on the
// U-Net modules (tools/train.py:44-62 -> SoftGroup.forward_train, softgroup.py:113-150; modules of
// softgroup/model/blocks.py:44-143 in train() mode): the reference -- and this package's module
// path -- go through the interpreter, the autograd graph and torch.nn.BatchNorm1d once per layer,
// ~400 launches and ~7 ms of host time per step for the tiny U-Net of the refinement head alone.
// Here the forward is one C call and the backward is another:
//   * the forward walks the UBlock recursion (the same one as unet_exec.hip) and records every
//     operation on a TAPE: BatchNorm1d(batch statistics)+ReLU, sparse conv (+ residual), concat.
//     Activations, statistics, gather tables and plans stay in the caller's arena (bump allocated,
//     nothing is recycled before the backward has run);
//   * the backward replays the tape in reverse.  Input gradient of a conv = the forward conv kernel
//     on the transposed rulebook (SubM: the same table with mirrored offsets; strided conv <->
//     inverse conv: each other's tables, which the pyramid build has already made) with the
//     transposed weights; weight gradient = sg_spconv_wgrad (fixed-order chunk sums); BatchNorm1d
//     gradient from two column sums.  Gradients of a tensor with several consumers are chained
//     through the `residual` input of the kernels (out = previous + contribution), never in place.
//   * column sums (BatchNorm statistics, its gradient sums) are accumulated in fp64 per thread,
//     combined in a fixed order, and finalised by the last workgroup to arrive (one launch);
//     nothing uses floating-point atomics: the step is bit-reproducible.
// Arena need is computed exactly by a dry run of forward + backward once the level row counts are
// known (the only host synchronisation of the step, inside the shared index build).
#include <iterator>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"
#include "unet_common.h"

namespace sg {

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
constexpr int kStatBlocksMax = 512;
constexpr int kStatGroup = 16;          // workgroups whose partials one workgroup adds in the first stage
constexpr int kStatCounters = 1 + kStatBlocksMax / kStatGroup;      // counters of one column_sums call

__device__ __forceinline__ void st_agent(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Column sums of two per-element quantities over [rows, c] (c % 4 == 0): thread (lane, g) walks rows
// lane, lane + lanes, ... of its workgroup's row range for the channel group g (4 channels), FOUR rows in
// flight (their loads are independent; one row at a time the walk was a chain of memory round trips: 25-27 us
// per call whatever the size, 4.2 ms of the 19 ms of a training step -- profiles/r06_train_profile.txt), fp64
// accumulators; lanes meet in LDS in lane order; workgroup partials go to `partial` [gridDim.x][c][2] with
// agent-scope stores; they meet in two stages (below): the last workgroup of every group of kStatGroup adds its
// group's partials, the last group to finish adds the group sums and calls `fin(channel, sum0, sum1)`.  Every
// order is fixed: the sums are bit-reproducible.
template <typename Elem, typename Fin>
__device__ __forceinline__ void column_sums(int64_t rows, int c, double *partial, unsigned *counter, Elem elem,
                                            Fin fin) {
  __shared__ double sh[256 * 8];
  __shared__ bool is_last;
  const int c4 = c >> 2;
  const int lanes = 256 / c4 > 0 ? 256 / c4 : 1;
  const int tid = threadIdx.x;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = per * blockIdx.x, r1 = r0 + per < rows ? r0 + per : rows;
  for (int g0 = 0; g0 < c4; g0 += 256) {       // (c4 > 256 only for c > 1024: several passes)
    const int g = g0 + (c4 >= 256 ? tid : tid % c4);
    const int lane = c4 >= 256 ? 0 : tid / c4;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < lanes && g < c4) {
      int64_t r = r0 + lane;
      for (; r + 3 * lanes < r1; r += 4 * lanes) {
        double a[4][4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) elem(r + u * lanes, g, a[u], b[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s[j] += a[u][j];
            s[4 + j] += b[u][j];
          }
      }
      for (; r < r1; r += lanes) {
        double a[4], b[4];
        elem(r, g, a, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] += a[j];
          s[4 + j] += b[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[tid * 8 + j] = s[j];
    __syncthreads();
    if (lane == 0 && g < c4) {
      const int gl = c4 >= 256 ? tid : tid % c4;
      double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int l = 0; l < (c4 >= 256 ? 1 : lanes); ++l)
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] += sh[(l * (c4 >= 256 ? 0 : c4) + gl) * 8 + j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st_agent(&partial[(static_cast<int64_t>(blockIdx.x) * c + g * 4 + j) * 2 + 0], t[j]);
        st_agent(&partial[(static_cast<int64_t>(blockIdx.x) * c + g * 4 + j) * 2 + 1], t[4 + j]);
      }
    }
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's partial stores are out before the arrival below
  __syncthreads();
  // ---- two-stage meeting: the last workgroup of every group of kStatGroup adds its group's partials, the
  //      last group to finish adds the group sums (one stage over ~500 partials was a 7-round chain of loads)
  const int nb = static_cast<int>(gridDim.x);
  const int ngroups = (nb + kStatGroup - 1) / kStatGroup;
  const int grp = blockIdx.x / kStatGroup;
  const int gfirst = grp * kStatGroup, gcount = min(kStatGroup, nb - gfirst);
  double *gpartial = partial + static_cast<int64_t>(nb) * c * 2;      // [ngroups][c][2]
  const int cc = c < 256 ? c : 256;             // channels per pass
  const int nq = 256 / cc;                      // threads per channel
  // rows [first, first + count) of src[.][c][2] summed per channel: thread (channel, q) the rows q, q + nq, ...
  // in ascending order, eight loads in flight; the nq sums meet in LDS in q order
  auto sum_rows = [&](const double *src, int first, int count, auto &&sink) {
    for (int ch0 = 0; ch0 < c; ch0 += cc) {
      const int ch = ch0 + tid % cc, q = tid / cc;
      double s0 = 0, s1 = 0;
      if (q < nq && ch < c) {
        int i = q;
        for (; i + 7 * nq < count; i += 8 * nq) {
          double v0[8], v1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v0[u] = ld_agent(&src[(static_cast<int64_t>(first + i + u * nq) * c + ch) * 2 + 0]);
            v1[u] = ld_agent(&src[(static_cast<int64_t>(first + i + u * nq) * c + ch) * 2 + 1]);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            s0 += v0[u];
            s1 += v1[u];
          }
        }
        for (; i < count; i += nq) {
          s0 += ld_agent(&src[(static_cast<int64_t>(first + i) * c + ch) * 2 + 0]);
          s1 += ld_agent(&src[(static_cast<int64_t>(first + i) * c + ch) * 2 + 1]);
        }
      }
      sh[tid * 2] = s0;
      sh[tid * 2 + 1] = s1;
      __syncthreads();
      if (q == 0 && ch < c) {
        double t0 = 0, t1 = 0;
        for (int k = 0; k < nq; ++k) {
          t0 += sh[(k * cc + tid) * 2];
          t1 += sh[(k * cc + tid) * 2 + 1];
        }
        sink(ch, t0, t1);
      }
      __syncthreads();
    }
  };
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counter + 1 + grp, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = prev == static_cast<unsigned>(gcount - 1);
  }
  __syncthreads();
  if (!is_last) return;
  if (ngroups == 1) {
    sum_rows(partial, 0, nb, fin);
    return;
  }
  sum_rows(partial, gfirst, gcount, [&](int ch, double t0, double t1) {
    st_agent(&gpartial[(static_cast<int64_t>(grp) * c + ch) * 2 + 0], t0);
    st_agent(&gpartial[(static_cast<int64_t>(grp) * c + ch) * 2 + 1], t1);
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = prev == static_cast<unsigned>(ngroups - 1);
  }
  __syncthreads();
  if (!is_last) return;
  sum_rows(gpartial, 0, ngroups, fin);
}

// BatchNorm1d, train() mode: batch mean / biased variance per channel; running statistics updated
// (torch.nn.functional.batch_norm semantics: unbiased variance into running_var); the affine the
// apply kernel uses: y = x * scale + shift with scale = weight * invstd, shift = bias - mean * scale
__global__ void __launch_bounds__(256) bn_stats_kernel(const float *__restrict__ x, int64_t rows, int c,
                                                      const float *__restrict__ weight,
                                                      const float *__restrict__ bias, float *running_mean,
                                                      float *running_var, float momentum, float eps,
                                                      double *partial, unsigned *counter, float *mean_out,
                                                      float *invstd_out, float *scale_out, float *shift_out) {
  column_sums(
      rows, c, partial, counter,
      [&](int64_t r, int g, double (&a)[4], double (&b)[4]) {
        const float4 v = reinterpret_cast<const float4 *>(x)[r * (c >> 2) + g];
        a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = a[j] * a[j];      // exact in fp64
      },
      [&](int ch, double s, double q) {
        const double n = static_cast<double>(rows);
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0) var = 0;
        const float invstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
        const float meanf = static_cast<float>(mean);
        mean_out[ch] = meanf;
        invstd_out[ch] = invstd;
        const float sc = weight[ch] * invstd;
        scale_out[ch] = sc;
        shift_out[ch] = bias[ch] - meanf * sc;
        if (running_mean != nullptr) {
          const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
          running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * meanf;
          running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * static_cast<float>(unbiased);
        }
      });
}

// gradient of y = relu(bn(x)) wrt the BatchNorm inputs: dy = g where y > 0;
// g_bias = sum dy, g_weight = sum dy * xhat; the apply kernel needs per channel
// coef = (weight * invstd, g_bias / N, g_weight / N)
__global__ void __launch_bounds__(256) bn_bwd_sums_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                         const float *__restrict__ x, int64_t rows, int c,
                                                         const float *__restrict__ mean,
                                                         const float *__restrict__ invstd,
                                                         const float *__restrict__ weight, double *partial,
                                                         unsigned *counter, float *g_weight, float *g_bias,
                                                         float *coef) {
  column_sums(
      rows, c, partial, counter,
      [&](int64_t r, int gidx, double (&a)[4], double (&b)[4]) {
        const int64_t at = r * (c >> 2) + gidx;
        const float4 gv = reinterpret_cast<const float4 *>(g)[at];
        const float4 yv = reinterpret_cast<const float4 *>(y)[at];
        const float4 xv = reinterpret_cast<const float4 *>(x)[at];
        const float4 m = reinterpret_cast<const float4 *>(mean)[gidx];
        const float4 is = reinterpret_cast<const float4 *>(invstd)[gidx];
        const float d0 = yv.x > 0.f ? gv.x : 0.f, d1 = yv.y > 0.f ? gv.y : 0.f;
        const float d2 = yv.z > 0.f ? gv.z : 0.f, d3 = yv.w > 0.f ? gv.w : 0.f;
        a[0] = d0; a[1] = d1; a[2] = d2; a[3] = d3;
        b[0] = static_cast<double>(d0) * ((xv.x - m.x) * is.x); b[1] = static_cast<double>(d1) * ((xv.y - m.y) * is.y);
        b[2] = static_cast<double>(d2) * ((xv.z - m.z) * is.z); b[3] = static_cast<double>(d3) * ((xv.w - m.w) * is.w);
      },
      [&](int ch, double s1, double s2) {
        if (g_bias != nullptr) g_bias[ch] = static_cast<float>(s1);
        if (g_weight != nullptr) g_weight[ch] = static_cast<float>(s2);
        const double n = static_cast<double>(rows);
        coef[ch] = weight[ch] * invstd[ch];
        coef[c + ch] = static_cast<float>(s1 / n);
        coef[2 * c + ch] = static_cast<float>(s2 / n);
      });
}

// dx = add + weight*invstd * (dy - g_bias/N - xhat * g_weight/N)
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float4 *__restrict__ g, const float4 *__restrict__ y,
                                                          const float4 *__restrict__ x, int64_t rows, int c4,
                                                          const float4 *__restrict__ mean,
                                                          const float4 *__restrict__ invstd,
                                                          const float4 *__restrict__ coef,
                                                          const float4 *__restrict__ add, float4 *__restrict__ dx) {
  const int64_t total = rows * c4;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int gi = static_cast<int>(t % c4);
    const float4 gv = g[t], yv = y[t], xv = x[t];
    const float4 m = mean[gi], is = invstd[gi], a = coef[gi], b = coef[c4 + gi], cc = coef[2 * c4 + gi];
    float4 o = add != nullptr ? add[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x += a.x * ((yv.x > 0.f ? gv.x : 0.f) - b.x - (xv.x - m.x) * is.x * cc.x);
    o.y += a.y * ((yv.y > 0.f ? gv.y : 0.f) - b.y - (xv.y - m.y) * is.y * cc.y);
    o.z += a.z * ((yv.z > 0.f ? gv.z : 0.f) - b.z - (xv.z - m.z) * is.z * cc.z);
    o.w += a.w * ((yv.w > 0.f ? gv.w : 0.f) - b.w - (xv.w - m.w) * is.w * cc.w);
    dx[t] = o;
  }
}

__global__ void __launch_bounds__(256) cat2_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                  int64_t rows, int ca4, int cb4, float4 *__restrict__ out) {
  const int c4 = ca4 + cb4;
  const int64_t total = rows * c4;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t r = t / c4;
    const int c = static_cast<int>(t - r * c4);
    out[t] = c < ca4 ? a[r * ca4 + c] : b[r * cb4 + (c - ca4)];
  }
}
// the reverse: ga = g[:, :ca] (+ adda), gb = g[:, ca:] (+ addb)
__global__ void __launch_bounds__(256) split2_kernel(const float4 *__restrict__ g, int64_t rows, int ca4, int cb4,
                                                    const float4 *__restrict__ adda,
                                                    const float4 *__restrict__ addb, float4 *__restrict__ ga,
                                                    float4 *__restrict__ gb) {
  const int c4 = ca4 + cb4;
  const int64_t total = rows * c4;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t r = t / c4;
    const int c = static_cast<int>(t - r * c4);
    float4 v = g[t];
    if (c < ca4) {
      const int64_t at = r * ca4 + c;
      if (adda != nullptr) { const float4 o = adda[at]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      ga[at] = v;
    } else {
      const int64_t at = r * cb4 + (c - ca4);
      if (addb != nullptr) { const float4 o = addb[at]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      gb[at] = v;
    }
  }
}
__global__ void __launch_bounds__(256) add2_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                  int64_t n4, float4 *__restrict__ out) {
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < n4; t += gridDim.x * 256LL) {
    const float4 u = a[t], v = b[t];
    out[t] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}
// ---------------------------------------------------------------------------------------------
// tape
// ---------------------------------------------------------------------------------------------
enum PlanKind { PK_SUBM = 0, PK_DOWN = 1, PK_UP = 2, PK_IDENT = 3 };

struct Tensor {
  float *p = nullptr;
  int rows = 0, c = 0;
  float *g = nullptr;      // backward: gradient accumulated so far
  bool needs_grad = true;
};

struct Op {
  enum Kind { BN, CONV, CAT } kind = BN;
  int in = -1, in2 = -1, out = -1;      // in2: CAT's second input, CONV's residual
  sg_train_bn bn{};
  float *mean = nullptr, *invstd = nullptr;
  sg_train_conv cv{};
  int level = 0, plan_kind = PK_SUBM, cin = 0, cout = 0;
};

struct Tape {
  std::vector<Tensor> t;
  std::vector<Op> ops;
  LevelIdx li[SG_PYRAMID_MAX_LEVELS];
  const int32_t *nbr_t[SG_PYRAMID_MAX_LEVELS][4] = {};
  int n_levels = 0, arithmetic = 0;
  char *arena = nullptr;
  size_t arena_bytes = 0, off = 0;       // bump offset where the backward continues
  unsigned *counters = nullptr;
  int counters_used = 0;
  int root = -1, result = -1;
};
constexpr int kCounters = 16384;

struct TrainExec {
  Tape &tp;
  Arena ar;
  sg_stream_t stream;
  bool dry;
  const char *who;

  TrainExec(Tape &tape, void *arena, size_t bytes, size_t start, sg_stream_t s, bool dry_, const char *w)
      : tp(tape), ar(arena, dry_ ? ~static_cast<size_t>(0) >> 1 : bytes), stream(s), dry(dry_), who(w) {
    ar.off = start;
  }

  template <typename T>
  T *take(size_t count) {
    T *r = ar.take<T>(count ? count : 1);
    return r;
  }
#define SG_TALLOC(var, T, count)                                                       \
  T *var = take<T>(count);                                                             \
  if (var == nullptr) {                                                                \
    set_error("%s: arena too small (%zu bytes)", who, ar.cap);                         \
    return SG_ERR_WORKSPACE;                                                           \
  }

  hipStream_t hs() const { return as_stream(stream); }
  const Plan &plan(int level, int kind) const {
    const LevelIdx &I = tp.li[level];
    return kind == PK_SUBM ? I.subm : kind == PK_DOWN ? I.down : kind == PK_UP ? I.up : I.ident;
  }
  // plan of the transposed convolution: SubM / 1x1 their own, strided <-> inverse each other's
  const Plan &plan_t(int level, int kind) const {
    return plan(level, kind == PK_DOWN ? PK_UP : kind == PK_UP ? PK_DOWN : kind);
  }

  int new_tensor(float *p, int rows, int c) {
    Tensor t;
    t.p = p; t.rows = rows; t.c = c;
    tp.t.push_back(t);
    return static_cast<int>(tp.t.size()) - 1;
  }
  int alloc_tensor(int rows, int c, int *id) {
    SG_TALLOC(p, float, static_cast<size_t>(rows) * c);
    *id = new_tensor(p, rows, c);
    return SG_OK;
  }
  unsigned *next_counter() {      // (kStatCounters words: the top counter and one per group of workgroups)
    if (tp.counters_used + kStatCounters > kCounters) tp.counters_used = 0;
    unsigned *c = tp.counters + tp.counters_used;
    tp.counters_used += kStatCounters;
    return c;
  }
  static int stat_blocks(int64_t rows, int c) {
    const int c4 = c >> 2;
    const int lanes = 256 / c4 > 0 ? 256 / c4 : 1;
    int64_t b = (rows + static_cast<int64_t>(lanes) * 8 - 1) / (static_cast<int64_t>(lanes) * 8);
    if (b < 1) b = 1;
    return static_cast<int>(b > kStatBlocksMax ? kStatBlocksMax : b);
  }

  // ---- every conv weight of a pass packed by ONE launch per 64 (spconv_pack_weights): the forward walks the
  //      descriptor, the backward the tape; conv_launch finds the packed copy by (weights, mode)
  std::map<std::pair<const float *, int>, float *> packed;
  std::vector<PackJob> jobs;
  int pack_job(const float *w, int cout, int kvol, int cin, int mode) {
    if (w == nullptr || packed.count({w, mode})) return SG_OK;
    SG_TALLOC(wp, float, sg_spconv_packed_weight_elems(kvol, cin, cout));
    packed[{w, mode}] = wp;
    jobs.push_back(PackJob{w, wp, cout, kvol, cin, mode});
    return SG_OK;
  }
  int pack_flush() {
    int rc = SG_OK;
    if (!dry && !jobs.empty()) rc = spconv_pack_weights(jobs.data(), static_cast<int>(jobs.size()), stream);
    jobs.clear();
    return rc;
  }
  int prepack_block(const sg_unet_train_block &b, int l) {
    SG_TRY(pack_job(b.c1.w, b.cout, plan(l, PK_SUBM).kvol, b.cin, 0));
    if (b.ci.w != nullptr) SG_TRY(pack_job(b.ci.w, b.cout, plan(l, PK_IDENT).kvol, b.cin, 0));
    SG_TRY(pack_job(b.c2.w, b.cout, plan(l, PK_SUBM).kvol, b.cout, 0));
    return SG_OK;
  }
  int prepack_forward(const sg_unet_train_desc *d) {
    const int c0 = d->levels[0].planes;
    if (d->input.w != nullptr) SG_TRY(pack_job(d->input.w, c0, plan(0, PK_SUBM).kvol, d->input_cin, 0));
    for (int l = 0; l < d->n_levels; ++l) {
      const sg_unet_train_level &L = d->levels[l];
      for (int i = 0; i < L.n_blocks; ++i) SG_TRY(prepack_block(L.blocks[i], l));
      if (l + 1 < d->n_levels) {
        const int c = L.planes, c2 = d->levels[l + 1].planes;
        SG_TRY(pack_job(L.down.w, c2, plan(l, PK_DOWN).kvol, c, 0));
        SG_TRY(pack_job(L.up.w, c, plan(l, PK_UP).kvol, c2, 0));
        for (int i = 0; i < L.n_blocks; ++i) SG_TRY(prepack_block(L.tail[i], l));
      }
    }
    return pack_flush();
  }
  int prepack_backward() {
    for (const Op &op : tp.ops) {
      if (op.kind != Op::CONV || !tp.t[op.in].needs_grad) continue;
      // (the transposed conv: this conv's Cout = the layer's Cin -- conv_launch's arguments in backward_op)
      SG_TRY(pack_job(op.cv.w, op.cin, plan_t(op.level, op.plan_kind).kvol, op.cout, op.plan_kind == PK_SUBM ? 3 : 2));
    }
    return pack_flush();
  }

  // ---- raw conv launch (forward convs and input gradients): packs `w` with `pack_mode` first (unless prepacked)
  int conv_launch(const float *in, int in_rows, const Plan &p, int cin, int cout, const float *w_raw, int pack_mode,
                  const float *residual, float *out) {
    float *wp = nullptr;
    const auto hit = packed.find({w_raw, pack_mode});
    const bool prepacked = hit != packed.end();
    if (prepacked) {
      wp = hit->second;
    } else {
      wp = take<float>(sg_spconv_packed_weight_elems(p.kvol, cin, cout));
      if (wp == nullptr) {
        set_error("%s: arena too small (%zu bytes)", who, ar.cap);
        return SG_ERR_WORKSPACE;
      }
    }
    const size_t m = ar.mark();
    const size_t nb = sg_spconv_conv_workspace_bytes(p.rows, cout);
    void *ws = nullptr;
    if (nb > 256) {
      ws = take<char>(nb);
      if (ws == nullptr) {
        set_error("%s: arena too small (%zu bytes)", who, ar.cap);
        return SG_ERR_WORKSPACE;
      }
    }
    int rc = SG_OK;
    if (!dry && p.rows > 0) {
      if (!prepacked) rc = sg_spconv_pack_weight(w_raw, cout, p.kvol, cin, pack_mode, wp, stream);
      if (rc == SG_OK)
        rc = sg_spconv_gather_conv_f32(in, in_rows, p.nbr, p.rows, p.kvol, cin, cout, wp, nullptr, nullptr, residual,
                                       nullptr, nullptr, nullptr, p.order, p.tile_mask, p.nbr_tiles, out, ws,
                                       ws ? nb : 0, stream);
    }
    ar.release(m);      // (scratch only; the packed weights stay: the kernel reads them asynchronously)
    return rc;
  }

  // ---- forward ops
  int bn_relu(int x, const sg_train_bn &bn, int *out_id, float *out_ptr = nullptr) {
    const Tensor X = tp.t[x];
    Op op;
    op.kind = Op::BN; op.in = x; op.bn = bn;
    SG_TALLOC(stats, float, 4 * static_cast<size_t>(X.c));
    op.mean = stats; op.invstd = stats + X.c;
    float *scale = stats + 2 * X.c, *shift = stats + 3 * X.c;
    int y;
    if (out_ptr != nullptr) y = new_tensor(out_ptr, X.rows, X.c);
    else SG_TRY(alloc_tensor(X.rows, X.c, &y));
    op.out = y;
    const int nb = stat_blocks(X.rows, X.c);
    SG_TALLOC(partial, double, static_cast<size_t>(nb + (nb + kStatGroup - 1) / kStatGroup) * X.c * 2);
    if (!dry && X.rows > 0) {
      bn_stats_kernel<<<nb, 256, 0, hs()>>>(X.p, X.rows, X.c, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                            bn.momentum, bn.eps, partial, next_counter(), op.mean, op.invstd, scale,
                                            shift);
      SG_TRY(sg_bn_relu_f32(X.p, scale, shift, X.rows, X.c, 1, tp.t[y].p, stream));
    }
    tp.ops.push_back(op);
    *out_id = y;
    return SG_OK;
  }

  int conv(int x, int level, int kind, const sg_train_conv &cv, int cin, int cout, int residual, int *out_id) {
    const Tensor X = tp.t[x];
    const Plan &p = plan(level, kind);
    Op op;
    op.kind = Op::CONV; op.in = x; op.in2 = residual; op.cv = cv; op.level = level; op.plan_kind = kind;
    op.cin = cin; op.cout = cout;
    int y;
    SG_TRY(alloc_tensor(p.rows, cout, &y));
    op.out = y;
    SG_TRY(conv_launch(X.p, X.rows, p, cin, cout, cv.w, 0, residual >= 0 ? tp.t[residual].p : nullptr, tp.t[y].p));
    tp.ops.push_back(op);
    *out_id = y;
    return SG_OK;
  }

  int cat(int a, int b, int *out_id) {
    const Tensor A = tp.t[a], B = tp.t[b];
    Op op;
    op.kind = Op::CAT; op.in = a; op.in2 = b;
    int y;
    SG_TRY(alloc_tensor(A.rows, A.c + B.c, &y));
    op.out = y;
    if (!dry && A.rows > 0)
      cat2_kernel<<<grid_for(static_cast<int64_t>(A.rows) * ((A.c + B.c) / 4), 256), 256, 0, hs()>>>(
          reinterpret_cast<const float4 *>(A.p), reinterpret_cast<const float4 *>(B.p), A.rows, A.c / 4, B.c / 4,
          reinterpret_cast<float4 *>(tp.t[y].p));
    tp.ops.push_back(op);
    *out_id = y;
    return SG_OK;
  }

  // ResidualBlock (blocks.py:44-79)
  int block(const sg_unet_train_block &b, int x, int level, int *out_id) {
    int xa, h, ha, sc = x, y;
    SG_TRY(bn_relu(x, b.bn1, &xa));
    SG_TRY(conv(xa, level, PK_SUBM, b.c1, b.cin, b.cout, -1, &h));
    SG_TRY(bn_relu(h, b.bn2, &ha));
    if (b.ci.w != nullptr) SG_TRY(conv(x, level, PK_IDENT, b.ci, b.cin, b.cout, -1, &sc));
    SG_TRY(conv(ha, level, PK_SUBM, b.c2, b.cout, b.cout, sc, &y));
    *out_id = y;
    return SG_OK;
  }

  // UBlock (blocks.py:82-143)
  int level(const sg_unet_train_desc *d, int l, int x, int *out_id) {
    const sg_unet_train_level &L = d->levels[l];
    int cur = x;
    for (int i = 0; i < L.n_blocks; ++i) SG_TRY(block(L.blocks[i], cur, l, &cur));
    if (l + 1 < d->n_levels) {
      const int c = L.planes, c2 = d->levels[l + 1].planes;
      int da, y, z, za, upf, ct;
      SG_TRY(bn_relu(cur, L.down_bn, &da));
      SG_TRY(conv(da, l, PK_DOWN, L.down, c, c2, -1, &y));
      SG_TRY(level(d, l + 1, y, &z));
      SG_TRY(bn_relu(z, L.up_bn, &za));
      SG_TRY(conv(za, l, PK_UP, L.up, c2, c, -1, &upf));
      SG_TRY(cat(cur, upf, &ct));
      cur = ct;
      for (int i = 0; i < L.n_blocks; ++i) SG_TRY(block(L.tail[i], cur, l, &cur));
    }
    *out_id = cur;
    return SG_OK;
  }

  int forward(const sg_unet_train_desc *d, const float *feats, int num_rows, float *out) {
    const int c0 = d->levels[0].planes;
    int x = new_tensor(const_cast<float *>(feats), num_rows, d->input.w != nullptr ? d->input_cin : c0);
    tp.root = x;
    packed.clear();
    SG_TRY(prepack_forward(d));
    if (d->input.w != nullptr) SG_TRY(conv(x, 0, PK_SUBM, d->input, d->input_cin, c0, -1, &x));
    int y;
    SG_TRY(level(d, 0, x, &y));
    if (d->out_bn.weight != nullptr) {
      SG_TRY(bn_relu(y, d->out_bn, &y, out));
    } else if (!dry && num_rows > 0) {
      if (hipMemcpyAsync(out, tp.t[y].p, static_cast<size_t>(num_rows) * c0 * 4, hipMemcpyDeviceToDevice, hs()) !=
          hipSuccess) {
        set_error("%s: copying the result failed", who);
        return SG_ERR_LAUNCH;
      }
    }
    tp.result = y;
    return SG_OK;
  }

  // ---- backward: gradient `g` arrives at tensor `id`
  int accumulate(int id, const float *g) {
    Tensor &T = tp.t[id];
    if (!T.needs_grad) return SG_OK;
    if (T.g == nullptr) {
      T.g = const_cast<float *>(g);      // alias: gradient buffers are never written twice
      return SG_OK;
    }
    SG_TALLOC(s, float, static_cast<size_t>(T.rows) * T.c);
    const int64_t n4 = static_cast<int64_t>(T.rows) * T.c / 4;
    if (!dry && n4 > 0)
      add2_kernel<<<grid_for(n4, 256), 256, 0, hs()>>>(reinterpret_cast<const float4 *>(T.g),
                                                       reinterpret_cast<const float4 *>(g), n4,
                                                       reinterpret_cast<float4 *>(s));
    T.g = s;
    return SG_OK;
  }

  int backward_op(const Op &op) {
    const Tensor O = tp.t[op.out];
    if (O.g == nullptr && !dry) return SG_OK;      // nothing downstream asked for this value
    if (op.kind == Op::BN) {
      Tensor &X = tp.t[op.in];
      const bool want_params = op.bn.g_weight != nullptr || op.bn.g_bias != nullptr;
      if (!X.needs_grad && !want_params) return SG_OK;
      SG_TALLOC(coef, float, 3 * static_cast<size_t>(X.c));
      const int nb = stat_blocks(X.rows, X.c);
      SG_TALLOC(partial, double, static_cast<size_t>(nb + (nb + kStatGroup - 1) / kStatGroup) * X.c * 2);
      if (!dry && X.rows > 0)
        bn_bwd_sums_kernel<<<nb, 256, 0, hs()>>>(O.g, O.p, X.p, X.rows, X.c, op.mean, op.invstd, op.bn.weight, partial,
                                                 next_counter(), op.bn.g_weight, op.bn.g_bias, coef);
      if (!X.needs_grad) return SG_OK;
      SG_TALLOC(dx, float, static_cast<size_t>(X.rows) * X.c);
      if (!dry && X.rows > 0)
        bn_bwd_apply_kernel<<<grid_for(static_cast<int64_t>(X.rows) * (X.c / 4), 256), 256, 0, hs()>>>(
            reinterpret_cast<const float4 *>(O.g), reinterpret_cast<const float4 *>(O.p),
            reinterpret_cast<const float4 *>(X.p), X.rows, X.c / 4, reinterpret_cast<const float4 *>(op.mean),
            reinterpret_cast<const float4 *>(op.invstd), reinterpret_cast<const float4 *>(coef),
            reinterpret_cast<const float4 *>(X.g), reinterpret_cast<float4 *>(dx));
      X.g = dx;
      return SG_OK;
    }
    if (op.kind == Op::CAT) {
      Tensor &A = tp.t[op.in], &B = tp.t[op.in2];
      SG_TALLOC(ga, float, static_cast<size_t>(A.rows) * A.c);
      SG_TALLOC(gb, float, static_cast<size_t>(B.rows) * B.c);
      if (!dry && A.rows > 0)
        split2_kernel<<<grid_for(static_cast<int64_t>(A.rows) * ((A.c + B.c) / 4), 256), 256, 0, hs()>>>(
            reinterpret_cast<const float4 *>(O.g), A.rows, A.c / 4, B.c / 4, reinterpret_cast<const float4 *>(A.g),
            reinterpret_cast<const float4 *>(B.g), reinterpret_cast<float4 *>(ga), reinterpret_cast<float4 *>(gb));
      A.g = ga;
      B.g = gb;
      return SG_OK;
    }
    // CONV: out = residual + conv(in)
    if (op.in2 >= 0) SG_TRY(accumulate(op.in2, O.g));
    Tensor &X = tp.t[op.in];
    const Plan &p = plan(op.level, op.plan_kind);
    if (op.cv.g_w != nullptr) {
      const size_t n = static_cast<size_t>(p.kvol) * op.cin * op.cout;
      const size_t m = ar.mark();
      const size_t nb = sg_spconv_wgrad_workspace_bytes(p.rows, p.kvol, op.cin, op.cout);
      SG_TALLOC(ws, char, nb);
      if (!dry) {
        if (p.rows > 0) {      // (the reduction writes the parameter's layout [Cout][K][Cin])
          SG_TRY(spconv_wgrad_layout(X.p, 0, O.g, 0, tp.nbr_t[op.level][op.plan_kind], p.rows, p.kvol, op.cin, op.cout,
                                     op.cv.g_w, 1, ws, nb, stream));
        } else if (hipMemsetAsync(op.cv.g_w, 0, n * 4, hs()) != hipSuccess) {
          return check_launch(who);
        }
      }
      ar.release(m);
    }
    if (!X.needs_grad) return SG_OK;
    // input gradient: the forward kernel over the transposed rulebook, out = (gradient so far) + conv
    const Plan &pt = plan_t(op.level, op.plan_kind);
    SG_TALLOC(gx, float, static_cast<size_t>(X.rows) * X.c);
    SG_TRY(conv_launch(O.g, O.rows, pt, op.cout, op.cin, op.cv.w, op.plan_kind == PK_SUBM ? 3 : 2, X.g, gx));
    X.g = gx;
    return SG_OK;
  }

  int backward(const float *g_out, float *g_feats) {
    for (Tensor &T : tp.t) T.g = nullptr;
    tp.t[tp.root].needs_grad = g_feats != nullptr || dry;
    // (dry run: any non-null address, so that aliasing and chaining are counted as they will happen)
    if (dry && g_out == nullptr) g_out = reinterpret_cast<const float *>(tp.arena);
    tp.t[tp.result].g = const_cast<float *>(g_out);
    for (auto it = packed.begin(); it != packed.end();) it = it->first.second >= 2 ? packed.erase(it) : std::next(it);
    SG_TRY(prepack_backward());
    for (size_t i = tp.ops.size(); i-- > 0;) SG_TRY(backward_op(tp.ops[i]));
    if (g_feats != nullptr && !dry) {
      const Tensor &R = tp.t[tp.root];
      const size_t bytes = static_cast<size_t>(R.rows) * R.c * 4;
      hipError_t e = R.g != nullptr ? hipMemcpyAsync(g_feats, R.g, bytes, hipMemcpyDeviceToDevice, hs())
                                    : hipMemsetAsync(g_feats, 0, bytes, hs());
      if (e != hipSuccess) {
        set_error("%s: writing the input gradient failed", who);
        return SG_ERR_LAUNCH;
      }
    }
    return check_launch(who);
  }
};

// deep copy of the descriptor (the caller's arrays need not outlive the forward call)
struct DescCopy {
  sg_unet_train_desc d{};
  std::vector<sg_unet_train_level> levels;
  std::vector<std::vector<sg_unet_train_block>> blocks, tails;
  explicit DescCopy(const sg_unet_train_desc *src) : d(*src) {
    levels.assign(src->levels, src->levels + src->n_levels);
    blocks.resize(src->n_levels);
    tails.resize(src->n_levels);
    for (int l = 0; l < src->n_levels; ++l) {
      blocks[l].assign(src->levels[l].blocks, src->levels[l].blocks + src->levels[l].n_blocks);
      levels[l].blocks = blocks[l].data();
      if (src->levels[l].tail != nullptr) {
        tails[l].assign(src->levels[l].tail, src->levels[l].tail + src->levels[l].n_blocks);
        levels[l].tail = tails[l].data();
      }
    }
    d.levels = levels.data();
  }
};

struct TapeBox {
  Tape tape;
  DescCopy desc;
  explicit TapeBox(const sg_unet_train_desc *d) : desc(d) {}
};

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_unet_train_arena_hint(const sg_unet_train_desc *d, int num_rows) {
  if (d == nullptr || d->n_levels < 1) return 0;
  // index tables (bound) + ~40 live [rows, 2 * planes] buffers on the outermost level, a quarter per
  // level below (real scenes shrink 2-4x per level); sg_unet_train_forward reports the exact need
  size_t total = unet_index_bytes(d->n_levels, num_rows) + (8 << 20);
  const size_t rows = static_cast<size_t>(num_rows > 0 ? num_rows : 1);
  size_t r = rows;
  for (int l = 0; l < d->n_levels; ++l) {
    total += r * 40 * 2 * static_cast<size_t>(d->levels[l].planes) * 4 + (4 << 20);
    r = r / 3 + 1;
  }
  return total;
}

int sg_unet_train_forward(const sg_unet_train_desc *d, const float *feats, const int32_t *indices, int num_rows,
                          const int32_t *spatial_shape_host, float *out, void *arena, size_t arena_bytes,
                          size_t *arena_needed, void **tape_out, sg_stream_t stream) {
  const char *who = "sg_unet_train_forward";
  SG_REQUIRE(d != nullptr && d->n_levels >= 1 && d->levels != nullptr && tape_out != nullptr, "%s: bad descriptor",
             who);
  SG_REQUIRE(d->n_levels <= SG_PYRAMID_MAX_LEVELS, "%s: at most %d levels", who, SG_PYRAMID_MAX_LEVELS);
  SG_REQUIRE(num_rows > 0, "%s: no input rows", who);
  SG_REQUIRE(d->arithmetic == 0 || d->arithmetic == 2, "%s: arithmetic must be 0 or 2", who);
  for (int l = 0; l < d->n_levels; ++l) {
    const sg_unet_train_level &L = d->levels[l];
    SG_REQUIRE(L.planes % 4 == 0 && L.n_blocks >= 1 && L.blocks != nullptr, "%s: level %d: planes must be a multiple of 4",
               who, l);
    SG_REQUIRE((l + 1 == d->n_levels) || (L.tail != nullptr && L.down.w != nullptr && L.up.w != nullptr &&
                                          L.down_bn.weight != nullptr && L.up_bn.weight != nullptr),
               "%s: level %d: incomplete descriptor", who, l);
  }
  *tape_out = nullptr;
  if (arena_needed) *arena_needed = 0;
  struct ArithScope {
    int keep;
    explicit ArithScope(int a) : keep(t_conv_arith) { if (a > 0) t_conv_arith = a; }
    ~ArithScope() { t_conv_arith = keep; }
  } arith_scope(d->arithmetic);
  const int L = d->n_levels;
  TapeBox *box = new TapeBox(d);
  Tape &tp = box->tape;
  struct Drop {      // the box dies with this call unless it is handed out
    TapeBox *b;
    ~Drop() { delete b; }
  } drop{box};
  tp.n_levels = L;
  tp.arithmetic = d->arithmetic;
  tp.arena = static_cast<char *>(arena);
  tp.arena_bytes = arena_bytes;
  std::unique_lock<std::mutex> guard;
  size_t index_bytes = 0;
  SG_TRY(unet_build_index(who, L, indices, num_rows, spatial_shape_host, arena, arena_bytes, stream, tp.li,
                          &index_bytes, &guard));
  // ---- exact arena need: dry run of the forward and of the backward over the known level sizes
  size_t start = align_up(index_bytes, 4096);
  auto fixed_part = [&](TrainExec &ex) -> int {      // counters + transposed tables, then the tape
    unsigned *ctr = ex.take<unsigned>(kCounters);
    if (ctr == nullptr) return SG_ERR_WORKSPACE;
    tp.counters = ctr;
    for (int l = 0; l < L; ++l) {
      const LevelIdx &I = tp.li[l];
      const Plan *pl[4] = {&I.subm, &I.down, &I.up, &I.ident};
      for (int k = 0; k < 4; ++k) {
        tp.nbr_t[l][k] = nullptr;
        if (pl[k]->nbr == nullptr || pl[k]->rows == 0) continue;
        if (k == PK_IDENT) {      // K = 1: the table is its own transpose
          tp.nbr_t[l][k] = pl[k]->nbr;
          continue;
        }
        int32_t *t = ex.take<int32_t>(static_cast<size_t>(pl[k]->rows) * pl[k]->kvol);
        if (t == nullptr) return SG_ERR_WORKSPACE;
        tp.nbr_t[l][k] = t;
        if (!ex.dry) SG_TRY(sg_spconv_transpose_table(pl[k]->nbr, pl[k]->rows, pl[k]->kvol, t, stream));
      }
    }
    return SG_OK;
  };
  {
    Tape scratch_tape = tp;      // (levels' plans are needed by the dry run; the ops it records are dropped)
    TrainExec dryx(scratch_tape, arena, arena_bytes, start, stream, true, who);
    SG_TRY(fixed_part(dryx));
    SG_TRY(dryx.forward(&box->desc.d, feats, num_rows, out));
    SG_TRY(dryx.backward(nullptr, nullptr));
    const size_t need = align_up(dryx.ar.peak, 4096) + 4096;
    if (arena_needed) *arena_needed = need;
    if (need > arena_bytes) {
      set_error("%s: arena too small (%zu bytes, this input needs %zu)", who, arena_bytes, need);
      return SG_ERR_WORKSPACE;
    }
  }
  TrainExec ex(tp, arena, arena_bytes, start, stream, false, who);
  SG_TRY(fixed_part(ex));
  if (hipMemsetAsync(tp.counters, 0, kCounters * sizeof(unsigned), as_stream(stream)) != hipSuccess) {
    set_error("%s: clearing the arrival counters failed", who);
    return SG_ERR_LAUNCH;
  }
  SG_TRY(ex.forward(&box->desc.d, feats, num_rows, out));
  tp.off = ex.ar.off;
  SG_TRY(check_launch(who));
  drop.b = nullptr;
  *tape_out = box;
  return SG_OK;
}

int sg_unet_train_backward(void *tape, const float *g_out, float *g_feats, sg_stream_t stream) {
  const char *who = "sg_unet_train_backward";
  SG_REQUIRE(tape != nullptr && g_out != nullptr, "%s: bad arguments", who);
  TapeBox *box = static_cast<TapeBox *>(tape);
  struct Drop {
    TapeBox *b;
    ~Drop() { delete b; }
  } drop{box};
  Tape &tp = box->tape;
  struct ArithScope {
    int keep;
    explicit ArithScope(int a) : keep(t_conv_arith) { if (a > 0) t_conv_arith = a; }
    ~ArithScope() { t_conv_arith = keep; }
  } arith_scope(tp.arithmetic);
  TrainExec ex(tp, tp.arena, tp.arena_bytes, tp.off, stream, false, who);
  return ex.backward(g_out, g_feats);
}

void sg_unet_train_release(void *tape) {
  delete static_cast<TapeBox *>(tape);
}

}  // extern "C"
