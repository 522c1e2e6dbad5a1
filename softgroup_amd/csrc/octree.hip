// octree.hip -- SoftGroup++ ball query over the exported 3-level octree.
// Replaces octree_ball_query/octree_ball_query.cu:14-147 (one thread per point walking 585
// nodes with 2.3 KB + 4 KB of per-thread scratch).
//
// One wave per query point: level 1 is tested by 8 lanes, level 2 by all 64 lanes at once,
// the 512 leaves in 8 rounds of 64; the active-leaf sets live in 64-bit ballots.  Leaves are
// then visited in export order and their points streamed 64 at a time; ballot + popcount
// prefix keeps the reference's neighbour order (leaf order, then within-leaf order) and its
// "first 1000" cap without any per-thread array or sort.
#include "common.h"

namespace sg {

constexpr int kOctMids = SG_OCTREE_NUM_NODES - SG_OCTREE_NUM_LEAVES;  // 73
constexpr int kOctCap = SG_BALLQUERY_MAX_NEIGHBORS;

// box/sphere test, expression-for-expression octree_ball_query.cu:14-44
__device__ __forceinline__ bool box_hit(const float *__restrict__ b, float cx, float cy, float cz,
                                        float r) {
  const float x = b[0], y = b[1], z = b[2], w = b[3], h = b[4], l = b[5];
  const float dist_x = fabsf(__fsub_rn(x, cx)), dist_y = fabsf(__fsub_rn(y, cy)),
              dist_z = fabsf(__fsub_rn(z, cz));
  const float hw = w / 2, hh = h / 2, hl = l / 2;  // exact halving
  if (dist_x > __fadd_rn(hw, r)) return false;
  if (dist_y > __fadd_rn(hh, r)) return false;
  if (dist_z > __fadd_rn(hl, r)) return false;
  if (dist_x <= hw) return true;
  if (dist_y <= hh) return true;
  if (dist_z <= hl) return true;
  const float dx = __fsub_rn(dist_x, hw), dy = __fsub_rn(dist_y, hh), dz = __fsub_rn(dist_z, hl);
  return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy))) <= __fmul_rn(r, r);
}

template <bool FILL>
__global__ void __launch_bounds__(256) octree_query_kernel(const float *__restrict__ points,
                                                          const float *__restrict__ boxes,
                                                          const int32_t *__restrict__ pt_inds,
                                                          const int32_t *__restrict__ pt_start_len,
                                                          int n, float radius,
                                                          int32_t *__restrict__ start_len,
                                                          int32_t *__restrict__ idx_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float r2 = __fmul_rn(radius, radius);
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const float cx = points[3 * i], cy = points[3 * i + 1], cz = points[3 * i + 2];
    const uint64_t a1 = __ballot(lane < 8 && box_hit(boxes + (1 + lane) * 6, cx, cy, cz, radius));
    const uint64_t a2 =
        __ballot(((a1 >> (lane >> 3)) & 1) && box_hit(boxes + (9 + lane) * 6, cx, cy, cz, radius));
    const int64_t out_start = FILL ? start_len[2 * i] : 0;
    int count = 0;
    for (int r = 0; r < 8; ++r) {
      uint64_t leaves = __ballot(((a2 >> (r * 8 + (lane >> 3))) & 1) &&
                                 box_hit(boxes + (kOctMids + r * 64 + lane) * 6, cx, cy, cz, radius));
      while (leaves) {
        const int b = __ffsll(static_cast<long long>(leaves)) - 1;
        leaves &= leaves - 1;
        const int leaf = r * 64 + b;
        const int st = pt_start_len[2 * leaf], len = pt_start_len[2 * leaf + 1];
        for (int j0 = 0; j0 < len; j0 += 64) {
          const int j = j0 + lane;
          bool ok = false;
          int p = 0;
          if (j < len) {
            p = pt_inds[st + j];
            const float dx = __fsub_rn(cx, points[3 * p]), dy = __fsub_rn(cy, points[3 * p + 1]),
                        dz = __fsub_rn(cz, points[3 * p + 2]);
            ok = __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy))) < r2;
          }
          const uint64_t bal = __ballot(ok);
          if (FILL) {
            const int pos = count + mask_prefix(bal);
            if (ok && pos < kOctCap) idx_out[out_start + pos] = p;
          }
          count += __popcll(bal);
        }
      }
    }
    if (!FILL && lane == 0) start_len[2 * i + 1] = min(count, kOctCap);
  }
}

}  // namespace sg

using namespace sg;

extern "C" {

int sg_octree_ballquery_count(const float *points, const float *boxes, const int32_t *pt_inds,
                              const int32_t *pt_start_len, int n, float radius, int32_t *start_len,
                              sg_stream_t stream) {
  SG_REQUIRE(n >= 0, "sg_octree_ballquery_count: n < 0");
  if (n == 0) return SG_OK;
  octree_query_kernel<false><<<grid_for(n, 4, 256 * 16), 256, 0, as_stream(stream)>>>(
      points, boxes, pt_inds, pt_start_len, n, radius, start_len, nullptr);
  return check_launch("sg_octree_ballquery_count");
}

int sg_octree_ballquery_fill(const float *points, const float *boxes, const int32_t *pt_inds,
                             const int32_t *pt_start_len, int n, float radius,
                             const int32_t *start_len, int32_t *idx, sg_stream_t stream) {
  SG_REQUIRE(n >= 0, "sg_octree_ballquery_fill: n < 0");
  if (n == 0) return SG_OK;
  octree_query_kernel<true><<<grid_for(n, 4, 256 * 16), 256, 0, as_stream(stream)>>>(
      points, boxes, pt_inds, pt_start_len, n, radius, const_cast<int32_t *>(start_len), idx);
  return check_launch("sg_octree_ballquery_fill");
}

}  // extern "C"
