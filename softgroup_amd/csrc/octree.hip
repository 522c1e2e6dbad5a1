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
#include "radix_sort.h"

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

// STASH (sg_scan_grouping_pp: the count pass also parks the first kOctStash accepted ids of every query in
// `stash[i][.]`, in list order; a list that short -- the usual case at SoftGroup++'s radii on level voxels -- is
// then copied by octree_unstash_kernel instead of being walked a second time: 2 x 137 us -> 137 + ~15 us per class)
constexpr int kOctStash = 64;
template <bool FILL, bool STASH = false>
__global__ void __launch_bounds__(256) octree_query_kernel(const float *__restrict__ points,
                                                          const float *__restrict__ boxes,
                                                          const int32_t *__restrict__ pt_inds,
                                                          const int32_t *__restrict__ pt_start_len,
                                                          int n, float radius,
                                                          int32_t *__restrict__ start_len,
                                                          int32_t *__restrict__ idx_out,
                                                          int32_t *__restrict__ stash = nullptr) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float r2 = __fmul_rn(radius, radius);
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const float cx = points[3 * i], cy = points[3 * i + 1], cz = points[3 * i + 2];
    const uint64_t a1 = __ballot(lane < 8 && box_hit(boxes + (1 + lane) * 6, cx, cy, cz, radius));
    const uint64_t a2 =
        __ballot(((a1 >> (lane >> 3)) & 1) && box_hit(boxes + (9 + lane) * 6, cx, cy, cz, radius));
    const int64_t out_start = FILL ? start_len[2 * i] : 0;
    if (FILL && STASH && start_len[2 * i + 1] <= kOctStash) continue;      // (copied from the stash: wave-uniform)
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
          if (STASH && !FILL) {
            const int pos = count + mask_prefix(bal);
            if (ok && pos < kOctStash) stash[static_cast<int64_t>(i) * kOctStash + pos] = p;
          }
          count += __popcll(bal);
        }
      }
    }
    if (!FILL && lane == 0) start_len[2 * i + 1] = min(count, kOctCap);
  }
}


__global__ void __launch_bounds__(256) octree_unstash_kernel(const int32_t *__restrict__ stash,
                                                            const int32_t *__restrict__ start_len, int n,
                                                            int32_t *__restrict__ idx_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const int len = start_len[2 * i + 1];
    if (len <= kOctStash && lane < len) idx_out[start_len[2 * i] + lane] = stash[static_cast<int64_t>(i) * kOctStash + lane];
  }
}

// ------------------------------------------------------------------------------------------------
// Octree build on the device (reference: host C++, octree_ball_query.cpp:8-147, reached through
// functions.py:14-33 with a .cpu() of the class's coordinates per class and scan).  Same products,
// bit for bit: the root box from the coordinates' extent ((max + min) / 2, max - min in fp32), the 585
// boxes by the reference's float expressions (cpp:60-82), every point's leaf by its `<` tests
// (cpp:52-57), leaves holding their points in ascending index order (a stable sort of the point
// indices by leaf).  No host round trip: extent, boxes, leaves, sort and leaf ranges are seven
// launches on the caller's stream.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ordered_bits(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered(uint32_t o) {
  return __builtin_bit_cast(float, (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// ext[0..2] = min, ext[3..5] = max of the coordinates, as order-preserving integers (memset to ff.. / 0)
__global__ void __launch_bounds__(256) oct_extent_kernel(const float *__restrict__ pts, int n, uint32_t *ext) {
  __shared__ uint32_t red[6][4];
  uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const uint32_t o = ordered_bits(pts[3LL * i + a]);
      mn[a] = min(mn[a], o);
      mx[a] = max(mx[a], o);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], static_cast<uint32_t>(__shfl_xor(static_cast<int>(mn[a]), o, 64)));
      mx[a] = max(mx[a], static_cast<uint32_t>(__shfl_xor(static_cast<int>(mx[a]), o, 64)));
    }
    if ((threadIdx.x & 63) == 0) {
      red[a][threadIdx.x >> 6] = mn[a];
      red[3 + a][threadIdx.x >> 6] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    atomicMin(&ext[threadIdx.x], min(min(red[threadIdx.x][0], red[threadIdx.x][1]),
                                     min(red[threadIdx.x][2], red[threadIdx.x][3])));
    atomicMax(&ext[3 + threadIdx.x], max(max(red[3 + threadIdx.x][0], red[3 + threadIdx.x][1]),
                                         max(red[3 + threadIdx.x][2], red[3 + threadIdx.x][3])));
  }
}

// one workgroup of 512: root from the extent, then level by level (thread = child)
__global__ void __launch_bounds__(512) oct_boxes_kernel(const uint32_t *__restrict__ ext, float *__restrict__ boxes) {
  if (threadIdx.x < 3) {
    const float mn = from_ordered(ext[threadIdx.x]), mx = from_ordered(ext[3 + threadIdx.x]);
    boxes[threadIdx.x] = __fmul_rn(__fadd_rn(mx, mn), 0.5f);        // (max + min) / 2
    boxes[3 + threadIdx.x] = __fsub_rn(mx, mn);
  }
  __syncthreads();
  int first = 0, width = 1;
  for (int l = 0; l < 3; ++l) {
    const int child = threadIdx.x;
    if (child < width * 8) {
      const int path = child >> 3, oct = child & 7;
      const float *pa = boxes + (first + path) * 6;
      const float w = __fmul_rn(pa[3], 0.5f), h = __fmul_rn(pa[4], 0.5f), d = __fmul_rn(pa[5], 0.5f);   // x / 2, exact
      float *c = boxes + (first + width + child) * 6;
      c[0] = (oct & 1) ? __fadd_rn(pa[0], __fmul_rn(w, 0.5f)) : __fsub_rn(pa[0], __fmul_rn(w, 0.5f));
      c[1] = (oct & 2) ? __fadd_rn(pa[1], __fmul_rn(h, 0.5f)) : __fsub_rn(pa[1], __fmul_rn(h, 0.5f));
      c[2] = (oct & 4) ? __fadd_rn(pa[2], __fmul_rn(d, 0.5f)) : __fsub_rn(pa[2], __fmul_rn(d, 0.5f));
      c[3] = w; c[4] = h; c[5] = d;
    }
    __threadfence_block();
    __syncthreads();
    first += width;
    width *= 8;
  }
}

__global__ void __launch_bounds__(256) oct_leaf_kernel(const float *__restrict__ pts, int n,
                                                      const float *__restrict__ boxes, uint32_t *__restrict__ key,
                                                      int32_t *__restrict__ val, int32_t *__restrict__ hist) {
  __shared__ float bx[73 * 3];            // centres of the 1 + 8 + 64 inner nodes
  __shared__ int h[512];
  for (int i = threadIdx.x; i < 73 * 3; i += 256) bx[i] = boxes[(i / 3) * 6 + i % 3];
  for (int i = threadIdx.x; i < 512; i += 256) h[i] = 0;
  __syncthreads();
  const int firsts[3] = {0, 1, 9};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float x = pts[3LL * i], y = pts[3LL * i + 1], z = pts[3LL * i + 2];
    int path = 0;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const float *b = bx + (firsts[l] + path) * 3;
      path = path * 8 + ((x < b[0] ? 0 : 1) | (y < b[1] ? 0 : 2) | (z < b[2] ? 0 : 4));
    }
    key[i] = static_cast<uint32_t>(path);
    val[i] = i;
    atomicAdd(&h[path], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

__global__ void __launch_bounds__(512) oct_ranges_kernel(const int32_t *__restrict__ hist,
                                                        const int32_t *__restrict__ sorted, int n,
                                                        int32_t *__restrict__ pt_inds,
                                                        int32_t *__restrict__ pt_start_len) {
  __shared__ int part[8];
  for (int i = blockIdx.x * 512 + threadIdx.x; i < n; i += gridDim.x * 512) pt_inds[i] = sorted[i];
  if (blockIdx.x != 0) return;
  const int c = hist[threadIdx.x];
  const int incl = wave_incl_scan(c);
  if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = incl;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < (threadIdx.x >> 6); ++w) before += part[w];
  pt_start_len[2 * threadIdx.x] = before + incl - c;
  pt_start_len[2 * threadIdx.x + 1] = c;
}

// ------------------------------------------------------------------------------------------------
// Pyramid inverse map (SoftGroup.pyramid_inverse_map, softgroup.py:500-507): proposals over the
// level voxels of a class -> proposals over its points.  The reference builds a dense int
// [nProposal, n] matrix and takes its nonzero; proposals of one class are disjoint, so: voxel ->
// proposal table, point -> proposal through the point's voxel, a STABLE sort of the points by
// proposal (points ascending inside a proposal, the reference's row order) and a count per proposal.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pim_scatter_kernel(const int32_t *__restrict__ pairs, int64_t S,
                                                         int32_t *__restrict__ prop_of_voxel) {
  for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < S; e += gridDim.x * 256LL)
    prop_of_voxel[pairs[2 * e + 1]] = pairs[2 * e];
}
__global__ void __launch_bounds__(256) pim_key_kernel(const int32_t *__restrict__ prop_of_voxel,
                                                     const int32_t *__restrict__ l2p, int n, int n_prop,
                                                     uint32_t *__restrict__ key, int32_t *__restrict__ val) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int p = prop_of_voxel[l2p[i]];
    key[i] = p >= 0 ? static_cast<uint32_t>(p) : static_cast<uint32_t>(n_prop);
    val[i] = i;
  }
}
// rows in sorted order; the offsets are the positions where the sorted key changes (a proposal's
// rows start where its key first appears; keys nobody has -- none in practice, every proposal owns
// a voxel with a point -- get the position of the next larger key), n_out = first unassigned row
__global__ void __launch_bounds__(256) pim_emit_kernel(const uint32_t *__restrict__ key, const int32_t *__restrict__ val,
                                                      int n, int n_prop, int32_t *__restrict__ out_idx,
                                                      int32_t *__restrict__ out_off, int32_t *__restrict__ n_out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t k = key[i];
    if (k < static_cast<uint32_t>(n_prop)) {
      out_idx[2LL * i] = static_cast<int32_t>(k);
      out_idx[2LL * i + 1] = val[i];
    }
    const int prev = i > 0 ? static_cast<int>(key[i - 1]) : -1;
    for (int p = prev + 1; p <= static_cast<int>(k) && p <= n_prop; ++p) out_off[p] = i;
    if (k >= static_cast<uint32_t>(n_prop) && prev < n_prop) *n_out = i;
    if (i == n - 1 && k < static_cast<uint32_t>(n_prop)) {
      for (int p = static_cast<int>(k) + 1; p <= n_prop; ++p) out_off[p] = n;
      *n_out = n;
    }
  }
  if (n == 0 && blockIdx.x == 0)
    for (int p = threadIdx.x; p <= n_prop; p += 256) out_off[p] = 0;
}

// sg_octree_ballquery_count / _fill with the stash (internal: sg_scan_grouping_pp); stash = int32 [n * 64]
size_t octree_stash_bytes(int n) { return align_up(static_cast<size_t>(n > 0 ? n : 1) * kOctStash * sizeof(int32_t)); }
int octree_ballquery_count_stash(const float *points, const float *boxes, const int32_t *pt_inds,
                                 const int32_t *pt_start_len, int n, float radius, int32_t *start_len, int32_t *stash,
                                 hipStream_t stream) {
  if (n == 0) return SG_OK;
  octree_query_kernel<false, true><<<grid_for(n, 4, 256 * 16), 256, 0, stream>>>(points, boxes, pt_inds, pt_start_len, n,
                                                                                 radius, start_len, nullptr, stash);
  return check_launch("octree_ballquery_count_stash");
}
int octree_ballquery_fill_stash(const float *points, const float *boxes, const int32_t *pt_inds,
                                const int32_t *pt_start_len, int n, float radius, const int32_t *start_len,
                                const int32_t *stash, int32_t *idx, hipStream_t stream) {
  if (n == 0) return SG_OK;
  octree_unstash_kernel<<<grid_for(n, 4, 256 * 16), 256, 0, stream>>>(stash, start_len, n, idx);
  octree_query_kernel<true, true><<<grid_for(n, 4, 256 * 16), 256, 0, stream>>>(
      points, boxes, pt_inds, pt_start_len, n, radius, const_cast<int32_t *>(start_len), idx, nullptr);
  return check_launch("octree_ballquery_fill_stash");
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

size_t sg_octree_build_workspace_bytes(int n) {
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  return 2 * align_up(nn * 4) + align_up(1024 * 4) + radix_sort_workspace_bytes(n) + 512;
}

int sg_octree_build(const float *points, int n, float *boxes, int32_t *pt_inds, int32_t *pt_start_len, void *ws,
                    size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(n >= 0 && boxes && pt_start_len, "sg_octree_build: bad arguments");
  SG_REQUIRE(ws != nullptr && ws_bytes >= sg_octree_build_workspace_bytes(n), "sg_octree_build: workspace too small");
  hipStream_t stream = as_stream(stream_);
  Workspace a(ws, ws_bytes);
  uint32_t *key = a.take<uint32_t>(n > 0 ? n : 1);
  int32_t *val = a.take<int32_t>(n > 0 ? n : 1);
  int32_t *hist = a.take<int32_t>(1024);            // [512] leaf counts, [512..517] extent
  uint32_t *ext = reinterpret_cast<uint32_t *>(hist + 512);
  const size_t rs_bytes = radix_sort_workspace_bytes(n);
  void *rs_ws = a.take<char>(rs_bytes);
  hipMemsetAsync(hist, 0, 512 * 4, stream);
  hipMemsetAsync(ext, 0xff, 3 * 4, stream);
  hipMemsetAsync(ext + 3, 0, 3 * 4, stream);
  if (n > 0) oct_extent_kernel<<<grid_for(n, 256, 1024), 256, 0, stream>>>(points, n, ext);
  oct_boxes_kernel<<<1, 512, 0, stream>>>(ext, boxes);
  uint32_t *ks = key;
  int32_t *vs = val;
  if (n > 0) {
    oct_leaf_kernel<<<grid_for(n, 256, 1024), 256, 0, stream>>>(points, n, boxes, key, val, hist);
    const int rc = radix_sort_pairs(key, val, n, 9, rs_ws, rs_bytes, stream, &ks, &vs);
    if (rc != SG_OK) return rc;
  }
  oct_ranges_kernel<<<grid_for(n > 0 ? n : 1, 512, 1024), 512, 0, stream>>>(hist, vs, n, pt_inds, pt_start_len);
  return check_launch("sg_octree_build");
}

size_t sg_pyramid_inverse_map_workspace_bytes(int n_points, int n_voxels, int n_prop) {
  const size_t nn = static_cast<size_t>(n_points > 0 ? n_points : 1);
  return 2 * align_up(nn * 4) + align_up(static_cast<size_t>(n_voxels > 0 ? n_voxels : 1) * 4) +
         align_up((static_cast<size_t>(n_prop) + 2) * 4) + radix_sort_workspace_bytes(n_points) + 512;
}

int sg_pyramid_inverse_map(const int32_t *proposals_idx, int64_t num_pairs, int n_prop, const int32_t *l2p_map,
                           int n_points, int n_voxels, int32_t *out_idx, int32_t *out_offsets, int32_t *n_out_dev,
                           void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(num_pairs >= 0 && n_prop >= 0 && n_points >= 0 && n_voxels >= 0 && out_offsets && n_out_dev,
             "sg_pyramid_inverse_map: bad arguments");
  SG_REQUIRE(ws != nullptr && ws_bytes >= sg_pyramid_inverse_map_workspace_bytes(n_points, n_voxels, n_prop),
             "sg_pyramid_inverse_map: workspace too small");
  hipStream_t stream = as_stream(stream_);
  Workspace a(ws, ws_bytes);
  uint32_t *key = a.take<uint32_t>(n_points > 0 ? n_points : 1);
  int32_t *val = a.take<int32_t>(n_points > 0 ? n_points : 1);
  int32_t *pov = a.take<int32_t>(n_voxels > 0 ? n_voxels : 1);
  const size_t rs_bytes = radix_sort_workspace_bytes(n_points);
  void *rs_ws = a.take<char>(rs_bytes);
  hipMemsetAsync(pov, 0xff, static_cast<size_t>(n_voxels > 0 ? n_voxels : 1) * 4, stream);
  if (num_pairs > 0)
    pim_scatter_kernel<<<grid_for(num_pairs, 256, 2048), 256, 0, stream>>>(proposals_idx, num_pairs, pov);
  uint32_t *ks = key;
  int32_t *vs = val;
  if (n_points > 0) {
    pim_key_kernel<<<grid_for(n_points, 256, 2048), 256, 0, stream>>>(pov, l2p_map, n_points, n_prop, key, val);
    int bits = 1;
    while ((1 << bits) <= n_prop) ++bits;
    const int rc = radix_sort_pairs(key, val, n_points, bits, rs_ws, rs_bytes, stream, &ks, &vs);
    if (rc != SG_OK) return rc;
  }
  hipMemsetAsync(n_out_dev, 0, 4, stream);
  pim_emit_kernel<<<grid_for(n_points > 0 ? n_points : 1, 256, 2048), 256, 0, stream>>>(ks, vs, n_points, n_prop, out_idx,
                                                                                       out_offsets, n_out_dev);
  return check_launch("sg_pyramid_inverse_map");
}

}  // extern "C"
