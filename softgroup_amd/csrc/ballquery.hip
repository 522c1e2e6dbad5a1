// ballquery.hip -- radius-r neighbour lists (CSR) on a hashed uniform grid.
// Replaces the reference's O(n^2) brute-force kernel (bfs_cluster/bfs_cluster.cu:15-66:
// every point scans every point of its batch, 4 KB of per-thread scratch).
//
// Contract kept from the reference: per point the ASCENDING list of the
// indices k of the same batch with d2 < r^2 (strict, self included), capped at the 1000
// smallest; d2 = fma(dz,dz, fma(dy,dy, dx*dx)) in fp32 on the original coordinates.
//
// MI355X design: cell = r*(1+1e-6) so all neighbours live in the 27 surrounding cells;
// cells are found through an open-addressing table keyed by (batch, cx, cy, cz); points are
// counting-sorted by cell into a float4 {x,y,z,id} array so the candidate scan is a
// contiguous 16-B/lane stream.  One wave per query point: lanes 0..26 probe the 27 cells,
// then all 64 lanes stream the candidates (flat over the 27 ranges); accepted ids are compacted with ballot+popcount
// into LDS and rank-sorted there (ascending) before one coalesced write.  Two passes
// (count -> scan -> fill) make the CSR layout deterministic (the reference's atomic cursor
// does not).
#include "common.h"
#include "scan.h"

namespace sg {

constexpr int32_t kBqEmpty = 0x7f7f7f7f;
constexpr int kBqBuf = 2048;  // ids staged per wave in LDS (8 KB)
constexpr int kBqCap = SG_BALLQUERY_MAX_NEIGHBORS;
constexpr int kBqKeep = 64;   // lists up to this length are finished by the COUNT pass (sorted ids parked in `keep`)

struct BqWs {
  int4 *cell;        // [n] (b, cx, cy, cz)
  int32_t *table;    // [cap] representative point of the cell in this slot
  int32_t *count;    // [cap]
  int32_t *start;    // [cap]
  int32_t *cursor;   // [cap]
  int32_t *slot_of;  // [n]
  float4 *sorted;    // [n] x,y,z,id
  int32_t *keep;     // [n][kBqKeep] ascending ids of the points whose list has <= kBqKeep entries
  void *scan_ws;
  size_t scan_bytes;
  uint32_t cap;
};

__device__ __forceinline__ uint32_t cell_slot(int b, int cx, int cy, int cz, uint32_t mask) {
  uint64_t h = (static_cast<uint64_t>(static_cast<uint32_t>(b)) << 32) | static_cast<uint32_t>(cx);
  h = mix64(h) ^ ((static_cast<uint64_t>(static_cast<uint32_t>(cy)) << 32) | static_cast<uint32_t>(cz));
  return static_cast<uint32_t>(mix64(h)) & mask;
}

__global__ void __launch_bounds__(256) bq_insert_kernel(const float *__restrict__ xyz,
                                                       const int32_t *__restrict__ batch_idxs,
                                                       int n, double inv_cell, int4 *__restrict__ cell,
                                                       int32_t *table, uint32_t mask, int32_t *count,
                                                       int32_t *__restrict__ slot_of) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int4 c;
  c.x = batch_idxs ? batch_idxs[i] : 0;
  c.y = static_cast<int>(floor(static_cast<double>(xyz[3 * i + 0]) * inv_cell));
  c.z = static_cast<int>(floor(static_cast<double>(xyz[3 * i + 1]) * inv_cell));
  c.w = static_cast<int>(floor(static_cast<double>(xyz[3 * i + 2]) * inv_cell));
  cell[i] = c;
  uint32_t s = cell_slot(c.x, c.y, c.z, c.w, mask);
  while (true) {
    int32_t cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kBqEmpty) {
      cur = atomicCAS(&table[s], kBqEmpty, i);
      if (cur == kBqEmpty) break;
    }
    // the representative's cell may not be visible through the plain array yet: recompute it
    int4 o;
    o.x = batch_idxs ? batch_idxs[cur] : 0;
    o.y = static_cast<int>(floor(static_cast<double>(xyz[3 * cur + 0]) * inv_cell));
    o.z = static_cast<int>(floor(static_cast<double>(xyz[3 * cur + 1]) * inv_cell));
    o.w = static_cast<int>(floor(static_cast<double>(xyz[3 * cur + 2]) * inv_cell));
    if (o.x == c.x && o.y == c.y && o.z == c.z && o.w == c.w) break;
    s = (s + 1) & mask;
  }
  slot_of[i] = static_cast<int32_t>(s);
  atomicAdd(&count[s], 1);
}

__global__ void __launch_bounds__(256) bq_scatter_kernel(const float *__restrict__ xyz, int n,
                                                        const int32_t *__restrict__ slot_of,
                                                        const int32_t *__restrict__ start,
                                                        int32_t *cursor, float4 *__restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = slot_of[i];
  const int pos = start[s] + atomicAdd(&cursor[s], 1);
  sorted[pos] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float(i));
}

__device__ __forceinline__ float dist2(float ox, float oy, float oz, float x, float y, float z) {
  const float dx = __fsub_rn(ox, x), dy = __fsub_rn(oy, y), dz = __fsub_rn(oz, z);
  // contraction of the reference build (LLVM DAG combiner shared by NVVM and AMDGPU: the left
  // product of dx*dx + dy*dy is fused, the right one rounded), pinned by tests/test_ref_gpu_kernels.py
  return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
}

// number of accepted candidates with id < limit (limit = INT_MAX counts all)
// The candidates of a query are the concatenation of its 27 cells' point ranges.  They are walked
// FLAT, 64 at a time: lane t of a chunk finds its cell by a 5-step binary search over the cells'
// inclusive prefix counts (held by lanes 0..26, read with shuffles), so a query with ~150
// candidates costs 3 coalesced-ish load rounds instead of 27 dependent ones.
struct BqCells {
  int start, excl, incl, total;   // per lane (cell = lane < 27): range start, prefix counts; wave total
};
__device__ __forceinline__ BqCells bq_cells(int my_start, int my_count) {
  BqCells c;
  c.start = my_start;
  c.incl = wave_incl_scan(my_count);
  c.excl = c.incl - my_count;
  c.total = __shfl(c.incl, 63, 64);
  return c;
}
// index into `sorted` of flat candidate t (t < total)
__device__ __forceinline__ int bq_candidate(const BqCells &c, int t) {
  int lo = 0, hi = 26;
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    const int mid = (lo + hi) >> 1;
    const bool right = __shfl(c.incl, mid, 64) <= t;
    lo = right ? mid + 1 : lo;
    hi = right ? hi : mid;
  }
  return __shfl(c.start, lo, 64) + (t - __shfl(c.excl, lo, 64));
}

__device__ __forceinline__ int bq_scan_count(const float4 *__restrict__ sorted, const BqCells &c,
                                             float ox, float oy, float oz, float r2, int limit) {
  int total = 0;
  for (int t0 = 0; t0 < c.total; t0 += 64) {
    const int t = t0 + lane_id();
    const int at = bq_candidate(c, min(t, c.total - 1));
    if (t < c.total) {
      const float4 p = sorted[at];
      total += (dist2(ox, oy, oz, p.x, p.y, p.z) < r2 && __float_as_int(p.w) < limit) ? 1 : 0;
    }
  }
  return wave_sum(total);
}

template <bool FILL>
__global__ void __launch_bounds__(256) bq_query_kernel(const float *__restrict__ xyz, int n,
                                                      float r2, const int4 *__restrict__ cell,
                                                      const int32_t *__restrict__ table,
                                                      const int32_t *__restrict__ count,
                                                      const int32_t *__restrict__ start,
                                                      uint32_t mask, const float4 *__restrict__ sorted,
                                                      int32_t *__restrict__ start_len,
                                                      int32_t *__restrict__ idx_out, int32_t *__restrict__ keep) {
  __shared__ __attribute__((aligned(16))) int32_t buf_all[FILL ? 4 * kBqBuf : 4 * kBqKeep];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int32_t *buf = buf_all + wave * (FILL ? kBqBuf : kBqKeep);
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    if (FILL) {
      // short lists were finished by the count pass: one coalesced copy, no candidate walk
      const int out_len = start_len[2 * i + 1];
      if (out_len <= kBqKeep) {
        if (lane < out_len) idx_out[static_cast<int64_t>(start_len[2 * i]) + lane] = keep[static_cast<int64_t>(i) * kBqKeep + lane];
        continue;
      }
    }
    const float ox = xyz[3 * i], oy = xyz[3 * i + 1], oz = xyz[3 * i + 2];
    const int4 c = cell[i];
    // lanes 0..26 resolve the 27 neighbour cells
    int my_start = 0, my_count = 0;
    if (lane < 27) {
      const int cx = c.y + lane / 9 - 1, cy = c.z + (lane / 3) % 3 - 1, cz = c.w + lane % 3 - 1;
      uint32_t s = cell_slot(c.x, cx, cy, cz, mask);
      while (true) {
        const int32_t rep = table[s];
        if (rep == kBqEmpty) break;
        const int4 o = cell[rep];
        if (o.x == c.x && o.y == cx && o.z == cy && o.w == cz) {
          my_start = start[s];
          my_count = count[s];
          break;
        }
        s = (s + 1) & mask;
      }
    }
    const BqCells cells = bq_cells(my_start, my_count);
    if (!FILL) {
      // count, and keep the first kBqKeep accepted ids: a list that short (the usual case: ~36
      // neighbours per point at r = 0.04 on 2 cm voxels) is sorted here and parked in `keep`, and the
      // fill pass only copies it -- one candidate walk per point instead of two
      int m = 0;
      for (int t0 = 0; t0 < cells.total; t0 += 64) {
        const int t = t0 + lane;
        const int at = bq_candidate(cells, min(t, cells.total - 1));
        bool ok = false;
        int id = 0;
        if (t < cells.total) {
          const float4 p = sorted[at];
          id = __float_as_int(p.w);
          ok = dist2(ox, oy, oz, p.x, p.y, p.z) < r2;
        }
        const uint64_t bal = __ballot(ok);
        const int pos = m + mask_prefix(bal);
        if (ok && pos < kBqKeep) buf[pos] = id;
        m += __popcll(bal);
      }
      if (lane == 0) start_len[2 * i + 1] = min(m, kBqCap);
      if (m <= kBqKeep) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int e = lane < m ? buf[lane] : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += (buf[j] < e);
        if (lane < m) keep[static_cast<int64_t>(i) * kBqKeep + rank] = e;
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    // ---- fill: collect accepted ids (below `limit`) into LDS, in arrival order
    const int out_start = start_len[2 * i], out_len = start_len[2 * i + 1];
    int limit = 0x7fffffff;
    int m = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      m = 0;
      for (int t0 = 0; t0 < cells.total; t0 += 64) {
        const int t = t0 + lane;
        const int at = bq_candidate(cells, min(t, cells.total - 1));
        bool ok = false;
        int id = 0;
        if (t < cells.total) {
          const float4 p = sorted[at];
          id = __float_as_int(p.w);
          ok = dist2(ox, oy, oz, p.x, p.y, p.z) < r2 && id < limit;
        }
        const uint64_t bal = __ballot(ok);
        const int pos = m + mask_prefix(bal);
        if (ok && pos < kBqBuf) buf[pos] = id;
        m += __popcll(bal);
      }
      if (m <= kBqBuf) break;
      // Rare: more accepted candidates than the LDS stage holds.  Find the id threshold below
      // which exactly `out_len` (= 1000) accepted ids lie, then collect only those.
      int lo = 0, hi = n;  // smallest T with count(id < T) >= out_len
      while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        const int cmid = bq_scan_count(sorted, cells, ox, oy, oz, r2, mid);
        if (cmid >= out_len) hi = mid; else lo = mid + 1;
      }
      limit = lo;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- rank sort (ids are distinct): out[rank(e)] = e, keep rank < out_len (1000 smallest)
    for (int e0 = 0; e0 < m; e0 += 64) {
      const int ei = e0 + lane;
      const int e = ei < m ? buf[ei] : 0x7fffffff;
      int rank = 0;
      int j = 0;
      for (; j + 4 <= m; j += 4) {
        const int4 q = *reinterpret_cast<const int4 *>(buf + j);
        rank += (q.x < e) + (q.y < e) + (q.z < e) + (q.w < e);
      }
      for (; j < m; ++j) rank += (buf[j] < e);
      if (ei < m && rank < out_len) idx_out[static_cast<int64_t>(out_start) + rank] = e;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

static size_t bq_cap(int n) {
  size_t cap = 1024;
  while (cap < static_cast<size_t>(n) * 2) cap <<= 1;
  return cap;
}

static bool bq_carve(void *ws, size_t ws_bytes, int n, BqWs *w) {
  Workspace a(ws, ws_bytes);
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  w->cap = static_cast<uint32_t>(bq_cap(n));
  w->cell = a.take<int4>(nn);
  w->sorted = a.take<float4>(nn);
  w->table = a.take<int32_t>(w->cap);
  w->count = a.take<int32_t>(w->cap);
  w->start = a.take<int32_t>(w->cap);
  w->cursor = a.take<int32_t>(w->cap);
  w->slot_of = a.take<int32_t>(nn);
  w->scan_bytes = scan_workspace_bytes(w->cap);
  w->scan_ws = a.take<char>(w->scan_bytes);
  w->keep = a.take<int32_t>(nn * kBqKeep);
  return w->scan_ws != nullptr && w->keep != nullptr;
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_ballquery_workspace_bytes(int n) {
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  const size_t cap = bq_cap(n);
  return 2 * align_up(nn * 16) + 4 * align_up(cap * 4) + align_up(nn * 4) +
         align_up(scan_workspace_bytes(cap)) + align_up(nn * kBqKeep * 4) + 256;
}

int sg_ballquery_build_grid(const float *xyz, const int32_t *batch_idxs, int n, float radius,
                            void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(n >= 0 && radius > 0.f, "sg_ballquery_build_grid: bad arguments (n=%d r=%g)", n, radius);
  hipStream_t stream = as_stream(stream_);
  BqWs w;
  if (!bq_carve(ws, ws_bytes, n, &w)) {
    set_error("sg_ballquery_build_grid: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  if (n == 0) return SG_OK;
  {
    FillList f;
    f.add(w.table, static_cast<size_t>(w.cap) * 4, 0x7f);
    f.add(w.count, static_cast<size_t>(w.cap) * 4, 0);
    f.add(w.cursor, static_cast<size_t>(w.cap) * 4, 0);
    fill_many(f, stream);
  }
  const double inv_cell = 1.0 / (static_cast<double>(radius) * (1.0 + 1e-6));
  const int grid = (n + 255) / 256;
  bq_insert_kernel<<<grid, 256, 0, stream>>>(xyz, batch_idxs, n, inv_cell, w.cell, w.table,
                                             w.cap - 1, w.count, w.slot_of);
  const int32_t *count = w.count;
  int32_t *start = w.start;
  auto in = [count] __device__(int64_t s) { return count[s]; };
  auto out = [start] __device__(int64_t s, int v) { start[s] = v; };
  int rc = exclusive_scan(in, out, w.cap, nullptr, w.scan_ws, w.scan_bytes, stream);
  if (rc != SG_OK) return rc;
  bq_scatter_kernel<<<grid, 256, 0, stream>>>(xyz, n, w.slot_of, w.start, w.cursor, w.sorted);
  return check_launch("sg_ballquery_build_grid");
}

int sg_ballquery_count(const float *xyz, const int32_t *batch_idxs, int n, float radius,
                       int32_t *start_len, int32_t *meta, void *ws, size_t ws_bytes,
                       sg_stream_t stream_) {
  (void)batch_idxs;
  (void)meta;
  SG_REQUIRE(n >= 0 && radius > 0.f, "sg_ballquery_count: bad arguments");
  BqWs w;
  if (!bq_carve(ws, ws_bytes, n, &w)) {
    set_error("sg_ballquery_count: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  if (n == 0) return SG_OK;
  const float r2 = radius * radius;  // bfs_cluster.cu:26
  bq_query_kernel<false><<<grid_for(n, 4, 256 * 16), 256, 0, as_stream(stream_)>>>(
      xyz, n, r2, w.cell, w.table, w.count, w.start, w.cap - 1, w.sorted, start_len, nullptr, w.keep);
  return check_launch("sg_ballquery_count");
}

int sg_ballquery_fill(const float *xyz, const int32_t *batch_idxs, int n, float radius,
                      const int32_t *start_len, int32_t *idx, void *ws, size_t ws_bytes,
                      sg_stream_t stream_) {
  (void)batch_idxs;
  SG_REQUIRE(n >= 0 && radius > 0.f, "sg_ballquery_fill: bad arguments");
  BqWs w;
  if (!bq_carve(ws, ws_bytes, n, &w)) {
    set_error("sg_ballquery_fill: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  if (n == 0) return SG_OK;
  const float r2 = radius * radius;
  bq_query_kernel<true><<<grid_for(n, 4, 256 * 16), 256, 0, as_stream(stream_)>>>(
      xyz, n, r2, w.cell, w.table, w.count, w.start, w.cap - 1, w.sorted,
      const_cast<int32_t *>(start_len), idx, w.keep);
  return check_launch("sg_ballquery_fill");
}

}  // extern "C"
