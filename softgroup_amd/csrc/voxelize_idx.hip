// voxelize_idx.hip -- device build of the point->voxel index products.
// Replaces the reference's single-threaded CPU hash (voxelize/voxelize.cpp:11-165), which the
// model calls inline on the critical path (softgroup.py:494,703) with D2H/H2D around it.
//
// Bit-exact products on a parallel machine:
//   voxel id      = first-seen order  -> each key is owned by its MINIMUM point index
//                   (open-addressing table, atomicMin on the slot), id = rank of the owner
//                   among owners in point order (device-wide prefix sum);
//   rule row      = [count, point indices ascending, 0 pad]  -> rank of a point inside its
//                   voxel = number of smaller indices in the voxel's CSR segment;
//   output_coords = coords of the owner (= rule[1]).
#include "common.h"
#include "scan.h"

namespace sg {

constexpr int32_t kEmpty = 0x7f7f7f7f;  // hipMemset(0x7f) pattern, larger than any point index

struct VKey {
  int32_t b, x, y, z;
};
__device__ __forceinline__ VKey load_vkey(const int64_t *__restrict__ coords, int i, int ncol) {
  const int64_t *r = coords + static_cast<int64_t>(i) * ncol;
  VKey k;
  if (ncol == 3) { k.b = 0; k.x = static_cast<int32_t>(r[0]); k.y = static_cast<int32_t>(r[1]); k.z = static_cast<int32_t>(r[2]); }
  else { k.b = static_cast<int32_t>(r[0]); k.x = static_cast<int32_t>(r[1]); k.y = static_cast<int32_t>(r[2]); k.z = static_cast<int32_t>(r[3]); }
  return k;
}
__device__ __forceinline__ bool same(const VKey &a, const VKey &b) {
  return a.b == b.b && a.x == b.x && a.y == b.y && a.z == b.z;
}
__device__ __forceinline__ uint32_t vkey_slot(const VKey &k, uint32_t cap_mask) {
  uint64_t h = (static_cast<uint64_t>(static_cast<uint32_t>(k.b)) << 32) | static_cast<uint32_t>(k.x);
  h = mix64(h) ^ ((static_cast<uint64_t>(static_cast<uint32_t>(k.y)) << 32) | static_cast<uint32_t>(k.z));
  return static_cast<uint32_t>(mix64(h)) & cap_mask;
}

__global__ void __launch_bounds__(256) vox_insert_kernel(const int64_t *__restrict__ coords, int n,
                                                        int ncol, int32_t *table, uint32_t cap_mask,
                                                        int32_t *__restrict__ slot_of) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const VKey k = load_vkey(coords, i, ncol);
  uint32_t s = vkey_slot(k, cap_mask);
  while (true) {
    int32_t cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kEmpty) {
      cur = atomicCAS(&table[s], kEmpty, i);
      if (cur == kEmpty) break;  // claimed; a smaller index of the same key may still lower it
    }
    // whoever sits in the slot has the slot's key (only same-key points ever replace it)
    if (same(load_vkey(coords, cur, ncol), k)) {
      atomicMin(&table[s], i);
      break;
    }
    s = (s + 1) & cap_mask;
  }
  slot_of[i] = static_cast<int32_t>(s);
}

__global__ void __launch_bounds__(256) vox_owner_kernel(const int32_t *__restrict__ table,
                                                       const int32_t *__restrict__ slot_of, int n,
                                                       int32_t *__restrict__ owner) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) owner[i] = table[slot_of[i]];
}

__global__ void __launch_bounds__(256) vox_map_kernel(const int32_t *__restrict__ owner,
                                                     const int32_t *__restrict__ rank, int n,
                                                     int pooled, int32_t *__restrict__ input_map,
                                                     int32_t *count, int32_t *last, int32_t *meta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int c = 0;
  if (i < n) {
    const int v = rank[owner[i]];
    input_map[i] = v;
    c = atomicAdd(&count[v], 1) + 1;
    atomicMax(&last[v], i);
  }
  if (pooled) {
    c = wave_max(c);  // the final count of every voxel is seen by whoever adds last
    if (lane_id() == 0 && c > 1) atomicMax(&meta[1], c);
  }
}

__global__ void __launch_bounds__(256) vox_scatter_kernel(const int32_t *__restrict__ input_map,
                                                         const int32_t *__restrict__ csr_off, int n,
                                                         int32_t *cursor, int32_t *__restrict__ seg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int v = input_map[i];
  seg[csr_off[v] + atomicAdd(&cursor[v], 1)] = i;
}

__global__ void __launch_bounds__(256) vox_fill_kernel(const int64_t *__restrict__ coords, int n,
                                                      int ncol, int mode,
                                                      const int32_t *__restrict__ input_map,
                                                      const int32_t *__restrict__ csr_off,
                                                      const int32_t *__restrict__ count,
                                                      const int32_t *__restrict__ last,
                                                      const int32_t *__restrict__ seg,
                                                      int max_active, int64_t *__restrict__ out_coords,
                                                      int32_t *__restrict__ out_map) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int v = input_map[i];
  int32_t *row = out_map + static_cast<int64_t>(v) * (max_active + 1);
  const int cnt = count[v];
  const int32_t *sg_ = seg + csr_off[v];
  int rank = 0;
  for (int j = 0; j < cnt; ++j) rank += (sg_[j] < i);
  const bool pooled = (mode == 3 || mode == 4);
  if (pooled) {
    row[1 + rank] = i;
    if (rank == 0) row[0] = cnt;
  } else if ((mode == 2) ? (i == last[v]) : (rank == 0)) {
    row[0] = 1;
    row[1] = i;
  }
  const bool writes_coords = (mode == 2) ? (i == last[v]) : (rank == 0);
  if (writes_coords) {
    const int64_t *src = coords + static_cast<int64_t>(i) * ncol;
    int64_t *dst = out_coords + static_cast<int64_t>(v) * ncol;
    for (int c = 0; c < ncol; ++c) dst[c] = src[c];
  }
}

struct VoxWs {
  int32_t *table, *slot_of, *owner, *rank, *count, *last, *csr_off, *cursor, *seg;
  void *scan_ws;
  size_t scan_bytes;
  uint32_t cap;
};

static size_t vox_cap(int n) {
  size_t cap = 1024;
  while (cap < static_cast<size_t>(n) * 2) cap <<= 1;
  return cap;
}

static bool vox_carve(void *ws, size_t ws_bytes, int n, VoxWs *w) {
  Workspace a(ws, ws_bytes);
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  w->cap = static_cast<uint32_t>(vox_cap(n));
  w->table = a.take<int32_t>(w->cap);
  w->slot_of = a.take<int32_t>(nn);
  w->owner = a.take<int32_t>(nn);
  w->rank = a.take<int32_t>(nn);
  w->count = a.take<int32_t>(nn);
  w->last = a.take<int32_t>(nn);
  w->csr_off = a.take<int32_t>(nn);
  w->cursor = a.take<int32_t>(nn);
  w->seg = a.take<int32_t>(nn);
  w->scan_bytes = scan_workspace_bytes(n);
  w->scan_ws = a.take<char>(w->scan_bytes);
  return w->scan_ws != nullptr;
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_voxelize_idx_workspace_bytes(int n) {
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  return align_up(vox_cap(n) * 4) + 8 * align_up(nn * 4) + align_up(scan_workspace_bytes(n)) + 256;
}

int sg_voxelize_idx_build(const int64_t *coords, int n, int ncol, int mode, int32_t *input_map,
                          int32_t *meta, void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(n >= 0 && (ncol == 3 || ncol == 4) && mode >= 0 && mode <= 4,
             "sg_voxelize_idx_build: bad arguments (n=%d ncol=%d mode=%d)", n, ncol, mode);
  hipStream_t stream = as_stream(stream_);
  VoxWs w;
  if (!vox_carve(ws, ws_bytes, n, &w)) {
    set_error("sg_voxelize_idx_build: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  const int32_t init_meta[2] = {0, 1};
  hipMemcpyAsync(meta, init_meta, sizeof(init_meta), hipMemcpyHostToDevice, stream);
  if (n == 0) return SG_OK;
  const int grid = (n + 255) / 256;
  {
    FillList f;
    f.add(w.table, static_cast<size_t>(w.cap) * 4, 0x7f);
    f.add(w.count, static_cast<size_t>(n) * 4, 0);
    f.add(w.last, static_cast<size_t>(n) * 4, 0);
    fill_many(f, stream);
  }
  vox_insert_kernel<<<grid, 256, 0, stream>>>(coords, n, ncol, w.table, w.cap - 1, w.slot_of);
  vox_owner_kernel<<<grid, 256, 0, stream>>>(w.table, w.slot_of, n, w.owner);
  const int32_t *owner = w.owner;
  int32_t *rank = w.rank;
  auto in = [owner] __device__(int64_t i) { return owner[i] == static_cast<int32_t>(i) ? 1 : 0; };
  auto out = [rank] __device__(int64_t i, int v) { rank[i] = v; };
  int rc = exclusive_scan(in, out, n, meta, w.scan_ws, w.scan_bytes, stream);
  if (rc != SG_OK) return rc;
  vox_map_kernel<<<grid, 256, 0, stream>>>(w.owner, w.rank, n, (mode == 3 || mode == 4) ? 1 : 0,
                                           input_map, w.count, w.last, meta);
  return check_launch("sg_voxelize_idx_build");
}

int sg_voxelize_idx_fill(const int64_t *coords, int n, int ncol, int mode, const int32_t *input_map,
                         int num_voxels, int max_active, int64_t *out_coords, int32_t *out_map,
                         void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(n >= 0 && num_voxels >= 0 && num_voxels <= n && max_active >= 1,
             "sg_voxelize_idx_fill: bad arguments");
  hipStream_t stream = as_stream(stream_);
  VoxWs w;
  if (!vox_carve(ws, ws_bytes, n, &w)) {
    set_error("sg_voxelize_idx_fill: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  if (n == 0 || num_voxels == 0) return SG_OK;
  {
    FillList f;
    f.add(out_map, static_cast<size_t>(num_voxels) * (max_active + 1) * 4, 0);
    f.add(w.cursor, static_cast<size_t>(num_voxels) * 4, 0);
    fill_many(f, stream);
  }
  const int32_t *count = w.count;
  int32_t *csr = w.csr_off;
  auto in = [count] __device__(int64_t v) { return count[v]; };
  auto out = [csr] __device__(int64_t v, int x) { csr[v] = x; };
  int rc = exclusive_scan(in, out, num_voxels, nullptr, w.scan_ws, w.scan_bytes, stream);
  if (rc != SG_OK) return rc;
  const int grid = (n + 255) / 256;
  vox_scatter_kernel<<<grid, 256, 0, stream>>>(input_map, w.csr_off, n, w.cursor, w.seg);
  vox_fill_kernel<<<grid, 256, 0, stream>>>(coords, n, ncol, mode, input_map, w.csr_off, w.count,
                                            w.last, w.seg, max_active, out_coords, out_map);
  return check_launch("sg_voxelize_idx_fill");
}

}  // extern "C"
