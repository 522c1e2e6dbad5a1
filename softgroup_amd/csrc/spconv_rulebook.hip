// spconv_rulebook.hip -- active-site hashing and gather tables for the sparse convolutions.
// Replaces the indice-pair generation of the un-vendored spconv 2.1 library for the three conv
// flavours the reference model uses: SubMConv3d k3 p1, SparseConv3d k2 s2,
// SparseInverseConv3d k2.  Tables are OUTPUT-STATIONARY: nbr[j*K + k] = input row that feeds
// output row j through kernel offset k (-1 = inactive), so every output row is written once.
//
// Coordinate hash: 64-bit key = linearised (batch, d0, d1, d2), open addressing, table of
// >= 2M slots (<= 3 MB at 124k voxels: L2 resident on every XCD).  The tile plan orders rows by
// their neighbour bit mask (stable LSD radix sort, radix_sort.h) so that a 32-row MFMA tile
// only visits offsets that some row of the tile really has.
#include <stdlib.h>

#include "common.h"
#include "radix_sort.h"
#include "scan.h"

namespace sg {

constexpr uint64_t kKeyEmpty = ~0ULL;

struct HashWs {
  uint64_t *keys;
  int32_t *vals;
  int32_t *owner, *rank;  // [M]
  void *scan_ws;
  size_t scan_bytes;
  uint32_t cap;
};

static size_t hash_cap(int n) {
  size_t cap = 1024;
  while (cap < static_cast<size_t>(n) * 2) cap <<= 1;
  return cap;
}
static bool hash_carve(void *ws, size_t ws_bytes, int n, HashWs *w) {
  Workspace a(ws, ws_bytes);
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  w->cap = static_cast<uint32_t>(hash_cap(n));
  w->keys = a.take<uint64_t>(w->cap);
  w->vals = a.take<int32_t>(w->cap);
  w->owner = a.take<int32_t>(nn);
  w->rank = a.take<int32_t>(nn);
  w->scan_bytes = scan_workspace_bytes(n);
  w->scan_ws = a.take<char>(w->scan_bytes);
  return w->scan_ws != nullptr;
}

struct Shape3 {
  int s0, s1, s2;
};
__device__ __forceinline__ uint64_t lin_key(int b, int x, int y, int z, Shape3 s) {
  return ((static_cast<uint64_t>(b) * s.s0 + x) * s.s1 + y) * s.s2 + z;
}

// insert key -> min(value); returns the slot
__device__ __forceinline__ uint32_t hash_insert_min(uint64_t *keys, int32_t *vals, uint32_t mask,
                                                    uint64_t key, int32_t val) {
  uint32_t s = static_cast<uint32_t>(mix64(key)) & mask;
  while (true) {
    unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&keys[s]),
                                        static_cast<unsigned long long>(kKeyEmpty),
                                        static_cast<unsigned long long>(key));
    if (prev == kKeyEmpty || prev == key) {
      atomicMin(&vals[s], val);
      return s;
    }
    s = (s + 1) & mask;
  }
}
__device__ __forceinline__ int32_t hash_find(const uint64_t *__restrict__ keys,
                                             const int32_t *__restrict__ vals, uint32_t mask,
                                             uint64_t key) {
  uint32_t s = static_cast<uint32_t>(mix64(key)) & mask;
  while (true) {
    const uint64_t k = keys[s];
    if (k == key) return vals[s];
    if (k == kKeyEmpty) return -1;
    s = (s + 1) & mask;
  }
}

__global__ void __launch_bounds__(256) subm_insert_kernel(const int32_t *__restrict__ indices, int M,
                                                         Shape3 shape, uint64_t *keys,
                                                         int32_t *vals, uint32_t mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int4 c = reinterpret_cast<const int4 *>(indices)[i];
  hash_insert_min(keys, vals, mask, lin_key(c.x, c.y, c.z, c.w, shape), i);
}

__global__ void __launch_bounds__(256) subm_lookup_kernel(const int32_t *__restrict__ indices, int M,
                                                         Shape3 shape,
                                                         const uint64_t *__restrict__ keys,
                                                         const int32_t *__restrict__ vals,
                                                         uint32_t mask, int32_t *__restrict__ nbr) {
  const int64_t total = static_cast<int64_t>(M) * 27;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int j = static_cast<int>(t / 27), k = static_cast<int>(t - static_cast<int64_t>(j) * 27);
    const int4 c = reinterpret_cast<const int4 *>(indices)[j];
    const int x = c.y + k / 9 - 1, y = c.z + (k / 3) % 3 - 1, z = c.w + k % 3 - 1;
    int32_t r = -1;
    if (k == 13) r = j;
    else if (x >= 0 && y >= 0 && z >= 0 && x < shape.s0 && y < shape.s1 && z < shape.s2)
      r = hash_find(keys, vals, mask, lin_key(c.x, x, y, z, shape));
    nbr[t] = r;
  }
}

// ---- strided k2 s2: output site = c//2, first-seen numbering (owner = min input row)
__global__ void __launch_bounds__(256) down_insert_kernel(const int32_t *__restrict__ indices, int M,
                                                         Shape3 oshape, uint64_t *keys,
                                                         int32_t *vals, uint32_t mask,
                                                         int32_t *__restrict__ slot_of) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int4 c = reinterpret_cast<const int4 *>(indices)[i];
  const int x = c.y >> 1, y = c.z >> 1, z = c.w >> 1;
  if (x >= oshape.s0 || y >= oshape.s1 || z >= oshape.s2) {  // odd extent: last plane dropped
    slot_of[i] = -1;
    return;
  }
  slot_of[i] = static_cast<int32_t>(
      hash_insert_min(keys, vals, mask, lin_key(c.x, x, y, z, oshape), i));
}
__global__ void __launch_bounds__(256) down_owner_kernel(const int32_t *__restrict__ vals,
                                                        const int32_t *__restrict__ slot_of, int M,
                                                        int32_t *__restrict__ owner) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < M) owner[i] = slot_of[i] < 0 ? -1 : vals[slot_of[i]];
}
__global__ void __launch_bounds__(256) down_map_kernel(const int32_t *__restrict__ owner,
                                                      const int32_t *__restrict__ rank, int M,
                                                      int32_t *__restrict__ in2out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < M) in2out[i] = owner[i] < 0 ? -1 : rank[owner[i]];
}
__global__ void __launch_bounds__(256) down_fill_kernel(const int32_t *__restrict__ indices, int M,
                                                       const int32_t *__restrict__ in2out,
                                                       const int32_t *__restrict__ owner,
                                                       int32_t *__restrict__ out_indices,
                                                       int32_t *__restrict__ child) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int o = in2out[i];
  if (o < 0) return;
  const int4 c = reinterpret_cast<const int4 *>(indices)[i];
  child[o * 8 + (c.y & 1) * 4 + (c.z & 1) * 2 + (c.w & 1)] = i;
  if (owner[i] == i) reinterpret_cast<int4 *>(out_indices)[o] = make_int4(c.x, c.y >> 1, c.z >> 1, c.w >> 1);
}
__global__ void __launch_bounds__(256) inverse_rulebook_kernel(const int32_t *__restrict__ indices,
                                                              const int32_t *__restrict__ in2out,
                                                              int M, int32_t *__restrict__ inv) {
  const int64_t total = static_cast<int64_t>(M) * 8;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int i = static_cast<int>(t >> 3), k = static_cast<int>(t & 7);
    const int4 c = reinterpret_cast<const int4 *>(indices)[i];
    const int kk = (c.y & 1) * 4 + (c.z & 1) * 2 + (c.w & 1);
    inv[t] = (k == kk) ? in2out[i] : -1;
  }
}

// ---- active sites of EVERY level of a U-Net from the finest level's coordinates alone: level l+1
// holds the distinct (c >> 1) of level l that fall inside shape_l / 2 (same drop rule as
// down_insert_kernel), so one pass over the finest voxels with one small key-only hash table per
// coarser level gives all the row counts at once.  The executor reads them back with a single
// host sync and can then size and enqueue the whole U-Net without waiting for the GPU again.
__global__ void __launch_bounds__(256) pyramid_count_kernel(const int32_t *__restrict__ indices, int M,
                                                           Shape3 shape, int n_levels,
                                                           uint64_t *__restrict__ keys, uint32_t cap,
                                                           int32_t *__restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool alive = i < M;
  int4 c = make_int4(0, 0, 0, 0);
  if (alive) c = reinterpret_cast<const int4 *>(indices)[i];
  Shape3 s = shape;
  const uint32_t mask = cap - 1;
  for (int l = 1; l < n_levels; ++l) {
    const Shape3 os{s.s0 / 2, s.s1 / 2, s.s2 / 2};
    c.y >>= 1; c.z >>= 1; c.w >>= 1;
    alive = alive && c.y < os.s0 && c.z < os.s1 && c.w < os.s2;
    bool fresh = false;
    if (alive) {
      const uint64_t key = lin_key(c.x, c.y, c.z, c.w, os);
      uint64_t *tab = keys + static_cast<size_t>(l - 1) * cap;
      uint32_t slot = static_cast<uint32_t>(mix64(key)) & mask;
      while (true) {
        const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&tab[slot]),
                                                  static_cast<unsigned long long>(kKeyEmpty),
                                                  static_cast<unsigned long long>(key));
        if (prev == kKeyEmpty) { fresh = true; break; }
        if (prev == key) break;
        slot = (slot + 1) & mask;
      }
    }
    const uint64_t b = __ballot(fresh);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&counts[l], __popcll(b));
    s = os;
  }
}

// ---- tile plan
// Rows are sorted by their neighbour mask so that the 32 rows of a tile share as many kernel
// offsets as possible.  The sort key is the mask with its bits PERMUTED by how common each offset
// is in this layer: the rarest offset becomes the most significant bit, the most common one the
// least significant.  Rows that have a rare offset end up together (so few tiles pay for it) and
// neighbouring keys differ in offsets almost every row has anyway.  Measured on the S2 scene:
// 8-10 % fewer (tile, offset) pairs than sorting by the raw mask on the two big U-Net levels.
__global__ void __launch_bounds__(256) plan_mask_kernel(const int32_t *__restrict__ nbr, int M, int K,
                                                       uint32_t *__restrict__ mask,
                                                       int32_t *__restrict__ row,
                                                       int32_t *__restrict__ freq) {
  __shared__ int cnt[32];
  if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  uint32_t m = 0;
  if (j < M) {
    for (int k = 0; k < K; ++k) m |= (nbr[static_cast<int64_t>(j) * K + k] >= 0 ? 1u : 0u) << k;
    mask[j] = m;
    row[j] = j;
  }
  for (int k = 0; k < K; ++k) {
    const int c = __popcll(__ballot((m >> k) & 1u));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt[k], c);
  }
  __syncthreads();
  if (threadIdx.x < K && cnt[threadIdx.x]) atomicAdd(&freq[threadIdx.x], cnt[threadIdx.x]);
}

// position of offset k in the sort key: number of offsets that are more common (ties: the lower
// offset counts as more common), i.e. the most common offset is bit 0, the rarest bit K-1
__device__ __forceinline__ void plan_bit_positions(const int32_t *__restrict__ freq, int K, int *pos) {
  if (threadIdx.x < 32) {
    const int k = threadIdx.x;
    int p = 0;
    if (k < K) {
      const int fk = freq[k];
      for (int o = 0; o < K; ++o) {
        const int fo = freq[o];
        p += (fo > fk || (fo == fk && o < k)) ? 1 : 0;
      }
    }
    pos[k] = p;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) plan_key_kernel(const int32_t *__restrict__ freq, int M, int K,
                                                      uint32_t *__restrict__ mask_to_key) {
  __shared__ int pos[32];
  plan_bit_positions(freq, K, pos);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const uint32_t m = mask_to_key[j];
  uint32_t key = 0;
  for (int k = 0; k < K; ++k) key |= ((m >> k) & 1u) << pos[k];
  mask_to_key[j] = key;
}

__global__ void __launch_bounds__(256) plan_tiles_kernel(const uint32_t *__restrict__ key_sorted,
                                                        const int32_t *__restrict__ freq, int M, int K,
                                                        uint32_t *__restrict__ tile_mask) {
  __shared__ int pos[32];
  plan_bit_positions(freq, K, pos);
  const int j = blockIdx.x * 256 + threadIdx.x;  // 256 rows = 8 tiles of 32 per block
  uint32_t key = j < M ? key_sorted[j] : 0u;
  // OR over each aligned group of 32 lanes (commutes with the bit permutation), then back to offsets
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) key |= __shfl_xor(key, o, 64);
  if ((threadIdx.x & 31) == 0 && j < M) {
    uint32_t m = 0;
    for (int k = 0; k < K; ++k) m |= ((key >> pos[k]) & 1u) << k;
    tile_mask[j >> 5] = m;
  }
}

// tiles in descending order of work (number of kernel offsets present): with the heaviest
// tiles dispatched first and workgroups handed to CUs round-robin, every SIMD ends up with a mix
// of heavy and light waves instead of a tail of heavy ones (longest-processing-time-first).
__global__ void __launch_bounds__(1024) plan_tile_order_kernel(const uint32_t *__restrict__ tile_mask,
                                                              int num_tiles,
                                                              int32_t *__restrict__ tile_order) {
  // one workgroup: counting sort of the tiles by popcount, descending (order inside a bucket is
  // irrelevant for load balance, so positions come from LDS atomics)
  __shared__ int hist[33], base[33];
  if (threadIdx.x < 33) hist[threadIdx.x] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < num_tiles; t += 1024) atomicAdd(&hist[__popc(tile_mask[t])], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 32; b >= 0; --b) { base[b] = run; run += hist[b]; }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < num_tiles; t += 1024)
    tile_order[atomicAdd(&base[__popc(tile_mask[t])], 1)] = t;
}

// final layout, tile t' = tile_order[t'] of the mask-sorted sequence: rows of the tile (-1 pad),
// its mask, and its gather-table rows copied contiguously (the conv kernel then reads one
// linear 32*K block per tile instead of chasing order[] -> nbr[]).
__global__ void __launch_bounds__(256) plan_emit_kernel(const int32_t *__restrict__ nbr, int M, int K,
                                                       const int32_t *__restrict__ row_sorted,
                                                       const uint32_t *__restrict__ tile_mask_sorted,
                                                       const int32_t *__restrict__ tile_order,
                                                       int num_tiles, int32_t *__restrict__ order,
                                                       uint32_t *__restrict__ tile_mask,
                                                       int32_t *__restrict__ nbr_tiles) {
  const int per_tile = 32 * K;
  for (int t2 = blockIdx.x; t2 < num_tiles; t2 += gridDim.x) {
    const int t = tile_order[t2];
    if (threadIdx.x == 0) tile_mask[t2] = tile_mask_sorted[t];
    if (threadIdx.x < 32) {
      const int pos = t * 32 + threadIdx.x;
      order[t2 * 32 + threadIdx.x] = pos < M ? row_sorted[pos] : -1;
    }
    for (int e = threadIdx.x; e < per_tile; e += 256) {
      const int r = e / K, k = e - r * K;
      const int pos = t * 32 + r;
      nbr_tiles[static_cast<int64_t>(t2) * per_tile + e] =
          pos < M ? nbr[static_cast<int64_t>(row_sorted[pos]) * K + k] : -1;
    }
  }
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_spconv_hash_workspace_bytes(int M) {
  const size_t nn = static_cast<size_t>(M > 0 ? M : 1);
  return align_up(hash_cap(M) * 8) + align_up(hash_cap(M) * 4) + 3 * align_up(nn * 4) +
         align_up(scan_workspace_bytes(M)) + 256;
}

static int build_hash_common(HashWs *w, void *ws, size_t ws_bytes, int M, hipStream_t stream,
                             const char *who) {
  if (!hash_carve(ws, ws_bytes, M, w)) {
    set_error("%s: workspace too small", who);
    return SG_ERR_WORKSPACE;
  }
  hipMemsetAsync(w->keys, 0xff, static_cast<size_t>(w->cap) * 8, stream);
  hipMemsetAsync(w->vals, 0x7f, static_cast<size_t>(w->cap) * 4, stream);
  return SG_OK;
}

int sg_spconv_subm_rulebook(const int32_t *indices, int M, const int32_t *shape_host, int32_t *nbr,
                            void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(M >= 0 && shape_host, "sg_spconv_subm_rulebook: bad arguments");
  if (M == 0) return SG_OK;
  hipStream_t stream = as_stream(stream_);
  HashWs w;
  int rc = build_hash_common(&w, ws, ws_bytes, M, stream, "sg_spconv_subm_rulebook");
  if (rc != SG_OK) return rc;
  const Shape3 shape{shape_host[0], shape_host[1], shape_host[2]};
  subm_insert_kernel<<<(M + 255) / 256, 256, 0, stream>>>(indices, M, shape, w.keys, w.vals,
                                                         w.cap - 1);
  subm_lookup_kernel<<<grid_for(static_cast<int64_t>(M) * 27, 256, 256 * 32), 256, 0, stream>>>(
      indices, M, shape, w.keys, w.vals, w.cap - 1, nbr);
  return check_launch("sg_spconv_subm_rulebook");
}

int sg_spconv_down_build(const int32_t *indices, int M, const int32_t *shape_host, int32_t *in2out,
                         int32_t *meta, void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(M >= 0 && shape_host, "sg_spconv_down_build: bad arguments");
  hipStream_t stream = as_stream(stream_);
  hipMemsetAsync(meta, 0, 4, stream);
  if (M == 0) return SG_OK;
  HashWs w;
  int rc = build_hash_common(&w, ws, ws_bytes, M, stream, "sg_spconv_down_build");
  if (rc != SG_OK) return rc;
  const Shape3 oshape{shape_host[0] / 2, shape_host[1] / 2, shape_host[2] / 2};
  const int grid = (M + 255) / 256;
  int32_t *slot_of = w.rank;  // reused: slot ids are dead once owners are known
  down_insert_kernel<<<grid, 256, 0, stream>>>(indices, M, oshape, w.keys, w.vals, w.cap - 1,
                                               slot_of);
  down_owner_kernel<<<grid, 256, 0, stream>>>(w.vals, slot_of, M, w.owner);
  const int32_t *owner = w.owner;
  int32_t *rank = w.rank;
  rc = exclusive_scan(
      [owner] __device__(int64_t i) { return owner[i] == static_cast<int32_t>(i) ? 1 : 0; },
      [rank] __device__(int64_t i, int v) { rank[i] = v; }, M, meta, w.scan_ws, w.scan_bytes,
      stream);
  if (rc != SG_OK) return rc;
  down_map_kernel<<<grid, 256, 0, stream>>>(w.owner, w.rank, M, in2out);
  return check_launch("sg_spconv_down_build");
}

int sg_spconv_down_fill(const int32_t *indices, int M, const int32_t *in2out, int M_out,
                        int32_t *out_indices, int32_t *child, void *ws, size_t ws_bytes,
                        sg_stream_t stream_) {
  SG_REQUIRE(M >= 0 && M_out >= 0, "sg_spconv_down_fill: bad arguments");
  if (M == 0 || M_out == 0) return SG_OK;
  hipStream_t stream = as_stream(stream_);
  HashWs w;
  if (!hash_carve(ws, ws_bytes, M, &w)) {
    set_error("sg_spconv_down_fill: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  hipMemsetAsync(child, 0xff, static_cast<size_t>(M_out) * 8 * 4, stream);
  down_fill_kernel<<<(M + 255) / 256, 256, 0, stream>>>(indices, M, in2out, w.owner, out_indices,
                                                       child);
  return check_launch("sg_spconv_down_fill");
}

int sg_spconv_inverse_rulebook(const int32_t *indices_fine, const int32_t *in2out, int M,
                               int32_t *inv_nbr, sg_stream_t stream) {
  SG_REQUIRE(M >= 0, "sg_spconv_inverse_rulebook: bad arguments");
  if (M == 0) return SG_OK;
  inverse_rulebook_kernel<<<grid_for(static_cast<int64_t>(M) * 8, 256, 256 * 32), 256, 0,
                            as_stream(stream)>>>(indices_fine, in2out, M, inv_nbr);
  return check_launch("sg_spconv_inverse_rulebook");
}

size_t sg_spconv_level_rows_workspace_bytes(int M, int n_levels) {
  return static_cast<size_t>(n_levels > 1 ? n_levels - 1 : 0) * align_up(hash_cap(M) * 8) + 256;
}

int sg_spconv_level_rows(const int32_t *indices, int M, const int32_t *shape_host, int n_levels,
                         int32_t *counts, void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(M >= 0 && shape_host && n_levels >= 1 && n_levels <= 16 && counts,
             "sg_spconv_level_rows: bad arguments");
  hipStream_t stream = as_stream(stream_);
  hipMemsetAsync(counts, 0, static_cast<size_t>(n_levels) * 4, stream);
  if (M == 0 || n_levels == 1) return check_launch("sg_spconv_level_rows");
  SG_REQUIRE(ws != nullptr && ws_bytes >= sg_spconv_level_rows_workspace_bytes(M, n_levels),
             "sg_spconv_level_rows: workspace too small");
  const size_t cap = hash_cap(M);
  SG_REQUIRE(align_up(cap * 8) == cap * 8, "sg_spconv_level_rows: internal table alignment");
  hipMemsetAsync(ws, 0xff, static_cast<size_t>(n_levels - 1) * cap * 8, stream);
  const Shape3 shape{shape_host[0], shape_host[1], shape_host[2]};
  pyramid_count_kernel<<<(M + 255) / 256, 256, 0, stream>>>(indices, M, shape, n_levels,
                                                           static_cast<uint64_t *>(ws),
                                                           static_cast<uint32_t>(cap), counts);
  return check_launch("sg_spconv_level_rows");
}

size_t sg_spconv_plan_workspace_bytes(int M) {
  const size_t nn = static_cast<size_t>(M > 0 ? M : 1);
  const size_t nt = (nn + 31) / 32;
  return 2 * align_up(nn * 4) + 2 * align_up(nt * 4) + align_up(32 * 4) + radix_sort_workspace_bytes(M) + 256;
}

int sg_spconv_plan(const int32_t *nbr, int M, int K, int32_t *order, uint32_t *tile_mask,
                   int32_t *nbr_tiles, void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(M >= 0 && K >= 1 && K <= 32, "sg_spconv_plan: bad arguments (M=%d K=%d)", M, K);
  if (M == 0) return SG_OK;
  hipStream_t stream = as_stream(stream_);
  const int num_tiles = (M + 31) / 32;
  Workspace a(ws, ws_bytes);
  uint32_t *mask = a.take<uint32_t>(M);
  int32_t *row = a.take<int32_t>(M);
  uint32_t *tmask = a.take<uint32_t>(num_tiles);
  int32_t *torder = a.take<int32_t>(num_tiles);
  int32_t *freq = a.take<int32_t>(32);
  const size_t rs_bytes = radix_sort_workspace_bytes(M);
  void *rs_ws = a.take<char>(rs_bytes);
  if (!rs_ws) {
    set_error("sg_spconv_plan: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  const int grid = (M + 255) / 256;
  hipMemsetAsync(freq, 0, 32 * 4, stream);
  plan_mask_kernel<<<grid, 256, 0, stream>>>(nbr, M, K, mask, row, freq);
  static const bool raw_env = getenv("SG_PLAN_RAW") != nullptr;     // developer knob: sort by the raw mask
  if (raw_env) hipMemsetAsync(freq, 0, 32 * 4, stream);             // equal counts -> identity permutation
  plan_key_kernel<<<grid, 256, 0, stream>>>(freq, M, K, mask);
  uint32_t *ms;
  int32_t *rs;
  int rc = radix_sort_pairs(mask, row, M, K, rs_ws, rs_bytes, stream, &ms, &rs);
  if (rc != SG_OK) return rc;
  plan_tiles_kernel<<<grid, 256, 0, stream>>>(ms, freq, M, K, tmask);
  plan_tile_order_kernel<<<1, 1024, 0, stream>>>(tmask, num_tiles, torder);
  plan_emit_kernel<<<min(num_tiles, 4096), 256, 0, stream>>>(nbr, M, K, rs, tmask, torder, num_tiles,
                                                            order, tile_mask, nbr_tiles);
  return check_launch("sg_spconv_plan");
}

}  // extern "C"
