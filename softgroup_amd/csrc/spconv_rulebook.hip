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

#include <mutex>

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

// insert key -> min(value); returns the slot.  Test before the atomics: the coarse levels of a
// pyramid have a handful of sites that EVERY level-0 row inserts into -- thousands of same-address
// atomics in a row (pyr_insert_kernel 134 us on the bench scene, most of it at the three deepest
// levels).  A plain load that already shows the key with a value <= ours makes both atomics
// redundant (values only ever decrease; a stale read just means the atomics run as before).
__device__ __forceinline__ uint32_t hash_insert_min(uint64_t *keys, int32_t *vals, uint32_t mask,
                                                    uint64_t key, int32_t val) {
  uint32_t s = static_cast<uint32_t>(mix64(key)) & mask;
  while (true) {
    const uint64_t seen = __hip_atomic_load(&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen == key) {
      if (__hip_atomic_load(&vals[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > val) atomicMin(&vals[s], val);
      return s;
    }
    if (seen == kKeyEmpty) {
      unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long *>(&keys[s]),
                                          static_cast<unsigned long long>(kKeyEmpty),
                                          static_cast<unsigned long long>(key));
      if (prev == kKeyEmpty || prev == key) {
        atomicMin(&vals[s], val);
        return s;
      }
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

// ---- tile plan
// Rows are sorted by their neighbour mask so that the 32 rows of a tile share as many kernel
// offsets as possible.  The sort key is the mask with its bits PERMUTED by how common each offset
// is in this layer: the rarest offset becomes the most significant bit, the most common one the
// least significant.  Rows that have a rare offset end up together (so few tiles pay for it) and
// neighbouring keys differ in offsets almost every row has anyway.  Measured on the S2 scene:
// 8-10 % fewer (tile, offset) pairs than sorting by the raw mask on the two big U-Net levels.
// (the kernels are the segmented ones further down: a single table is a plan problem of one segment)

// =============================================================================================
// Whole-pyramid index build: every level of a U-Net in a handful of launches.
//
// A U-Net over M0 voxels needs, per level, a SubM rulebook + tile plan, the strided-conv pairs to
// the next level (+ plan) and the inverse-conv table (+ plan).  Built level by level that is ~55
// small launches and a host read-back per level (what spconv does: each strided conv sizes its
// output by a device->host copy); the deep levels hold 18..5000 voxels, so the forward is bound by
// the HOST issuing those launches, not by the GPU.  Everything below works on ALL levels at once:
//
//   * level l's sites are the distinct (c >> l) of the level-0 coordinates that survive the
//     odd-extent drop of every strided conv on the way (same rule as down_insert_kernel), and
//     spconv-style first-seen numbering down a chain of strided convs is the same as numbering
//     level l's sites by their smallest level-0 descendant.  So one pass over the level-0 voxels
//     inserts into one hash table per level (value = min level-0 row), one device-wide scan over
//     the [L][M0] owner flags numbers the sites of every level, and the row counts of all levels
//     come back to the host in ONE read-back (sg_spconv_pyramid_rows);
//   * coordinates, child / in2out / inverse tables of all levels are then written by two launches,
//     the SubM tables of all levels by one (sg_spconv_pyramid_build);
//   * the tile plans of all 3L-2 gather tables are ONE segmented problem: rows of all tables are
//     concatenated, the sort key gets the segment id in its top 5 bits (27 mask bits + 5 = 32), one
//     radix sort orders every segment, and the tile kernels run over the concatenation.  Within a
//     segment the result is identical to sg_spconv_plan's (same keys, same stable order).
// ---------------------------------------------------------------------------------------------
constexpr int kPyrMaxLevels = SG_PYRAMID_MAX_LEVELS;
constexpr int kPyrMaxSegs = 3 * kPyrMaxLevels;      // <= 32: the segment id lives in key bits 27..31
constexpr uint32_t kPlanKeyBits = 27;

__global__ void __launch_bounds__(256) pyr_insert_kernel(const int32_t *__restrict__ indices, int M0,
                                                        Shape3 shape, int L, uint64_t *keys,
                                                        int32_t *vals, uint32_t cap,
                                                        int32_t *__restrict__ slot) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M0) return;
  int4 c = reinterpret_cast<const int4 *>(indices)[i];
  Shape3 s = shape;
  bool alive = true;
  for (int l = 0; l < L; ++l) {
    if (l > 0) {
      const Shape3 os{s.s0 / 2, s.s1 / 2, s.s2 / 2};
      c.y >>= 1; c.z >>= 1; c.w >>= 1;
      alive = alive && c.y < os.s0 && c.z < os.s1 && c.w < os.s2;   // odd extent: last plane dropped
      s = os;
    }
    int32_t sl = -1;
    if (alive)
      sl = static_cast<int32_t>(hash_insert_min(keys + static_cast<size_t>(l) * cap,
                                                vals + static_cast<size_t>(l) * cap, cap - 1,
                                                lin_key(c.x, c.y, c.z, c.w, s), i));
    slot[static_cast<size_t>(l) * M0 + i] = sl;
  }
}

__global__ void pyr_rows_kernel(const int32_t *__restrict__ pos, int M0, int L, int32_t *rows) {
  const int l = threadIdx.x;
  if (l < L) rows[l] = pos[static_cast<size_t>(l + 1) * M0] - pos[static_cast<size_t>(l) * M0];
}

struct PyrOut {
  int32_t *indices[kPyrMaxLevels];
  int32_t *nbr[kPyrMaxLevels];
  int32_t *in2out[kPyrMaxLevels];
  int32_t *child[kPyrMaxLevels];
  int32_t *inv[kPyrMaxLevels];
  long long pre27[kPyrMaxLevels + 1];     // prefix of rows_l * 27
  Shape3 shape[kPyrMaxLevels];
};

// the owner (smallest level-0 descendant) of every site writes the site's row and coordinates
__global__ void __launch_bounds__(256) pyr_emit_kernel(const int32_t *__restrict__ indices0, int M0,
                                                      int L, const int32_t *__restrict__ vals,
                                                      int32_t *__restrict__ rowtab, uint32_t cap,
                                                      const int32_t *__restrict__ slot,
                                                      const int32_t *__restrict__ pos, PyrOut o) {
  const int64_t total = static_cast<int64_t>(L) * M0;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int l = static_cast<int>(t / M0), i = static_cast<int>(t - static_cast<int64_t>(l) * M0);
    const int sl = slot[t];
    if (sl < 0 || vals[static_cast<size_t>(l) * cap + sl] != i) continue;
    const int r = pos[t] - pos[static_cast<size_t>(l) * M0];
    rowtab[static_cast<size_t>(l) * cap + sl] = r;
    const int4 c = reinterpret_cast<const int4 *>(indices0)[i];
    if (o.indices[l] != nullptr)
      reinterpret_cast<int4 *>(o.indices[l])[r] = make_int4(c.x, c.y >> l, c.z >> l, c.w >> l);
  }
}

// strided-conv pairs between consecutive levels, from the owners of the finer level
__global__ void __launch_bounds__(256) pyr_link_kernel(const int32_t *__restrict__ indices0, int M0,
                                                      int L, const int32_t *__restrict__ vals,
                                                      const int32_t *__restrict__ rowtab,
                                                      uint32_t cap, const int32_t *__restrict__ slot,
                                                      const int32_t *__restrict__ pos, PyrOut o) {
  const int64_t total = static_cast<int64_t>(L - 1) * M0;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int l = static_cast<int>(t / M0), i = static_cast<int>(t - static_cast<int64_t>(l) * M0);
    const int sl = slot[t];
    if (sl < 0 || vals[static_cast<size_t>(l) * cap + sl] != i) continue;
    const int j = pos[t] - pos[static_cast<size_t>(l) * M0];
    const int s2 = slot[t + M0];
    const int up = s2 >= 0 ? rowtab[static_cast<size_t>(l + 1) * cap + s2] : -1;
    const int4 c = reinterpret_cast<const int4 *>(indices0)[i];
    const int k = (((c.y >> l) & 1) << 2) | (((c.z >> l) & 1) << 1) | ((c.w >> l) & 1);
    o.in2out[l][j] = up;
    if (up >= 0) o.child[l][static_cast<int64_t>(up) * 8 + k] = j;
    int4 *dst = reinterpret_cast<int4 *>(o.inv[l] + static_cast<int64_t>(j) * 8);
    dst[0] = make_int4(k == 0 ? up : -1, k == 1 ? up : -1, k == 2 ? up : -1, k == 3 ? up : -1);
    dst[1] = make_int4(k == 4 ? up : -1, k == 5 ? up : -1, k == 6 ? up : -1, k == 7 ? up : -1);
  }
}

// SubM gather tables of all levels
__global__ void __launch_bounds__(256) pyr_subm_kernel(int L, const uint64_t *__restrict__ keys,
                                                      const int32_t *__restrict__ rowtab,
                                                      uint32_t cap, PyrOut o) {
  const long long total = o.pre27[L];
  int l = 0;
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    while (t >= o.pre27[l + 1]) ++l;                       // t only grows
    const long long e = t - o.pre27[l];
    const int j = static_cast<int>(e / 27), k = static_cast<int>(e - static_cast<long long>(j) * 27);
    const int4 c = reinterpret_cast<const int4 *>(o.indices[l])[j];
    const Shape3 s = o.shape[l];
    const int x = c.y + k / 9 - 1, y = c.z + (k / 3) % 3 - 1, z = c.w + k % 3 - 1;
    int32_t r = -1;
    if (k == 13) r = j;
    else if (x >= 0 && y >= 0 && z >= 0 && x < s.s0 && y < s.s1 && z < s.s2)
      r = hash_find(keys + static_cast<size_t>(l) * cap, rowtab + static_cast<size_t>(l) * cap, cap - 1,
                    lin_key(c.x, x, y, z, s));
    o.nbr[l][e] = r;
  }
}

// ---- segmented tile plans
struct PlanSeg {
  const int32_t *nbr;
  int32_t *order;
  uint32_t *tile_mask;
  int32_t *nbr_tiles;
  int rows, K, row_base, tile_base, sb_base, sb_rows;      // sb_rows: rows per super-block of this table (multiple of 32)
  // radix path: the table's sort key = permuted mask (key_bits wide) | seg_code << key_bits; where its
  // sorted keys / row ids ended up (set by build_plans after the sorts)
  int key_bits, seg_code;
  const uint32_t *key_sorted;
  const int32_t *val_sorted;
};
struct PlanSegs {
  int n, total_rows, total_tiles, total_sbs;
  int n_big, big_rows;      // the first n_big tables (big_rows rows) are sorted by the device-wide radix sorts:
  int n_wide, wide_rows;    // the first n_wide of them (27-bit masks) by one sort, the rest (8-bit masks) by another
  PlanSeg s[kPyrMaxSegs];
};

// ---- spatially local tile plans (round 5; SG_PLAN_ORDER=1|2, NOT the default -- see build_plans).  With the rows of a table in Morton order (the executor's
// internal row order, unet_exec.hip) a SUPER-BLOCK of kSbRows consecutive rows is a compact region
// of the scene.  Rows are mask-sorted inside their super-block only (a local sort in LDS instead of
// a device-wide radix sort), so a tile's 32 rows and their neighbours lie in one region; the tiles
// of a table are dealt to the 8 XCDs in CONTIGUOUS ranges (position p of the emitted list runs on
// XCD p % 8 -- the conv kernel keeps unit -> XCD fixed -- and holds the (p / 8)-th tile of that
// XCD's range), so a region's rows and halo are gathered through ONE 4 MB L2; inside an XCD's range
// the tiles are emitted heaviest first (mode 1) or super-block by super-block, heaviest first inside
// (mode 2).  Tables of <= kSbRows rows come out exactly as under the global sort.
constexpr int kSbRowsMax = 16384;      // 128 KB of LDS for the sort
constexpr int kSbIndexBits = 14;
// rows per super-block of a table: SG_PLAN_SB (developer knob) or, by default, an eighth of the
// table -- one XCD's share -- capped by what the LDS sort holds
static int plan_sb_rows(int rows) {
  static const int env = getenv("SG_PLAN_SB") ? atoi(getenv("SG_PLAN_SB")) : 0;
  int sb = env > 0 ? env : (rows + 7) / 8;
  if (env <= 0 && sb < 4096) sb = 4096;       // (small tables: one super-block = sorted as a whole)
  sb = (sb + 31) / 32 * 32;
  if (sb < 32) sb = 32;
  if (sb > kSbRowsMax) sb = kSbRowsMax;
  return sb;
}

// mask -> sort key (bits permuted by offset frequency), sort of one super-block, tile masks
__global__ void __launch_bounds__(1024) plan_sort_sb_kernel(PlanSegs P, const uint32_t *__restrict__ mask,
                                                           const int32_t *__restrict__ bitpos,
                                                           int32_t *__restrict__ val_sorted,
                                                           uint32_t *__restrict__ tmask) {
  extern __shared__ __attribute__((aligned(16))) uint64_t e[];
  __shared__ int pos[32];
  int seg = P.n_big;      // (the tables before that are sorted by the radix sort)
  while (seg + 1 < P.n && static_cast<int>(blockIdx.x) >= P.s[seg + 1].sb_base) ++seg;
  const PlanSeg &S = P.s[seg];
  const int sb = blockIdx.x - S.sb_base;
  const int first = sb * S.sb_rows;
  const int cnt = min(S.sb_rows, S.rows - first);
  const int K = S.K;
  if (threadIdx.x < 32) pos[threadIdx.x] = bitpos[seg * 32 + threadIdx.x];
  int N = 64;
  while (N < cnt) N <<= 1;
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 1024) {
    uint64_t v = ~0ull;
    if (i < cnt) {
      uint32_t key = 0;
      for (uint32_t mm = mask[S.row_base + first + i]; mm; mm &= mm - 1) key |= 1u << pos[__ffs(static_cast<int>(mm)) - 1];
      v = (static_cast<uint64_t>(key) << kSbIndexBits) | static_cast<uint64_t>(i);     // index in the low bits: stable
    }
    e[i] = v;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < N / 2; t += 1024) {
        const int i = 2 * t - (t & (j - 1));
        const int l = i + j;
        const uint64_t a = e[i], b = e[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { e[i] = b; e[l] = a; }
      }
      __syncthreads();
    }
  }
  for (int i0 = 0; i0 < N; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    uint32_t key = 0;
    if (i < cnt) {
      const uint64_t v = e[i];
      val_sorted[S.row_base + first + i] = first + static_cast<int>(v & ((1u << kSbIndexBits) - 1u));     // row inside the table
      key = static_cast<uint32_t>(v >> kSbIndexBits);
    }
    // OR over each aligned group of 32 lanes (commutes with the bit permutation), then back to offsets
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) key |= __shfl_xor(key, o, 64);
    if ((threadIdx.x & 31) == 0 && i < cnt) {
      uint32_t m = 0;
      for (int k = 0; k < K; ++k) m |= ((key >> pos[k]) & 1u) << k;
      tmask[S.tile_base + sb * (S.sb_rows >> 5) + (i >> 5)] = m;
    }
  }
}

// Emission order of a table's tiles: grid (segment, XCD).  XCD x owns tiles [start, start + cnt) of
// the super-block-major sequence, cnt = the number of positions p < nt with p % 8 == x; its k-th
// tile goes to position 8 k + x.  Rank inside the range: descending number of offsets (mode 1) or
// (super-block, descending number of offsets) (mode 2), ties in ascending tile order -- by counting
// the tiles that come before (a few hundred per range; no atomics, deterministic).
__global__ void __launch_bounds__(1024) plan_tile_order_xcd_kernel(PlanSegs P, const uint32_t *__restrict__ tmask,
                                                                  int mode, int32_t *__restrict__ torder) {
  __shared__ int bucket[4096];
  __shared__ int hist[33];
  const PlanSeg &S = P.s[blockIdx.x];
  const int nt = (S.rows + 31) / 32;
  const int x = blockIdx.y, q = nt >> 3, r = nt & 7;
  const int cnt = q + (x < r ? 1 : 0), start = x * q + min(x, r);
  if (x == 0) {      // tiles per number of offsets, behind the segment's final tile masks
    if (threadIdx.x < 33) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += 1024) atomicAdd(&hist[__popc(tmask[S.tile_base + t])], 1);
    __syncthreads();
    if (threadIdx.x < SG_PLAN_HIST_WORDS) S.tile_mask[nt + threadIdx.x] = threadIdx.x < 33 ? hist[threadIdx.x] : 0;
  }
  const int sb_tiles = S.sb_rows >> 5;
  const int sb0 = start / sb_tiles;
  auto bucket_of = [&](int t) {
    const int b = 32 - __popc(tmask[S.tile_base + t]);
    return mode == 2 ? (t / sb_tiles - sb0) * 33 + b : b;
  };
  for (int c0 = 0; c0 < cnt; c0 += 4096) {      // (ranges above 4096 tiles: chunks keep their order)
    const int n = min(4096, cnt - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) bucket[i] = bucket_of(start + c0 + i);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
      const int bi = bucket[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const int bj = bucket[j];
        rank += (bj < bi || (bj == bi && j < i)) ? 1 : 0;
      }
      torder[S.tile_base + (c0 + rank) * 8 + x] = start + c0 + i;
    }
  }
}

// grid.y = segment (blocks past the segment's rows exit at once): the segment is block-uniform,
// its descriptor comes straight from the kernel arguments through scalar loads
__global__ void __launch_bounds__(256) plan_mask_all_kernel(PlanSegs P, uint32_t *__restrict__ mask,
                                                           int32_t *__restrict__ val,
                                                           int32_t *__restrict__ freq) {
  __shared__ int cnt[32];
  const int seg = blockIdx.y;
  const int rows = P.s[seg].rows, K = P.s[seg].K, row_base = P.s[seg].row_base;
  if (blockIdx.x * 256 >= rows) return;
  const int32_t *nbr = P.s[seg].nbr;
  if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  uint32_t m = 0;
  if (j < rows) {
    const int32_t *row = nbr + static_cast<int64_t>(j) * K;
    for (int k = 0; k < K; ++k) m |= (row[k] >= 0 ? 1u : 0u) << k;
    mask[row_base + j] = m;
    val[row_base + j] = row_base + j;
  }
  for (int k = 0; k < K; ++k) {
    const int c = __popcll(__ballot((m >> k) & 1u));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt[k], c);
  }
  __syncthreads();
  if (threadIdx.x < K && cnt[threadIdx.x]) atomicAdd(&freq[seg * 32 + threadIdx.x], cnt[threadIdx.x]);
}

// bit position of offset k in the sort key of its segment (plan_bit_positions per segment)
__global__ void plan_pos_all_kernel(PlanSegs P, const int32_t *__restrict__ freq, int32_t *__restrict__ bitpos) {
  const int seg = blockIdx.x, k = threadIdx.x;     // 32 threads
  const int K = P.s[seg].K;
  int p = 0;
  if (k < K) {
    const int fk = freq[seg * 32 + k];
    for (int o = 0; o < K; ++o) {
      const int fo = freq[seg * 32 + o];
      p += (fo > fk || (fo == fk && o < k)) ? 1 : 0;
    }
  }
  bitpos[seg * 32 + k] = p;
}

__global__ void __launch_bounds__(256) plan_key_all_kernel(PlanSegs P, const int32_t *__restrict__ bitpos,
                                                          uint32_t *__restrict__ mask_to_key) {
  __shared__ int pos[32];
  const int seg = blockIdx.y;
  const int rows = P.s[seg].rows;
  if (blockIdx.x * 256 >= rows) return;
  if (threadIdx.x < 32) pos[threadIdx.x] = bitpos[seg * 32 + threadIdx.x];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= rows) return;
  const int g = P.s[seg].row_base + j;
  uint32_t key = 0;
  for (uint32_t mm = mask_to_key[g]; mm; mm &= mm - 1) key |= 1u << pos[__ffs(static_cast<int>(mm)) - 1];
  mask_to_key[g] = key | (static_cast<uint32_t>(P.s[seg].seg_code) << P.s[seg].key_bits);
}

// 256 sorted rows = 8 tiles per block, OR over each aligned 32-lane group (as plan_tiles_kernel)
__global__ void __launch_bounds__(256) plan_tiles_all_kernel(PlanSegs P, const int32_t *__restrict__ bitpos,
                                                            uint32_t *__restrict__ tmask) {
  __shared__ int pos[32];
  const int seg = blockIdx.y;
  const int rows = P.s[seg].rows, K = P.s[seg].K;
  if (blockIdx.x * 256 >= rows) return;
  if (threadIdx.x < 32) pos[threadIdx.x] = bitpos[seg * 32 + threadIdx.x];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  uint32_t key = j < rows ? P.s[seg].key_sorted[P.s[seg].row_base + j] & ((1u << P.s[seg].key_bits) - 1u) : 0u;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) key |= __shfl_xor(key, o, 64);
  if ((threadIdx.x & 31) == 0 && j < rows) {
    uint32_t m = 0;
    for (int k = 0; k < K; ++k) m |= ((key >> pos[k]) & 1u) << k;
    tmask[P.s[seg].tile_base + (j >> 5)] = m;
  }
}

// per segment: tiles by descending number of offsets, ties in ascending tile order (stable,
// deterministic): one workgroup per segment, ranks from ballot match-any like the radix scatter
__global__ void __launch_bounds__(1024) plan_tile_order_all_kernel(PlanSegs P, const uint32_t *__restrict__ tmask,
                                                                  int32_t *__restrict__ torder) {
  __shared__ int hist[33], base[33];
  __shared__ int wcnt[16][33];
  const PlanSeg &S = P.s[blockIdx.x];
  const int nt = (S.rows + 31) / 32;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 33) hist[threadIdx.x] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < nt; t += 1024) atomicAdd(&hist[__popc(tmask[S.tile_base + t])], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 32; b >= 0; --b) { base[b] = run; run += hist[b]; }
  }
  // tiles per number of offsets, behind the segment's final tile masks (SG_PLAN_HIST_WORDS words)
  if (threadIdx.x < SG_PLAN_HIST_WORDS) S.tile_mask[nt + threadIdx.x] = threadIdx.x < 33 ? hist[threadIdx.x] : 0;
  __syncthreads();
  for (int t0 = 0; t0 < nt; t0 += 1024) {
    if (threadIdx.x < 16 * 33) (&wcnt[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const int t = t0 + threadIdx.x;
    const bool valid = t < nt;
    const int b = valid ? __popc(tmask[S.tile_base + t]) : 0;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
      const uint64_t bal = __ballot((b >> bit) & 1);
      peers &= ((b >> bit) & 1) ? bal : ~bal;
    }
    const int lane_rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && lane_rank == 0) wcnt[wave][b] = __popcll(peers);
    __syncthreads();
    if (valid) {
      int off = base[b] + lane_rank;
      for (int w = 0; w < wave; ++w) off += wcnt[w][b];
      torder[S.tile_base + off] = t;
    }
    __syncthreads();
    if (threadIdx.x < 33) {
      int add = 0;
      for (int w = 0; w < 16; ++w) add += wcnt[w][threadIdx.x];
      base[threadIdx.x] += add;
    }
    __syncthreads();
  }
}

// grid = (tile blocks, segments)
// (the radix path sorts table-global row ids -- row_base is subtracted --, the LDS sort writes local ones)
__global__ void __launch_bounds__(256) plan_emit_all_kernel(PlanSegs P, const int32_t *__restrict__ val_local,
                                                           const uint32_t *__restrict__ tmask,
                                                           const int32_t *__restrict__ torder) {
  const int seg = blockIdx.y;
  const int rows = P.s[seg].rows, K = P.s[seg].K, tile_base = P.s[seg].tile_base;
  const int val_base = P.s[seg].row_base;
  const bool local_vals = seg >= P.n_big;
  const int32_t *val_sorted = local_vals ? val_local : P.s[seg].val_sorted;
  const int row_base = local_vals ? 0 : val_base;
  const int nt = (rows + 31) / 32, per_tile = 32 * K;
  const int32_t *nbr = P.s[seg].nbr;
  int32_t *order = P.s[seg].order, *nbr_tiles = P.s[seg].nbr_tiles;
  uint32_t *tile_mask = P.s[seg].tile_mask;
  for (int t2 = blockIdx.x; t2 < nt; t2 += gridDim.x) {
    const int t = torder[tile_base + t2];
    if (threadIdx.x == 0) tile_mask[t2] = tmask[tile_base + t];
    if (threadIdx.x < 32) {
      const int pos = t * 32 + threadIdx.x;
      order[t2 * 32 + threadIdx.x] = pos < rows ? val_sorted[val_base + pos] - row_base : -1;
    }
    for (int e = threadIdx.x; e < per_tile; e += 256) {
      const int r = e / K, k = e - r * K;
      const int pos = t * 32 + r;
      nbr_tiles[static_cast<int64_t>(t2) * per_tile + e] =
          pos < rows ? nbr[static_cast<int64_t>(val_sorted[val_base + pos] - row_base) * K + k] : -1;
    }
  }
}

// scratch of build_plans for R rows in T tiles
static size_t plan_scratch_bytes(size_t R, size_t T) {
  return 2 * align_up(R * 4) + 2 * align_up(T * 4) + 2 * align_up(kPyrMaxSegs * 32 * 4) +
         radix_sort_workspace_bytes(static_cast<int64_t>(R)) + 512;
}

// Tile plans of the P.n tables described by P (row_base / tile_base / sb_base filled by the caller).
// SG_PLAN_ORDER: 0 (default) = device-wide radix sort by mask, heaviest tiles first over the whole
// table; 1 = super-block sort, XCD ranges, heaviest first inside a range; 2 = the same with
// super-block-major emission inside a range.  1 and 2 are the spatially local plans of round 5,
// measured SLOWER on every level of the bench scene (fewer L2 misses per item, but 17-58 % more
// (tile, offset) items than the global sort: profiles/r05_conv_locality.txt) -- kept as A/B knobs.
static int plan_order_mode() {
  static const int mode = getenv("SG_PLAN_ORDER") ? atoi(getenv("SG_PLAN_ORDER")) : 0;
  return mode;
}
// Lays the tables of P out for build_plans: big tables first (they share the radix sort, their
// segment index is part of its key), then the ones one workgroup sorts in LDS.
static void finalize_segs(PlanSegs &P) {
  const int mode = plan_order_mode();
  PlanSeg big[kPyrMaxSegs], small[kPyrMaxSegs];
  int nb = 0, ns = 0;
  // mode 0: every table goes through a radix sort -- the wide ones (K > 8: 27 mask bits) first, then
  // the narrow ones (strided / inverse convs: 8 mask bits), each group with its own sort
  P.n_wide = 0;
  P.wide_rows = 0;
  if (mode == 0) {
    for (int i = 0; i < P.n; ++i)
      if (P.s[i].K > 8) big[nb++] = P.s[i];
    P.n_wide = nb;
    for (int i = 0; i < P.n; ++i)
      if (P.s[i].K <= 8) big[nb++] = P.s[i];
    for (int i = 0; i < nb; ++i) {
      big[i].key_bits = i < P.n_wide ? static_cast<int>(kPlanKeyBits) : 8;
      big[i].seg_code = i < P.n_wide ? i : i - P.n_wide;
    }
  } else {
    for (int i = 0; i < P.n; ++i) small[ns++] = P.s[i];
  }
  int row_base = 0, tile_base = 0, sb_base = 0;
  P.n_big = nb;
  for (int i = 0; i < P.n; ++i) {
    PlanSeg &S = P.s[i];
    S = i < nb ? big[i] : small[i - nb];
    S.row_base = row_base;
    S.tile_base = tile_base;
    S.sb_base = sb_base;
    // mode 0: a small table is ONE super-block (= sorted as a whole); modes 1 / 2: plan_sb_rows
    S.sb_rows = mode == 0 ? (S.rows + 31) / 32 * 32 : plan_sb_rows(S.rows);
    row_base += S.rows;
    tile_base += (S.rows + 31) / 32;
    if (i >= nb) sb_base += (S.rows + S.sb_rows - 1) / S.sb_rows;
    if (i + 1 == nb) P.big_rows = row_base;
    if (i + 1 == P.n_wide) P.wide_rows = row_base;
  }
  if (nb == 0) P.big_rows = 0;
  P.total_rows = row_base;
  P.total_tiles = tile_base;
  P.total_sbs = sb_base;
}

static int build_plans(const PlanSegs &P, void *ws2, size_t ws2_bytes, bool zero_freq, hipStream_t stream,
                       const char *who) {
  Workspace wsp(ws2, ws2_bytes);
  const size_t R = static_cast<size_t>(P.total_rows);
  uint32_t *mask = wsp.take<uint32_t>(R);
  int32_t *val = wsp.take<int32_t>(R);
  uint32_t *tmask = wsp.take<uint32_t>(P.total_tiles);
  int32_t *torder = wsp.take<int32_t>(P.total_tiles);
  int32_t *freq = wsp.take<int32_t>(kPyrMaxSegs * 32);
  int32_t *bitpos = wsp.take<int32_t>(kPyrMaxSegs * 32);
  const size_t rs_bytes = radix_sort_workspace_bytes(static_cast<int64_t>(R));
  void *rs_ws = wsp.take<char>(rs_bytes);
  if (!rs_ws) {
    set_error("%s: build workspace too small", who);
    return SG_ERR_WORKSPACE;
  }
  if (zero_freq) hipMemsetAsync(freq, 0, kPyrMaxSegs * 32 * 4, stream);
  int max_rows = 0, max_big = 0;
  for (int i = 0; i < P.n; ++i) {
    max_rows = P.s[i].rows > max_rows ? P.s[i].rows : max_rows;
    if (i < P.n_big) max_big = P.s[i].rows > max_big ? P.s[i].rows : max_big;
  }
  const dim3 grid((max_rows + 255) / 256, P.n);
  plan_mask_all_kernel<<<grid, 256, 0, stream>>>(P, mask, val, freq);
  static const bool raw_env = getenv("SG_PLAN_RAW") != nullptr;     // developer knob: sort by the raw mask
  if (raw_env) hipMemsetAsync(freq, 0, kPyrMaxSegs * 32 * 4, stream);   // equal counts -> identity permutation
  plan_pos_all_kernel<<<P.n, 32, 0, stream>>>(P, freq, bitpos);
  const int mode = plan_order_mode();
  // ---- big tables: one device-wide radix sort, the table's index in the key bits above the mask
  PlanSegs Q = P;      // + where each table's sorted keys / row ids are
  if (P.n_big > 0) {
    const dim3 gbig((max_big + 255) / 256, P.n_big);
    plan_key_all_kernel<<<gbig, 256, 0, stream>>>(P, bitpos, mask);
    // two sorts, in stream order through the same scratch: the K = 27 tables (27 mask bits + table code:
    // 4 passes) and the K = 8 tables (8 + code: 2 passes) -- one sort of everything moved the 60 % of
    // the rows that belong to narrow tables through 4 passes as well
    struct Group { int first, count, row0, rows, key_bits; } groups[2] = {
        {0, P.n_wide, 0, P.wide_rows, static_cast<int>(kPlanKeyBits)},
        {P.n_wide, P.n_big - P.n_wide, P.wide_rows, P.big_rows - P.wide_rows, 8}};
    for (const Group &g : groups) {
      if (g.count == 0 || g.rows == 0) continue;
      int nbits = g.key_bits;
      for (int n = g.count - 1; n > 0; n >>= 1) ++nbits;
      uint32_t *ms;
      int32_t *vs;
      int rc = radix_sort_pairs(mask + g.row0, val + g.row0, static_cast<int64_t>(g.rows), nbits, rs_ws, rs_bytes,
                                stream, &ms, &vs);
      if (rc != SG_OK) return rc;
      // (a result left in the shared scratch would be overwritten by the next group's sort: both groups
      // take an even number of passes -- 4 and 2 -- and so end in their own mask / val ranges)
      if (ms != mask + g.row0) {
        hipMemcpyAsync(mask + g.row0, ms, static_cast<size_t>(g.rows) * 4, hipMemcpyDeviceToDevice, stream);
        hipMemcpyAsync(val + g.row0, vs, static_cast<size_t>(g.rows) * 4, hipMemcpyDeviceToDevice, stream);
      }
      for (int i = g.first; i < g.first + g.count; ++i) {
        Q.s[i].key_sorted = mask;      // (tables index with their global row_base)
        Q.s[i].val_sorted = val;
      }
    }
    plan_tiles_all_kernel<<<gbig, 256, 0, stream>>>(Q, bitpos, tmask);
  }
  // ---- modes 1 / 2: every table super-block by super-block, one workgroup per super-block sorts in LDS
  //      (measured as a replacement of the radix sort for the small tables of mode 0 too: a 16 384-entry
  //      bitonic sort takes one workgroup 79 us, more than the eight radix launches it would replace)
  if (P.total_sbs > 0) {
    int max_sb = 64;
    for (int i = P.n_big; i < P.n; ++i)
      while (max_sb < P.s[i].sb_rows && max_sb < P.s[i].rows) max_sb <<= 1;
    static std::once_flag once;
    std::call_once(once, [] {
      hipFuncSetAttribute(reinterpret_cast<const void *>(plan_sort_sb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          kSbRowsMax * 8);
    });
    plan_sort_sb_kernel<<<P.total_sbs, 1024, static_cast<size_t>(max_sb) * 8, stream>>>(Q, mask, bitpos, val, tmask);
  }
  if (mode == 0) plan_tile_order_all_kernel<<<P.n, 1024, 0, stream>>>(Q, tmask, torder);
  else plan_tile_order_xcd_kernel<<<dim3(P.n, 8), 1024, 0, stream>>>(Q, tmask, mode, torder);
  plan_emit_all_kernel<<<dim3(min((max_rows + 31) / 32, 2048), P.n), 256, 0, stream>>>(Q, val, tmask, torder);
  return check_launch(who);
}

struct PyrWs {
  uint64_t *keys;     // [L][cap]
  int32_t *vals;      // [L][cap] smallest level-0 row of the site
  int32_t *rowtab;    // [L][cap] row of the site inside its level
  int32_t *slot;      // [L][M0]
  int32_t *pos;       // [L*M0 + 1]
  void *scan_ws;
  size_t scan_bytes;
  uint32_t cap;
};
static bool pyr_carve(void *ws, size_t ws_bytes, int M0, int L, PyrWs *w) {
  Workspace a(ws, ws_bytes);
  const size_t nn = static_cast<size_t>(M0 > 0 ? M0 : 1);
  w->cap = static_cast<uint32_t>(hash_cap(M0));
  w->keys = a.take<uint64_t>(static_cast<size_t>(L) * w->cap);
  w->vals = a.take<int32_t>(static_cast<size_t>(L) * w->cap);
  w->rowtab = a.take<int32_t>(static_cast<size_t>(L) * w->cap);
  w->slot = a.take<int32_t>(static_cast<size_t>(L) * nn);
  w->pos = a.take<int32_t>(static_cast<size_t>(L) * nn + 1);
  w->scan_bytes = scan_workspace_bytes(static_cast<int64_t>(L) * nn);
  w->scan_ws = a.take<char>(w->scan_bytes);
  return w->scan_ws != nullptr;
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


// ---------------------------------------------------------------------------------------------
// whole-pyramid entry points (see the comment block above pyr_insert_kernel)
size_t sg_spconv_pyramid_workspace_bytes(int M0, int n_levels) {
  const size_t nn = static_cast<size_t>(M0 > 0 ? M0 : 1), L = static_cast<size_t>(n_levels > 0 ? n_levels : 1);
  const size_t cap = hash_cap(M0);
  return align_up(L * cap * 8) + 2 * align_up(L * cap * 4) + align_up(L * nn * 4) +
         align_up((L * nn + 1) * 4) + align_up(scan_workspace_bytes(static_cast<int64_t>(L * nn))) + 256;
}

int sg_spconv_pyramid_rows(const int32_t *indices, int M0, const int32_t *shape_host, int n_levels,
                           int32_t *rows_dev, void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(M0 >= 0 && shape_host && rows_dev && n_levels >= 1 && n_levels <= kPyrMaxLevels,
             "sg_spconv_pyramid_rows: bad arguments (M0=%d n_levels=%d)", M0, n_levels);
  hipStream_t stream = as_stream(stream_);
  if (M0 == 0) {
    hipMemsetAsync(rows_dev, 0, static_cast<size_t>(n_levels) * 4, stream);
    return check_launch("sg_spconv_pyramid_rows");
  }
  PyrWs w;
  if (!pyr_carve(ws, ws_bytes, M0, n_levels, &w)) {
    set_error("sg_spconv_pyramid_rows: workspace too small (%zu bytes)", ws_bytes);
    return SG_ERR_WORKSPACE;
  }
  const int L = n_levels;
  {
    FillList f;
    f.add(w.keys, static_cast<size_t>(L) * w.cap * 8, 0xff);
    f.add(w.vals, static_cast<size_t>(L) * w.cap * 4, 0x7f);
    fill_many(f, stream);
  }
  const Shape3 shape{shape_host[0], shape_host[1], shape_host[2]};
  pyr_insert_kernel<<<(M0 + 255) / 256, 256, 0, stream>>>(indices, M0, shape, L, w.keys, w.vals, w.cap,
                                                         w.slot);
  const int32_t *slot = w.slot, *vals = w.vals;
  int32_t *pos = w.pos;
  const int m0 = M0;
  const uint32_t cap = w.cap;
  const int64_t n = static_cast<int64_t>(L) * M0;
  int rc = exclusive_scan(
      [slot, vals, m0, cap] __device__(int64_t t) {
        const int64_t l = t / m0;
        const int sl = slot[t];
        return (sl >= 0 && vals[l * cap + sl] == static_cast<int32_t>(t - l * m0)) ? 1 : 0;
      },
      [pos] __device__(int64_t t, int v) { pos[t] = v; }, n, pos + n, w.scan_ws, w.scan_bytes, stream);
  if (rc != SG_OK) return rc;
  pyr_rows_kernel<<<1, 64, 0, stream>>>(w.pos, M0, L, rows_dev);
  return check_launch("sg_spconv_pyramid_rows");
}

static long long pyr_seg_rows(const sg_pyramid_level *lv, int L, int seg_kind, int l) {
  return seg_kind == 1 ? lv[l + 1].rows : lv[l].rows;   // 0 subm, 1 down (rows of the coarser level), 2 up
}

size_t sg_spconv_pyramid_build_workspace_bytes(const sg_pyramid_level *levels, int n_levels) {
  long long total = 0;
  for (int l = 0; l < n_levels; ++l) {
    total += levels[l].rows;
    if (l + 1 < n_levels) total += levels[l + 1].rows + levels[l].rows;
  }
  const size_t R = static_cast<size_t>(total > 0 ? total : 1);
  const size_t T = R / 32 + static_cast<size_t>(3 * n_levels) + 1;
  return plan_scratch_bytes(R, T);
}

int sg_spconv_pyramid_build(const int32_t *indices, int M0, const int32_t *shape_host, int n_levels,
                            const sg_pyramid_level *levels, void *ws, size_t ws_bytes, void *ws2,
                            size_t ws2_bytes, sg_stream_t stream_) {
  SG_REQUIRE(M0 >= 0 && shape_host && levels && n_levels >= 1 && n_levels <= kPyrMaxLevels,
             "sg_spconv_pyramid_build: bad arguments (M0=%d n_levels=%d)", M0, n_levels);
  SG_REQUIRE(levels[0].rows == M0, "sg_spconv_pyramid_build: levels[0].rows must be num_rows");
  if (M0 == 0) return SG_OK;
  hipStream_t stream = as_stream(stream_);
  const int L = n_levels;
  PyrWs w;
  if (!pyr_carve(ws, ws_bytes, M0, L, &w)) {
    set_error("sg_spconv_pyramid_build: workspace too small (%zu bytes)", ws_bytes);
    return SG_ERR_WORKSPACE;
  }
  SG_REQUIRE(ws2 != nullptr && ws2_bytes >= sg_spconv_pyramid_build_workspace_bytes(levels, L),
             "sg_spconv_pyramid_build: build workspace too small (%zu bytes)", ws2_bytes);
  PyrOut o;
  PlanSegs P;
  P.n = 0;
  auto add_seg = [&](const int32_t *nbr, int rows, int K, const sg_plan_ptrs &pl) {
    PlanSeg &S = P.s[P.n++];
    S.nbr = nbr; S.order = pl.order; S.tile_mask = pl.tile_mask; S.nbr_tiles = pl.nbr_tiles;
    S.rows = rows; S.K = K;
  };
  Shape3 s{shape_host[0], shape_host[1], shape_host[2]};
  o.pre27[0] = 0;
  for (int l = 0; l < kPyrMaxLevels; ++l) {
    const bool in = l < L;
    o.indices[l] = in ? levels[l].indices : nullptr;
    o.nbr[l] = in ? levels[l].nbr : nullptr;
    o.in2out[l] = in ? levels[l].in2out : nullptr;
    o.child[l] = in ? levels[l].child : nullptr;
    o.inv[l] = in ? levels[l].inv : nullptr;
    o.shape[l] = s;
    o.pre27[l + 1] = o.pre27[l] + (in ? static_cast<long long>(levels[l].rows) * 27 : 0);
    s = Shape3{s.s0 / 2, s.s1 / 2, s.s2 / 2};
  }
  for (int l = 0; l < L; ++l) {
    SG_REQUIRE(levels[l].rows >= 0 && levels[l].nbr && levels[l].indices && levels[l].subm.order,
               "sg_spconv_pyramid_build: level %d: missing output pointers", l);
    if (l + 1 < L)
      SG_REQUIRE(levels[l].in2out && levels[l].child && levels[l].inv && levels[l].down.order &&
                     levels[l].up.order, "sg_spconv_pyramid_build: level %d: missing output pointers", l);
  }
  // ---- tile plans of all gather tables as one segmented problem
  for (int l = 0; l < L; ++l) {
    if (levels[l].rows > 0) add_seg(levels[l].nbr, levels[l].rows, 27, levels[l].subm);
    if (l + 1 < L) {
      if (levels[l + 1].rows > 0) add_seg(levels[l].child, levels[l + 1].rows, 8, levels[l].down);
      if (levels[l].rows > 0) add_seg(levels[l].inv, levels[l].rows, 8, levels[l].up);
    }
  }
  if (P.n == 0) return check_launch("sg_spconv_pyramid_build");
  finalize_segs(P);
  // (freq sits where build_plans carves it: the fill below clears it together with the child tables)
  int32_t *freq;
  {
    Workspace wsp(ws2, ws2_bytes);
    const size_t R = static_cast<size_t>(P.total_rows);
    wsp.take<uint32_t>(R);
    wsp.take<int32_t>(R);
    wsp.take<uint32_t>(P.total_tiles);
    wsp.take<int32_t>(P.total_tiles);
    freq = wsp.take<int32_t>(kPyrMaxSegs * 32);
    if (!freq) {
      set_error("sg_spconv_pyramid_build: build workspace too small");
      return SG_ERR_WORKSPACE;
    }
  }
  {      // one launch: the strided tables' "no child" marks and the offset histogram of the plans
    static_assert(kPyrMaxLevels + 1 <= kFillMax, "one fill region per level + the histogram");
    FillList f;
    for (int l = 0; l + 1 < L; ++l)
      if (levels[l + 1].rows > 0) f.add(levels[l].child, static_cast<size_t>(levels[l + 1].rows) * 8 * 4, 0xff);
    f.add(freq, kPyrMaxSegs * 32 * 4, 0);
    fill_many(f, stream);
  }
  // ---- coordinates, strided pairs, SubM tables of all levels
  const int g_all = grid_for(static_cast<int64_t>(L) * M0, 256, 4096);
  pyr_emit_kernel<<<g_all, 256, 0, stream>>>(indices, M0, L, w.vals, w.rowtab, w.cap, w.slot, w.pos, o);
  if (L > 1) {
    pyr_link_kernel<<<grid_for(static_cast<int64_t>(L - 1) * M0, 256, 4096), 256, 0, stream>>>(
        indices, M0, L, w.vals, w.rowtab, w.cap, w.slot, w.pos, o);
  }
  pyr_subm_kernel<<<grid_for(o.pre27[L], 256, 8192), 256, 0, stream>>>(L, w.keys, w.rowtab, w.cap, o);
  return build_plans(P, ws2, ws2_bytes, false, stream, "sg_spconv_pyramid_build");
}

size_t sg_spconv_plan_workspace_bytes(int M) {
  const size_t nn = static_cast<size_t>(M > 0 ? M : 1);
  return plan_scratch_bytes(nn, (nn + 31) / 32 + 1);
}

int sg_spconv_plan(const int32_t *nbr, int M, int K, int32_t *order, uint32_t *tile_mask,
                   int32_t *nbr_tiles, void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(M >= 0 && K >= 1 && K <= 27, "sg_spconv_plan: bad arguments (M=%d K=%d)", M, K);
  if (M == 0) return SG_OK;
  PlanSegs P;
  P.n = 1;
  P.s[0] = PlanSeg{nbr, order, tile_mask, nbr_tiles, M, K, 0, 0, 0, 0, 0, 0, nullptr, nullptr};
  finalize_segs(P);
  return build_plans(P, ws, ws_bytes, true, as_stream(stream_), "sg_spconv_plan");
}

}  // extern "C"
