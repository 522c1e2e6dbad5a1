// radix_sort.h -- stable LSD radix sort of (uint32 key, int32 value) pairs, 8 bits per pass.
// Used to order sparse-conv output rows by their neighbour bit mask (spconv_rulebook.hip).
// Per pass: block histograms (digit-major) -> device-wide exclusive scan -> stable scatter.
// Ranks inside a wave come from ballot "match-any" masks, so the scatter needs no sorting in
// LDS and keeps the pass stable.
#pragma once
#include <mutex>

#include "common.h"
#include "scan.h"

namespace sg {

constexpr int kRsBlock = 256;
constexpr int kRsItems = 8;
constexpr int kRsTile = kRsBlock * kRsItems;
constexpr int kRsBuckets = 256;

inline size_t radix_sort_workspace_bytes(int64_t n) {
  const int64_t nblk = (n + kRsTile - 1) / kRsTile;
  const int64_t hist = nblk * kRsBuckets + 1;
  return align_up(hist * 4) + align_up(scan_workspace_bytes(hist) + 4 * kRsBuckets * 4) +
         2 * align_up(n * 4) + 256;
}

static __global__ void __launch_bounds__(kRsBlock) rs_hist_kernel(const uint32_t *__restrict__ keys,
                                                          int64_t n, int shift, int nblk, int items,
                                                          int32_t *__restrict__ hist,
                                                          int32_t *totals) {
  __shared__ int h[kRsBuckets];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * items * kRsBlock;
  for (int r = 0; r < items; ++r) {
    const int64_t i = base + r * kRsBlock + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xff], 1);
  }
  __syncthreads();
  hist[static_cast<int64_t>(threadIdx.x) * nblk + blockIdx.x] = h[threadIdx.x];
  if (totals && h[threadIdx.x]) atomicAdd(&totals[threadIdx.x], h[threadIdx.x]);
}

// Stable scatter of one pass.  The block's tile is first sorted by digit INSIDE LDS (rank of a pair =
// start of its digit in the tile + pairs of that digit in earlier rounds / waves / lanes), then written
// out in tile order: consecutive threads hold consecutive pairs of a digit's run, i.e. coalesced stores,
// and nothing waits for a store before the kernel ends.  (The first version stored every round's pairs
// straight to their final, scattered addresses; `__syncthreads()` waits for outstanding stores on this
// target, so each of a block's 8-16 rounds paid a full store round trip: 23-34 us per pass over 220-550 k
// pairs, profiles/r05_index_build.txt.)  Dynamic LDS: items * 256 * 8 bytes.
static __global__ void __launch_bounds__(kRsBlock) rs_scatter_kernel(const uint32_t *__restrict__ keys,
                                                             const int32_t *__restrict__ vals,
                                                             int64_t n, int shift, int nblk, int items,
                                                             const int32_t *__restrict__ hist,
                                                             const int32_t *__restrict__ totals,
                                                             uint32_t *__restrict__ keys_out,
                                                             int32_t *__restrict__ vals_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
  uint32_t *skey = reinterpret_cast<uint32_t *>(rs_lds);
  int32_t *sval = reinterpret_cast<int32_t *>(rs_lds) + items * kRsBlock;
  __shared__ int base[kRsBuckets];        // global position of the tile's first pair of each digit
  __shared__ int dstart[kRsBuckets];      // start of the digit's run inside the tile
  __shared__ int run[kRsBuckets];         // pairs of the digit placed so far (earlier rounds)
  __shared__ int wcnt[4][kRsBuckets];
  __shared__ int lds4[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tile = static_cast<int64_t>(blockIdx.x) * items * kRsBlock;
  const int in_tile = static_cast<int>(n - tile < static_cast<int64_t>(items) * kRsBlock ? n - tile
                                                                                       : static_cast<int64_t>(items) * kRsBlock);
  // ---- digit counts of this tile (LDS atomics; order does not matter for counts)
  run[threadIdx.x] = 0;
  __syncthreads();
  for (int e = threadIdx.x; e < in_tile; e += kRsBlock) atomicAdd(&run[(keys[tile + e] >> shift) & 0xff], 1);
  __syncthreads();
  const int mine = run[threadIdx.x];
  {
    const int incl = block_incl_scan_256(mine, lds4, nullptr);
    dstart[threadIdx.x] = incl - mine;
  }
  if (totals) {
    // small inputs: hist holds raw per-block counts; this block's base of digit d =
    // (exclusive prefix of the digit totals) + (counts of d in the blocks before this one)
    // (16 independent loads per step: as a plain loop the compiler waits for every load before the
    // next -- ~0.25 us each -- and the LAST block of a 113-block pass spent 25 us here, which was the
    // length of the whole kernel whatever the number of pairs)
    const int32_t *row = hist + static_cast<int64_t>(threadIdx.x) * nblk;
    const int nb = static_cast<int>(blockIdx.x);
    int before = 0, b = 0;
    for (; b + 16 <= nb; b += 16) {
      int v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = row[b + q];
#pragma unroll
      for (int q = 0; q < 16; ++q) before += v[q];
    }
    for (; b < nb; ++b) before += row[b];
    const int tot = totals[threadIdx.x];
    const int incl = block_incl_scan_256(tot, lds4, nullptr);
    base[threadIdx.x] = incl - tot + before;
  } else {
    base[threadIdx.x] = hist[static_cast<int64_t>(threadIdx.x) * nblk + blockIdx.x];
  }
  run[threadIdx.x] = 0;
  __syncthreads();
  // ---- rounds of 256 pairs: rank inside the tile, pair into its LDS slot
  uint32_t pk[8];
  int32_t pv[8];
  for (int r = 0; r < items; ++r) {
    if ((r & 7) == 0) {       // the loads of 8 rounds are requested together (items is a multiple of 8)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int64_t j = tile + static_cast<int64_t>(r + q) * kRsBlock + threadIdx.x;
        pk[q] = j < n ? keys[j] : 0u;
        pv[q] = j < n ? vals[j] : 0;
      }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const bool valid = tile + static_cast<int64_t>(r) * kRsBlock + threadIdx.x < n;
    uint32_t key = 0u;
    int32_t val = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q)        // (static register indices: no scratch)
      if ((r & 7) == q) { key = pk[q]; val = pv[q]; }
    const int d = (key >> shift) & 0xff;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bal = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? bal : ~bal;
    }
    const int lane_rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && lane_rank == 0) wcnt[wave][d] = __popcll(peers);
    __syncthreads();
    if (valid) {
      int off = dstart[d] + run[d] + lane_rank;
      for (int w = 0; w < wave; ++w) off += wcnt[w][d];
      skey[off] = key;
      sval[off] = val;
    }
    __syncthreads();
    run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] +
                        wcnt[3][threadIdx.x];
  }
  __syncthreads();
  // ---- tile order out: pair e of the tile (digit d, e - dstart[d] into the digit's run) -> base[d] + that
  for (int e = threadIdx.x; e < in_tile; e += kRsBlock) {
    const uint32_t key = skey[e];
    const int d = (key >> shift) & 0xff;
    const int off = base[d] + (e - dstart[d]);
    keys_out[off] = key;
    vals_out[off] = sval[e];
  }
}

// Sorts by the low `num_bits` of the keys.  keys/vals are clobbered; the sorted result is in
// (*keys_sorted, *vals_sorted), which alias either the inputs or the workspace buffers.

static inline int radix_sort_pairs(uint32_t *keys, int32_t *vals, int64_t n, int num_bits, void *ws,
                            size_t ws_bytes, hipStream_t stream, uint32_t **keys_sorted,
                            int32_t **vals_sorted) {
  *keys_sorted = keys;
  *vals_sorted = vals;
  if (n <= 1 || num_bits <= 0) return SG_OK;
  if (ws_bytes < radix_sort_workspace_bytes(n)) {
    set_error("radix_sort_pairs: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  // keys per workgroup: 2048, doubled (up to 16k) while that keeps the sort at <= 256 workgroups,
  // the size up to which a pass is 2 launches instead of 5 (see below)
  int items = kRsItems;
  while (items < 64 && (n + static_cast<int64_t>(items) * kRsBlock - 1) / (static_cast<int64_t>(items) * kRsBlock) > 256)
    items *= 2;
  const int64_t tile = static_cast<int64_t>(items) * kRsBlock;
  const int nblk = static_cast<int>((n + tile - 1) / tile);
  const size_t lds = static_cast<size_t>(tile) * 8;       // the scatter kernel's tile staging (<= 128 KB)
  // The tile staging needs up to 128 KB of dynamic LDS + ~7 KB static: gfx950's 160 KB per workgroup.
  // The attribute is per device (and per translation unit, like the static kernel it configures):
  // set once for every device this process sorts on, result checked.
  {
    static std::mutex mu;
    static uint64_t done_mask = 0;      // bit d: device d configured
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (dev < 0 || dev >= 64 || !((done_mask >> dev) & 1)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(rs_scatter_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 64 * kRsBlock * 8) != hipSuccess) {
        (void)hipGetLastError();
        set_error("radix_sort_pairs: %d bytes of dynamic LDS are not available on this device (gfx950: 160 KB "
                  "per workgroup)", 64 * kRsBlock * 8);
        return SG_ERR_LAUNCH;
      }
      if (dev >= 0 && dev < 64) done_mask |= 1ull << dev;
    }
  }
  const int64_t hist_n = static_cast<int64_t>(nblk) * kRsBuckets;
  Workspace a(ws, ws_bytes);
  int32_t *hist = a.take<int32_t>(hist_n + 1);
  const size_t sbytes = scan_workspace_bytes(hist_n) + 4 * kRsBuckets * 4;
  void *sws = a.take<char>(sbytes);
  uint32_t *k2 = a.take<uint32_t>(n);
  int32_t *v2 = a.take<int32_t>(n);
  uint32_t *kin = keys, *kout = k2;
  int32_t *vin = vals, *vout = v2;
  const int n_pass = (num_bits + 7) / 8;
  // up to 256 blocks (512k keys) the per-block bases are rebuilt inside the scatter kernel from
  // raw counts + digit totals: 2 launches per pass instead of 5
  const bool local_scan = nblk <= 256 && n_pass <= 4;
  int32_t *totals = static_cast<int32_t *>(sws);    // [n_pass][256], lives in the scan scratch
  if (local_scan) hipMemsetAsync(totals, 0, static_cast<size_t>(n_pass) * kRsBuckets * 4, stream);
  for (int shift = 0, pass = 0; shift < num_bits; shift += 8, ++pass) {
    if (local_scan) {
      int32_t *t = totals + pass * kRsBuckets;
      rs_hist_kernel<<<nblk, kRsBlock, 0, stream>>>(kin, n, shift, nblk, items, hist, t);
      rs_scatter_kernel<<<nblk, kRsBlock, lds, stream>>>(kin, vin, n, shift, nblk, items, hist, t, kout, vout);
      if (const int rc = check_launch("radix_sort_pairs(scatter)"); rc != SG_OK) return rc;
    } else {
      rs_hist_kernel<<<nblk, kRsBlock, 0, stream>>>(kin, n, shift, nblk, items, hist, nullptr);
      int32_t *h = hist;
      int rc = exclusive_scan([h] __device__(int64_t i) { return h[i]; },
                              [h] __device__(int64_t i, int v) { h[i] = v; }, hist_n, nullptr, sws,
                              sbytes, stream);
      if (rc != SG_OK) return rc;
      rs_scatter_kernel<<<nblk, kRsBlock, lds, stream>>>(kin, vin, n, shift, nblk, items, hist, nullptr,
                                                         kout, vout);
      if (const int rc = check_launch("radix_sort_pairs(scatter)"); rc != SG_OK) return rc;
    }
    uint32_t *tk = kin; kin = kout; kout = tk;
    int32_t *tv = vin; vin = vout; vout = tv;
  }
  *keys_sorted = kin;
  *vals_sorted = vin;
  return check_launch("radix_sort_pairs");
}

}  // namespace sg
