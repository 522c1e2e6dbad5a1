// radix_sort.h -- stable LSD radix sort of (uint32 key, int32 value) pairs, 8 bits per pass.
// Used to order sparse-conv output rows by their neighbour bit mask (spconv_rulebook.hip).
// Per pass: block histograms (digit-major) -> device-wide exclusive scan -> stable scatter.
// Ranks inside a wave come from ballot "match-any" masks, so the scatter needs no sorting in
// LDS and keeps the pass stable.
#pragma once
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

static __global__ void __launch_bounds__(kRsBlock) rs_scatter_kernel(const uint32_t *__restrict__ keys,
                                                             const int32_t *__restrict__ vals,
                                                             int64_t n, int shift, int nblk, int items,
                                                             const int32_t *__restrict__ hist,
                                                             const int32_t *__restrict__ totals,
                                                             uint32_t *__restrict__ keys_out,
                                                             int32_t *__restrict__ vals_out) {
  __shared__ int base[kRsBuckets];
  __shared__ int wcnt[4][kRsBuckets];
  __shared__ int lds4[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (totals) {
    // small inputs: hist holds raw per-block counts; this block's base of digit d =
    // (exclusive prefix of the digit totals) + (counts of d in the blocks before this one)
    const int64_t row = static_cast<int64_t>(threadIdx.x) * nblk;
    int before = 0;
    for (int b = 0; b < static_cast<int>(blockIdx.x); ++b) before += hist[row + b];
    const int tot = totals[threadIdx.x];
    const int incl = block_incl_scan_256(tot, lds4, nullptr);
    base[threadIdx.x] = incl - tot + before;
  } else {
    base[threadIdx.x] = hist[static_cast<int64_t>(threadIdx.x) * nblk + blockIdx.x];
  }
  const int64_t tile = static_cast<int64_t>(blockIdx.x) * items * kRsBlock;
  for (int r = 0; r < items; ++r) {
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = tile + r * kRsBlock + threadIdx.x;
    const bool valid = i < n;
    const uint32_t key = valid ? keys[i] : 0u;
    const int32_t val = valid ? vals[i] : 0;
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
      int off = base[d] + lane_rank;
      for (int w = 0; w < wave; ++w) off += wcnt[w][d];
      keys_out[off] = key;
      vals_out[off] = val;
    }
    __syncthreads();
    base[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] +
                         wcnt[3][threadIdx.x];
    __syncthreads();
  }
}

// Sorts by the low `num_bits` of the keys.  keys/vals are clobbered; the sorted result is in
// (*keys_sorted, *vals_sorted), which alias either the inputs or the workspace buffers.
inline int radix_sort_pairs(uint32_t *keys, int32_t *vals, int64_t n, int num_bits, void *ws,
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
      rs_scatter_kernel<<<nblk, kRsBlock, 0, stream>>>(kin, vin, n, shift, nblk, items, hist, t, kout, vout);
    } else {
      rs_hist_kernel<<<nblk, kRsBlock, 0, stream>>>(kin, n, shift, nblk, items, hist, nullptr);
      int32_t *h = hist;
      int rc = exclusive_scan([h] __device__(int64_t i) { return h[i]; },
                              [h] __device__(int64_t i, int v) { h[i] = v; }, hist_n, nullptr, sws,
                              sbytes, stream);
      if (rc != SG_OK) return rc;
      rs_scatter_kernel<<<nblk, kRsBlock, 0, stream>>>(kin, vin, n, shift, nblk, items, hist, nullptr,
                                                       kout, vout);
    }
    uint32_t *tk = kin; kin = kout; kout = tk;
    int32_t *tv = vin; vin = vout; vout = tv;
  }
  *keys_sorted = kin;
  *vals_sorted = vin;
  return check_launch("radix_sort_pairs");
}

}  // namespace sg
