// scan.h -- device-wide exclusive prefix sum over int32 values produced by a functor.
// Two launches up to 4 M values (block reduce; block scan + carry, the carry summed by the block itself),
// three beyond (block reduce, single-block scan of the block sums, block scan + carry);
// the inputs here are tiny next to the gather/scatter traffic, so this is launch-bound and
// deliberately simple.  Deterministic.
#pragma once
#include "common.h"

namespace sg {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;  // 2048 values per workgroup

inline size_t scan_workspace_bytes(int64_t n) {
  return align_up(((n + kScanTile - 1) / kScanTile + 1) * sizeof(int32_t));
}

// block-wide inclusive scan of one value per thread (256 threads = 4 waves)
__device__ __forceinline__ int block_incl_scan_256(int v, int *lds4, int *block_total) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  int s = wave_incl_scan(v);
  if (l == 63) lds4[w] = s;
  __syncthreads();
  int carry = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < w) carry += lds4[i];
  if (block_total) *block_total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return s + carry;
}

template <typename In>
__global__ void __launch_bounds__(kScanBlock) scan_reduce_kernel(In in, int64_t n,
                                                                int32_t *block_sums) {
  __shared__ int lds4[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
  int v = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    int64_t idx = base + i * kScanBlock + threadIdx.x;
    if (idx < n) v += in(idx);
  }
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// exclusive scan of block_sums in place by one workgroup; total -> *total_out (may be null)
void launch_scan_block_sums(int32_t *block_sums, int num_blocks, int32_t *total_out,
                            hipStream_t stream);  // core.hip

// RAW = true: block_sums holds the blocks' own sums (scan_reduce_kernel's output, no single-block scan
// in between): the block adds up the sums of the blocks before it by itself -- <= kScanRawBlocks loads
// spread over 256 threads -- and the last block writes the total.  Two launches instead of three for
// everything up to kScanRawBlocks * 2048 values (a scan here is launch-bound: ~4.5 us per launch, ten
// scans per scan of the model).
constexpr int kScanRawBlocks = 2048;
template <typename In, typename Out, bool RAW = false>
__global__ void __launch_bounds__(kScanBlock) scan_apply_kernel(In in, Out out, int64_t n,
                                                               const int32_t *block_sums,
                                                               int32_t *total_out = nullptr) {
  __shared__ int lds4[4];
  __shared__ int carry_s;
  int carry_in = 0;
  if (RAW) {
    int v = 0;
    for (int b = threadIdx.x; b < static_cast<int>(blockIdx.x); b += kScanBlock) v += block_sums[b];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    carry_in = carry_s;
    if (total_out != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
      *total_out = carry_in + block_sums[blockIdx.x];
  }
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
  // thread-contiguous items so the scan order is the index order
  const int64_t first = base + static_cast<int64_t>(threadIdx.x) * kScanItems;
  int vals[kScanItems];
  int tsum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    int64_t idx = first + i;
    vals[i] = idx < n ? in(idx) : 0;
    tsum += vals[i];
  }
  int incl = block_incl_scan_256(tsum, lds4, nullptr);
  int run = (RAW ? carry_in : block_sums[blockIdx.x]) + incl - tsum;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    int64_t idx = first + i;
    if (idx < n) out(idx, run);
    run += vals[i];
  }
}

// exclusive scan: out(i, sum_{j<i} in(j)); total written to total_out (device, may be null)
template <typename In, typename Out>
int exclusive_scan(In in, Out out, int64_t n, int32_t *total_out, void *ws, size_t ws_bytes,
                   hipStream_t stream) {
  if (ws_bytes < scan_workspace_bytes(n)) {
    set_error("exclusive_scan: workspace too small (%zu < %zu)", ws_bytes, scan_workspace_bytes(n));
    return SG_ERR_WORKSPACE;
  }
  int32_t *block_sums = static_cast<int32_t *>(ws);
  int num_blocks = static_cast<int>((n + kScanTile - 1) / kScanTile);
  if (num_blocks == 0) {
    if (total_out) hipMemsetAsync(total_out, 0, sizeof(int32_t), stream);
    return SG_OK;
  }
  scan_reduce_kernel<<<num_blocks, kScanBlock, 0, stream>>>(in, n, block_sums);
  if (num_blocks <= kScanRawBlocks) {
    scan_apply_kernel<In, Out, true><<<num_blocks, kScanBlock, 0, stream>>>(in, out, n, block_sums, total_out);
  } else {
    launch_scan_block_sums(block_sums, num_blocks, total_out, stream);
    scan_apply_kernel<In, Out, false><<<num_blocks, kScanBlock, 0, stream>>>(in, out, n, block_sums);
  }
  return check_launch("exclusive_scan");
}

}  // namespace sg
