// eval_ops.hip -- the GPU half of instance evaluation: prediction x ground-truth
// intersection counts for ScanNetEval.assign_instances_for_scan
// (softgroup/evaluation/instance_eval.py:228-309), which the reference computes with one
// np.logical_and + count_nonzero over all N points per (prediction, GT instance) pair in a
// multiprocessing pool.  Here every mask point is visited once: masks arrive as runs (what the RLE
// strings hold), a thread per mask point looks up the point's GT slot and the wave adds one
// atomic per distinct (prediction, slot) it sees.
//   counts[p, slot]: slot < n_gt = GT instance index, slot == n_gt = "void" (label not evaluated)
// HBM-bound: total mask points * 4 B gathered + runs.
#include "common.h"

namespace sg {

__global__ void __launch_bounds__(256) eval_intersections_kernel(const int32_t *__restrict__ run_start,
                                                                const int64_t *__restrict__ run_off,
                                                                const int32_t *__restrict__ run_pred,
                                                                int n_runs, int64_t total_points,
                                                                const int32_t *__restrict__ gt_slot,
                                                                int n_slots, int32_t *__restrict__ counts) {
  const int lane = threadIdx.x & 63;
  for (int64_t t0 = (blockIdx.x * 256LL + threadIdx.x) - lane; t0 < total_points; t0 += gridDim.x * 256LL) {
    const int64_t t = t0 + lane;
    const bool valid = t < total_points;
    int key = -1;
    if (valid) {
      int lo = 0, hi = n_runs;                 // last run with run_off[r] <= t
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (run_off[mid] <= t) lo = mid; else hi = mid;
      }
      const int point = run_start[lo] + static_cast<int>(t - run_off[lo]);
      key = run_pred[lo] * n_slots + gt_slot[point];
    }
    uint64_t todo = __ballot(valid);
    while (todo) {                             // one atomic per distinct (prediction, slot) of the wave
      const int leader = __ffsll(static_cast<long long>(todo)) - 1;
      const int k = __shfl(key, leader, 64);
      const uint64_t same = __ballot(valid && key == k) & todo;
      if (lane == leader) atomicAdd(&counts[k], __popcll(same));
      todo &= ~same;
    }
  }
}

}  // namespace sg

using namespace sg;

extern "C" {

int sg_eval_intersections(const int32_t *run_start, const int64_t *run_off, const int32_t *run_pred,
                          int n_runs, int64_t total_points, const int32_t *gt_slot, int n_pred,
                          int n_slots, int32_t *counts, sg_stream_t stream_) {
  SG_REQUIRE(n_runs >= 0 && total_points >= 0 && n_pred >= 0 && n_slots >= 1,
             "sg_eval_intersections: bad arguments");
  hipStream_t stream = as_stream(stream_);
  hipMemsetAsync(counts, 0, static_cast<size_t>(n_pred) * n_slots * 4, stream);
  if (n_runs == 0 || total_points == 0) return check_launch("sg_eval_intersections");
  eval_intersections_kernel<<<grid_for(total_points, 256, 8192), 256, 0, stream>>>(
      run_start, run_off, run_pred, n_runs, total_points, gt_slot, n_slots, counts);
  return check_launch("sg_eval_intersections");
}

}  // extern "C"
