// instances.hip -- instance extraction for the inference result (the step right after the hot
// path).  Replaces the dense-mask part of SoftGroup.get_instances
// (softgroup/model/softgroup.py:537-604): for every instance class i and proposal p the reference
// builds an int32 [nProposal, N] matrix, sets mask[p, point] = 1 where mask_scores[:, i] > thr,
// filters proposals (cls score, >= min_npoint points), copies the matrix to the host and
// run-length encodes each row in Python (util/rle.py:5-19).
//
// Here, for ALL classes at once:
//   sg_instance_npoint   npoint[p, i] = #{pairs of proposal p with mask_scores[e, i] > thr}
//                        (pairs arrive grouped by proposal: one atomic per (wave, proposal, class))
//   -- host: keep[p, i] = cls_prob > cls_thr && npoint >= min_npoint, kept (class, proposal) pairs
//      numbered class-major (the reference's output order), tiny [nP, nc] work --
//   sg_instance_runs     one N-bit row per KEPT instance only (32x18 less memory than the
//                        reference's per-class int32 matrices), set by atomicOr; run starts / ends
//                        are the 0->1 / 1->0 transitions of the bit rows, numbered by one
//                        device-wide scan of the per-word start counts; since every run is closed
//                        inside its own row, the r-th start and the r-th end of the scan order
//                        belong together.  Output: starts[], ends[] (exclusive), bounds[k] = first
//                        run of instance k -- exactly what sg_rle_format_host turns into the
//                        reference's "start len start len ..." strings.
// HBM-bound streaming work: S*nc score reads + n_kept*N/8 bitmap bytes (written once, read 3x).
#include "common.h"
#include "scan.h"

namespace sg {

// Every wave walks a CONTIGUOUS range of the (proposal, point) pairs, 64 at a time; pairs of one
// proposal are consecutive, so a chunk is usually one proposal: its per-class counts (popcount of
// the wave's ballot) are kept by lane = class and flushed with one atomic per class when the
// proposal changes -- a 149 k-point proposal costs nc atomics per wave instead of nc per 64 points
// (all of them on the same nc words).  A chunk that holds a boundary falls back to one atomic per
// run.  Every lane reads ITS pair's row of class scores once.
__global__ void __launch_bounds__(256) instance_npoint_kernel(const int32_t *__restrict__ pairs,
                                                             const float *__restrict__ mask_scores,
                                                             int64_t S, int stride, float thr, int nc,
                                                             int32_t *__restrict__ npoint) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * 4;
  const int64_t per = (((S + n_waves - 1) / n_waves) + 63) & ~63LL;
  const int64_t begin = (static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * per;
  const int64_t end = begin + per < S ? begin + per : S;
  int cur = -1, acc = 0;      // lane i < nc: points of proposal `cur` above the threshold in class i
  auto flush = [&]() {
    if (cur >= 0 && lane < nc && acc) atomicAdd(&npoint[static_cast<int64_t>(cur) * nc + lane], acc);
    acc = 0;
  };
  for (int64_t e0 = begin; e0 < end; e0 += 64) {
    const int64_t e = e0 + lane;
    const bool valid = e < end;
    const int p = valid ? pairs[2 * e] : -1;
    const int p0 = __shfl(p, 0, 64);
    const float *row = mask_scores + (valid ? e : 0) * stride;
    if (nc <= 64 && __ballot(valid && p != p0) == 0ull) {          // one proposal in this chunk
      if (p0 != cur) {
        flush();
        cur = p0;
      }
      for (int i = 0; i < nc; ++i) {
        const int cnt = __popcll(__ballot(valid && row[i] > thr));
        if (lane == i) acc += cnt;
      }
      continue;
    }
    flush();
    cur = -1;
    const int p_prev = __shfl_up(p, 1, 64);
    const bool head = valid && (lane == 0 || p != p_prev);
    const uint64_t heads = __ballot(head);
    uint64_t span = 0;
    if (head) {
      const uint64_t above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
      const int stop = above ? __ffsll(static_cast<long long>(above)) - 1 : 64;
      span = (stop == 64 ? ~0ull : ((1ull << stop) - 1ull)) & ~((1ull << lane) - 1ull);
    }
    for (int i = 0; i < nc; ++i) {
      const uint64_t ons = __ballot(valid && row[i] > thr);
      if (head) {
        const int cnt = __popcll(ons & span);
        if (cnt) atomicAdd(&npoint[static_cast<int64_t>(p) * nc + i], cnt);
      }
    }
  }
  flush();
}

__global__ void __launch_bounds__(256) instance_bitmap_kernel(const int32_t *__restrict__ pairs,
                                                             const float *__restrict__ mask_scores,
                                                             int64_t S, int stride, float thr, int nc,
                                                             const int32_t *__restrict__ inst_of,
                                                             int n_prop, int words,
                                                             uint32_t *__restrict__ bits) {
  for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < S; e += gridDim.x * 256LL) {
    const int2 pq = reinterpret_cast<const int2 *>(pairs)[e];
    const float *row = mask_scores + e * stride;      // (the pair's class scores: read once, not per class)
    for (int i = 0; i < nc; ++i) {
      if (!(row[i] > thr)) continue;
      const int k = inst_of[static_cast<int64_t>(i) * n_prop + pq.x];
      if (k < 0) continue;
      atomicOr(&bits[static_cast<int64_t>(k) * words + (pq.y >> 5)], 1u << (pq.y & 31));
    }
  }
}

__device__ __forceinline__ uint32_t run_starts(const uint32_t *__restrict__ bits, int64_t t, int j) {
  const uint32_t w = bits[t];
  const uint32_t carry = j > 0 ? bits[t - 1] >> 31 : 0u;
  return w & ~((w << 1) | carry);
}

__global__ void __launch_bounds__(256) instance_runs_emit_kernel(const uint32_t *__restrict__ bits,
                                                                int64_t total_words, int words,
                                                                const int32_t *__restrict__ base,
                                                                int32_t *__restrict__ starts,
                                                                int32_t *__restrict__ ends,
                                                                int64_t *__restrict__ bounds,
                                                                int64_t capacity) {
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total_words; t += gridDim.x * 256LL) {
    const int j = static_cast<int>(t % words);
    const uint32_t w = bits[t];
    const uint32_t carry = j > 0 ? bits[t - 1] >> 31 : 0u;
    const uint32_t next = j + 1 < words ? bits[t + 1] & 1u : 0u;
    uint32_t sb = w & ~((w << 1) | carry);
    uint32_t eb = w & ~((w >> 1) | (next << 31));
    const int b0 = base[t];
    if (j == 0) bounds[t / words] = b0;
    int r = b0;
    while (sb) {
      const int b = __ffs(static_cast<int>(sb)) - 1;
      sb &= sb - 1;
      if (r < capacity) starts[r] = j * 32 + b;
      ++r;
    }
    // a run that is open across the lower word boundary started earlier: its end is numbered one
    // below this word's first start
    r = b0 - static_cast<int>(carry & w & 1u);
    while (eb) {
      const int b = __ffs(static_cast<int>(eb)) - 1;
      eb &= eb - 1;
      if (r < capacity) ends[r] = j * 32 + b + 1;
      ++r;
    }
  }
}


// ---- RLE text on the device: "start len start len ..." (1-based starts, util/rle.py:5-19) of every
// kept instance straight from the run arrays, so that neither the runs (8 B each) nor a host-side
// integer formatter are on the result path: one length pass, one scan, one digit pass, and the
// text + one offset per instance go to the host.
__device__ __forceinline__ int dec_digits(uint32_t v) {
  return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5
       : v < 1000000u ? 6 : v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
__device__ __forceinline__ void put_digits(uint8_t *p, uint32_t v, int n) {
  for (int i = n - 1; i >= 0; --i) {
    p[i] = static_cast<uint8_t>('0' + v % 10u);
    v /= 10u;
  }
}
// every run takes digits(start+1) + 1 + digits(len) + 1 bytes (a space after both numbers; the
// space after an instance's last run is not part of its text)
__device__ __forceinline__ int run_text_len(const int32_t *starts, const int32_t *ends, int64_t r) {
  return dec_digits(static_cast<uint32_t>(starts[r] + 1)) + dec_digits(static_cast<uint32_t>(ends[r] - starts[r])) + 2;
}
__global__ void __launch_bounds__(256) rle_text_kernel(const int32_t *__restrict__ starts,
                                                      const int32_t *__restrict__ ends,
                                                      const int32_t *__restrict__ off,
                                                      const int64_t *__restrict__ n_runs_p,
                                                      int64_t capacity, uint8_t *__restrict__ text) {
  const int64_t n_runs = *n_runs_p < capacity ? *n_runs_p : capacity;
  for (int64_t r = blockIdx.x * 256LL + threadIdx.x; r < n_runs; r += gridDim.x * 256LL) {
    const uint32_t s = static_cast<uint32_t>(starts[r] + 1), l = static_cast<uint32_t>(ends[r] - starts[r]);
    const int ds = dec_digits(s), dl = dec_digits(l);
    uint8_t *p = text + off[r];
    put_digits(p, s, ds);
    p[ds] = ' ';
    put_digits(p + ds + 1, l, dl);
    p[ds + 1 + dl] = ' ';
  }
}
__global__ void __launch_bounds__(256) rle_offsets_kernel(const int64_t *__restrict__ bounds, int n_inst,
                                                         const int32_t *__restrict__ off,
                                                         const int32_t *__restrict__ total,
                                                         int64_t *__restrict__ text_off) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g > n_inst) return;
  const int64_t b = bounds[g];
  text_off[g] = b < bounds[n_inst] ? off[b] : *total;
}

__global__ void instance_runs_total_kernel(const int32_t *__restrict__ total, int n_kept,
                                           int64_t *__restrict__ bounds) {
  bounds[n_kept] = *total;
}

// ---- panoptic fusion (SoftGroup.panoptic_fusion, softgroup/model/softgroup.py:606-639) on the bit
// rows of the kept instances: in the given order (descending confidence, decided by the caller like
// the reference's np.argsort) an instance is skipped when more than skip_iou of its points are
// already taken (intersect / (npoint + 1e-5) in double, like numpy), otherwise its free points get
// the next panoptic id and the instance's class.  The order makes it a sequential scan over the
// instances: ONE workgroup walks them, every thread owning a fixed set of 32-point words of the
// `taken` row (no hazards between threads; two block reductions per instance).  The reference
// decodes every RLE string to a dense N-vector on the host (and round 3 re-parsed the RLE text):
// 21 ms per LiDAR sweep with ~1000 instances, profiles/r04_kitti_host_profile.txt.
constexpr int kFuseThreads = 1024;
constexpr int kFuseWpt = 8;          // 32-point words a thread owns in the register path (N <= 262 144 points)
__global__ void __launch_bounds__(kFuseThreads) panoptic_fusion_kernel(
    const uint32_t *__restrict__ bits, int words, int n_inst, const int32_t *__restrict__ order,
    const int32_t *__restrict__ label_id, const int64_t *__restrict__ semantic_preds, int n_points,
    int cls_offset, double skip_iou, int semantic_classes, int thing_class_min, uint32_t *__restrict__ taken,
    uint32_t *__restrict__ ids, int32_t *__restrict__ label_of_id, uint32_t *__restrict__ out) {
  __shared__ int red[2][2][kFuseThreads / 64];          // [parity][inter | count][wave]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < n_points; i += kFuseThreads) ids[i] = 0u;
  int next_id = 1;
  if (words <= kFuseWpt * kFuseThreads) {
    // ---- register path: thread t owns words t, t + 1024, ... of the `taken` row for the whole walk;
    //      the next instance's row is on its way while this one is decided (the order is known up
    //      front), so an instance costs one block reduction, not two memory round trips
    uint32_t tk[kFuseWpt], rw[kFuseWpt], nx[kFuseWpt];
    auto fetch = [&](int r, uint32_t (&dst)[kFuseWpt]) {
      const uint32_t *row = bits + static_cast<int64_t>(order[r]) * words;
#pragma unroll
      for (int j = 0; j < kFuseWpt; ++j) {
        const int w = threadIdx.x + j * kFuseThreads;
        dst[j] = w < words ? row[w] : 0u;
      }
    };
#pragma unroll
    for (int j = 0; j < kFuseWpt; ++j) tk[j] = nx[j] = 0u;
    if (n_inst > 0) fetch(0, nx);
    for (int r = 0; r < n_inst; ++r) {
#pragma unroll
      for (int j = 0; j < kFuseWpt; ++j) rw[j] = nx[j];
      if (r + 1 < n_inst) fetch(r + 1, nx);
      int inter = 0, cnt = 0;
#pragma unroll
      for (int j = 0; j < kFuseWpt; ++j) {
        cnt += __popc(rw[j]);
        inter += __popc(rw[j] & tk[j]);
      }
      inter = wave_sum(inter);
      cnt = wave_sum(cnt);
      const int par = r & 1;                           // double-buffered: one barrier per instance
      if (lane == 0) {
        red[par][0][wave] = inter;
        red[par][1][wave] = cnt;
      }
      __syncthreads();
      long long ti = 0, tc = 0;
#pragma unroll
      for (int v = 0; v < kFuseThreads / 64; ++v) {
        ti += red[par][0][v];
        tc += red[par][1][v];
      }
      const bool paste_it = !(static_cast<double>(ti) / (static_cast<double>(tc) + 1e-5) > skip_iou);   // every thread, same numbers
      if (paste_it) {
        if (threadIdx.x == 0) label_of_id[next_id] = label_id[order[r]] + cls_offset;
#pragma unroll
        for (int j = 0; j < kFuseWpt; ++j) {
          uint32_t paste = rw[j] & ~tk[j];
          tk[j] |= paste;
          const int w = threadIdx.x + j * kFuseThreads;
          while (paste) {
            const int bb = __ffs(static_cast<int>(paste)) - 1;
            paste &= paste - 1;
            ids[w * 32 + bb] = static_cast<uint32_t>(next_id);
          }
        }
        ++next_id;
      }
    }
  } else {
    for (int w = threadIdx.x; w < words; w += kFuseThreads) taken[w] = 0u;
    for (int r = 0; r < n_inst; ++r) {
      const int k = order[r];
      const uint32_t *row = bits + static_cast<int64_t>(k) * words;
      int inter = 0, cnt = 0;
      for (int w = threadIdx.x; w < words; w += kFuseThreads) {
        const uint32_t bw = row[w];
        cnt += __popc(bw);
        inter += __popc(bw & taken[w]);
      }
      inter = wave_sum(inter);
      cnt = wave_sum(cnt);
      const int par = r & 1;
      if (lane == 0) {
        red[par][0][wave] = inter;
        red[par][1][wave] = cnt;
      }
      __syncthreads();
      long long ti = 0, tc = 0;
      for (int v = 0; v < kFuseThreads / 64; ++v) {
        ti += red[par][0][v];
        tc += red[par][1][v];
      }
      const bool paste_it = !(static_cast<double>(ti) / (static_cast<double>(tc) + 1e-5) > skip_iou);
      if (paste_it) {
        if (threadIdx.x == 0) label_of_id[next_id] = label_id[k] + cls_offset;
        for (int w = threadIdx.x; w < words; w += kFuseThreads) {
          uint32_t paste = row[w] & ~taken[w];
          if (paste == 0u) continue;
          taken[w] |= paste;
          while (paste) {
            const int bb = __ffs(static_cast<int>(paste)) - 1;
            paste &= paste - 1;
            ids[w * 32 + bb] = static_cast<uint32_t>(next_id);
          }
        }
        ++next_id;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  // encode: class | id << 16; thing classes that nobody claimed -> ignore (semantic_classes)
  for (int i = threadIdx.x; i < n_points; i += kFuseThreads) {
    const uint32_t id = ids[i];
    const uint32_t cls = id ? static_cast<uint32_t>(label_of_id[id]) : static_cast<uint32_t>(semantic_preds[i]);
    uint32_t v = (cls & 0xFFFFu) | (id << 16);
    if (cls >= static_cast<uint32_t>(thing_class_min) && id == 0u) v = static_cast<uint32_t>(semantic_classes);
    out[i] = v;
  }
}

}  // namespace sg

using namespace sg;

extern "C" {

int sg_instance_npoint(const int32_t *proposals_idx, const float *mask_scores, int64_t S, int stride,
                       int n_classes, float mask_thr, int n_prop, int32_t *npoint,
                       sg_stream_t stream_) {
  SG_REQUIRE(S >= 0 && stride >= n_classes && n_classes >= 1 && n_prop >= 0,
             "sg_instance_npoint: bad arguments");
  hipStream_t stream = as_stream(stream_);
  hipMemsetAsync(npoint, 0, static_cast<size_t>(n_prop) * n_classes * 4, stream);
  if (S == 0 || n_prop == 0) return check_launch("sg_instance_npoint");
  const int grid = grid_for((S + 1023) / 1024, 4, 1024);      // >= 1024 pairs per wave
  instance_npoint_kernel<<<grid, 256, 0, stream>>>(proposals_idx, mask_scores, S, stride, mask_thr,
                                                  n_classes, npoint);
  return check_launch("sg_instance_npoint");
}

static int64_t instance_words(int n_points) { return (static_cast<int64_t>(n_points) + 31) / 32; }

size_t sg_instance_runs_workspace_bytes(int n_kept, int n_points) {
  const int64_t tw = static_cast<int64_t>(n_kept > 0 ? n_kept : 1) * instance_words(n_points);
  return align_up(tw * 4) * 2 + align_up(scan_workspace_bytes(tw)) + 512;
}

int sg_instance_runs(const int32_t *proposals_idx, const float *mask_scores, int64_t S, int stride,
                     int n_classes, float mask_thr, const int32_t *inst_of, int n_prop, int n_kept,
                     int n_points, int32_t *starts, int32_t *ends, int64_t *bounds,
                     int64_t runs_capacity, void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(S >= 0 && stride >= n_classes && n_classes >= 1 && n_prop >= 0 && n_kept >= 0 &&
                 n_points >= 0 && runs_capacity >= 0,
             "sg_instance_runs: bad arguments");
  hipStream_t stream = as_stream(stream_);
  if (n_kept == 0) return SG_OK;
  const int words = static_cast<int>(instance_words(n_points));
  const int64_t tw = static_cast<int64_t>(n_kept) * words;
  SG_REQUIRE(tw < (1LL << 31), "sg_instance_runs: %d instances x %d points exceed the bitmap limit",
             n_kept, n_points);
  SG_REQUIRE(ws != nullptr && ws_bytes >= sg_instance_runs_workspace_bytes(n_kept, n_points),
             "sg_instance_runs: workspace too small");
  Workspace a(ws, ws_bytes);
  uint32_t *bits = a.take<uint32_t>(tw);
  int32_t *base = a.take<int32_t>(tw);
  const size_t sbytes = scan_workspace_bytes(tw);
  void *sws = a.take<char>(sbytes);
  int32_t *total = a.take<int32_t>(64);
  hipMemsetAsync(bits, 0, static_cast<size_t>(tw) * 4, stream);
  if (S > 0 && n_prop > 0) {
    const int grid = grid_for(S, 256, 4096);
    instance_bitmap_kernel<<<grid, 256, 0, stream>>>(proposals_idx, mask_scores, S, stride, mask_thr,
                                                    n_classes, inst_of, n_prop, words, bits);
  }
  const uint32_t *cb = bits;
  const int w = words;
  int rc = exclusive_scan(
      [cb, w] __device__(int64_t t) { return __popc(run_starts(cb, t, static_cast<int>(t % w))); },
      [base] __device__(int64_t t, int v) { base[t] = v; }, tw, total, sws, sbytes, stream);
  if (rc != SG_OK) return rc;
  instance_runs_emit_kernel<<<grid_for(tw, 256, 4096), 256, 0, stream>>>(bits, tw, words, base, starts,
                                                                        ends, bounds, runs_capacity);
  instance_runs_total_kernel<<<1, 1, 0, stream>>>(total, n_kept, bounds);
  return check_launch("sg_instance_runs");
}


size_t sg_rle_format_device_workspace_bytes(int64_t run_capacity) {
  const int64_t n = run_capacity > 0 ? run_capacity : 1;
  return align_up(static_cast<size_t>(n) * 4) + align_up(scan_workspace_bytes(n)) + 512;
}

int64_t sg_rle_format_device_text_bytes(int64_t run_capacity, int64_t length) {
  int d = 1;
  for (int64_t v = length + 1; v >= 10; v /= 10) ++d;
  return (run_capacity > 0 ? run_capacity : 1) * (2LL * d + 2);
}

// the run count is read on the device (bounds[n_inst]): nothing has to come back to the host
// between sg_instance_runs and this call
int sg_rle_format_device(const int32_t *starts, const int32_t *ends, const int64_t *bounds, int n_inst,
                         int64_t run_capacity, int64_t length, uint8_t *text, int64_t text_capacity,
                         int64_t *text_off, void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(n_inst >= 0 && run_capacity >= 0 && bounds && text_off && length >= 0 && length < (1LL << 31),
             "sg_rle_format_device: bad arguments");
  const int64_t need = sg_rle_format_device_text_bytes(run_capacity, length);
  SG_REQUIRE(need < (1LL << 31), "sg_rle_format_device: %lld runs exceed the 2 GiB text limit",
             static_cast<long long>(run_capacity));
  SG_REQUIRE(text != nullptr && text_capacity >= need,
             "sg_rle_format_device: text buffer %lld < %lld bytes (sg_rle_format_device_text_bytes)",
             static_cast<long long>(text_capacity), static_cast<long long>(need));
  SG_REQUIRE(ws != nullptr && ws_bytes >= sg_rle_format_device_workspace_bytes(run_capacity),
             "sg_rle_format_device: workspace too small");
  hipStream_t stream = as_stream(stream_);
  const int64_t n = run_capacity > 0 ? run_capacity : 1;
  Workspace a(ws, ws_bytes);
  int32_t *off = a.take<int32_t>(n);
  const size_t sbytes = scan_workspace_bytes(n);
  void *sws = a.take<char>(sbytes);
  int32_t *total = a.take<int32_t>(64);
  const int64_t *n_runs_p = bounds + n_inst;
  int rc = exclusive_scan(
      [starts, ends, n_runs_p] __device__(int64_t r) { return r < *n_runs_p ? run_text_len(starts, ends, r) : 0; },
      [off] __device__(int64_t r, int v) { off[r] = v; }, n, total, sws, sbytes, stream);
  if (rc != SG_OK) return rc;
  rle_text_kernel<<<grid_for(n, 256, 4096), 256, 0, stream>>>(starts, ends, off, n_runs_p, n, text);
  rle_offsets_kernel<<<(n_inst + 256) / 256, 256, 0, stream>>>(bounds, n_inst, off, total, text_off);
  return check_launch("sg_rle_format_device");
}

size_t sg_panoptic_fusion_workspace_bytes(int n_inst, int n_points) {
  return align_up(static_cast<size_t>(instance_words(n_points)) * 4) + align_up(static_cast<size_t>(n_points > 0 ? n_points : 1) * 4) +
         align_up((static_cast<size_t>(n_inst) + 2) * 4) + 256;
}

int sg_panoptic_fusion(const uint32_t *bits, int n_inst, int n_points, const int32_t *order,
                       const int32_t *label_id, const int64_t *semantic_preds, int cls_offset,
                       double skip_iou, int semantic_classes, int thing_class_min, uint32_t *out, void *ws,
                       size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(n_inst >= 0 && n_points >= 0 && out != nullptr && semantic_preds != nullptr,
             "sg_panoptic_fusion: bad arguments");
  SG_REQUIRE(n_inst < 65536, "sg_panoptic_fusion: %d instances exceed the 16-bit panoptic id", n_inst);
  SG_REQUIRE(ws != nullptr && ws_bytes >= sg_panoptic_fusion_workspace_bytes(n_inst, n_points),
             "sg_panoptic_fusion: workspace too small");
  if (n_points == 0) return SG_OK;
  Workspace a(ws, ws_bytes);
  const int words = static_cast<int>(instance_words(n_points));
  uint32_t *taken = a.take<uint32_t>(words);
  uint32_t *ids = a.take<uint32_t>(n_points);
  int32_t *label_of_id = a.take<int32_t>(static_cast<size_t>(n_inst) + 2);
  panoptic_fusion_kernel<<<1, kFuseThreads, 0, as_stream(stream_)>>>(
      bits, words, n_inst, order, label_id, semantic_preds, n_points, cls_offset, skip_iou, semantic_classes,
      thing_class_min, taken, ids, label_of_id, out);
  return check_launch("sg_panoptic_fusion");
}

}  // extern "C"
