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
#include <stdlib.h>

#include <mutex>

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
      // (eight class scores of the lane's row in flight, then their ballots: one score at a time was a chain of
      //  nc dependent loads per 64 pairs -- 74 us for the bench scene's 24 k pairs on its 6 workgroups)
      for (int i0 = 0; i0 < nc; i0 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row[min(i0 + j, nc - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cnt = __popcll(__ballot(valid && i0 + j < nc && v[j] > thr));
          if (lane == i0 + j) acc += cnt;
        }
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
    for (int i0 = 0; i0 < nc; i0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = row[min(i0 + j, nc - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint64_t ons = __ballot(valid && i0 + j < nc && v[j] > thr);
        if (head && i0 + j < nc) {
          const int cnt = __popcll(ons & span);
          if (cnt) atomicAdd(&npoint[static_cast<int64_t>(p) * nc + i0 + j], cnt);
        }
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
    for (int i0 = 0; i0 < nc; i0 += 8) {                // (eight scores in flight, then the instances of the classes above
      float v[8];                                       //  the threshold in flight: two round trips per 8 classes, not 16)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = row[min(i0 + j, nc - 1)];
      int k[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool on = i0 + j < nc && v[j] > thr;
        k[j] = inst_of[static_cast<int64_t>(on ? i0 + j : 0) * n_prop + pq.x];
        k[j] = on ? k[j] : -1;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k[j] >= 0) atomicOr(&bits[static_cast<int64_t>(k[j]) * words + (pq.y >> 5)], 1u << (pq.y & 31));
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
// the next panoptic id and the instance's class.  The order makes it a sequential walk over the
// instances.  The reference decodes every RLE string to a dense N-vector on the host (21 ms per
// LiDAR sweep with ~1000 instances); round 4 walked the dense bit rows with one workgroup and a
// barrier per instance (2.2 ms: 1088 instances x 15 KB of mostly zero words).  A mask is SPARSE --
// ~100 points of 120 000 -- so:
//   1. panoptic_summary_kernel (whole chip): per instance one bit per 32-point word, "word != 0";
//   2. panoptic_walk_kernel: ONE WAVE walks the instances; the instance's non-zero words are dealt
//      to the lanes from a list in LDS, the `taken` row lives in LDS, the two counts meet by DPP
//      reductions -- no barrier; a SECOND wave runs a few instances ahead and touches the lines the
//      walker is about to read, so the walker waits for cache hits, not for the fabric;
//   3. panoptic_assign_kernel (whole chip): every point takes the id of the first pasted instance that
//      holds it (atomicMin over the visiting rank) -- the walker itself issues NO global store: stores
//      share the loads' in-order counter, the next instance's first load would wait for them;
//   4. panoptic_encode_kernel (whole chip): class | id << 16.
constexpr int kFuseSumPerLane = 16;                       // summary words a lane owns: rows of <= 32 768 words
constexpr int kFuseTakenWords = kFuseSumPerLane * 64 * 32;      // = 32 768 words of `taken` in LDS (128 KB, 1 M points)
constexpr int kFuseListCap = 8192;                           // non-zero words of one instance kept as a list (16 KB)

__global__ void __launch_bounds__(256) panoptic_summary_kernel(const uint32_t *__restrict__ bits, int words,
                                                              int sum_words, int n_inst,
                                                              uint32_t *__restrict__ summary) {
  // one thread per (instance, word): the wave's 64 "non-zero" flags are two summary words
  const int64_t padded = static_cast<int64_t>(sum_words) * 32;
  const int64_t total = static_cast<int64_t>(n_inst) * padded;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int64_t k = t / padded;
    const int w = static_cast<int>(t - k * padded);
    const uint32_t v = w < words ? bits[k * words + w] : 0u;
    const uint64_t bal = __ballot(v != 0u);
    const int lane = threadIdx.x & 63;
    if ((lane & 31) == 0) summary[k * sum_words + (w >> 5)] = static_cast<uint32_t>(bal >> (lane & 32));
  }
}

template <int SPL>      // summary words per lane (1, 2, 4, 8, 16)
__global__ void __launch_bounds__(128) panoptic_walk_kernel(const uint32_t *__restrict__ bits, int words, int sum_words,
                                                           const uint32_t *__restrict__ summary, int n_inst,
                                                           const int32_t *__restrict__ order,
                                                           const int32_t *__restrict__ label_id, int cls_offset,
                                                           double skip_iou, uint32_t *__restrict__ id_of_rank,
                                                           int32_t *__restrict__ label_of_id) {
  extern __shared__ __attribute__((aligned(16))) uint32_t taken[];      // [words rounded up]
  __shared__ unsigned short list[kFuseListCap];                         // non-zero words of the instance at hand
  __shared__ uint32_t pasted[2048];                                     // one bit per visited instance (n_inst < 65 536)
  __shared__ int progress;                                              // instance the walker is at
  const int lane = threadIdx.x & 63;
  const int padded = sum_words * 32;
  for (int w = threadIdx.x; w < padded; w += 128) taken[w] = 0u;
  for (int w = threadIdx.x; w < 2048; w += 128) pasted[w] = 0u;
  if (threadIdx.x == 0) progress = 0;
  __syncthreads();
  auto fetch_summary = [&](int r, uint32_t (&dst)[SPL]) {
    const uint32_t *row = summary + static_cast<int64_t>(order[r]) * sum_words;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      const int c = lane + 64 * j;
      dst[j] = c < sum_words ? row[c] : 0u;
    }
  };
  if (threadIdx.x >= 64) {
    // ---- wave 1, the PREFETCHER: it touches what the walker will read -- summary words and the
    //      128-byte line of every non-zero 32-word group of the next instances -- a bounded distance
    //      ahead.  Loads of one wave return in order, so a wave that waits for an L1 hit cannot have a
    //      fabric round trip outstanding without waiting for that too: the slow loads need a wave of
    //      their own.  Nothing is handed over but cache lines (the CU's L1, the XCD's L2).
    constexpr int kAhead = 12;
    uint32_t sink = 0;
    for (int r = 0; r < n_inst; ++r) {
      while (r > __hip_atomic_load(&progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + kAhead)
        __builtin_amdgcn_s_sleep(8);
      uint32_t sm[SPL];
      fetch_summary(r, sm);
      const uint32_t *row = bits + static_cast<int64_t>(order[r]) * words;
#pragma unroll
      for (int j = 0; j < SPL; ++j)
        if (sm[j]) {
          const int w0 = (lane + 64 * j) * 32;
          sink ^= row[w0 + __ffs(static_cast<int>(sm[j])) - 1];
          sink ^= row[min(w0 + 31 - __clz(static_cast<int>(sm[j])) + 0, words - 1)];      // (a group may straddle two lines)
        }
    }
    asm volatile("" ::"v"(sink));
  } else {
  // (`order` 64 entries at a time in a register per lane, read back by v_readlane: a load of order[r]
  // in every iteration would be one more dependent round trip per instance)
  // ---- wave 0, the WALKER: loads only -- a store in flight (the pasted points' ids, as the first
  //      version wrote them here) would be waited for by the next instance's first load, a fabric
  //      round trip per instance; the ids are assigned by panoptic_assign_kernel from the decisions
  uint32_t s_cur[SPL], s_nxt[SPL];
#pragma unroll
  for (int j = 0; j < SPL; ++j) s_cur[j] = s_nxt[j] = 0u;
  int ord_cur = lane < n_inst ? order[lane] : 0;            // order[64 b + lane] of the block at hand
  int ord_nxt = 64 + lane < n_inst ? order[64 + lane] : 0;  // ... and of the next block
  auto summary_of = [&](int k, uint32_t (&dst)[SPL]) {
    const uint32_t *row = summary + static_cast<int64_t>(k) * sum_words;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
      const int c = lane + 64 * j;
      dst[j] = c < sum_words ? row[c] : 0u;
    }
  };
  if (n_inst > 0) summary_of(__builtin_amdgcn_readlane(ord_cur, 0), s_cur);
  // (Tried: requesting an instance's words one instance ahead -- a software pipeline over the walk, 1 or
  // 4 words per lane in registers: 959 / 1103 us against 827 us for this loop on the KITTI-like sweep.
  // What an instance costs is its instruction stream -- list build, 2 DPP reductions, LDS traffic --
  // not its one round trip to an L1 / L2 that the prefetcher wave keeps warm.)
  for (int r = 0; r < n_inst; ++r) {
    if (lane == 0) __hip_atomic_store(&progress, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int k = __builtin_amdgcn_readlane(ord_cur, r & 63);
    if ((r & 63) == 63) {         // uniform: next block of the order
      ord_cur = ord_nxt;
      ord_nxt = r + 65 + lane < n_inst ? order[r + 65 + lane] : 0;
    }
    if (r + 1 < n_inst) summary_of(__builtin_amdgcn_readlane(ord_cur, (r + 1) & 63), s_nxt);
    const uint32_t *row = bits + static_cast<int64_t>(k) * words;
    // ---- the instance's non-zero words as a compact list in LDS (lane l's words behind those of the
    //      lanes before it), then handed out round-robin: a mask that is one contiguous run of the
    //      cloud puts all its words into ONE lane's summary word -- walked lane by lane that is up to
    //      32 dependent loads; from the list every lane gets T / 64 of them
    int mine = 0;
#pragma unroll
    for (int j = 0; j < SPL; ++j) mine += __popc(s_cur[j]);
    const int before = wave_incl_scan(mine) - mine;
    const int T = __builtin_amdgcn_readlane(before + mine, 63);
    int inter = 0, cnt = 0, w_first = -1;
    uint32_t bw_first = 0u;
    const bool listed = T <= kFuseListCap;      // uniform
    if (listed) {
      int at = before;
#pragma unroll
      for (int j = 0; j < SPL; ++j) {
        const int w0 = (lane + 64 * j) * 32;
        for (uint32_t m = s_cur[j]; m; m &= m - 1) list[at++] = static_cast<unsigned short>(w0 + __ffs(static_cast<int>(m)) - 1);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): one wave, LDS in program order -- no barrier
      for (int i = lane; i < T; i += 64) {
        const int w = list[i];
        const uint32_t bw = row[w];
        cnt += __popc(bw);
        inter += __popc(bw & taken[w]);
        if (i == lane) { w_first = w; bw_first = bw; }      // (kept for the paste: most masks are <= 64 words)
      }
    } else {
#pragma unroll
      for (int j = 0; j < SPL; ++j) {
        const int w0 = (lane + 64 * j) * 32;
        for (uint32_t m = s_cur[j]; m; m &= m - 1) {
          const int w = w0 + __ffs(static_cast<int>(m)) - 1;
          const uint32_t bw = row[w];
          cnt += __popc(bw);
          inter += __popc(bw & taken[w]);
        }
      }
    }
    const long long ti = wave_sum(inter), tc = wave_sum(cnt);
    // intersect / (npoint + 1e-5) > skip_iou as numpy evaluates it (double).  The quotient can only
    // round across the threshold when ti - skip_iou * (tc + 1e-5) is within a few ulps of zero: away
    // from that (always, in practice -- the 1e-5 makes exact ties impossible) the sign of the
    // difference decides and the ~40-instruction double division stays off the chain.
    const double den = static_cast<double>(tc) + 1e-5;
    const double diff = static_cast<double>(ti) - skip_iou * den;
    const bool paste_it = fabs(diff) > 1e-9 * den ? !(diff > 0.0)
                                                  : !(static_cast<double>(ti) / den > skip_iou);   // every lane, same numbers
    if (paste_it) {
      if (lane == 0) pasted[r >> 5] |= 1u << (r & 31);
      auto paste_word = [&](int w) { taken[w] |= row[w]; };      // (an L1 hit: read a moment ago)
      if (listed) {
        if (w_first >= 0) taken[w_first] |= bw_first;
        for (int i = lane + 64; i < T; i += 64) paste_word(list[i]);
      } else {
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
          const int w0 = (lane + 64 * j) * 32;
          for (uint32_t m = s_cur[j]; m; m &= m - 1) paste_word(w0 + __ffs(static_cast<int>(m)) - 1);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < SPL; ++j) s_cur[j] = s_nxt[j];
  }
  if (lane == 0) __hip_atomic_store(&progress, n_inst + (1 << 20), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  // ---- decisions -> panoptic ids: the r-th visited instance, if pasted, gets 1 + (pasted before it)
  if (threadIdx.x < 64) {
    int base = 0;
    for (int c0 = 0; c0 * 32 < n_inst; c0 += 64) {
      const int c = c0 + lane;
      const uint32_t bitsw = c < 2048 ? pasted[c] : 0u;
      const int pc = __popc(bitsw);
      const int incl = wave_incl_scan(pc);
      int id = base + incl - pc;
      for (int b = 0; b < 32; ++b) {
        const int r = c * 32 + b;
        if (r >= n_inst) break;
        const bool on = (bitsw >> b) & 1u;
        if (on) {
          ++id;
          label_of_id[id] = label_id[order[r]] + cls_offset;
        }
        id_of_rank[r] = on ? static_cast<uint32_t>(id) : 0u;
      }
      base += __builtin_amdgcn_readlane(incl, 63);
    }
  }
}

// the pasted instances' points: a point takes the id of the FIRST pasted instance (in visiting order)
// whose mask holds it -- what the sequential paste of the reference leaves behind
__global__ void __launch_bounds__(256) panoptic_assign_kernel(const uint32_t *__restrict__ bits, int words, int n_inst,
                                                             const int32_t *__restrict__ order,
                                                             const uint32_t *__restrict__ id_of_rank,
                                                             uint32_t *__restrict__ first) {
  // one thread per (visited instance, word of its row): coalesced over the row, the few non-zero words
  // of a pasted instance spread over as many threads (per summary word, the first version, a thread
  // walked up to 32 words x 32 bits of atomics alone: 130 us)
  const int64_t total = static_cast<int64_t>(n_inst) * words;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int r = static_cast<int>(t / words), w = static_cast<int>(t - static_cast<int64_t>(r) * words);
    if (id_of_rank[r] == 0u) continue;
    for (uint32_t bw = bits[static_cast<int64_t>(order[r]) * words + w]; bw; bw &= bw - 1)
      atomicMin(&first[w * 32 + __ffs(static_cast<int>(bw)) - 1], static_cast<uint32_t>(r));
  }
}

// rows too long for the LDS `taken` row (> 1 M points): the dense walk of round 4, one workgroup
constexpr int kFuseThreads = 1024;
__global__ void __launch_bounds__(kFuseThreads) panoptic_walk_dense_kernel(
    const uint32_t *__restrict__ bits, int words, int n_inst, const int32_t *__restrict__ order,
    const int32_t *__restrict__ label_id, int cls_offset, double skip_iou, uint32_t *__restrict__ taken,
    uint32_t *__restrict__ ids, int32_t *__restrict__ label_of_id) {
  __shared__ int red[2][2][kFuseThreads / 64];          // [parity][inter | count][wave]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int next_id = 1;
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
      ++next_id;      // (uniform: every thread takes the same decision from the same sums)
    }
  }
}

// encode: class | id << 16; thing classes that nobody claimed -> ignore (semantic_classes)
// (`id_of_rank` != null: ids[] holds the rank of the first pasted instance of the point, 0xffffffff = none)
__global__ void __launch_bounds__(256) panoptic_encode_kernel(const uint32_t *__restrict__ ids,
                                                             const uint32_t *__restrict__ id_of_rank,
                                                             const int32_t *__restrict__ label_of_id,
                                                             const int64_t *__restrict__ semantic_preds, int n_points,
                                                             int semantic_classes, int thing_class_min,
                                                             uint32_t *__restrict__ out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_points; i += gridDim.x * 256) {
    uint32_t id = ids[i];
    if (id_of_rank) id = id == 0xffffffffu ? 0u : id_of_rank[id];
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
  // up to 4096 waves of >= 64 pairs each (>= 1024 pairs per wave -- few atomics on a giant proposal's words -- once
  // there are that many pairs; with fewer, more waves: the walk is a chain of memory round trips per 64 pairs)
  const int grid = grid_for((S + 63) / 64, 4, 1024);
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

static int fuse_summary_words(int n_points) { return static_cast<int>((instance_words(n_points) + 31) / 32); }

size_t sg_panoptic_fusion_workspace_bytes(int n_inst, int n_points) {
  return align_up(static_cast<size_t>(instance_words(n_points)) * 4) + align_up(static_cast<size_t>(n_points > 0 ? n_points : 1) * 4) +
         align_up((static_cast<size_t>(n_inst) + 2) * 4) +
         align_up(static_cast<size_t>(n_inst > 0 ? n_inst : 1) * fuse_summary_words(n_points) * 4) +
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
  hipStream_t stream = as_stream(stream_);
  Workspace a(ws, ws_bytes);
  const int words = static_cast<int>(instance_words(n_points));
  const int sum_words = fuse_summary_words(n_points);
  uint32_t *taken = a.take<uint32_t>(words);
  uint32_t *ids = a.take<uint32_t>(n_points);
  int32_t *label_of_id = a.take<int32_t>(static_cast<size_t>(n_inst) + 2);
  uint32_t *summary = a.take<uint32_t>(static_cast<size_t>(n_inst > 0 ? n_inst : 1) * sum_words);
  uint32_t *id_of_rank = a.take<uint32_t>(static_cast<size_t>(n_inst) + 2);
  // developer A/B knob, read per call so that a test can run both walks in one process: round 4's dense walk
  const char *dense_s = getenv("SG_PANOPTIC_DENSE");
  const bool dense_env = dense_s != nullptr && dense_s[0] != '\0' && dense_s[0] != '0';
  const bool sparse = n_inst > 0 && words <= kFuseTakenWords && !dense_env;
  hipMemsetAsync(ids, sparse ? 0xff : 0, static_cast<size_t>(n_points) * 4, stream);
  if (sparse) {
    panoptic_summary_kernel<<<grid_for(static_cast<int64_t>(n_inst) * sum_words * 32, 256, 4096), 256, 0, stream>>>(
        bits, words, sum_words, n_inst, summary);
    const size_t lds = static_cast<size_t>(sum_words) * 32 * 4;
    const int spl = (sum_words + 63) / 64;
#define SG_WALK(SPL)                                                                                                  \
  do {                                                                                                                \
    static std::once_flag once;                                                                                       \
    std::call_once(once, [] {                                                                                         \
      hipFuncSetAttribute(reinterpret_cast<const void *>(panoptic_walk_kernel<SPL>),                                  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (kFuseTakenWords + 64) * 4);                    \
    });                                                                                                               \
    panoptic_walk_kernel<SPL><<<1, 128, lds, stream>>>(bits, words, sum_words, summary, n_inst, order, label_id,      \
                                                      cls_offset, skip_iou, id_of_rank, label_of_id);                \
  } while (0)
    if (spl <= 1) SG_WALK(1);
    else if (spl <= 2) SG_WALK(2);
    else if (spl <= 4) SG_WALK(4);
    else if (spl <= 8) SG_WALK(8);
    else SG_WALK(16);
#undef SG_WALK
    panoptic_assign_kernel<<<grid_for(static_cast<int64_t>(n_inst) * words, 256, 8192), 256, 0, stream>>>(
        bits, words, n_inst, order, id_of_rank, ids);
  } else if (n_inst > 0) {
    panoptic_walk_dense_kernel<<<1, kFuseThreads, 0, stream>>>(bits, words, n_inst, order, label_id, cls_offset,
                                                              skip_iou, taken, ids, label_of_id);
  }
  panoptic_encode_kernel<<<grid_for(n_points, 256, 2048), 256, 0, stream>>>(ids, sparse ? id_of_rank : nullptr, label_of_id, semantic_preds, n_points,
                                                                          semantic_classes, thing_class_min, out);
  return check_launch("sg_panoptic_fusion");
}

}  // extern "C"
