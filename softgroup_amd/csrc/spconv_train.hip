// spconv_train.hip -- the training side of the sparse convolution (BASELINE config 3: bf16
// autocast, tools/train.py:47 + spconv's autograd): a bf16 forward / input-gradient kernel on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation, and a deterministic weight gradient.
//
//   forward / dgrad   out[j,:] = bf16( sum_k in[nbr[j,k],:] . W[k] )          in, W, out: bf16
//   wgrad             dW[k][ci][co] = sum_j in[nbr[j,k]][ci] * g[j][co]       in, g: fp32 | bf16
//
// Where the fp32 inference kernel (spconv_conv.hip) is bound by the matrix pipe, the bf16 MFMA
// retires a 16-channel slice of a 32x32 tile in 32 cycles -- 16x faster per byte of operand -- so
// this kernel is bound by operand delivery (gather + weights through the CU's vector memory
// path), not by MFMA.  The design follows from that:
//   * one wave = one (32-row tile, up to 4 x 32 output columns, offset range) unit: the gathered
//     A fragment (one 16-B load per lane and slice) is reused for all of the unit's column blocks;
//   * the tile's gather block sits in LDS (wave-private), absent neighbours are buffer loads past
//     the end (return 0), a 3-deep operand ring keeps two slices of loads in flight;
//   * weights are packed [K][ceil16(Cin)/8][Cout][8] bf16 (zero padded): a lane's 8 reduction steps
//     for its column are one 16-B read, a wave reads 1 KB fully coalesced per (slice, column block);
//   * deep U-Net levels (few tiles) split the kernel offsets over several units; fp32 partial sums
//     are reduced in a fixed order.
// Tried in round 3 and not kept: a workgroup-shared variant (four consecutive tiles per workgroup
// walking the union of their offsets in lockstep, the step's weight block fetched once per
// workgroup and staged in LDS, double buffered, one barrier per step; gathers line-wise through a
// swizzled LDS transpose).  Correct (tests/test_train_gpu.py green) but slower where it applies:
// 32->32 x 124 k rows 30.6 us against 20.2 us, 64->64 x 77 k rows 48.8 against 42.5, 96->96 41.4
// against 42.6 -- a step is only 2-6 MFMAs of 32 cycles, the barrier per step costs more than the
// weight loads it saves (profiles/r03_train_conv_wg.txt).
// Weight gradient: grid = (row chunks, K); a workgroup walks its chunk two rows per fp32 MFMA
// (v_mfma_f32_32x32x2_f32: the products are exact, accumulation fp32 -- bf16 operands are widened
// on load); every lane loads VI consecutive input channels and VO consecutive gradient channels
// with one vector load each and feeds VI x VO MFMAs from them.  Chunk partials go to a workspace
// and are added in chunk order by a second kernel: bit-identical from run to run (no atomics).
#include <stdlib.h>

#include "common.h"

namespace sg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kTRows = 32;
constexpr int kTMaxK = 27;
constexpr unsigned kPast = 0x80000000u;   // byte offset past every (< 2 GB) buffer: loads return 0
constexpr int kRing = 3;

// round to nearest even, NaN kept quiet (the conversion torch's .to(torch.bfloat16) performs)
__device__ __forceinline__ uint16_t to_bf16(float x) {
  uint32_t u = __builtin_bit_cast(uint32_t, x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}
__device__ __forceinline__ float from_bf16(uint16_t h) {
  return __builtin_bit_cast(float, static_cast<uint32_t>(h) << 16);
}

struct Bf16Args {
  const uint16_t *in;
  const int32_t *nbr;          // [M_out][K] (used when the plan arrays are absent)
  const uint16_t *w;           // packed bf16
  const int32_t *order;        // plan (all three or none)
  const uint32_t *tile_mask;
  const int32_t *nbr_tiles;
  uint16_t *out;               // [M_out][Cout] bf16 (ksplit == 1)
  float *partial;              // [ksplit][M_out][Cout] fp32 (ksplit > 1)
  int M_out, K, Cin, Cout;
  int col_units, blocks_per_unit, ksplit, k_per_split;
  unsigned in_bytes, w_bytes;
};

template <int NBW, bool VEC>
__global__ void __launch_bounds__(256) gather_conv_bf16_kernel(Bf16Args p) {
  extern __shared__ __attribute__((aligned(16))) int32_t nbr_all[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int32_t *nbr_lds = nbr_all + wave * kTRows * kTMaxK;
  const int arow = lane & 31, ahalf = lane >> 5;
  const int c8p = (p.Cin + 15) / 16 * 2;     // 8-channel blocks of the padded reduction range
  const int n_slices = c8p / 2;

  const int num_tiles = (p.M_out + kTRows - 1) / kTRows;
  const int units_per_tile = p.col_units * p.ksplit;
  const long long unit = static_cast<long long>(blockIdx.x) * 4 + wave;
  const bool valid = unit < static_cast<long long>(num_tiles) * units_per_tile;
  const int tile = valid ? static_cast<int>(unit / units_per_tile) : 0;
  const int sub = valid ? static_cast<int>(unit % units_per_tile) : 0;
  const int cu = sub % p.col_units, ks = sub / p.col_units;
  const int nb0 = cu * p.blocks_per_unit;
  const int nbw = min(p.blocks_per_unit, (p.Cout + 31) / 32 - nb0);
  const int k_lo = ks * p.k_per_split, k_hi = min(p.K, k_lo + p.k_per_split);

  int my_row = -1;
  if (valid) {
    const int pos = tile * kTRows + arow;
    if (p.order) my_row = p.order[pos];         // plan order, padded with -1 to whole tiles
    else if (pos < p.M_out) my_row = pos;
    if (p.nbr_tiles) {
      const int32_t *blk = p.nbr_tiles + static_cast<long long>(tile) * kTRows * p.K;
      for (int e = lane; e < kTRows * p.K; e += 64) {
        const int r = e / p.K;
        nbr_lds[r * kTMaxK + (e - r * p.K)] = blk[e];
      }
    } else {
      for (int e = lane; e < kTRows * p.K; e += 64) {
        const int r = e / p.K, k = e - r * p.K;
        const int row = __shfl(my_row, r, 64);
        nbr_lds[r * kTMaxK + k] = row >= 0 ? p.nbr[static_cast<long long>(row) * p.K + k] : -1;
      }
    }
  }
  __syncthreads();          // the only barrier: gather blocks visible
  if (!valid) return;

  uint32_t mask = p.tile_mask ? p.tile_mask[tile] : 0xffffffffu;
  mask &= ((k_hi >= 32 ? 0u : (1u << k_hi)) - 1u) & ~((1u << k_lo) - 1u);
  mask = __builtin_amdgcn_readfirstlane(mask);

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.w), 0, p.w_bytes, 0x00020000);

  const int col = nb0 * 32 + arow;
  // surplus column blocks (n >= nbw) re-read the last real one: every load is unconditional
  unsigned v_w[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n) {
    const int c = min(col + min(n, nbw - 1) * 32, p.Cout - 1);
    v_w[n] = static_cast<unsigned>(ahalf * p.Cout + c) * 16u;
  }

  struct Slice { f4 a; f4 b[NBW]; };
  auto load = [&](int k, int s, Slice &x) {
    const int src = nbr_lds[arow * kTMaxK + k];
    if (VEC) {
      const unsigned v_a = src >= 0 ? static_cast<unsigned>(src * p.Cin + ahalf * 8) * 2u : kPast;
      x.a = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_a, s * 32, 0));
    } else {             // any Cin: element loads, channels past Cin are zeros
      uint32_t w4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = s * 16 + ahalf * 8 + 2 * j;
        const uint32_t lo = (src >= 0 && c < p.Cin) ? p.in[static_cast<long long>(src) * p.Cin + c] : 0u;
        const uint32_t hi = (src >= 0 && c + 1 < p.Cin) ? p.in[static_cast<long long>(src) * p.Cin + c + 1] : 0u;
        w4[j] = lo | (hi << 16);
      }
      x.a = __builtin_bit_cast(f4, w4);
    }
    const int s_w = (k * c8p + s * 2) * p.Cout * 16;
#pragma unroll
    for (int n = 0; n < NBW; ++n)
      x.b[n] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, v_w[n], s_w, 0));
  };
  auto advance = [&](int &k, int &s) {
    if (++s == n_slices) {
      s = 0;
      const uint32_t rest = mask & ~((2u << k) - 1u);
      k = rest ? __builtin_ctz(rest) : k;
    }
  };

  f32x16 acc[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  auto compute = [&](Slice &x) {
    const bf16x8 a = __builtin_bit_cast(bf16x8, x.a);
#pragma unroll
    for (int n = 0; n < NBW; ++n)
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, x.b[n]), acc[n], 0, 0, 0);
  };

  // items = (offset in mask) x slice; the ring holds items i .. i+kRing-2 while item i multiplies.
  // Past the end the last item is re-read and dropped, so loads stay unconditional.
  const int rem = __builtin_popcount(mask) * n_slices;
  if (rem > 0) {
    Slice S[kRing];
    int kp = __builtin_ctz(mask), sp = 0, issued = 1;
    load(kp, sp, S[0]);
#pragma unroll
    for (int i = 1; i < kRing - 1; ++i) {
      if (issued < rem) advance(kp, sp);
      ++issued;
      load(kp, sp, S[i]);
    }
    for (int g = rem / kRing; g > 0; --g) {
#pragma unroll
      for (int i = 0; i < kRing; ++i) {
        if (issued < rem) advance(kp, sp);
        ++issued;
        load(kp, sp, S[(i + kRing - 1) % kRing]);
        __builtin_amdgcn_sched_barrier(0);
        compute(S[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int tail = rem % kRing;
#pragma unroll
    for (int i = 0; i < kRing - 1; ++i)
      if (i < tail) compute(S[i]);
  }

  // acc[n][reg] -> tile row (reg&3) + 8*(reg>>2) + 4*half, column col + 32 n
  const bool final_out = p.ksplit == 1;
  float *part = final_out ? nullptr : p.partial + static_cast<long long>(ks) * p.M_out * p.Cout;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int r = (reg & 3) + 8 * (reg >> 2) + 4 * ahalf;
    const int row = __shfl(my_row, r, 64);
    if (row < 0) continue;
    const long long off = static_cast<long long>(row) * p.Cout + col;
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      if (n < nbw && col + n * 32 < p.Cout) {
        if (final_out) p.out[off + n * 32] = to_bf16(acc[n][reg]);
        else part[off + n * 32] = acc[n][reg];
      }
    }
  }
}

// fixed-order sum of the offset-split partials -> bf16
__global__ void __launch_bounds__(256) conv_reduce_bf16_kernel(const float *__restrict__ partial, int ksplit,
                                                              long long n, uint16_t *__restrict__ out) {
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < n; t += gridDim.x * 256LL) {
    float a = partial[t];
    for (int s = 1; s < ksplit; ++s) a += partial[s * n + t];
    out[t] = to_bf16(a);
  }
}

// src fp32 [Cout][K][Cin] (src_kio == 0) or [K][Cin][Cout] (1) -> bf16 [K][ceil16(Cin)/8][Cout][8]
__global__ void __launch_bounds__(256) pack_weight_bf16_kernel(const float *__restrict__ w, int cout, int K,
                                                              int cin, int src_kio,
                                                              uint16_t *__restrict__ out) {
  const int c8p = (cin + 15) / 16 * 2;
  const int64_t total = static_cast<int64_t>(K) * c8p * cout * 8;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int j = static_cast<int>(t & 7);
    int64_t r = t >> 3;
    const int co = static_cast<int>(r % cout);
    r /= cout;
    const int blk = static_cast<int>(r % c8p), k = static_cast<int>(r / c8p);
    const int ci = blk * 8 + j;
    float v = 0.f;
    if (ci < cin)
      v = src_kio ? w[(static_cast<int64_t>(k) * cin + ci) * cout + co]
                  : w[(static_cast<int64_t>(co) * K + k) * cin + ci];
    out[t] = to_bf16(v);
  }
}

template <int NBW>
static void launch_bf16(const Bf16Args &a, int grid, size_t lds, bool vec, hipStream_t stream) {
  if (vec) gather_conv_bf16_kernel<NBW, true><<<grid, 256, lds, stream>>>(a);
  else gather_conv_bf16_kernel<NBW, false><<<grid, 256, lds, stream>>>(a);
}

// ---------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------
// Operand loads of the weight gradient's matrix loop, in two halves: `fetch` issues the load of an address that
// is always readable (the caller clamps it) and returns the RAW words; `value` -- called one trip later, right
// before the MFMAs -- converts and masks.  Until round 6 a load sat under its validity branch and was converted
// where it was issued: the compiler then waits for ALL outstanding loads at the top of every trip, so no trip's
// loads ever overlapped its predecessor's MFMAs, whatever the depth of the ring.
template <typename T, int V>
struct RowVec;
template <>
struct RowVec<float, 1> {
  typedef float Raw;
  static __device__ __forceinline__ Raw fetch(const float *p) { return *p; }
  static __device__ __forceinline__ void value(Raw r, bool ok, float (&v)[1]) { v[0] = ok ? r : 0.f; }
};
template <>
struct RowVec<float, 2> {
  typedef float2 Raw;
  static __device__ __forceinline__ Raw fetch(const float *p) { return *reinterpret_cast<const float2 *>(p); }
  static __device__ __forceinline__ void value(Raw r, bool ok, float (&v)[2]) {
    v[0] = ok ? r.x : 0.f;
    v[1] = ok ? r.y : 0.f;
  }
};
template <>
struct RowVec<uint16_t, 1> {
  // the aligned 32-bit word that holds the element (a 16-bit load merges into its destination register and
  // serialises behind it) with the element moved to the high half; the neighbour half is inside the same row or,
  // for the very last element of an odd-sized tensor, inside the allocation's alignment padding
  typedef uint2 Raw;      // (word, shift)
  static __device__ __forceinline__ Raw fetch(const uint16_t *p) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    return make_uint2(*reinterpret_cast<const uint32_t *>(addr & ~static_cast<uintptr_t>(3)), (addr & 2) ? 0u : 16u);
  }
  static __device__ __forceinline__ void value(Raw r, bool ok, float (&v)[1]) {
    v[0] = ok ? __builtin_bit_cast(float, (r.x << r.y) & 0xffff0000u) : 0.f;
  }
};
template <>
struct RowVec<uint16_t, 2> {
  typedef uint32_t Raw;
  static __device__ __forceinline__ Raw fetch(const uint16_t *p) { return *reinterpret_cast<const uint32_t *>(p); }
  static __device__ __forceinline__ void value(Raw r, bool ok, float (&v)[2]) {
    v[0] = ok ? __builtin_bit_cast(float, r << 16) : 0.f;
    v[1] = ok ? __builtin_bit_cast(float, r & 0xffff0000u) : 0.f;
  }
};

// partial [chunks][K][Cin][Cout]; every (chunk, k) block is written (zeros when the chunk has no
// pair for offset k), so the reduction needs no flags.
//   1. the chunk's rows that HAVE a neighbour at offset k (one coalesced read of the transposed
//      gather table nbr_t[K][M]) are compacted, in ascending row order, into LDS (at 2 cm voxels only 14 % of the (row, offset) pairs of the finest level exist:
//      the matrix loop runs over existing pairs only);
//   2. the (ci, co) range is cut into supertiles of 32 VI x 32 VO channels.  With >= 3 supertiles
//      each wave owns whole supertiles (round-robin over the 4 waves x gridDim.z workgroups); with 1
//      or 2 the compacted rows are split over 4 or 2 waves whose accumulators meet in LDS in a
//      fixed order ((w0 + w1) + (w2 + w3)).
#ifndef SG_WGRAD_DEPTH
#define SG_WGRAD_DEPTH 2
#endif
constexpr int kWgDepth = SG_WGRAD_DEPTH;      // trips of 8 rows whose loads are in flight per wave
template <typename TA, typename TG, int VI, int VO>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const TA *__restrict__ in, const TG *__restrict__ g_out,
                                                        const int32_t *__restrict__ nbr_t, int M_out, int K,
                                                        int Cin, int Cout, int chunk_rows, int csplit,
                                                        float *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) int32_t wg_lds[];
  int32_t *row_c = wg_lds, *src_c = wg_lds + chunk_rows;
  float *red = reinterpret_cast<float *>(wg_lds + 2 * chunk_rows);    // [2][VI*VO*16][64]
  __shared__ int wcount[4];
#ifdef SG_WGRAD_OFFSET_MAJOR      // (A/B build, measured no faster: 72 / 135 / 166 us against 66 / 125 / 165 on levels 0-2 --
  const int k = blockIdx.x, vchunk = blockIdx.y;      //  the offsets of one row chunk as consecutive workgroups)
#else
  // the centre offset first: every row has that neighbour (itself), the other 26 offsets ~1 in 7 at 2 cm voxels, so
  // its workgroups carry 3-7 times the pairs of the others and should not be the last to start
  const int k = (static_cast<int>(blockIdx.y) + (K >> 1)) % K, vchunk = blockIdx.x;
#endif
  // ... and its chunks are cut into `csplit` pieces, one workgroup each (the other offsets: one workgroup per
  // chunk, the pieces' other workgroups leave at once): a centre workgroup with all of a chunk's rows was the
  // longest chain of the launch
  // (the other offsets' workgroups are the FIRST gridDim / csplit of their row of the grid: consecutive workgroups go
  //  round the XCDs, every csplit-th one would have put all of them on 8 / csplit of the 8)
  const bool centre = csplit > 1 && k == (K >> 1);
#ifdef SG_WGRAD_OFFSET_MAJOR
  const int n_chunks = gridDim.y / csplit;
#else
  const int n_chunks = gridDim.x / csplit;
#endif
  if (!centre && vchunk >= n_chunks) return;
  const int chunk = centre ? vchunk / csplit : vchunk, sub = centre ? vchunk % csplit : 0;
  int r0 = chunk * chunk_rows;
  int nrows = min(chunk_rows, M_out - r0);
  if (centre) {
    const int piece = (chunk_rows + csplit - 1) / csplit;
    nrows = max(0, min(piece, nrows - sub * piece));
    r0 += sub * piece;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // every thread's (<= 8) gather-table words are requested before the first one is used
  constexpr int kMaxPer = 8;           // chunk_rows <= 2048
  int sv[kMaxPer];
#pragma unroll
  for (int u = 0; u < kMaxPer; ++u) {
    const int r = u * 256 + threadIdx.x;
    sv[u] = r < nrows ? nbr_t[static_cast<long long>(k) * M_out + r0 + r] : -1;
  }
  int total = 0;
#pragma unroll
  for (int u = 0; u < kMaxPer; ++u) {
    if (u * 256 >= nrows) break;       // uniform
    const int s = sv[u];
    const uint64_t bal = __ballot(s >= 0);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = wcount[w];
      before += w < wave ? c : 0;
      all += c;
    }
    if (s >= 0) {
      const int pos = total + before + mask_prefix(bal);
      row_c[pos] = u * 256 + threadIdx.x;
      src_c[pos] = s;
    }
    total += all;
    __syncthreads();
  }
  float *dst = partial + (static_cast<long long>(vchunk) * K + k) * Cin * Cout;
  if (total == 0) {
    for (int t = threadIdx.x; t < Cin * Cout; t += 256) dst[t] = 0.f;
    return;
  }
  const int col = lane & 31, half = lane >> 5;
  const int nsi = (Cin + 32 * VI - 1) / (32 * VI), nso = (Cout + 32 * VO - 1) / (32 * VO);
  const int nst = nsi * nso;
  const int RS = nst >= 3 ? 1 : (nst == 2 ? 2 : 4);     // row split; 4 / RS waves share the supertiles
  const int ST = 4 / RS;
  const int sw = wave % ST, rs = wave / ST;
  int per = (total + RS - 1) / RS;
  per = (per + 1) & ~1;
  const int begin = min(total, rs * per), end = min(total, begin + per);
  const int st_step = RS == 1 ? 4 * gridDim.z : ST;
  for (int st = RS == 1 ? blockIdx.z * 4 + wave : sw; st < nst; st += st_step) {
    const int sib = st / nso, sob = st % nso;
    const int ci0 = sib * 32 * VI + VI * col, co0 = sob * 32 * VO + VO * col;
    const bool ci_ok = ci0 + VI <= Cin, co_ok = co0 + VO <= Cout;
    const int ci_ld = ci_ok ? ci0 : 0, co_ld = co_ok ? co0 : 0;      // (lanes beyond the channels read channel 0)
    const int last = total - 1;
    f32x16 acc[VI][VO];
#pragma unroll
    for (int c = 0; c < VI; ++c)
#pragma unroll
      for (int d = 0; d < VO; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][d][r] = 0.f;
    // 4 pairs of existing rows per trip (8 vector loads, 4 VI VO MFMAs); the loads of trip t+1 are
    // in flight while trip t multiplies
    typedef typename RowVec<TA, VI>::Raw RawA;
    typedef typename RowVec<TG, VO>::Raw RawB;
    auto ld = [&](int t, RawA (&a)[4], RawB (&b)[4], int &okm) {
      okm = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = t + 2 * u + half;
        const bool ok = it < end;
        const int itc = ok ? it : last;                  // (past the end: the chunk's last pair again, value dropped)
        const int s = src_c[itc], r = row_c[itc];
        a[u] = RowVec<TA, VI>::fetch(in + static_cast<long long>(s) * Cin + ci_ld);
        b[u] = RowVec<TG, VO>::fetch(g_out + static_cast<long long>(r0 + r) * Cout + co_ld);
        okm |= (ok ? 1 : 0) << u;
      }
    };
    // (a pair past the end, or a lane beyond the input channels, multiplies ZERO by a real gradient row; output
    //  channels beyond Cout are computed and not stored)
    auto mm = [&](const RawA (&ra)[4], const RawB (&rb)[4], int okm) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float a[VI], b[VO];
        RowVec<TA, VI>::value(ra[u], ((okm >> u) & 1) != 0 && ci_ok, a);
        RowVec<TG, VO>::value(rb[u], true, b);
#pragma unroll
        for (int c = 0; c < VI; ++c)
#pragma unroll
          for (int d = 0; d < VO; ++d)
            acc[c][d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], b[d], acc[c][d], 0, 0, 0);
      }
    };
    {
      // kWgDepth trips in the ring: the loads of kWgDepth - 1 of them are in flight while one multiplies.  With the
      // loads out of their branches (RowVec above): sum over the 7 levels 0.517 -> 0.465 ms fp32 rows, 0.559 -> 0.467
      // bf16 rows at depth 2; depth 3 the same (0.468 / 0.481), depth 4 slower (0.522: registers) -- what is left is
      // not the latency of the gathered rows
      RawA a[kWgDepth][4];
      RawB b[kWgDepth][4];
      int okm[kWgDepth];
      int t = begin;
#pragma unroll
      for (int i = 0; i + 1 < kWgDepth; ++i) ld(t + 8 * i, a[i], b[i], okm[i]);
      while (t < end) {
#pragma unroll
        for (int i = 0; i < kWgDepth; ++i) {
          constexpr int kNext = kWgDepth - 1;
          ld(t + 8 * kNext, a[(i + kNext) % kWgDepth], b[(i + kNext) % kWgDepth], okm[(i + kNext) % kWgDepth]);
          __builtin_amdgcn_sched_barrier(0);
          mm(a[i], b[i], okm[i]);
          __builtin_amdgcn_sched_barrier(0);
          t += 8;
          if (t >= end) break;
        }
      }
    }
    if (RS > 1) {            // uniform per workgroup: every wave runs exactly one supertile
      constexpr int kBuf = VI * VO * 16 * 64;
      auto put = [&](float *buf) {
#pragma unroll
        for (int c = 0; c < VI; ++c)
#pragma unroll
          for (int d = 0; d < VO; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) buf[((c * VO + d) * 16 + r) * 64 + lane] = acc[c][d][r];
      };
      auto add = [&](const float *buf) {
#pragma unroll
        for (int c = 0; c < VI; ++c)
#pragma unroll
          for (int d = 0; d < VO; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][d][r] += buf[((c * VO + d) * 16 + r) * 64 + lane];
      };
      if (RS == 2) {         // waves (rs, sw): sw = wave & 1
        if (rs == 1) put(red + sw * kBuf);
        __syncthreads();
        if (rs == 0) add(red + sw * kBuf);
      } else {               // RS == 4: (w0 + w1) + (w2 + w3)
        if (wave & 1) put(red + (wave >> 1) * kBuf);
        __syncthreads();
        if (!(wave & 1)) add(red + (wave >> 1) * kBuf);
        __syncthreads();
        if (wave == 2) put(red);
        __syncthreads();
        if (wave == 0) add(red);
      }
      if (rs != 0) continue;           // (no further barrier follows)
    }
    // acc[c][d][reg] = dW[ci = sib*32*VI + VI*((reg&3) + 8*(reg>>2) + 4*half) + c][co0 + d]
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;
#pragma unroll
      for (int c = 0; c < VI; ++c) {
        const int ci = sib * 32 * VI + VI * i + c;
        if (ci < Cin && co_ok) {
#pragma unroll
          for (int d = 0; d < VO; ++d) dst[static_cast<long long>(ci) * Cout + co0 + d] = acc[c][d][reg];
        }
      }
    }
  }
}

// dw[o] = sum over chunks, in a fixed order: thread (o, q) adds the chunks c = q mod 4 in ascending
// order (4 loads in flight), the four partial sums meet in LDS as (s0 + s1) + (s2 + s3)
// (oki_cin > 0: dw is the PARAMETER's layout [Cout][K][Cin] instead of [K][Cin][Cout] -- the training tape's
//  extra transposing launch per layer, 86 per step, folded in)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ partial, int chunks,
                                                          long long n, float *__restrict__ dw, int oki_k,
                                                          int oki_cin, int oki_cout, int csplit, int cico) {
  __shared__ float sh[4][64];
  const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
  const long long o = blockIdx.x * 64LL + l;
  float a = 0.f;
  if (o < n) {
    // (partial holds chunks * csplit slots per offset: the centre offset fills all of them, the others the first `chunks`)
    const bool centre = csplit > 1 && static_cast<int>(o / cico) == (oki_k >> 1);
    const int cnt = centre ? chunks * csplit : chunks;
    const long long step = n;
    int c = q;
    for (; c + 12 < cnt; c += 16) {
      const float v0 = partial[c * step + o], v1 = partial[(c + 4) * step + o];
      const float v2 = partial[(c + 8) * step + o], v3 = partial[(c + 12) * step + o];
      a += v0;
      a += v1;
      a += v2;
      a += v3;
    }
    for (; c < cnt; c += 4) a += partial[c * step + o];
  }
  sh[q][l] = a;
  __syncthreads();
  if (q == 0 && o < n) {
    long long at = o;
    if (oki_cin > 0) {
      const int co = static_cast<int>(o % oki_cout);
      const long long r = o / oki_cout;
      const int ci = static_cast<int>(r % oki_cin), k = static_cast<int>(r / oki_cin);
      at = (static_cast<long long>(co) * oki_k + k) * oki_cin + ci;
    }
    dw[at] = (sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l]);
  }
}

// gather table [M][K] -> [K][M]: the weight gradient walks one offset's column at a time, and a
// column of the row-major table costs a 128-B line per 4 useful bytes
__global__ void __launch_bounds__(256) transpose_table_kernel(const int32_t *__restrict__ nbr, int M, int K,
                                                             int32_t *__restrict__ nbr_t) {
  __shared__ int32_t tile[64 * kTMaxK];
  const int r0 = blockIdx.x * 64;
  const int rows = min(64, M - r0);
  for (int e = threadIdx.x; e < rows * K; e += 256) tile[e] = nbr[static_cast<long long>(r0) * K + e];
  __syncthreads();
  const int r = threadIdx.x & 63;
  for (int k = threadIdx.x >> 6; k < K; k += 4)
    if (r < rows) nbr_t[static_cast<long long>(k) * M + r0 + r] = tile[r * K + k];
}

struct WgradShape {
  int vi, vo, nst, zs, rows, chunks, csplit;
};
#ifndef SG_WGRAD_CSPLIT
#define SG_WGRAD_CSPLIT 4
#endif
// one decomposition for the workspace query and the launch: ~1024 workgroups
static WgradShape wgrad_shape(int M_out, int K, int Cin, int Cout) {
  WgradShape w;
  // two channels per lane from 64 channels up: one vector load feeds two MFMAs.  (For odd multiples
  // of 32 the last 64-channel supertile is half empty; one channel per lane tiles exactly but
  // doubles the loads per MFMA and measured slower: the loop is bound by load issue / latency.)
  w.vi = (Cin % 2 == 0 && Cin >= 64) ? 2 : 1;
  w.vo = (Cout % 2 == 0 && Cout >= 64) ? 2 : 1;
  w.nst = ((Cin + 32 * w.vi - 1) / (32 * w.vi)) * ((Cout + 32 * w.vo - 1) / (32 * w.vo));
  w.zs = w.nst >= 3 ? (w.nst + 3) / 4 : 1;
  const int chunks_want = (1024 + K * w.zs - 1) / (K * w.zs);
  int r = (M_out + chunks_want - 1) / chunks_want;
  r = (r + 1) & ~1;
  if (r < 128) r = 128;
  if (r > 2048) r = 2048;
  w.rows = r;
  w.chunks = (M_out + r - 1) / r;
  w.csplit = (K == 27 && r >= 64 * SG_WGRAD_CSPLIT) ? SG_WGRAD_CSPLIT : 1;      // (SubM: the centre offset is every row)
  return w;
}

template <typename TA, typename TG>
static void launch_wgrad(const void *in, const void *g, const int32_t *nbr, int M_out, int K, int Cin,
                         int Cout, const WgradShape &w, float *partial, hipStream_t stream) {
#ifdef SG_WGRAD_OFFSET_MAJOR
  const dim3 grid(K, w.chunks * w.csplit, w.zs);
#else
  const dim3 grid(w.chunks * w.csplit, K, w.zs);
#endif
  const size_t lds = static_cast<size_t>(w.rows) * 8 + (w.nst < 3 ? 2u * w.vi * w.vo * 16 * 64 * 4 : 0u);
  const TA *a = static_cast<const TA *>(in);
  const TG *b = static_cast<const TG *>(g);
  const int rows = w.rows;
  if (w.vi == 2 && w.vo == 2) conv_wgrad_kernel<TA, TG, 2, 2><<<grid, 256, lds, stream>>>(a, b, nbr, M_out, K, Cin, Cout, rows, w.csplit, partial);
  else if (w.vi == 2) conv_wgrad_kernel<TA, TG, 2, 1><<<grid, 256, lds, stream>>>(a, b, nbr, M_out, K, Cin, Cout, rows, w.csplit, partial);
  else if (w.vo == 2) conv_wgrad_kernel<TA, TG, 1, 2><<<grid, 256, lds, stream>>>(a, b, nbr, M_out, K, Cin, Cout, rows, w.csplit, partial);
  else conv_wgrad_kernel<TA, TG, 1, 1><<<grid, 256, lds, stream>>>(a, b, nbr, M_out, K, Cin, Cout, rows, w.csplit, partial);
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_spconv_packed_weight_elems_bf16(int kvol, int cin, int cout) {
  return static_cast<size_t>(kvol) * ((cin + 15) / 16 * 2) * cout * 8;
}

int sg_spconv_pack_weight_bf16(const float *w, int cout, int kvol, int cin, int src_is_kio, uint16_t *w_k8,
                               sg_stream_t stream) {
  SG_REQUIRE(cout > 0 && kvol > 0 && cin > 0 && w && w_k8, "sg_spconv_pack_weight_bf16: bad arguments");
  const int64_t total = static_cast<int64_t>(sg_spconv_packed_weight_elems_bf16(kvol, cin, cout));
  pack_weight_bf16_kernel<<<grid_for(total, 256), 256, 0, as_stream(stream)>>>(w, cout, kvol, cin,
                                                                              src_is_kio, w_k8);
  return check_launch("sg_spconv_pack_weight_bf16");
}

// workspace of the offset-split path (deep levels only): K * M_out * Cout floats
size_t sg_spconv_conv_bf16_workspace_bytes(int M_out, int Cout) {
  const int num_tiles = (M_out + kTRows - 1) / kTRows;
  if (num_tiles >= 1024) return 256;
  return static_cast<size_t>(kTMaxK) * M_out * Cout * sizeof(float) + 256;
}

int sg_spconv_gather_conv_bf16(const uint16_t *in, int num_in_rows, const int32_t *nbr, int M_out, int K,
                               int Cin, int Cout, const uint16_t *w_k8, const int32_t *order,
                               const uint32_t *tile_mask, const int32_t *nbr_tiles, uint16_t *out,
                               void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(M_out >= 0 && K >= 1 && K <= kTMaxK && Cin >= 1 && Cout >= 1 && num_in_rows >= 0,
             "sg_spconv_gather_conv_bf16: bad arguments (M_out=%d K=%d Cin=%d Cout=%d)", M_out, K, Cin, Cout);
  SG_REQUIRE((order == nullptr) == (tile_mask == nullptr) && (order == nullptr) == (nbr_tiles == nullptr),
             "sg_spconv_gather_conv_bf16: order, tile_mask and nbr_tiles come together (sg_spconv_plan)");
  SG_REQUIRE(order != nullptr || nbr != nullptr, "sg_spconv_gather_conv_bf16: no gather table");
  if (M_out == 0) return SG_OK;
  const long long in_bytes = static_cast<long long>(num_in_rows) * Cin * 2;
  const long long w_bytes = static_cast<long long>(sg_spconv_packed_weight_elems_bf16(K, Cin, Cout)) * 2;
  SG_REQUIRE(in_bytes < (1LL << 31) && w_bytes < (1LL << 31),
             "sg_spconv_gather_conv_bf16: input (%lld B) or weights (%lld B) exceed the 2 GiB buffer range",
             in_bytes, w_bytes);
  hipStream_t stream = as_stream(stream_);
  const int NB = (Cout + 31) / 32;
  const int num_tiles = (M_out + kTRows - 1) / kTRows;
  // column blocks per unit: as even as possible, at most 4 (64 accumulator registers)
  const int col_units = (NB + 3) / 4;
  const int bpu = (NB + col_units - 1) / col_units;
  int ksplit = 1;
  const long long waves = static_cast<long long>(num_tiles) * col_units;
  if (waves < 1024) {
    const long long want = (1024 + waves - 1) / waves;
    ksplit = static_cast<int>(want < K ? want : K);
    const size_t need = static_cast<size_t>(ksplit) * M_out * Cout * sizeof(float);
    if (ksplit > 1 && (ws == nullptr || ws_bytes < need)) ksplit = 1;
  }
  const int k_per_split = (K + ksplit - 1) / ksplit;
  ksplit = (K + k_per_split - 1) / k_per_split;

  Bf16Args a;
  a.in = in; a.nbr = nbr; a.w = w_k8; a.order = order; a.tile_mask = tile_mask; a.nbr_tiles = nbr_tiles;
  a.out = out; a.partial = static_cast<float *>(ws);
  a.M_out = M_out; a.K = K; a.Cin = Cin; a.Cout = Cout;
  a.col_units = col_units; a.blocks_per_unit = bpu; a.ksplit = ksplit; a.k_per_split = k_per_split;
  a.in_bytes = static_cast<unsigned>(in_bytes);
  a.w_bytes = static_cast<unsigned>(w_bytes);
  const long long units = static_cast<long long>(num_tiles) * col_units * ksplit;
  const int grid = static_cast<int>((units + 3) / 4);
  const size_t lds = 4 * kTRows * kTMaxK * sizeof(int32_t);
  const bool vec = Cin % 16 == 0;
  switch (bpu) {
    case 1: launch_bf16<1>(a, grid, lds, vec, stream); break;
    case 2: launch_bf16<2>(a, grid, lds, vec, stream); break;
    case 3: launch_bf16<3>(a, grid, lds, vec, stream); break;
    default: launch_bf16<4>(a, grid, lds, vec, stream); break;
  }
  if (ksplit > 1) {
    const long long n = static_cast<long long>(M_out) * Cout;
    conv_reduce_bf16_kernel<<<grid_for(n, 256), 256, 0, stream>>>(static_cast<const float *>(ws), ksplit, n, out);
  }
  return check_launch("sg_spconv_gather_conv_bf16");
}

int sg_spconv_transpose_table(const int32_t *nbr, int M, int K, int32_t *nbr_t, sg_stream_t stream) {
  SG_REQUIRE(M >= 0 && K >= 1 && K <= kTMaxK, "sg_spconv_transpose_table: bad arguments");
  if (M == 0) return SG_OK;
  transpose_table_kernel<<<(M + 63) / 64, 256, 0, as_stream(stream)>>>(nbr, M, K, nbr_t);
  return check_launch("sg_spconv_transpose_table");
}

size_t sg_spconv_wgrad_workspace_bytes(int M_out, int K, int Cin, int Cout) {
  if (M_out <= 0 || K <= 0 || Cin <= 0 || Cout <= 0) return 256;
  const WgradShape w = wgrad_shape(M_out, K, Cin, Cout);
  return static_cast<size_t>(w.chunks) * w.csplit * K * Cin * Cout * sizeof(float) + 256;
}

// dw_kio [K][Cin][Cout] fp32 is overwritten.  `in` is the input the forward conv gathered from
// (fp32, or bf16 when in_bf16), g_out the gradient of its output (fp32 | bf16).
int sg_spconv_wgrad(const void *in, int in_bf16, const void *g_out, int g_bf16, const int32_t *nbr_t,
                    int M_out, int K, int Cin, int Cout, float *dw_kio, void *ws, size_t ws_bytes,
                    sg_stream_t stream_) {
  return sg::spconv_wgrad_layout(in, in_bf16, g_out, g_bf16, nbr_t, M_out, K, Cin, Cout, dw_kio, 0, ws, ws_bytes,
                                 stream_);
}

}  // extern "C"

namespace sg {
// out_oki != 0: the gradient lands in the parameter's layout [Cout][K][Cin]
int spconv_wgrad_layout(const void *in, int in_bf16, const void *g_out, int g_bf16, const int32_t *nbr_t, int M_out,
                        int K, int Cin, int Cout, float *dw_kio, int out_oki, void *ws, size_t ws_bytes,
                        sg_stream_t stream_) {
  SG_REQUIRE(M_out >= 0 && K >= 1 && K <= kTMaxK && Cin >= 1 && Cout >= 1 && dw_kio,
             "sg_spconv_wgrad: bad arguments");
  hipStream_t stream = as_stream(stream_);
  const long long n = static_cast<long long>(K) * Cin * Cout;
  if (M_out == 0) {
    hipMemsetAsync(dw_kio, 0, static_cast<size_t>(n) * 4, stream);
    return check_launch("sg_spconv_wgrad");
  }
  SG_REQUIRE(ws != nullptr && ws_bytes >= sg_spconv_wgrad_workspace_bytes(M_out, K, Cin, Cout),
             "sg_spconv_wgrad: workspace too small (sg_spconv_wgrad_workspace_bytes)");
  const WgradShape w = wgrad_shape(M_out, K, Cin, Cout);
  float *partial = static_cast<float *>(ws);
  if (in_bf16 && g_bf16) launch_wgrad<uint16_t, uint16_t>(in, g_out, nbr_t, M_out, K, Cin, Cout, w, partial, stream);
  else if (in_bf16) launch_wgrad<uint16_t, float>(in, g_out, nbr_t, M_out, K, Cin, Cout, w, partial, stream);
  else if (g_bf16) launch_wgrad<float, uint16_t>(in, g_out, nbr_t, M_out, K, Cin, Cout, w, partial, stream);
  else launch_wgrad<float, float>(in, g_out, nbr_t, M_out, K, Cin, Cout, w, partial, stream);
  wgrad_reduce_kernel<<<static_cast<int>((n + 63) / 64), 256, 0, stream>>>(partial, w.chunks, n, dw_kio, K,
                                                                           out_oki ? Cin : 0, Cout, w.csplit, Cin * Cout);
  return check_launch("sg_spconv_wgrad");
}
}  // namespace sg
