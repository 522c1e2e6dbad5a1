// spconv_conv.hip -- sparse convolution as an output-stationary implicit GEMM on fp32 MFMA.
// Replaces the conv kernels of the un-vendored spconv 2.1 library for SubMConv3d /
// SparseConv3d(k2,s2) / SparseInverseConv3d (reference call sites: softgroup/model/softgroup.py:61,
// softgroup/model/blocks.py:57-70,101-119).  One kernel serves all three: it only sees a gather
// table nbr[M_out, K] (spconv_rulebook.hip) and weights re-laid out as [K][Cin][Cout].
//
//   out[j,:] = residual[j,:] + sum_k  act(in[nbr[j,k],:]) . W[k]        act = relu(x*s + b) | id
//
// MI355X mapping (v2: wave-private register pipeline, no LDS staging of operands, no barriers
// in the main loop -- sparse tiles have ragged depth, so lock-stepping waves wastes the machine)
//   * work unit = one wave = (tile of 32 output rows in neighbour-mask-sorted order) x (up to NBW
//     32-column blocks of Cout) x (a slice of the kernel offsets when the layer is too small to
//     fill 256 CUs otherwise).  The tile's 27-bit mask says which offsets exist at all.
//   * v_mfma_f32_32x32x2_f32 (exact fp32 = fmaf chain, 64 FLOP/clk/SIMD).  The reduction index
//     of a 16-channel slice is permuted so that lane (h, i) owns channels c0+8h .. c0+8h+7 of
//     gathered row i: its A operands for 8 MFMA steps are ONE contiguous 32-B read of that row
//     (two dwordx4), and its B operands are coalesced 128-B reads of W[k][c][32 cols].
//   * operands for slice t+1 are loaded into a second register set while slice t's MFMAs issue
//     (an fp32 MFMA occupies the SIMD for 64 cycles, so one wave-wide load per MFMA is cheap);
//     several waves per SIMD interleave freely because nothing synchronises them.
//   * fused eval-BatchNorm+ReLU on the gathered rows (scale/shift broadcast from LDS), fused
//     residual add, every output row written once, 128 B per half-wave.
//   * tiny layers (deep U-Net levels: 18..800 rows) split the kernel offsets over several waves;
//     partial sums go to a workspace and are reduced in a fixed order (deterministic).
// HBM traffic per layer ~ P*Cin*4 (gathered rows) + M*Cout*4 (stores) + index tables, i.e. the
// gather/scatter bytes B_gs of SURVEY 8(d); the weights (<= 8 MB) are served from L2/MALL.
#include <stdlib.h>

#include "common.h"

namespace sg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileRows = 32;
constexpr int kWavesPerWg = 4;
constexpr int kCk = 16;        // channels per pipeline slice (8 per half-wave)
constexpr int kMaxK = 27;

struct ConvArgs {
  const float *in;
  const int32_t *nbr;
  const float *w;         // [K][Cin][Cout]
  const float *bn_scale;  // [Cin] or null
  const float *bn_shift;
  const float *residual;  // [M_out][Cout] or null (ignored when writing partials)
  const int32_t *order;   // [M_out] or null
  const uint32_t *tile_mask;
  const int32_t *nbr_tiles;   // [num_tiles][32][K] gather rows in plan order, or null
  float *out;             // [M_out][Cout], or partials [ksplit][M_out][Cout]
  int M_out, K, Cin, Cout;
  int col_units;          // wave units along Cout
  int blocks_per_unit;    // 32-col blocks per unit (last unit may have fewer)
  int ksplit;             // slices of the kernel-offset range
  int k_per_split;
};

template <int NBW, bool VEC>
__global__ void __launch_bounds__(256) gather_conv_v2_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // LDS: per-wave neighbour table [32][K] + bn scale/shift [2][CinPad]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cin_pad = (p.Cin + kCk - 1) / kCk * kCk;
  float *bn_lds = reinterpret_cast<float *>(smem_raw);            // [2][cin_pad]
  int32_t *nbr_lds = reinterpret_cast<int32_t *>(bn_lds + 2 * cin_pad) + wave * kTileRows * kMaxK;

  if (p.bn_scale) {
    for (int c = threadIdx.x; c < cin_pad; c += 256) {
      bn_lds[c] = c < p.Cin ? p.bn_scale[c] : 0.f;
      bn_lds[cin_pad + c] = c < p.Cin ? p.bn_shift[c] : 0.f;
    }
  }

  const int num_tiles = (p.M_out + kTileRows - 1) / kTileRows;
  const int units_per_tile = p.col_units * p.ksplit;
  const long long unit = static_cast<long long>(blockIdx.x) * kWavesPerWg + wave;
  const bool valid = unit < static_cast<long long>(num_tiles) * units_per_tile;
  int tile = valid ? static_cast<int>(unit / units_per_tile) : 0;
  const int sub = valid ? static_cast<int>(unit % units_per_tile) : 0;
  const int cu = sub % p.col_units, ks = sub / p.col_units;
  const int nb0 = cu * p.blocks_per_unit;
  const int nbw = valid ? min(p.blocks_per_unit, (p.Cout + 31) / 32 - nb0) : 0;
  const int k_lo = ks * p.k_per_split, k_hi = min(p.K, k_lo + p.k_per_split);

  const int arow = lane & 31, ahalf = lane >> 5;
  // my output row (both halves hold the same rows) and the tile's neighbour table
  int my_row = -1;
  if (valid) {
    const int pos = tile * kTileRows + arow;
    if (p.order) my_row = p.order[pos];       // plan order: padded with -1 to whole tiles
    else if (pos < p.M_out) my_row = pos;
  }
  if (valid) {
    for (int e = lane; e < kTileRows * p.K; e += 64) {
      const int r = e / p.K, k = e - r * p.K;
      const int row = __shfl(my_row, r, 64);
      nbr_lds[r * kMaxK + k] = row >= 0 ? p.nbr[static_cast<long long>(row) * p.K + k] : -1;
    }
  }
  __syncthreads();  // bn_lds + nbr_lds visible (only barrier of the kernel)
  if (!valid) return;

  uint32_t mask = p.tile_mask ? p.tile_mask[tile] : 0xffffffffu;
  mask &= (k_hi >= 32 ? 0xffffffffu : ((1u << k_hi) - 1u)) & ~((1u << k_lo) - 1u);

  f32x16 acc[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  const int n_slices = cin_pad / kCk;
  const int col = nb0 * 32 + arow;

  // Column offsets of this unit's blocks; surplus blocks (n >= nbw) re-read the last real one so
  // that every load below is unconditional (hipcc branches around predicated loads).  All element
  // offsets are 32-bit (host checks the tensors are < 2^31 elements).
  int coff[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n) coff[n] = min(col + min(n, nbw - 1) * 32, p.Cout - 1);

  // raw loads of slice (k, s): A = 8 consecutive channels of the gathered row, B = 8 x NBW weights
  auto load_raw = [&](int k, int s, float (&a)[8], float (&b)[NBW][8], bool &present) {
    const int c = s * kCk + ahalf * 8;
    const int src = nbr_lds[arow * kMaxK + k];
    present = src >= 0;
    const float *row = p.in + static_cast<unsigned>((present ? src : 0) * p.Cin + c);
    if (VEC) {
      const float4 v0 = *reinterpret_cast<const float4 *>(row);
      const float4 v1 = *reinterpret_cast<const float4 *>(row + 4);
      a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
      a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = row[max(0, min(j, p.Cin - 1 - c))];
    }
    const int wbase = (k * p.Cin + c) * p.Cout;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int wrow = VEC ? wbase + j * p.Cout : (k * p.Cin + min(c + j, p.Cin - 1)) * p.Cout;
#pragma unroll
      for (int n = 0; n < NBW; ++n) b[n][j] = p.w[static_cast<unsigned>(wrow + coff[n])];
    }
  };
  // fused BatchNorm+ReLU and zeroing of absent neighbours / padded channels, applied when the
  // slice becomes current (so the wait for its loads sits AFTER the previous MFMA block)
  auto finish = [&](int s, const float (&raw)[8], bool present, float (&a)[8]) {
    const int c = s * kCk + ahalf * 8;
    if (p.bn_scale) {
      const float4 s0 = *reinterpret_cast<const float4 *>(bn_lds + c);
      const float4 s1 = *reinterpret_cast<const float4 *>(bn_lds + c + 4);
      const float4 h0 = *reinterpret_cast<const float4 *>(bn_lds + cin_pad + c);
      const float4 h1 = *reinterpret_cast<const float4 *>(bn_lds + cin_pad + c + 4);
      const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = fmaxf(fmaf(raw[j], sc[j], sh[j]), 0.f);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = raw[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (present && (VEC || c + j < p.Cin)) ? a[j] : 0.f;
  };

  float a_cur[8], a_nxt[8];
  float b_cur[NBW][8], b_nxt[NBW][8];
  bool pres_nxt = false;

  // flattened iteration space: (offset k in mask) x (slice s)
  int k = mask ? __builtin_ctz(mask) : -1;
  int s = 0;
  if (k >= 0) load_raw(k, 0, a_nxt, b_nxt, pres_nxt);
  while (k >= 0) {
    finish(s, a_nxt, pres_nxt, a_cur);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int n = 0; n < NBW; ++n) b_cur[n][j] = b_nxt[n][j];
    // next (k, s) and its prefetch
    int k2 = k, s2 = s + 1;
    if (s2 == n_slices) {
      s2 = 0;
      const uint32_t rest = mask & ~((2u << k) - 1u);
      k2 = rest ? __builtin_ctz(rest) : -1;
    }
    if (k2 >= 0) load_raw(k2, s2, a_nxt, b_nxt, pres_nxt);
    __builtin_amdgcn_sched_barrier(0);   // loads are in flight ...
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int n = 0; n < NBW; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[j], b_cur[n][j], acc[n], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);   // ... while this block issues; consume them only after
    k = k2;
    s = s2;
  }

  // ---- epilogue: acc[n][reg] -> row (reg&3)+8*(reg>>2)+4*half, column nb0*32 + n*32 + arow
  float *out = p.out + (p.ksplit > 1 ? static_cast<long long>(ks) * p.M_out * p.Cout : 0);
  const bool add_res = p.residual != nullptr && p.ksplit == 1;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int r = (reg & 3) + 8 * (reg >> 2) + 4 * ahalf;
    const int row = __shfl(my_row, r, 64);
    if (row < 0) continue;
    const long long off = static_cast<long long>(row) * p.Cout + col;
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      if (n < nbw && col + n * 32 < p.Cout) {
        float v = acc[n][reg];
        if (add_res) v += p.residual[off + n * 32];
        out[off + n * 32] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// v3 main loop for Cin % 32 == 0: 32-channel slices (16 MFMA steps x NBW per iteration), operands
// fetched with buffer loads whose per-lane offsets are loop invariants (all per-iteration address
// arithmetic is scalar), two register sets used alternately (no copies), the prefetch of slice
// t+1 is issued before slice t's MFMA block so that its latency hides under >= 1024 cycles of
// matrix work.
// ---------------------------------------------------------------------------------------------

template <int NBW, int CK>
__global__ void __launch_bounds__(256) gather_conv_v3_kernel(ConvArgs p, unsigned in_bytes,
                                                            unsigned w_bytes) {
  constexpr int HC = CK / 2;   // channels per lane per slice (8 or 16)
  constexpr int NQ = HC / 4;   // dwordx4 loads per lane per slice
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // wave id through readfirstlane: everything derived from it is then provably wave-uniform, so
  // buffer-load scalar offsets stay in SGPRs (no waterfall loops)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float *red = reinterpret_cast<float *>(smem_raw);                      // [4][NBW][16][64]
  float *bn_lds = red + kWavesPerWg * NBW * 16 * 64;                      // [2][Cin]
  int32_t *nbr_lds = reinterpret_cast<int32_t *>(bn_lds + 2 * p.Cin);     // [32][kMaxK]
  int32_t *rows_lds = nbr_lds + kTileRows * kMaxK;                        // [32]
  if (p.bn_scale) {
    for (int c = threadIdx.x; c < p.Cin; c += 256) {
      bn_lds[c] = p.bn_scale[c];
      bn_lds[p.Cin + c] = p.bn_shift[c];
    }
  }
  // workgroup = (tile, column unit, offset range); its 4 waves split the tile's offsets
  const int units_per_tile = p.col_units * p.ksplit;
  int tile = blockIdx.x / units_per_tile;
  const int sub = blockIdx.x % units_per_tile;
  const int cu = sub % p.col_units, ks = sub / p.col_units;
  const int nb0 = cu * p.blocks_per_unit;
  const int nbw = min(p.blocks_per_unit, (p.Cout + 31) / 32 - nb0);
  const int k_lo = ks * p.k_per_split, k_hi = min(p.K, k_lo + p.k_per_split);
  const int arow = lane & 31, ahalf = lane >> 5;
  // plan layout: rows and gather-table block of the tile are contiguous, independent loads
  if (threadIdx.x < kTileRows) rows_lds[threadIdx.x] = p.order[tile * kTileRows + threadIdx.x];
  {
    const int32_t *src = p.nbr_tiles + static_cast<long long>(tile) * kTileRows * p.K;
    for (int e = threadIdx.x; e < kTileRows * p.K; e += 256) {
      const int r = e / p.K, k = e - r * p.K;
      nbr_lds[r * kMaxK + k] = src[e];
    }
  }
  __syncthreads();

  uint32_t wg_mask = p.tile_mask ? p.tile_mask[tile] : 0xffffffffu;
  wg_mask &= (k_hi >= 32 ? 0xffffffffu : ((1u << k_hi) - 1u)) & ~((1u << k_lo) - 1u);
  // offsets are dealt to the 4 waves round-robin by their rank among the set bits
  uint32_t mask = 0;
  {
    uint32_t m = wg_mask;
    int rank = 0;
    while (m) {
      const uint32_t low = m & (0u - m);
      if ((rank & (kWavesPerWg - 1)) == wave) mask |= low;
      m ^= low;
      ++rank;
    }
  }
  mask = __builtin_amdgcn_readfirstlane(mask);

  f32x16 acc[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  const int n_slices = p.Cin / CK;
  const int col = nb0 * 32 + arow;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, w_bytes, 0x00020000);
  // loop-invariant per-lane byte offsets of the weight reads (surplus blocks repeat the last one)
  int v_w[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n)
    v_w[n] = ((ahalf * HC) * p.Cout + min(col + min(n, nbw - 1) * 32, p.Cout - 1)) * 4;
  const int row_stride = p.Cout * 4;

  typedef float f4 __attribute__((ext_vector_type(4)));
  struct Slice {
    f4 a[NQ];         // HC channels of the gathered row
    float b[NBW][HC];
    bool present;
  };
  Slice S0, S1;

  auto load = [&](int k, int s, Slice &S) {
    const int src = nbr_lds[arow * kMaxK + k];
    S.present = src >= 0;
    const int v_a = ((S.present ? src : 0) * p.Cin + ahalf * HC) * 4;
    const int s_a = s * (CK * 4);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      S.a[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_a + q * 16, s_a, 0));
    const int s_w = (k * p.Cin + s * CK) * row_stride;
#pragma unroll
    for (int j = 0; j < HC; ++j)
#pragma unroll
      for (int n = 0; n < NBW; ++n)
        S.b[n][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, v_w[n], s_w + j * row_stride, 0));
  };
  auto compute = [&](int s, Slice &S) {
    float a[HC];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { a[4 * q] = S.a[q][0]; a[4 * q + 1] = S.a[q][1]; a[4 * q + 2] = S.a[q][2]; a[4 * q + 3] = S.a[q][3]; }
    if (p.bn_scale) {
      const int c = s * CK + ahalf * HC;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 sc = *reinterpret_cast<const float4 *>(bn_lds + c + 4 * q);
        const float4 sh = *reinterpret_cast<const float4 *>(bn_lds + p.Cin + c + 4 * q);
        a[4 * q] = fmaxf(fmaf(a[4 * q], sc.x, sh.x), 0.f);
        a[4 * q + 1] = fmaxf(fmaf(a[4 * q + 1], sc.y, sh.y), 0.f);
        a[4 * q + 2] = fmaxf(fmaf(a[4 * q + 2], sc.z, sh.z), 0.f);
        a[4 * q + 3] = fmaxf(fmaf(a[4 * q + 3], sc.w, sh.w), 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < HC; ++j) a[j] = S.present ? a[j] : 0.f;
#pragma unroll
    for (int j = 0; j < HC; ++j)
#pragma unroll
      for (int n = 0; n < NBW; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], S.b[n][j], acc[n], 0, 0, 0);
  };
  auto advance = [&](int &k, int &s) {
    if (++s == n_slices) {
      s = 0;
      const uint32_t rest = mask & ~((2u << k) - 1u);
      k = rest ? __builtin_ctz(rest) : -1;
    }
  };

  // Every load below is issued unconditionally (past the end the current slice is re-read and
  // dropped): with no branch between a prefetch and the MFMA block that precedes its use, hipcc's
  // wait-count insertion keeps exact counts (vmcnt(20*NBW-ish) instead of draining the prefetch).
  int k = mask ? __builtin_ctz(mask) : -1, s = 0;
  if (k >= 0) {
    load(k, s, S0);
    while (true) {
      int k1 = k, s1 = s;
      advance(k1, s1);
      const bool more1 = k1 >= 0;
      load(more1 ? k1 : k, more1 ? s1 : s, S1);
      __builtin_amdgcn_sched_barrier(0);   // prefetch of the next slice is in flight ...
      compute(s, S0);
      __builtin_amdgcn_sched_barrier(0);   // ... and is only consumed after this MFMA block
      if (!more1) break;
      int k2 = k1, s2 = s1;
      advance(k2, s2);
      const bool more2 = k2 >= 0;
      load(more2 ? k2 : k1, more2 ? s2 : s1, S0);
      __builtin_amdgcn_sched_barrier(0);
      compute(s1, S1);
      __builtin_amdgcn_sched_barrier(0);
      if (!more2) break;
      k = k2;
      s = s2;
    }
  }

  // ---- cross-wave reduction through LDS in a fixed order (w0+w1+w2+w3), then each wave stores
  //      4 of the 16 accumulator rows-groups: reg -> row (reg&3)+8*(reg>>2)+4*half
#pragma unroll
  for (int n = 0; n < NBW; ++n)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) red[((wave * NBW + n) * 16 + reg) * 64 + lane] = acc[n][reg];
  __syncthreads();
  float *out = p.out + (p.ksplit > 1 ? static_cast<long long>(ks) * p.M_out * p.Cout : 0);
  const bool add_res = p.residual != nullptr && p.ksplit == 1;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int reg = wave * 4 + rr;
    const int r = (reg & 3) + 8 * (reg >> 2) + 4 * ahalf;
    const int row = rows_lds[r];
    if (row < 0) continue;
    const long long off = static_cast<long long>(row) * p.Cout + col;
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      if (n < nbw && col + n * 32 < p.Cout) {
        float v = red[((0 * NBW + n) * 16 + reg) * 64 + lane];
#pragma unroll
        for (int w = 1; w < kWavesPerWg; ++w) v += red[((w * NBW + n) * 16 + reg) * 64 + lane];
        if (add_res) v += p.residual[off + n * 32];
        out[off + n * 32] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[k][ci][co] += sum_j act_in[nbr[j,k]][ci] * g_out[j][co]   (training path;
// reference: spconv's backward, called through autograd from tools/train.py:58).
// grid = (row chunks, K).  A workgroup walks its rows two at a time: the MFMA A operand is the
// transposed input row pair (lane (h,ci) <- in[row 2s+h][ci]), B the gradient row pair, so both
// are 128-B coalesced row reads.  Each wave owns (ci-block, co-block) pairs round-robin and
// accumulates 32x32 fp32 in registers; chunk partials are added with fp32 atomics (the sum over
// chunks is the only non-deterministic step; chunks are large, ~1k rows).
// ---------------------------------------------------------------------------------------------
constexpr int kWgradRows = 1024;

__global__ void __launch_bounds__(256) conv_wgrad_kernel(const float *__restrict__ in,
                                                        const float *__restrict__ g_out,
                                                        const int32_t *__restrict__ nbr, int M_out,
                                                        int K, int Cin, int Cout,
                                                        float *__restrict__ dw_kio) {
  __shared__ int32_t src_lds[kWgradRows];
  const int k = blockIdx.y;
  const int r0 = blockIdx.x * kWgradRows;
  const int nrows = min(kWgradRows, M_out - r0);
  int any = 0;
  for (int r = threadIdx.x; r < kWgradRows; r += 256) {
    const int s = r < nrows ? nbr[static_cast<long long>(r0 + r) * K + k] : -1;
    src_lds[r] = s;
    any |= (s >= 0);
  }
  if (!__syncthreads_or(any)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 31, half = lane >> 5;
  const int nbi = (Cin + 31) / 32, nbo = (Cout + 31) / 32;
  for (int pair = wave; pair < nbi * nbo; pair += 4) {
    const int cib = pair / nbo, cob = pair % nbo;
    const int ci = cib * 32 + col, co = cob * 32 + col;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int r = 0; r < nrows; r += 2) {
      const int rr = r + half;
      const int s = rr < nrows ? src_lds[rr] : -1;
      const float a = (s >= 0 && ci < Cin) ? in[static_cast<long long>(s) * Cin + ci] : 0.f;
      const float b = (s >= 0 && co < Cout) ? g_out[static_cast<long long>(r0 + rr) * Cout + co] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // acc[reg] = dW[ci = cib*32 + (reg&3)+8*(reg>>2)+4*half][co = cob*32 + col]
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int cir = cib * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
      if (cir < Cin && co < Cout && acc[reg] != 0.f)
        atomicAdd(&dw_kio[(static_cast<long long>(k) * Cin + cir) * Cout + co], acc[reg]);
    }
  }
}

// fixed-order reduction of the offset-split partial sums (+ residual)
__global__ void __launch_bounds__(256) conv_reduce_kernel(const float4 *__restrict__ partial,
                                                         const float4 *__restrict__ residual,
                                                         int ksplit, long long n4,
                                                         float4 *__restrict__ out) {
  for (long long t = blockIdx.x * 256LL + threadIdx.x; t < n4; t += gridDim.x * 256LL) {
    float4 a = partial[t];
    for (int s = 1; s < ksplit; ++s) {
      const float4 b = partial[s * n4 + t];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (residual) {
      const float4 r = residual[t];
      a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    }
    out[t] = a;
  }
}

// Scalar path of the same operator for channel counts the MFMA tiling does not cover
// (Cout % 4 != 0).  One thread per (row, cout).
__global__ void __launch_bounds__(256) gather_conv_scalar_kernel(
    const float *__restrict__ in, const int32_t *__restrict__ nbr, int M_out, int K, int Cin,
    int Cout, const float *__restrict__ w_kio, const float *__restrict__ bn_scale,
    const float *__restrict__ bn_shift, const float *__restrict__ residual,
    float *__restrict__ out) {
  const int64_t total = static_cast<int64_t>(M_out) * Cout;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int j = static_cast<int>(t / Cout), co = static_cast<int>(t - static_cast<int64_t>(j) * Cout);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int s = nbr[static_cast<int64_t>(j) * K + k];
      if (s < 0) continue;
      const float *x = in + static_cast<int64_t>(s) * Cin;
      const float *w = w_kio + static_cast<int64_t>(k) * Cin * Cout + co;
      for (int ci = 0; ci < Cin; ++ci) {
        float v = x[ci];
        if (bn_scale) v = fmaxf(fmaf(v, bn_scale[ci], bn_shift[ci]), 0.f);
        acc = fmaf(v, w[static_cast<int64_t>(ci) * Cout], acc);
      }
    }
    if (residual) acc += residual[t];
    out[t] = acc;
  }
}

template <int NBW>
static void launch_v2(const ConvArgs &a, int grid, size_t lds, bool vec, hipStream_t stream) {
  if (vec)
    gather_conv_v2_kernel<NBW, true><<<grid, 256, lds, stream>>>(a);
  else
    gather_conv_v2_kernel<NBW, false><<<grid, 256, lds, stream>>>(a);
}

}  // namespace sg

using namespace sg;

extern "C" {

// workspace for the offset-split path: ksplit_max * M_out * Cout floats
size_t sg_spconv_conv_workspace_bytes(int M_out, int Cout) {
  const int num_tiles = (M_out + kTileRows - 1) / kTileRows;
  if (num_tiles * ((Cout + 31) / 32) >= 1024) return 256;   // big layers never split
  return static_cast<size_t>(kMaxK) * M_out * Cout * sizeof(float) + 256;
}

int sg_spconv_gather_conv_f32(const float *in, int num_in_rows, const int32_t *nbr, int M_out,
                              int K, int Cin, int Cout, const float *w_kio, const float *bn_scale,
                              const float *bn_shift, const float *residual, const int32_t *order,
                              const uint32_t *tile_mask, const int32_t *nbr_tiles, float *out,
                              void *ws, size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(M_out >= 0 && K >= 1 && K <= kMaxK && Cin >= 1 && Cout >= 1,
             "sg_spconv_gather_conv_f32: bad arguments (M_out=%d K=%d Cin=%d Cout=%d)", M_out, K,
             Cin, Cout);
  SG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr),
             "sg_spconv_gather_conv_f32: bn_scale and bn_shift must come together");
  if (M_out == 0) return SG_OK;
  hipStream_t stream = as_stream(stream_);
  if (Cout % 4 != 0) {
    gather_conv_scalar_kernel<<<grid_for(static_cast<int64_t>(M_out) * Cout, 256, 256 * 32), 256, 0,
                                stream>>>(in, nbr, M_out, K, Cin, Cout, w_kio, bn_scale, bn_shift,
                                          residual, out);
    return check_launch("sg_spconv_gather_conv_f32(scalar)");
  }
  const int NB = (Cout + 31) / 32;
  const int num_tiles = (M_out + kTileRows - 1) / kTileRows;
  // ---- decomposition: aim at >= ~2048 waves; widest column block that still fills the chip
  const int target = 2048;
  const long long in_bytes_ll = static_cast<long long>(num_in_rows) * Cin * 4;
  const long long w_bytes_ll = static_cast<long long>(K) * Cin * Cout * 4;
  const bool use_v3 = Cin % 32 == 0 && in_bytes_ll < (1LL << 31) && num_in_rows > 0 &&
                      order != nullptr && tile_mask != nullptr && nbr_tiles != nullptr;
  const int waves_per_unit = use_v3 ? kWavesPerWg : 1;   // v3: a workgroup's 4 waves share a tile
  static const int bpu_env = getenv("SG_CONV_BPU") ? atoi(getenv("SG_CONV_BPU")) : 0;
  int bpu = 1;                                    // 32-column blocks per unit
  if (bpu_env) bpu = bpu_env < NB ? bpu_env : NB;
  while (bpu > 1 &&
         static_cast<long long>(num_tiles) * ((NB + bpu - 1) / bpu) * waves_per_unit < target)
    --bpu;
  int col_units = (NB + bpu - 1) / bpu;
  int ksplit = 1;
  long long waves = static_cast<long long>(num_tiles) * col_units * waves_per_unit;
  if (waves < target / 2) {
    { long long want = (target / 2 + waves - 1) / waves; ksplit = static_cast<int>(want < K ? want : K); }
    const size_t need = static_cast<size_t>(ksplit) * M_out * Cout * sizeof(float);
    if (ksplit > 1 && (ws == nullptr || ws_bytes < need)) ksplit = 1;   // no scratch: stay exact
  }
  const int k_per_split = (K + ksplit - 1) / ksplit;
  ksplit = (K + k_per_split - 1) / k_per_split;

  ConvArgs a;
  a.in = in; a.nbr = nbr; a.w = w_kio; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
  a.residual = residual; a.order = order; a.tile_mask = tile_mask; a.nbr_tiles = nbr_tiles;
  a.out = ksplit > 1 ? static_cast<float *>(ws) : out;
  a.M_out = M_out; a.K = K; a.Cin = Cin; a.Cout = Cout;
  a.col_units = col_units; a.blocks_per_unit = bpu; a.ksplit = ksplit; a.k_per_split = k_per_split;

  const int cin_pad = (Cin + kCk - 1) / kCk * kCk;
  const size_t lds = 2 * cin_pad * sizeof(float) + kWavesPerWg * kTileRows * kMaxK * sizeof(int32_t);
  const long long units = static_cast<long long>(num_tiles) * col_units * ksplit;
  const int grid = static_cast<int>((units + kWavesPerWg - 1) / kWavesPerWg);
  const bool vec = (Cin % kCk) == 0;
  if (use_v3 && bpu <= 2) {
    const size_t lds3 = (static_cast<size_t>(kWavesPerWg) * bpu * 16 * 64 + 2 * Cin) * sizeof(float) +
                        (kTileRows * kMaxK + kTileRows) * sizeof(int32_t);
    const unsigned ib = static_cast<unsigned>(in_bytes_ll), wb = static_cast<unsigned>(w_bytes_ll);
    static const int ck_env = getenv("SG_CONV_CK") ? atoi(getenv("SG_CONV_CK")) : 0;
    const int ck = ck_env ? ck_env : 16;
    const int grid3 = static_cast<int>(units);      // one workgroup per unit
    if (bpu == 1 && ck == 16) gather_conv_v3_kernel<1, 16><<<grid3, 256, lds3, stream>>>(a, ib, wb);
    else if (bpu == 1) gather_conv_v3_kernel<1, 32><<<grid3, 256, lds3, stream>>>(a, ib, wb);
    else if (ck == 16) gather_conv_v3_kernel<2, 16><<<grid3, 256, lds3, stream>>>(a, ib, wb);
    else gather_conv_v3_kernel<2, 32><<<grid3, 256, lds3, stream>>>(a, ib, wb);
  } else
  switch (bpu) {
    case 1: launch_v2<1>(a, grid, lds, vec, stream); break;
    case 2: launch_v2<2>(a, grid, lds, vec, stream); break;
    case 3: launch_v2<3>(a, grid, lds, vec, stream); break;
    default: launch_v2<4>(a, grid, lds, vec, stream); break;
  }
  if (ksplit > 1) {
    const long long n4 = static_cast<long long>(M_out) * Cout / 4;
    conv_reduce_kernel<<<grid_for(n4, 256), 256, 0, stream>>>(
        reinterpret_cast<const float4 *>(ws), reinterpret_cast<const float4 *>(residual), ksplit, n4,
        reinterpret_cast<float4 *>(out));
  }
  return check_launch("sg_spconv_gather_conv_f32");
}


// dw_kio [K][Cin][Cout] must be zero-filled by the caller; `in` is the (already activated) input
// the forward conv gathered from.
int sg_spconv_wgrad_f32(const float *in, const float *g_out, const int32_t *nbr, int M_out, int K,
                        int Cin, int Cout, float *dw_kio, sg_stream_t stream_) {
  SG_REQUIRE(M_out >= 0 && K >= 1 && K <= kMaxK && Cin >= 1 && Cout >= 1,
             "sg_spconv_wgrad_f32: bad arguments");
  if (M_out == 0) return SG_OK;
  dim3 grid((M_out + kWgradRows - 1) / kWgradRows, K);
  conv_wgrad_kernel<<<grid, 256, 0, as_stream(stream_)>>>(in, g_out, nbr, M_out, K, Cin, Cout, dw_kio);
  return check_launch("sg_spconv_wgrad_f32");
}

}  // extern "C"
