// spconv_conv.hip -- sparse convolution as an output-stationary implicit GEMM on fp32 MFMA.
// Replaces the conv kernels of the un-vendored spconv 2.1 library for SubMConv3d /
// SparseConv3d(k2,s2) / SparseInverseConv3d (reference call sites: softgroup/model/softgroup.py:61,
// softgroup/model/blocks.py:57-70,101-119).  One operator serves all three: it only sees a gather
// table nbr[M_out, K] (spconv_rulebook.hip) and packed weights.
//
//   out[j,:] = post( residual[j,:] + sum_k in[nbr[j,k],:] . W[k] ),   post = relu(x*s + b) | id
//
// `in` is already activated: the eval-mode BatchNorm1d+ReLU in front of every conv of the U-Net is
// either the `post` epilogue of the conv that produced `in` or one elementwise pass (sg_bn_relu_f32).
// An earlier version applied BN+ReLU to the gathered rows inside the matrix loop; per-wave phase
// tracing showed the kernel was bound by instruction ISSUE, not by MFMA or memory (113 instructions
// per 8 MFMAs, the SIMD issues ~1 per 4-5 cycles), so everything that is not load / MFMA moved out
// of the loop: absent neighbours are buffer loads past the end (hardware returns 0, no select),
// weights are packed so a lane's 8 B operands are one 32-B read, addresses are loop invariants.
//
// Weight layout "k8": [K][ceil(Cin/8)][Cout][8]  (zero padded): the 8 reduction steps a lane feeds
// to 8 consecutive MFMAs are contiguous, a wave reads 2 x 1 KB fully coalesced per slice.
//
// MI355X mapping
//   * v_mfma_f32_32x32x2_f32 (exact fp32 = fmaf chain, 64 FLOP/clk/SIMD).  Tile = 32 output rows in
//     neighbour-mask-sorted order x one 32-column block; the tile's 27-bit mask says which kernel
//     offsets exist at all.  The reduction index of a 16-channel slice is permuted so that lane
//     (h, i) owns channels c0+8h .. c0+8h+7 of gathered row i: A = one 32-B read of that row.
//   * main kernel (gather_conv_persistent_kernel, Cin % 16 == 0): persistent workgroups, see there.
//   * general kernel (gather_conv_tile_kernel): any Cin, plan optional, wave-private tiles.
//   * tiny layers (deep U-Net levels: 18..800 rows) split the kernel offsets over several units;
//     partial sums go to a workspace and are reduced in a fixed order (deterministic).
// HBM traffic per layer ~ P*Cin*4 (gathered rows) + M*Cout*4 (stores) + index tables, i.e. the
// gather/scatter bytes B_gs of SURVEY 8(d); the weights (<= 8 MB) are served from L2/MALL.
#include <stdlib.h>

#include <atomic>
#include <cstdio>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace sg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// ---- fp32 products on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate):
// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even; the
// subtractions are exact, the three parts carry 24+ significant bits, |x - (h+m+l)| <= 2^-26 |x|).
// a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh) + O(2^-24 |ab|): six bf16 products, each
// exact in the MFMA, accumulated in fp32, smallest terms first.  Weights are split once when they
// are packed; gathered activations are split in registers.
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f2 v = {x[2 * j], x[2 * j + 1]};
    const bf16x2 hh = __builtin_convertvector(v, bf16x2);
    const f2 r = v - __builtin_convertvector(hh, f2);
    const bf16x2 mm = __builtin_convertvector(r, bf16x2);
    const f2 r2 = r - __builtin_convertvector(mm, f2);
    const bf16x2 ll = __builtin_convertvector(r2, bf16x2);
    h[2 * j] = hh[0]; h[2 * j + 1] = hh[1];
    m[2 * j] = mm[0]; m[2 * j + 1] = mm[1];
    l[2 * j] = ll[0]; l[2 * j + 1] = ll[1];
  }
}
// h alone: the operand of the "bf16 operands" arithmetic (round to nearest even, as torch's .bfloat16())
__device__ __forceinline__ bf16x8 round_bf16(const float (&x)[8]) {
  bf16x8 h;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f2 v = {x[2 * j], x[2 * j + 1]};
    const bf16x2 hh = __builtin_convertvector(v, bf16x2);
    h[2 * j] = hh[0]; h[2 * j + 1] = hh[1];
  }
  return h;
}
__device__ __forceinline__ uint16_t bf16_rne(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  return static_cast<uint16_t>((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f32(uint16_t b) {
  return __builtin_bit_cast(float, static_cast<uint32_t>(b) << 16);
}

// Timing-only what-if builds (tools/build_alt.sh <name> -DSG_WHATIF=<bits>; wrong numbers, never the
// product): 1 = one weight plane loaded instead of three, 2 = no operand split conversions, 4 = one MFMA
// per slice instead of six, 8 = no LDS transpose of the gathered lines, 16 = linear rows instead of the
// gather table.  profiles/r06_conv_whatif.txt is the table they produced.
#ifndef SG_WHATIF
#define SG_WHATIF 0
#endif

constexpr int kTileRows = 32;
constexpr int kWavesPerWg = 4;
constexpr int kCk = 16;        // channels per pipeline slice (8 per half-wave)
constexpr int kMaxK = 27;
constexpr int kTicketStride = 32;        // unsigned words between two ticket counters (128 B)
constexpr unsigned kOob = 0x80000000u;   // byte offset past every (< 2 GB) buffer: loads return 0

// LDS-DMA piece: every active lane moves `16` / `4` bytes from its own global address to
// LDS[lds_dst + lane * size].  Written as asm on purpose: while a compiler-visible global_load_lds
// is (possibly) in flight hipcc turns every partial s_waitcnt vmcnt(N) of ordinary loads into
// vmcnt(0), which would serialise the operand ring of the matrix loop; an asm statement is outside
// its bookkeeping, and an untracked OLDER operation only ever makes a counted wait longer, never
// wrong.  The issuing wave waits for the data itself (s_waitcnt vmcnt(0), then the barrier).
// M0 = LDS base of the transfer; saved and restored inside the statement.
__device__ __forceinline__ void lds_dma_b128(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void lds_dma_b32(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct ConvArgs {
  const float *in;
  const int32_t *nbr;
  const float *w;            // k8 layout
  const float *post_scale;   // [Cout] or null
  const float *post_shift;
  const float *residual;     // [M_out][Cout] or null
  const float *act_scale;    // second output: out_act = relu(out * act_scale + act_shift), or null
  const float *act_shift;
  float *out_act;            // [M_out][Cout]
  const int32_t *order;      // [num_tiles*32] or null
  const uint32_t *tile_mask;
  const int32_t *nbr_tiles;  // [num_tiles][32][K] gather rows in plan order, or null
  float *out;                // [M_out][Cout], or partials [ksplit][M_out][Cout]
  int M_out, K, Cin, Cout;
  int col_units;             // units along Cout
  int blocks_per_unit;       // 32-col blocks per unit (general kernel only)
  int ksplit;                // slices of the kernel-offset range
  int k_per_split;
  int num_units;
  unsigned magic_upt, magic_cu, magic_nsl;   // reciprocals of units_per_tile, col_units, n_slices
  int out16, res16;            // SPLIT == 2 kernels: outputs / the residual are bf16 rows (else fp32)
  int dyn_rounds;              // with `queue`: static rounds before the hand-out starts (1 or 2)
  unsigned *queue;             // persistent kernel: 8 zeroed ticket counters of this launch (one per
                               // XCD, 128 B apart), or null = static hand-out
  unsigned long long *trace;   // developer tracing only (SG_CONV_TRACE): [units][4 waves][8] stamps
  // persistent kernel, ksplit > 1, in-launch combine (else null): zeroed arrival counters of THIS
  // launch, one per (tile, column unit), and the real output (`out` then holds the partial sums)
  unsigned *done;
  float *out_final;
};

// ---------------------------------------------------------------------------------------------
// General kernel: one wave = (tile, up to NBW column blocks, offset range).  No barriers after the
// prologue; operands of slice t+1 are loaded while slice t's MFMAs issue.
// ---------------------------------------------------------------------------------------------
template <int NBW, bool VEC>
__global__ void __launch_bounds__(256) gather_conv_tile_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int32_t *nbr_lds = reinterpret_cast<int32_t *>(smem_raw) + wave * kTileRows * kMaxK;
  const int cin_pad = (p.Cin + kCk - 1) / kCk * kCk;
  const int c8 = (p.Cin + 7) / 8;

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
  __syncthreads();  // nbr_lds visible (only barrier of the kernel)
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
  // Column of this unit's blocks; surplus blocks (n >= nbw) re-read the last real one so that every
  // load below is unconditional (hipcc branches around predicated loads).
  int coff[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n) coff[n] = min(col + min(n, nbw - 1) * 32, p.Cout - 1);

  // loads of slice (k, s): A = 8 consecutive channels of the gathered row, B = 8 x NBW weights
  auto load_raw = [&](int k, int s, float (&a)[8], float (&b)[NBW][8]) {
    const int c = s * kCk + ahalf * 8;
    const int src = nbr_lds[arow * kMaxK + k];
    const bool present = src >= 0;
    const float *row = p.in + static_cast<unsigned>((present ? src : 0) * p.Cin + c);
    if (VEC) {
      const float4 v0 = *reinterpret_cast<const float4 *>(row);
      const float4 v1 = *reinterpret_cast<const float4 *>(row + 4);
      a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
      a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = present ? a[j] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = row[max(0, min(j, p.Cin - 1 - c))];
        a[j] = (present && c + j < p.Cin) ? v : 0.f;
      }
    }
    const int blk = min(c >> 3, c8 - 1);       // padded slices re-read the last block (a == 0)
    const unsigned wbase = static_cast<unsigned>((k * c8 + blk) * p.Cout) * 8u;
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      const float4 w0 = *reinterpret_cast<const float4 *>(p.w + wbase + coff[n] * 8);
      const float4 w1 = *reinterpret_cast<const float4 *>(p.w + wbase + coff[n] * 8 + 4);
      b[n][0] = w0.x; b[n][1] = w0.y; b[n][2] = w0.z; b[n][3] = w0.w;
      b[n][4] = w1.x; b[n][5] = w1.y; b[n][6] = w1.z; b[n][7] = w1.w;
    }
  };

  float a_cur[8], a_nxt[8];
  float b_cur[NBW][8], b_nxt[NBW][8];

  // flattened iteration space: (offset k in mask) x (slice s)
  int k = mask ? __builtin_ctz(mask) : -1;
  int s = 0;
  if (k >= 0) load_raw(k, 0, a_nxt, b_nxt);
  while (k >= 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a_cur[j] = a_nxt[j];
#pragma unroll
      for (int n = 0; n < NBW; ++n) b_cur[n][j] = b_nxt[n][j];
    }
    int k2 = k, s2 = s + 1;
    if (s2 == n_slices) {
      s2 = 0;
      const uint32_t rest = mask & ~((2u << k) - 1u);
      k2 = rest ? __builtin_ctz(rest) : -1;
    }
    if (k2 >= 0) load_raw(k2, s2, a_nxt, b_nxt);
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
  const bool final_out = p.ksplit == 1;
  float *out = p.out + (final_out ? 0 : static_cast<long long>(ks) * p.M_out * p.Cout);
  const bool add_res = p.residual != nullptr && final_out;
  const bool post = p.post_scale != nullptr && final_out;
  const bool act = p.out_act != nullptr && final_out;
  float ps[NBW], pb[NBW], as[NBW], ab[NBW];
#pragma unroll
  for (int n = 0; n < NBW; ++n) {
    ps[n] = post ? p.post_scale[coff[n]] : 1.f;
    pb[n] = post ? p.post_shift[coff[n]] : 0.f;
    as[n] = act ? p.act_scale[coff[n]] : 1.f;
    ab[n] = act ? p.act_shift[coff[n]] : 0.f;
  }
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
        if (post) v = fmaxf(fmaf(v, ps[n], pb[n]), 0.f);
        out[off + n * 32] = v;
        if (act) p.out_act[off + n * 32] = fmaxf(fmaf(v, as[n], ab[n]), 0.f);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Main kernel: persistent workgroups (Cin % 16 == 0, plan present).
//   * grid = resident workgroups only (every CU holds the same number of them: a smaller grid with
//     whole rounds was measured slower, CUs with one workgroup more fall behind);
//     workgroup b takes one unit per round of G from the heaviest-first unit list, rounds running
//     forward or reversed after the Thue-Morse word, so every workgroup gets one unit of every
//     weight band and the totals are even.
//   * unit = (32-row tile, one 32-column block, offset range); its (offset, slice) items are
//     split evenly over the 4 waves (imbalance <= 1 slice); partial sums meet in LDS and are added
//     in a fixed order.
//   * the matrix loop holds nothing but loads and MFMAs: 2 dwordx4 of the gathered row, 2 dwordx4
//     of packed weights, 8 MFMAs per slice; lane offsets are loop invariants, per-slice address
//     arithmetic is scalar; an absent neighbour is a load past the end of the buffer (returns 0).
//   * all the metadata of the NEXT unit (gather-table block, rows, mask) is fetched while the
//     current unit's matrix loop runs and is published to the other LDS buffer under the same
//     barrier as the partial sums; the first operand loads and the residual rows of the next unit
//     are requested before the current unit's stores are issued, so memory latency (1.5-2 us
//     under load) is exposed once per workgroup, not three times per unit.
//   * units whose column blocks share a tile run on the same XCD (unit u -> XCD u % 8), so the
//     second column block finds the gathered rows in that XCD's L2.
// ---------------------------------------------------------------------------------------------
// LDS block of a unit, filled by LDS-DMA (lane-linear images): the tile's gather block as it lies
// in memory (row-major, stride K), the 32 row ids, the tile mask
constexpr int kRowsAt = kTileRows * kMaxK, kMaskAt = kRowsAt + kTileRows;
constexpr int kMetaInts = kMaskAt + 4;

//   * WV waves per workgroup (2, 4, 8 or 16) share a unit: the unit's items are split WV ways.  More
//     waves = shorter units and more rounds of units per workgroup slot, which is what evens out
//     layers with few units (22 649 rows x 96 columns: 2 124 units for 1 280 four-wave slots = 1.7
//     rounds, workgroups ended between 42 k and 110 k ticks; the deep levels used to split their
//     offsets over several workgroups and add the partial sums in a second kernel).
//   * AT (with SPLIT, CK = 32): the gathered rows are fetched in FULL 128-byte lines -- 8 lanes per
//     row, 4 coalesced loads for the tile's 32 rows x 32 channels -- and transposed into the MFMA
//     fragment layout through a wave-private 4 KB LDS block (XOR-swizzled, conflict-free both ways).
//     The fragment-shaped alternative (every lane 32 B of its own row) touches 32 different lines
//     per load instruction, and it is the CU's vector-memory path, not the matrix pipe, that this
//     kernel waits on once the products are bf16 MFMAs.
//   * CHAIN (conv_chain_kernel, below): the layer is one step of a multi-layer launch -- its inputs were
//     written by OTHER workgroups of the SAME launch (possibly on another XCD), so every activation
//     access goes past the CU's L1: sc1 loads of the gathered rows and the residual, sc1 (write-through)
//     stores of the outputs.  The grid barrier between two steps then needs no fence at all
//     (cdna_hip_programming.md, publish/consume recipe R1).  Same instructions otherwise: same numbers.
//   * R16 (with SPLIT == 2 and AT): the INPUT rows are bf16 in memory (arithmetic 3: bf16 activations between
//     the layers of a frozen backbone under autocast, what spconv does under tools/train.py:47).  A 32-channel
//     item of a row is 64 bytes: 4 lanes per row, 2 coalesced loads for the tile's 32 rows, a 2 KB transpose
//     block, and the 16-byte chunks that come out of it ARE MFMA operands -- no conversion at all.  p.out16 /
//     p.res16 (run-time, SPLIT == 2 only) say whether outputs / the residual are bf16 rows; sums stay fp32,
//     results are rounded to nearest even when stored.
template <int CK, int DEPTH, int TRACE, int NBW, int SPLIT, int WV, int AT, int CHAIN, int R16 = 0>
__device__ __forceinline__ void conv_layer_body(const ConvArgs &p, unsigned in_bytes, unsigned w_bytes) {
  static_assert(!R16 || (SPLIT == 2 && AT), "bf16 rows: bf16-operand arithmetic, line-wise gather");
  constexpr int AUX = CHAIN ? 16 : 0;      // buffer-instruction aux bits of the activation accesses (16 = sc1)
  static_assert(!SPLIT || CK == 16 || AT, "split-precision path: 16-channel slices");
  static_assert(!AT || (SPLIT && CK == 32), "line-wise gather: split path, 32-channel items");
  // (Tried: reading the weights as fp32 -- 4 B instead of the 6 B of three bf16 planes -- and
  // splitting them in registers like the activations: a third fewer weight loads, but 2.46 instead
  // of 2.28 ms of conv time per scan; the extra VALU work costs more than the loads it saves.)
  static_assert(WV == 2 || WV == 4 || WV == 8 || WV == 16, "2, 4, 8 or 16 waves per workgroup");
  constexpr int RR = 16 / WV;          // accumulator registers (row groups) each wave finalises
  constexpr int WV_SHIFT = WV == 2 ? 1 : WV == 4 ? 2 : WV == 8 ? 3 : 4;
  constexpr int HC = CK / 2;   // channels per lane per slice (8 or 16)
  constexpr int NQ = HC / 4;   // dwordx4 loads per lane per operand per slice
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // wave id through readfirstlane: everything derived from it is provably wave-uniform, so
  // buffer-load scalar offsets stay in SGPRs (no waterfall loops)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int arow = lane & 31, ahalf = lane >> 5;
  float *red = reinterpret_cast<float *>(smem_raw);                  // [4 waves][NBW][16][64]
  int32_t *meta_lds = reinterpret_cast<int32_t *>(red + WV * NBW * 16 * 64);   // [2][kMetaInts]

  const int G = gridDim.x;
  const int units_per_tile = p.col_units * p.ksplit;
  const int num_units = p.num_units;
  // (R16: an item is a whole 128-byte line of a bf16 row = 64 channels; a last item of 32 channels when Cin % 64 == 32)
  const int n_slices = R16 ? (p.Cin + 63) / 64 : p.Cin / CK;
  const int c8 = p.Cin / 8;
  const int tileK = kTileRows * p.K;       // words of a tile's gather block (row-major, stride K)
  const int nbr_base = arow * p.K;
  const bool final_out = p.ksplit == 1;
  const bool add_res = p.residual != nullptr && final_out;
  const bool post = p.post_scale != nullptr && final_out;
  const bool act = p.out_act != nullptr && final_out;
  // In-launch combine of the offset-split partial sums (ksplit > 1 and p.done): every unit stores
  // its partial tile write-through, then its workgroup draws from the (tile, column unit) counter;
  // the workgroup that draws ksplit - 1 adds the partial tiles in the fixed order 0 .. ksplit-1,
  // applies the epilogue and writes the output -- the same numbers as conv_reduce_kernel, without
  // its launch.  Hand-off as cdna_hip_programming.md prescribes for a split-K reducer: sc1 stores,
  // every wave drains them, workgroup barrier, ONE relaxed agent-scope fetch_add, sc1 loads by the
  // reducer; correct wherever the units of a tile run.
  const bool combine = p.done != nullptr && !final_out;
  const int num_tiles = (p.M_out + kTileRows - 1) / kTileRows;
  const unsigned out_bytes = static_cast<unsigned>(p.M_out) * p.Cout * 4u;      // one fp32 tile set (partial sums)
  const bool out16 = SPLIT == 2 && p.out16 != 0, res16 = SPLIT == 2 && p.res16 != 0;
  const unsigned fin_bytes = out16 ? out_bytes >> 1 : out_bytes;      // the layer's real outputs
  const unsigned res_bytes = res16 ? out_bytes >> 1 : out_bytes;

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(add_res ? p.residual : p.in), 0, add_res ? res_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, final_out ? fin_bytes : out_bytes * static_cast<unsigned>(p.ksplit), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_act = __builtin_amdgcn_make_buffer_rsrc(
      act ? p.out_act : p.out, 0, act ? fin_bytes : 0u, 0x00020000);
  // the reducer's operands (zero-sized unless this launch combines)
  const bool c_res = combine && p.residual != nullptr, c_act = combine && p.out_act != nullptr;
  const __amdgpu_buffer_rsrc_t rs_cres = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(c_res ? p.residual : p.in), 0, c_res ? res_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cout = __builtin_amdgcn_make_buffer_rsrc(
      combine ? p.out_final : p.out, 0, combine ? fin_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cact = __builtin_amdgcn_make_buffer_rsrc(
      c_act ? p.out_act : p.out, 0, c_act ? fin_bytes : 0u, 0x00020000);
  // 16-bit rows: element offsets are the fp32 byte offsets halved (the out-of-range sentinel stays out of range),
  // a 32-column block is 64 bytes
  auto ld_row = [&](const __amdgpu_buffer_rsrc_t &rs, unsigned off, int n, bool h16) -> float {
    if constexpr (SPLIT == 2) {
      if (h16)
        return bf16_f32(static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rs, off >> 1, n * 64, AUX)));
    }
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, n * 128, AUX));
  };
  auto st_row = [&](float v, const __amdgpu_buffer_rsrc_t &rs, unsigned off, unsigned base, int n, bool h16) {
    if constexpr (SPLIT == 2) {
      if (h16) {
        __builtin_amdgcn_raw_buffer_store_b16(static_cast<short>(bf16_rne(v)), rs, off >> 1, base + n * 64, AUX);
        return;
      }
    }
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, base + n * 128, AUX);
  };

  // x / d for the few small wave-uniform divisors of the unit arithmetic: one s_mul_hi with a
  // host-made reciprocal (exact for x * d < 2^32; magic 0 means d == 1)
  auto udiv = [](unsigned x, unsigned magic) { return magic ? __umulhi(x, magic) : x; };

  // unit -> (tile, column block, offset range).  Full groups of 8 x units_per_tile units are laid
  // out so that all units of a tile have the same u % 8 (same XCD).
  const int group = 8 * units_per_tile;
  const int full = num_units / group * group;
  struct Unit { int tile, cu, ks; };
  auto decode = [&](int u) {
    Unit d;
    int sub;
    if (u < full) {
      const int blk = udiv(u >> 3, p.magic_upt);
      const int i = u - blk * group;
      d.tile = blk * 8 + (i & 7);
      sub = i >> 3;
    } else {
      d.tile = udiv(u, p.magic_upt);
      sub = u - d.tile * units_per_tile;
    }
    d.ks = udiv(sub, p.magic_cu);
    d.cu = sub - d.ks * p.col_units;
    return d;
  };

  // ---- metadata of a unit goes from memory straight into LDS (global_load_lds: no registers, no
  //      publish pass), issued by wave 0 alone; lanes past the tile's block sit the DMA out (the
  //      destination of a lane is base + lane * size, so the image stays dense).
  int32_t *ctl = meta_lds + 2 * kMetaInts;           // [2] unit id held by each metadata buffer
  const unsigned meta_lds_addr = __builtin_amdgcn_readfirstlane(
      static_cast<unsigned>(reinterpret_cast<uintptr_t>(meta_lds)));       // LDS byte offset
  auto dma_meta = [&](int tile, int buf) {
    const unsigned dst = meta_lds_addr + static_cast<unsigned>(buf * kMetaInts) * 4u;
    const int32_t *blk = p.nbr_tiles + static_cast<long long>(tile) * tileK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = q * 256 + lane * 4;
      if (e < tileK) lds_dma_b128(blk + e, dst + q * 1024);
    }
    if (lane < kTileRows) lds_dma_b32(p.order + tile * kTileRows + lane, dst + kRowsAt * 4);
    if (lane == 0) lds_dma_b32(p.tile_mask + tile, dst + kMaskAt * 4);
  };
  (void)num_tiles;

  // SPLIT: b[n][pl] = the 8 bf16 weights of plane pl (h, m, l) for this lane's channel block.
  // SPLIT == 2 ("bf16 operands", the autocast arithmetic): only the h plane = bf16(w) is read and the
  // gathered rows are rounded to bf16 once -- ONE MFMA per product instead of six, fp32 sums.
  constexpr int NPL = SPLIT == 2 ? 1 : 3;
  constexpr int NB_ = R16 ? 4 : AT ? 2 * NPL : SPLIT ? NPL : NQ;   // AT: two 16-channel sub-slices x the planes; R16: four
  struct Slice { f4 a[NQ]; f4 b[NBW][NB_]; int nsl; };      // nsl (R16): 16-channel sub-slices of the item (4, or 2 at the end of a row)
  const int plane_bytes = p.K * c8 * p.Cout * 16;          // one bf16 plane of the packed weights
  const int planes_at = 2 * plane_bytes;                   // they follow the fp32 copy
  // per-unit context (wave-uniform scalars + the lane's column)
  struct Ctx {
    const int32_t *meta;   // LDS block of the unit
    uint32_t wg_mask;
    int col, ks, k, s, kp, sp, rem;   // (k, s) first item, (kp, sp) last item requested
    int pair;              // tile * col_units + column unit (arrival counter of the in-launch combine)
    int v_w;
    float ps[NBW], pb[NBW], as[NBW], ab[NBW];
    unsigned o_off[RR];    // byte offsets of the output rows this lane stores (kOob: padding row)
    bool col_ok;
  };

  Slice S[DEPTH];   // operand ring: DEPTH-1 slices of loads in flight behind the one multiplied
  float resv[NBW][RR];

  // lane (h, i) owns channels s*CK + h*HC .. + HC-1 of gathered row i: HC*4 contiguous bytes of the
  // row (CK = 32: the half-wave pair reads the row's whole 128-B line in one slice) and the
  // matching HC/8 packed weight blocks of its column
  f4 *tr_lds = reinterpret_cast<f4 *>(ctl + 4) + wave * 256;       // AT: this wave's 32 x 128 B block
  const unsigned row_pitch = static_cast<unsigned>(p.Cin) * 4u, lane_chunk = static_cast<unsigned>(lane & 7) * 16u;
  auto load = [&](const Ctx &c, int k, int s, Slice &S) {
    if constexpr (R16) {
      // load q: row 8q + lane/8 of the tile, 16-byte chunk lane%8 of the item's 128-byte line (64 bf16 channels;
      // of a 32-channel last item the upper four chunks belong to the next row and are not used)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned src = static_cast<unsigned>(c.meta[(8 * q + (lane >> 3)) * p.K + k]);
        const unsigned v_a = __umul24(src, row_pitch >> 1) + lane_chunk;
        S.a[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_a, s * 128, AUX));
      }
      // weights: sub-slice sl = channel blocks 8s + 2sl + h of the bf16 plane (a 32-channel last item re-reads its
      // first two sub-slices for the unused ones: every load unconditional and in range)
      const int s_w = planes_at + (k * c8 + s * 8) * p.Cout * 16;
      const int nsl = s * 64 + 64 <= p.Cin ? 4 : 2;
      S.nsl = nsl;
#pragma unroll
      for (int n = 0; n < NBW; ++n)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
          S.b[n][sl] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rs_w, c.v_w + n * 512, s_w + (sl < nsl ? sl : sl - 2) * (2 * p.Cout * 16), 0));
      return;
    }
    if constexpr (AT) {
      // load q: row 8q + lane/8 of the tile, 16-byte chunk lane%8 of the item's 128-byte line
#pragma unroll
      // byte offset of the lane's 16-B chunk: row * (Cin * 4) + chunk * 16 as ONE 24-bit multiply-add
      // (rows < 2^24 - 1, checked by the launch).  An absent neighbour (-1) needs no select: its low 24
      // bits times the row pitch lie past the end of the buffer, and an out-of-range buffer load
      // returns 0 (the range check looks at this offset, not at the scalar slice offset added to it).
      for (int q = 0; q < 4; ++q) {
        unsigned src = static_cast<unsigned>(c.meta[(8 * q + (lane >> 3)) * p.K + k]);
        if (SG_WHATIF & 16) src = static_cast<unsigned>(c.pair * 32 + 8 * q + (lane >> 3)) % static_cast<unsigned>(p.M_out);
        const unsigned v_a = __umul24(src, row_pitch) + lane_chunk;
        S.a[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_a, s * 128, AUX));
      }
      // weights: sub-slice sl = channel blocks 4s + 2sl + h of the three bf16 planes
      const int s_w = planes_at + (k * c8 + s * 4) * p.Cout * 16;
#pragma unroll
      for (int n = 0; n < NBW; ++n)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            if ((SG_WHATIF & 1) && pl > 0) { S.b[n][sl * NPL + pl] = S.b[n][sl * NPL]; continue; }
            S.b[n][sl * NPL + pl] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                                             rs_w, c.v_w + n * 512,
                                                             s_w + sl * (2 * p.Cout * 16) + pl * plane_bytes, 0));
          }
      return;
    }
    const int src = c.meta[nbr_base + k];
    const unsigned v_a = src >= 0 ? static_cast<unsigned>(src * p.Cin + ahalf * HC) * 4u : kOob;
    const int s_a = s * (CK * 4);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      S.a[q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_a + q * 16, s_a, AUX));
    if constexpr (SPLIT) {
      // bf16 planes, layout [K][Cin/8][Cout][8]: lane (h, col) reads block (k, 2s + h) of its column
      const int s_w = planes_at + (k * c8 + s * 2) * p.Cout * 16;
#pragma unroll
      for (int n = 0; n < NBW; ++n)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          S.b[n][pl] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rs_w, c.v_w + n * 512, s_w + pl * plane_bytes, 0));
    } else {
      const int s_w = (k * c8 + s * (CK / 8)) * p.Cout * 32;
      // column block n of the unit: the same packed block 32 columns (1 KB) further on
#pragma unroll
      for (int n = 0; n < NBW; ++n)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          S.b[n][q] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 rs_w, c.v_w + n * 1024 + (q & 1) * 16,
                                                 s_w + (q >> 1) * (p.Cout * 32), 0));
    }
  };

  auto advance = [&](const Ctx &c, int &k, int &s) {
    if (++s == n_slices) {
      s = 0;
      const uint32_t rest = c.wg_mask & ~((2u << k) - 1u);
      k = rest ? __builtin_ctz(rest) : k;
    }
  };

  // split the unit's (offset, slice) items over the waves, request the first operand slices, the
  // residual rows and the epilogue constants.  Needs the unit's metadata in LDS.
  auto setup = [&](const Unit &d, int buf, Ctx &c) {
    c.ks = d.ks;
    c.pair = d.tile * p.col_units + d.cu;
    c.meta = meta_lds + buf * kMetaInts;
    uint32_t m = static_cast<uint32_t>(c.meta[kMaskAt]);
    if (!final_out) {
      const int k_lo = d.ks * p.k_per_split, k_hi = min(p.K, k_lo + p.k_per_split);
      m &= ((1u << k_hi) - 1u) & ~((1u << k_lo) - 1u);      // K <= 27
    }
    m = __builtin_amdgcn_readfirstlane(m);
    c.wg_mask = m;
    c.col = d.cu * (32 * NBW) + arow;
    c.col_ok = c.col < p.Cout;
    const int colc = min(c.col, p.Cout - 1);
    c.v_w = SPLIT ? (ahalf * p.Cout + colc) * 16 : (ahalf * (HC / 8) * p.Cout + colc) * 32;
    const int items = __builtin_popcount(m) * n_slices;
    const int per = (items + WV - 1) >> WV_SHIFT;
    const int begin = min(items, wave * per);
    c.rem = min(items, begin + per) - begin;
    int rank = udiv(begin, p.magic_nsl);
    c.s = begin - rank * n_slices;
    uint32_t mm = m;
    for (; rank > 0; --rank) mm &= mm - 1u;
    c.k = mm ? __builtin_ctz(mm) : 0;
    load(c, c.k, c.s, S[0]);
    c.kp = c.k;
    c.sp = c.s;
#pragma unroll
    for (int i = 1; i < DEPTH - 1; ++i) {
      if (i < c.rem) advance(c, c.kp, c.sp);
      load(c, c.kp, c.sp, S[i]);
    }
    // this wave finalises accumulator registers wave*RR .. wave*RR+RR-1; register r of lane half h
    // is tile row (r & 3) + 8 * (r >> 2) + 4 * h
    int row4[RR];
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const int reg = wave * RR + rr;
      row4[rr] = c.meta[kRowsAt + (reg & 3) + 8 * (reg >> 2) + 4 * ahalf];
    }
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const unsigned off = (row4[rr] >= 0 && c.col_ok)
                               ? static_cast<unsigned>(row4[rr] * p.Cout + c.col) * 4u : kOob;
      c.o_off[rr] = off;   // kept for the epilogue: the unit's LDS block is not read after its loop
#pragma unroll
      for (int n = 0; n < NBW; ++n)
        resv[n][rr] = ld_row(rs_res, off, n, res16);
    }
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
      c.ps[n] = c.pb[n] = c.as[n] = c.ab[n] = 0.f;
      if (post) {            // uniform
        c.ps[n] = p.post_scale[colc + 32 * n];
        c.pb[n] = p.post_shift[colc + 32 * n];
      }
      if (act) {
        c.as[n] = p.act_scale[colc + 32 * n];
        c.ab[n] = p.act_shift[colc + 32 * n];
      }
    }
  };

  f32x16 acc[NBW];
  // (Tried: a second accumulator for the second sub-slice of an item, i.e. two independent MFMA
  // chains per wave: 2.31 against 2.28 ms per scan, not kept.  A wave alone on its SIMD spends its
  // ~0.75 us per item on the ~200 instructions of the item -- gathers, LDS transpose, conversions,
  // MFMAs, address arithmetic -- at one issue per 4-5 cycles, not on any single dependency chain.)
  auto compute = [&](Slice &S) {
    if constexpr (R16) {
      // rows as loaded -> LDS (the fp32 line-wise layout: row r at r * 128 B, chunk c at position c ^ ((r >> 1) & 7));
      // lane (h, i) then reads chunk 2 sl + h of row i: its 8 bf16 channels of sub-slice sl, an MFMA operand as it is
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 8 * q + (lane >> 3);
        tr_lds[r * 8 + ((lane & 7) ^ ((r >> 1) & 7))] = S.a[q];
      }
      f4 fr[4];
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) fr[sl] = tr_lds[arow * 8 + ((sl * 2 + ahalf) ^ ((arow >> 1) & 7))];
      const int nsl = __builtin_amdgcn_readfirstlane(S.nsl);
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        if (sl >= 2 && nsl == 2) break;      // (uniform: the 32-channel last item of a row)
#pragma unroll
        for (int n = 0; n < NBW; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[sl]),
                                                           __builtin_bit_cast(bf16x8, S.b[n][sl]), acc[n], 0, 0, 0);
      }
      return;
    }
    if constexpr (AT) {
      // rows as loaded -> LDS (row r at r * 128 B, chunk c at position c ^ ((r >> 1) & 7)), then
      // lane (h, i) reads the two chunks of its 8 channels of row i for each sub-slice
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 8 * q + (lane >> 3);
        if (!(SG_WHATIF & 8)) tr_lds[r * 8 + ((lane & 7) ^ ((r >> 1) & 7))] = S.a[q];
      }
      f4 fr[2][2];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          fr[sl][j] = (SG_WHATIF & 8) ? S.a[sl * 2 + j] : tr_lds[arow * 8 + ((sl * 4 + ahalf * 2 + j) ^ ((arow >> 1) & 7))];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const float af[8] = {fr[sl][0][0], fr[sl][0][1], fr[sl][0][2], fr[sl][0][3],
                             fr[sl][1][0], fr[sl][1][1], fr[sl][1][2], fr[sl][1][3]};
        if constexpr (SPLIT == 2) {
          const bf16x8 ah = round_bf16(af);
#pragma unroll
          for (int n = 0; n < NBW; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, S.b[n][sl]), acc[n],
                                                             0, 0, 0);
          continue;
        }
        bf16x8 ah, am, al;
        if (SG_WHATIF & 2) {
          ah = __builtin_bit_cast(bf16x8, fr[sl][0]);
          am = __builtin_bit_cast(bf16x8, fr[sl][1]);
          al = ah;
        } else {
          split3(af, ah, am, al);
        }
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, S.b[n][sl * 3 + 0]);
          const bf16x8 bm = __builtin_bit_cast(bf16x8, S.b[n][sl * 3 + 1]);
          const bf16x8 bl = __builtin_bit_cast(bf16x8, S.b[n][sl * 3 + 2]);
          if (SG_WHATIF & 4) {
            typedef int i4 __attribute__((ext_vector_type(4)));
            const i4 ax = __builtin_bit_cast(i4, al) ^ __builtin_bit_cast(i4, am) ^ __builtin_bit_cast(i4, ah);
            const i4 bx = __builtin_bit_cast(i4, bh) ^ __builtin_bit_cast(i4, bm) ^ __builtin_bit_cast(i4, bl);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ax), __builtin_bit_cast(bf16x8, bx),
                                                             acc[n], 0, 0, 0);
            continue;
          }
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
        }
      }
      return;
    }
    if constexpr (SPLIT) {
      const float af[8] = {S.a[0][0], S.a[0][1], S.a[0][2], S.a[0][3], S.a[1][0], S.a[1][1], S.a[1][2], S.a[1][3]};
      if constexpr (SPLIT == 2) {
        const bf16x8 ah = round_bf16(af);
#pragma unroll
        for (int n = 0; n < NBW; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, S.b[n][0]), acc[n], 0,
                                                           0, 0);
        return;
      }
      bf16x8 ah, am, al;
      split3(af, ah, am, al);
#pragma unroll
      for (int n = 0; n < NBW; ++n) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, S.b[n][0]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, S.b[n][1]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, S.b[n][2]);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
      }
      return;
    }
    // every operand of the slice is in registers before the matrix block starts: one wait, then
    // back-to-back MFMAs with nothing in between (the column blocks alternate: independent chains)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      asm volatile("" : "+v"(S.a[q]));
#pragma unroll
      for (int n = 0; n < NBW; ++n) asm volatile("" : "+v"(S.b[n][q]));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NBW; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(S.a[q][j], S.b[n][q][j], acc[n], 0, 0, 0);
  };

  unsigned long long stamp[8];
  auto mark = [&](int i) {
    if constexpr (TRACE) stamp[i] = __builtin_readcyclecounter();
  };

  // ---- work distribution.  The unit list is heaviest-first.  Default: the static snake (round r
  //      forward or reversed after the Thue-Morse word) all the way.  Optional (p.queue != null,
  //      SG_CONV_STATIC=0; measured neutral, profiles/r03_conv_experiments.txt): rounds 0 and 1
  //      static, from round 2 on the units are HANDED OUT: one ticket counter per XCD (unit u always runs on XCD u % 8, so the column
  //      blocks of a tile share that XCD's L2), drawn by lane 0 of wave 0 right after its matrix
  //      loop -- two units ahead, so the ticket's round trip (and the LDS-DMA of the next unit's
  //      metadata, which needs the ticket) hides behind a whole unit; nothing ever waits for it.
  //      (An earlier attempt drew tickets inside the matrix loop: the returning atomic sat in the
  //      same in-order return queue as the operand loads and stalled them for microseconds.)
  //      With ~3 units per workgroup the late finishers used to define the kernel's span (workgroup
  //      end times spread 109-146 k ticks on the 64->64 x 77 k-row layer); now whoever is ahead
  //      takes the remaining (lightest) units.  Results do not depend on who computes a unit.
  const int xcd = blockIdx.x & 7;
  // p.dyn_rounds = rounds handed out statically before the tickets start: 2 (the snake's first two rounds) or 1
  // (only a workgroup's FIRST unit is static, the first ticket is drawn in the prologue: list scheduling over the
  // heaviest-first list; measured no faster, see launch_persistent_split)
  const int dyn0 = (p.dyn_rounds == 1 ? 1 : 2) * G;     // first handed-out unit (G is a multiple of 8 when > 8)
  const bool draws = p.queue != nullptr && dyn0 < num_units;     // anything to hand out at all?
  const bool draw_first = draws && p.dyn_rounds == 1;
  auto barrier = [] {                          // LDS-only rendezvous: must not drain the ticket atomic
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // ---- prologue of the workgroup: metadata of its first unit
  if (static_cast<int>(blockIdx.x) >= num_units) return;      // (grid rounded up to a multiple of 8)
  int round = 0, u = blockIdx.x, buf = 0;
  Unit du = decode(u);
  unsigned ticket = 0;                         // wave 0, lane 0: the last ticket drawn
  bool drawing = true;                         // wave 0: no out-of-range ticket seen yet
  if (wave == 0) {
    if (lane == 0 && draw_first)
      ticket = __hip_atomic_fetch_add(p.queue + xcd * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    dma_meta(du.tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  barrier();
  Ctx c;
  setup(du, 0, c);

  while (true) {
    mark(0);
    // wave 0: which unit comes after this one, and its metadata on the way into the other buffer
    int un_w0 = -1;
    if (wave == 0) {
      if ((round == 0 && !draw_first) || !draws) {      // (no queue: the static snake all the way, A/B knob)
        // (a reversed round mirrors the GROUP of 8 only: unit u stays on XCD u % 8 = blockIdx.x % 8,
        // which is what the tile plan's per-XCD ranges rely on; G is a multiple of 8 when > 8)
        const int r = round + 1;
        const int b = static_cast<int>(blockIdx.x);
        const int rev = (G & 7) == 0 ? G - 8 - (b & ~7) + (b & 7) : G - 1 - b;
        un_w0 = r * G + ((__builtin_popcount(r) & 1) ? rev : b);
      } else if (drawing) {
        const unsigned t = __builtin_amdgcn_readfirstlane(ticket);     // drawn one unit ago
        un_w0 = dyn0 + static_cast<int>(t) * 8 + xcd;
      }
      if (un_w0 >= num_units || un_w0 < 0) {
        un_w0 = -1;
        drawing = false;
      }
      if (un_w0 >= 0) dma_meta(decode(un_w0).tile, buf ^ 1);
      // (this slot and that buffer were last read two barriers ago; next read: after barrier B)
      if (lane == 0) ctl[buf ^ 1] = un_w0;
    }
#pragma unroll
    for (int n = 0; n < NBW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // ---- matrix loop over this wave's items.  S[0..DEPTH-2] already hold items 0..DEPTH-2
    //      (setup).  Past the end the last item is re-read and dropped, so every load is
    //      unconditional and the wait counts stay exact; whole groups of DEPTH items keep the loop
    //      single-exit (no register rotation), the tail is straight-line code.
    {
      int kp = c.kp, sp = c.sp, issued = DEPTH - 1;
      const int rem = c.rem;
      for (int g = rem / DEPTH; g > 0; --g) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
          int k2 = kp, s2 = sp;
          advance(c, k2, s2);
          const bool more = issued < rem;
          kp = more ? k2 : kp;
          sp = more ? s2 : sp;
          ++issued;
          load(c, kp, sp, S[(i + DEPTH - 1) % DEPTH]);
          __builtin_amdgcn_sched_barrier(0);   // the new slice's loads are in flight ...
          compute(S[i]);
          __builtin_amdgcn_sched_barrier(0);   // ... and are only consumed DEPTH-1 blocks later
        }
      }
      const int tail = rem % DEPTH;            // the ring is back at S[0]
#pragma unroll
      for (int i = 0; i < DEPTH - 1; ++i)
        if (i < tail) compute(S[i]);
    }
    if constexpr (TRACE) { asm volatile("" : "+v"(acc[0][0])); }
    mark(1);

    // wave 0: the metadata DMA (issued a whole matrix loop ago) has landed; draw the ticket for
    // the unit after next.  Its answer is first looked at one unit from now.
    if (wave == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // (drawn every unit of a launch that hands units out, also after the queue has run dry: a
      // draw nobody looks at is cheaper than a data-dependent branch around it)
      if (lane == 0 && draws)
        ticket = __hip_atomic_fetch_add(p.queue + xcd * kTicketStride, 1u, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- partial sums meet in LDS (the next unit's identity and metadata are there already)
    barrier();                           // the previous unit's epilogue has read `red`
    mark(2);
#pragma unroll
    for (int n = 0; n < NBW; ++n)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) red[((wave * NBW + n) * 16 + reg) * 64 + lane] = acc[n][reg];
    barrier();
    mark(3);
    const int un = __builtin_amdgcn_readfirstlane(ctl[buf ^ 1]);
    const bool has_next = un >= 0;
    const Unit dn = decode(has_next ? un : u);

    // ---- epilogue operands first (all LDS reads in flight together), then the next unit's first
    //      operand loads, then the stores: fixed-order sum w0+w1+w2+w3 (+ residual, post); each
    //      wave stores 4 row groups, padding rows go past the end of the buffer (dropped)
    float v[NBW][RR], va[NBW][RR];
    unsigned o_off[RR];
    {
      float part[NBW][RR][WV];
#pragma unroll
      for (int n = 0; n < NBW; ++n)
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) {
          const int reg = wave * RR + rr;
#pragma unroll
          for (int w = 0; w < WV; ++w) part[n][rr][w] = red[((w * NBW + n) * 16 + reg) * 64 + lane];
        }
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) {
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          float t = part[n][rr][0];          // fixed order: wave 0 + wave 1 + ... (deterministic)
#pragma unroll
          for (int w = 1; w < WV; ++w) t += part[n][rr][w];
          t += resv[n][rr];    // residual rows of THIS unit (requested by its setup, one unit ago)
          if (post) t = fmaxf(fmaf(t, c.ps[n], c.pb[n]), 0.f);
          v[n][rr] = t;
          va[n][rr] = fmaxf(fmaf(t, c.as[n], c.ab[n]), 0.f);
        }
        o_off[rr] = c.o_off[rr];
      }
    }
    const unsigned o_base = final_out ? 0u : static_cast<unsigned>(c.ks) * out_bytes;
    const int rem_done = c.rem;
    const int pair_done = c.pair, col_done = c.col;
    setup(dn, has_next ? (buf ^ 1) : buf, c);
    mark(4);
    if (combine) {      // uniform: partial tile, write-through (visible to whichever workgroup reduces)
#pragma unroll
      for (int rr = 0; rr < RR; ++rr)
#pragma unroll
        for (int n = 0; n < NBW; ++n)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[n][rr]), rs_out, o_off[rr],
                                                o_base + n * 128, 16);
    } else {
#pragma unroll
    for (int rr = 0; rr < RR; ++rr)
#pragma unroll
      for (int n = 0; n < NBW; ++n) st_row(v[n][rr], rs_out, o_off[rr], o_base, n, out16 && final_out);
    }
    if (act) {          // uniform; stores only (a zero-sized buffer would drop them anyway)
#pragma unroll
      for (int rr = 0; rr < RR; ++rr)
#pragma unroll
        for (int n = 0; n < NBW; ++n) st_row(va[n][rr], rs_act, o_off[rr], 0u, n, out16);
    }
    if (combine) {      // uniform
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's partial stores are out
      barrier();
      if (wave == 0 && lane == 0) {
        const unsigned arrived = __hip_atomic_fetch_add(p.done + pair_done, 1u, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
        ctl[3] = arrived == static_cast<unsigned>(p.ksplit - 1);
      }
      barrier();
      if (__builtin_amdgcn_readfirstlane(ctl[3]) != 0) {         // this workgroup reduces the tile
        const int colc = min(col_done, p.Cout - 1);
        float ps[NBW], pb[NBW], as[NBW], ab[NBW];
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
          ps[n] = pb[n] = as[n] = ab[n] = 0.f;
          if (p.post_scale) { ps[n] = p.post_scale[colc + 32 * n]; pb[n] = p.post_shift[colc + 32 * n]; }
          if (c_act) { as[n] = p.act_scale[colc + 32 * n]; ab[n] = p.act_shift[colc + 32 * n]; }
        }
        // partial tiles 0 .. ksplit-1 in that order (what conv_reduce_kernel does), four at a time in
        // flight; sc1 loads: served past this CU's L1 and this XCD's L2 lines of other XCDs' data
        auto part = [&](int ks, int rr, int n) {
          return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                               rs_out, o_off[rr], static_cast<unsigned>(ks) * out_bytes + n * 128, 16));
        };
        float t[NBW][RR];
#pragma unroll
        for (int rr = 0; rr < RR; ++rr)
#pragma unroll
          for (int n = 0; n < NBW; ++n) t[n][rr] = part(0, rr, n);
        int ks = 1;
        for (; ks + 3 < p.ksplit; ks += 4) {
          float q[4][NBW][RR];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rr = 0; rr < RR; ++rr)
#pragma unroll
              for (int n = 0; n < NBW; ++n) q[j][n][rr] = part(ks + j, rr, n);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rr = 0; rr < RR; ++rr)
#pragma unroll
              for (int n = 0; n < NBW; ++n) t[n][rr] += q[j][n][rr];
        }
        for (; ks < p.ksplit; ++ks) {
          float q[NBW][RR];
#pragma unroll
          for (int rr = 0; rr < RR; ++rr)
#pragma unroll
            for (int n = 0; n < NBW; ++n) q[n][rr] = part(ks, rr, n);
#pragma unroll
          for (int rr = 0; rr < RR; ++rr)
#pragma unroll
            for (int n = 0; n < NBW; ++n) t[n][rr] += q[n][rr];
        }
        float res[NBW][RR];     // (a zero-sized descriptor without a residual: + 0)
#pragma unroll
        for (int rr = 0; rr < RR; ++rr)
#pragma unroll
          for (int n = 0; n < NBW; ++n)
            res[n][rr] = ld_row(rs_cres, o_off[rr], n, res16);
#pragma unroll
        for (int rr = 0; rr < RR; ++rr)
#pragma unroll
          for (int n = 0; n < NBW; ++n) {
            float x = t[n][rr] + res[n][rr];
            if (p.post_scale) x = fmaxf(fmaf(x, ps[n], pb[n]), 0.f);
            st_row(x, rs_cout, o_off[rr], 0u, n, out16);
            // (a zero-sized descriptor without a second output: the store is dropped)
            st_row(fmaxf(fmaf(x, as[n], ab[n]), 0.f), rs_cact, o_off[rr], 0u, n, out16);
          }
      }
    }
    mark(5);
    if constexpr (TRACE) {
      if (lane == 0 && p.trace && wave < kWavesPerWg) {       // (the first four waves of a unit)
        unsigned long long *t = p.trace + (static_cast<long long>(u) * kWavesPerWg + wave) * 8;
        for (int i = 0; i < 6; ++i) t[i] = stamp[i];
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        t[6] = static_cast<unsigned long long>(blockIdx.x);
        t[7] = (static_cast<unsigned long long>(rem_done) << 48) | (static_cast<unsigned long long>(xcc & 0xf) << 32) | hw;
      }
    }
    if (!has_next) break;
    u = un;
    ++round;
    buf ^= 1;
  }
}

template <int CK, int DEPTH, int TRACE, int WPE, int NBW = 1, int SPLIT = 0, int WV = 4, int AT = 0, int R16 = 0>
__global__ void __launch_bounds__(64 * WV, WPE) gather_conv_persistent_kernel(ConvArgs p, unsigned in_bytes,
                                                                             unsigned w_bytes) {
  conv_layer_body<CK, DEPTH, TRACE, NBW, SPLIT, WV, AT, 0, R16>(p, in_bytes, w_bytes);
}

// ---------------------------------------------------------------------------------------------
// Multi-layer launch ("conv chain"): the layers of the deep U-Net levels -- a few thousand rows and
// fewer; 43 of the 65 conv launches of a backbone forward and the whole tiny U-Net, each 8-17 us of
// launch for ~5 us of work in flight (profiles/r05_conv_by_grid.txt, r06_conv_trace_base.txt) -- run
// as ONE persistent launch that walks a step list: conv layers (conv_layer_body, CHAIN form), the
// skip concat and the stray BatchNorm+ReLU, with a grid barrier between two steps.  Topology walked:
// softgroup/model/blocks.py:82-143 (unet_exec.hip records the steps in launch order).
//   * one decomposition for every layer of a chain: 32-channel items, line-wise gather, 8 waves per
//     unit, one 32-column block, offsets split over units exactly as the single launch splits them
//     (ksplit, in-launch combine).  The single-launch path uses the SAME decomposition for these
//     layers (t_chain.mode == 1), so a chain and the launches it replaces give the same bits.
//   * grid = one 512-thread workgroup per CU; co-residency is what the barrier needs, so chain
//     launches of one process never overlap each other (chain_launch orders them with events) and
//     the barrier's wait is bounded: on a timeout every workgroup leaves, the launch's results are
//     garbage and the next library call on that device reports it (g_chain_abort).
//   * the step list travels in the kernel arguments (<= 4 KB: kChainMax steps per launch; a longer
//     chain is cut into several launches -- a boundary costs what a barrier costs).
//   * visibility between steps: activations are stored write-through (sc1) and loaded past the L1
//     (sc1); every wave drains its stores before the workgroup arrives at the barrier; counters and
//     generation words are agent-scope atomics.  No fences (cdna_hip_programming.md, recipe R1).
// ---------------------------------------------------------------------------------------------
struct ChainStep {
  const float *in, *w, *post_scale, *post_shift, *residual, *act_scale, *act_shift;
  float *out_act;
  const int32_t *order;
  const uint32_t *tile_mask;
  const int32_t *nbr_tiles;
  float *out;
  unsigned *done;
  float *out_final;
  int M_out, K, Cin, Cout, col_units, ksplit, k_per_split, num_units;
  unsigned magic_upt, magic_cu, magic_nsl, in_bytes, w_bytes;
  int kind;        // 0 conv | 1 concat: out = [in | residual] (Cin | Cout channels), out_act = relu(out * act_scale + act_shift)
                   // | 2 BatchNorm + ReLU: out = relu(in * post_scale + post_shift), Cin channels
  int pad_[2];
};
constexpr int kChainMax = 22;
struct ChainArgs {
  ChainStep s[kChainMax];
  unsigned *bar;           // zeroed barrier block (kChainBarWords)
  unsigned *abort_host;    // pinned host word: set when a barrier timed out
  unsigned long long *trace;   // developer tracing (SG_CHAIN_TRACE): [step][workgroup][4] real-time stamps, or null
  int n;
};
static_assert(sizeof(ChainArgs) <= 4096, "the step list must fit the kernel-argument segment");
constexpr int kChainWV = 8;                      // waves per workgroup
constexpr int kChainBarWords = 18 * 32;          // 8 XCD counters, top counter, 8 generation words, abort: 128 B each
constexpr unsigned long long kChainTimeout = 200000000ull;     // 2 s of the 100 MHz real-time counter

// XCD-hierarchical grid barrier (MI355X_MICROARCH.md, row barrier-xcd): a workgroup arrives at the counter
// of ITS group (blockIdx % 8: the XCD under the default dispatch order; any placement is correct), the
// last of a group arrives at the top counter, the last of all publishes the epoch to every group's
// generation word; everyone polls its own group's word with relaxed loads and a sleep.
__device__ __forceinline__ bool chain_barrier(unsigned *bar, unsigned epoch, unsigned *abort_host, int *flag_lds,
                                              unsigned long long *stamp) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave: its write-through stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
    const unsigned x = blockIdx.x & 7, nx = (gridDim.x + 7u - x) >> 3;
    int ok = 1;
    const unsigned old = __hip_atomic_fetch_add(bar + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nx * epoch - 1u) {
      const unsigned groups = gridDim.x < 8u ? gridDim.x : 8u;
      const unsigned o2 = __hip_atomic_fetch_add(bar + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o2 == groups * epoch - 1u)
        for (unsigned j = 0; j < groups; ++j)
          __hip_atomic_store(bar + (9 + j) * 32, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (__hip_atomic_load(bar + (9 + x) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0u) {
        if (__hip_atomic_load(bar + 17 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > kChainTimeout) {
          __hip_atomic_store(bar + 17 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          ok = 0;
          break;
        }
      }
    }
    *flag_lds = ok;
    if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  return *flag_lds != 0;
}

// the elementwise steps of a chain (float4 granularity; activations past the L1 like the conv steps)
__device__ __forceinline__ void chain_elementwise(const ChainStep &s) {
  const int ca4 = s.Cin >> 2, cb4 = s.kind == 1 ? s.Cout >> 2 : 0, c4 = ca4 + cb4;
  const unsigned total = static_cast<unsigned>(s.M_out) * c4;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(s.in), 0, static_cast<unsigned>(s.M_out) * ca4 * 16u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(cb4 ? s.residual : s.in), 0, static_cast<unsigned>(s.M_out) * cb4 * 16u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(s.out, 0, total * 16u, 0x00020000);
  const __amdgpu_buffer_rsrc_t roa = __builtin_amdgcn_make_buffer_rsrc(
      s.out_act ? s.out_act : s.out, 0, s.out_act ? total * 16u : 0u, 0x00020000);
  const float *sc = s.kind == 1 ? s.act_scale : s.post_scale, *sh = s.kind == 1 ? s.act_shift : s.post_shift;
  for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const unsigned r = t / c4, c = t - r * c4;
    f4 v;
    if (c < static_cast<unsigned>(ca4))
      v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ra, (r * ca4 + c) * 16u, 0, 16));
    else
      v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rb, (r * cb4 + (c - ca4)) * 16u, 0, 16));
    f4 a = v;
    if (sc != nullptr) {
      const f4 k = *reinterpret_cast<const f4 *>(sc + 4 * c), h = *reinterpret_cast<const f4 *>(sh + 4 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = fmaxf(fmaf(v[j], k[j], h[j]), 0.f);
    }
    if (s.kind == 1) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ro, t * 16u, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, a), roa, t * 16u, 0, 16);
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, a), ro, t * 16u, 0, 16);
    }
  }
}

template <int SPLIT>
__global__ void __launch_bounds__(64 * kChainWV, 1) conv_chain_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // (the word conv_layer_body leaves unused in its control block: red | meta[2] | ctl[4] | transpose blocks)
  int *flag_lds = reinterpret_cast<int *>(smem_raw) + kChainWV * 16 * 64 + 2 * kMetaInts + 2;
#pragma nounroll
  for (int i = 0; i < a.n; ++i) {
    const ChainStep &s = a.s[i];
    unsigned long long *stamp = a.trace ? a.trace + (static_cast<size_t>(i) * gridDim.x + blockIdx.x) * 4 : nullptr;
    if (stamp && threadIdx.x == 0) stamp[0] = __builtin_amdgcn_s_memrealtime();
    if (s.kind == 0) {
      ConvArgs p;
      p.in = s.in; p.nbr = nullptr; p.w = s.w; p.post_scale = s.post_scale; p.post_shift = s.post_shift;
      p.residual = s.residual; p.act_scale = s.act_scale; p.act_shift = s.act_shift; p.out_act = s.out_act;
      p.order = s.order; p.tile_mask = s.tile_mask; p.nbr_tiles = s.nbr_tiles; p.out = s.out;
      p.M_out = s.M_out; p.K = s.K; p.Cin = s.Cin; p.Cout = s.Cout;
      p.col_units = s.col_units; p.blocks_per_unit = 1; p.ksplit = s.ksplit; p.k_per_split = s.k_per_split;
      p.num_units = s.num_units;
      p.magic_upt = s.magic_upt; p.magic_cu = s.magic_cu; p.magic_nsl = s.magic_nsl;
      p.queue = nullptr; p.dyn_rounds = 2; p.out16 = 0; p.res16 = 0; p.trace = nullptr; p.done = s.done; p.out_final = s.out_final;
      conv_layer_body<32, 2, 0, 1, SPLIT, kChainWV, 1, 1>(p, s.in_bytes, s.w_bytes);
    } else {
      chain_elementwise(s);
    }
    if (stamp && threadIdx.x == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();
    if (i + 1 < a.n && !chain_barrier(a.bar, static_cast<unsigned>(i + 1), a.abort_host, flag_lds, stamp)) return;
  }
}

// fixed-order reduction of the offset-split partial sums (+ residual, post)
__global__ void __launch_bounds__(256) conv_reduce_kernel(const float4 *__restrict__ partial,
                                                         const float4 *__restrict__ residual,
                                                         const float *__restrict__ post_scale,
                                                         const float *__restrict__ post_shift,
                                                         const float *__restrict__ act_scale,
                                                         const float *__restrict__ act_shift,
                                                         float4 *__restrict__ out_act,
                                                         int ksplit, long long n4, int cout4,
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
    if (post_scale) {
      const int c = static_cast<int>(t % cout4) * 4;
      a.x = fmaxf(fmaf(a.x, post_scale[c], post_shift[c]), 0.f);
      a.y = fmaxf(fmaf(a.y, post_scale[c + 1], post_shift[c + 1]), 0.f);
      a.z = fmaxf(fmaf(a.z, post_scale[c + 2], post_shift[c + 2]), 0.f);
      a.w = fmaxf(fmaf(a.w, post_scale[c + 3], post_shift[c + 3]), 0.f);
    }
    out[t] = a;
    if (out_act) {
      const int c = static_cast<int>(t % cout4) * 4;
      a.x = fmaxf(fmaf(a.x, act_scale[c], act_shift[c]), 0.f);
      a.y = fmaxf(fmaf(a.y, act_scale[c + 1], act_shift[c + 1]), 0.f);
      a.z = fmaxf(fmaf(a.z, act_scale[c + 2], act_shift[c + 2]), 0.f);
      a.w = fmaxf(fmaf(a.w, act_scale[c + 3], act_shift[c + 3]), 0.f);
      out_act[t] = a;
    }
  }
}

// Scalar path of the same operator for channel counts the MFMA tiling does not cover
// (Cout % 4 != 0).  One thread per (row, cout).
__global__ void __launch_bounds__(256) gather_conv_scalar_kernel(
    const float *__restrict__ in, const int32_t *__restrict__ nbr, int M_out, int K, int Cin,
    int Cout, const float *__restrict__ w_k8, const float *__restrict__ post_scale,
    const float *__restrict__ post_shift, const float *__restrict__ residual,
    const float *__restrict__ act_scale, const float *__restrict__ act_shift,
    float *__restrict__ out_act, float *__restrict__ out) {
  const int64_t total = static_cast<int64_t>(M_out) * Cout;
  const int c8 = (Cin + 7) / 8;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int j = static_cast<int>(t / Cout), co = static_cast<int>(t - static_cast<int64_t>(j) * Cout);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int s = nbr[static_cast<int64_t>(j) * K + k];
      if (s < 0) continue;
      const float *x = in + static_cast<int64_t>(s) * Cin;
      for (int ci = 0; ci < Cin; ++ci)
        acc = fmaf(x[ci], w_k8[((static_cast<int64_t>(k) * c8 + (ci >> 3)) * Cout + co) * 8 + (ci & 7)], acc);
    }
    if (residual) acc += residual[t];
    if (post_scale) acc = fmaxf(fmaf(acc, post_scale[co], post_shift[co]), 0.f);
    out[t] = acc;
    if (out_act) out_act[t] = fmaxf(fmaf(acc, act_scale[co], act_shift[co]), 0.f);
  }
}

// weight packing: src [Cout][K][Cin] (spconv "OKKKI", src_kio == 0) or [K][Cin][Cout]
// (src_kio == 1) -> [K][ceil(Cin/8)][Cout][8], zero padded.  src_kio == 2 / 3: the weights of the
// TRANSPOSED convolution (input gradient) straight from the forward layer's "OKKKI" tensor
// w[Cin][K][Cout] (this conv's Cin = that layer's Cout), 3 = kernel offsets mirrored (SubM)
__global__ void __launch_bounds__(256) pack_weight_kernel(const float *__restrict__ w, int cout, int K,
                                                         int cin, int src_kio,
                                                         float *__restrict__ out) {
  const int c8 = (cin + 7) / 8;
  const int64_t total = static_cast<int64_t>(K) * c8 * cout * 8;
  for (int64_t t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL) {
    const int j = static_cast<int>(t & 7);
    int64_t r = t >> 3;
    const int co = static_cast<int>(r % cout);
    r /= cout;
    const int blk = static_cast<int>(r % c8), k = static_cast<int>(r / c8);
    const int ci = blk * 8 + j;
    float v = 0.f;
    if (ci < cin) {
      if (src_kio == 0) v = w[(static_cast<int64_t>(co) * K + k) * cin + ci];
      else if (src_kio == 1) v = w[(static_cast<int64_t>(k) * cin + ci) * cout + co];
      else v = w[(static_cast<int64_t>(ci) * K + (src_kio == 3 ? K - 1 - k : k)) * cout + co];   // transposed conv
    }
    out[t] = v;
    // the same element as three bf16 planes behind the fp32 copy (split-precision MFMA path)
    uint16_t *planes = reinterpret_cast<uint16_t *>(out + total);
    const uint16_t h = bf16_rne(v);
    const float rest = v - bf16_f32(h);
    const uint16_t m = bf16_rne(rest);
    const uint16_t l = bf16_rne(rest - bf16_f32(m));
    planes[t] = h;
    planes[total + t] = m;
    planes[2 * total + t] = l;
  }
}

// many pack_weight calls in one launch (the training tape: every conv weight of the step, once for the forward
// layout and once for the input gradients' -- 171 launches of ~5 us each before): the jobs travel as kernel arguments
constexpr int kPackBatch = 64;
struct PackBatchArgs {
  const float *w[kPackBatch];
  float *out[kPackBatch];
  int cout[kPackBatch], K[kPackBatch], cin[kPackBatch], mode[kPackBatch];
  int block0[kPackBatch + 1];      // first workgroup of every job
  int n;
};
static_assert(sizeof(PackBatchArgs) <= 4096, "the batch travels as kernel arguments");
__global__ void __launch_bounds__(256) pack_weights_kernel(const PackBatchArgs a) {
  int j0 = 0, j1 = a.n;                                   // last job with block0 <= blockIdx.x
  while (j1 - j0 > 1) {
    const int mid = (j0 + j1) >> 1;
    if (a.block0[mid] <= static_cast<int>(blockIdx.x)) j0 = mid; else j1 = mid;
  }
  const float *__restrict__ w = a.w[j0];
  float *__restrict__ out = a.out[j0];
  const int cout = a.cout[j0], K = a.K[j0], cin = a.cin[j0], src_kio = a.mode[j0];
  const int c8 = (cin + 7) / 8;
  const int64_t total = static_cast<int64_t>(K) * c8 * cout * 8;
  const int64_t stride = static_cast<int64_t>(a.block0[j0 + 1] - a.block0[j0]) * 256;
  for (int64_t t = (blockIdx.x - a.block0[j0]) * 256LL + threadIdx.x; t < total; t += stride) {
    const int j = static_cast<int>(t & 7);
    int64_t r = t >> 3;
    const int co = static_cast<int>(r % cout);
    r /= cout;
    const int blk = static_cast<int>(r % c8), k = static_cast<int>(r / c8);
    const int ci = blk * 8 + j;
    float v = 0.f;
    if (ci < cin) {
      if (src_kio == 0) v = w[(static_cast<int64_t>(co) * K + k) * cin + ci];
      else if (src_kio == 1) v = w[(static_cast<int64_t>(k) * cin + ci) * cout + co];
      else v = w[(static_cast<int64_t>(ci) * K + (src_kio == 3 ? K - 1 - k : k)) * cout + co];
    }
    out[t] = v;
    uint16_t *planes = reinterpret_cast<uint16_t *>(out + total);
    const uint16_t h = bf16_rne(v);
    const float rest = v - bf16_f32(h);
    const uint16_t m = bf16_rne(rest);
    const uint16_t l = bf16_rne(rest - bf16_f32(m));
    planes[t] = h;
    planes[total + t] = m;
    planes[2 * total + t] = l;
  }
}

int spconv_pack_weights(const PackJob *jobs, int n, sg_stream_t stream) {
  for (int at = 0; at < n; at += kPackBatch) {
    PackBatchArgs a;
    a.n = n - at < kPackBatch ? n - at : kPackBatch;
    a.block0[0] = 0;
    for (int i = 0; i < a.n; ++i) {
      const PackJob &j = jobs[at + i];
      if (j.cout <= 0 || j.kvol <= 0 || j.cin <= 0 || j.mode < 0 || j.mode > 3 || j.w == nullptr || j.out == nullptr) {
        set_error("spconv_pack_weights: bad job %d", at + i);
        return SG_ERR_ARG;
      }
      a.w[i] = j.w; a.out[i] = j.out; a.cout[i] = j.cout; a.K[i] = j.kvol; a.cin[i] = j.cin; a.mode[i] = j.mode;
      const int64_t total = static_cast<int64_t>(j.kvol) * ((j.cin + 7) / 8) * j.cout * 8;
      const int64_t blocks = (total + 255) / 256;
      a.block0[i + 1] = a.block0[i] + static_cast<int>(blocks < 256 ? blocks : 256);
    }
    pack_weights_kernel<<<a.block0[a.n], 256, 0, as_stream(stream)>>>(a);
  }
  return check_launch("spconv_pack_weights");
}

template <int NBW>
static void launch_tile(const ConvArgs &a, int grid, size_t lds, bool vec, hipStream_t stream) {
  if (vec)
    gather_conv_tile_kernel<NBW, true><<<grid, 256, lds, stream>>>(a);
  else
    gather_conv_tile_kernel<NBW, false><<<grid, 256, lds, stream>>>(a);
}

// Optional timing of the conv launches with HIP events recorded inside the library, right around
// the kernel launches on the launch stream (bench.py's roofline measurement; off by default).
struct ConvProf {
  bool enabled = false;
  std::vector<hipEvent_t> pool;     // start/stop pairs, reused across sessions
  std::vector<int> dims;            // per call: M_out, K, Cin, Cout, num_in_rows
  size_t used = 0;
  hipEvent_t take() {
    if (used == pool.size()) {
      hipEvent_t e;
      hipEventCreate(&e);
      pool.push_back(e);
    }
    return pool[used++];
  }
};
static ConvProf g_conv_prof;

// arithmetic of the persistent kernel: -1 = from the environment (SG_CONV_SPLIT, default 1), 0 = fp32
// MFMA, 1 = split-precision bf16 MFMA (sg_spconv_set_arithmetic; tests compare the two in one process)
static std::atomic<int> g_arith_override{-1};
// arithmetic requested by the calling thread for the convs it launches (sg_unet_forward with
// sg_unet_desc.arithmetic set; common.h); -1 = none, the process-wide choice applies
thread_local int t_conv_arith = -1;
static std::atomic<int> g_combine_override{-1};   // in-launch combine of offset-split layers: -1 = SG_CONV_COMBINE (default 1)

// Ticket counters of the persistent kernel's unit hand-out: every launch gets its own zeroed block
// of 8 counters (one per XCD) out of a per-(device, stream) pool; the pool is cleared again, in
// stream order, when it has been used up.  Launches on one stream run in order, so a block is
// never shared by two kernels.  Every counter has a 128-byte line to itself: with the 8 counters
// of a launch in one line a draw took 13-34 k ticks (the line ping-pongs between the XCDs' L2s),
// with one line each 1.4 k (tools/micro/atomic_ticket.hip, profiles/r03_atomic_ticket.txt).
struct TicketPool {
  unsigned *dev = nullptr;
  size_t next = 0;
};
constexpr size_t kTicketBlockBytes = 8 * kTicketStride * sizeof(unsigned);
constexpr size_t kTicketBlocks = 1 << 12;     // x 1 KB = 4 MB per stream
static std::mutex g_ticket_mu;
static std::map<std::pair<int, hipStream_t>, TicketPool> g_ticket_pools;

// Arrival counters of the in-launch combine: every launch takes a fresh, zeroed range out of a
// per-(device, stream) pool; when the pool has been used up it is cleared again in stream order
// (launches of one stream do not overlap, so a range is never shared by two kernels, and a kernel
// that was aborted half way cannot leave counts behind for a later launch).
constexpr int kDoneCounters = 4096;                 // most (tile, column unit) pairs of one launch
constexpr size_t kDonePoolCounters = 1 << 20;       // x 4 B = 4 MB per stream
struct DonePool {
  unsigned *dev = nullptr;
  size_t next = 0;
};
static std::map<std::pair<int, hipStream_t>, DonePool> g_done_pools;

static unsigned *take_done(hipStream_t stream, size_t count) {
  int dev = 0;
  hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_ticket_mu);
  DonePool &dp = g_done_pools[{dev, stream}];
  if (dp.dev == nullptr) {
    if (hipMalloc(&dp.dev, kDonePoolCounters * sizeof(unsigned)) != hipSuccess) {
      dp.dev = nullptr;
      return nullptr;
    }
    dp.next = kDonePoolCounters;
  }
  if (dp.next + count > kDonePoolCounters) {
    hipMemsetAsync(dp.dev, 0, kDonePoolCounters * sizeof(unsigned), stream);
    dp.next = 0;
  }
  unsigned *r = dp.dev + dp.next;
  dp.next += count;
  return r;
}

// sg_stream_release: the caller's stream is idle and about to be destroyed
static void chain_release_stream(int dev, hipStream_t stream);
void conv_release_stream(int dev, hipStream_t stream) {
  chain_release_stream(dev, stream);
  std::lock_guard<std::mutex> lock(g_ticket_mu);
  auto d = g_done_pools.find({dev, stream});
  if (d != g_done_pools.end()) {
    if (d->second.dev) hipFree(d->second.dev);
    g_done_pools.erase(d);
  }
  auto t = g_ticket_pools.find({dev, stream});
  if (t != g_ticket_pools.end()) {
    if (t->second.dev) hipFree(t->second.dev);
    g_ticket_pools.erase(t);
  }
}

static unsigned *take_tickets(hipStream_t stream) {
  int dev = 0;
  hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_ticket_mu);
  TicketPool &tp = g_ticket_pools[{dev, stream}];
  if (tp.dev == nullptr) {
    if (hipMalloc(&tp.dev, kTicketBlocks * kTicketBlockBytes) != hipSuccess) return nullptr;
    tp.next = kTicketBlocks;
  }
  if (tp.next == kTicketBlocks) {
    hipMemsetAsync(tp.dev, 0, kTicketBlocks * kTicketBlockBytes, stream);
    tp.next = 0;
  }
  return tp.dev + (8 * kTicketStride) * tp.next++;
}


// ---------------------------------------------------------------------------------------------
// Conv chains, host side.  While a chain is open on the calling thread (conv_chain_begin .. _end, the
// U-Net executor around its deep levels) sg_spconv_gather_conv_f32 RECORDS the layers the chain kernel
// can take (mode 2) instead of launching them, or -- chains switched off (SG_CONV_CHAIN=0 /
// sg_spconv_set_chain(0)) -- launches them one by one with the chain's decomposition (mode 1): same
// numbers either way.  A layer the chain cannot take (no line-wise gather, no in-launch combine)
// flushes what was recorded and is launched as usual.
// ---------------------------------------------------------------------------------------------
struct ChainRec {
  int mode = 0;            // 0 closed | 1 open, single launches | 2 open, recording
  bool b16 = false;        // arithmetic of the recorded conv steps (one kernel instantiation per launch)
  bool arith_set = false;  // ... once a conv step has been recorded
  int conv_steps = 0;      // conv layers with K > 1 among the recorded steps (profile bookkeeping)
  hipStream_t stream = nullptr;
  ChainArgs args;
};
static thread_local ChainRec t_chain;
static std::atomic<int> g_chain_override{-1};       // -1 = SG_CONV_CHAIN (default 0)
static std::atomic<long long> g_chain_launches{0}, g_chain_steps{0};
static std::mutex g_chain_mu;                                         // order of the chain launches of a device
static std::map<int, hipEvent_t> g_chain_last;                        // device -> event behind its latest chain launch
static std::map<std::pair<int, hipStream_t>, hipEvent_t> g_chain_events;
static std::map<int, unsigned *> g_chain_abort;                       // device -> pinned word (set by a timed-out barrier)

static void chain_release_stream(int dev, hipStream_t stream) {      // (the stream is idle)
  std::lock_guard<std::mutex> lock(g_chain_mu);
  auto it = g_chain_events.find({dev, stream});
  if (it == g_chain_events.end()) return;
  auto last = g_chain_last.find(dev);
  if (last != g_chain_last.end() && last->second == it->second) last->second = nullptr;
  if (it->second) hipEventDestroy(it->second);
  g_chain_events.erase(it);
}

// Off by default (round 6 measurements, profiles/r06_conv_chain.txt): a chain step costs what a launch costs --
// the barrier is cheap (1.1-1.4 us after the last arrival + 0.2-0.6 us of store drain), but a small layer's
// 8-15 us are the chain of dependent memory round trips INSIDE the layer (metadata -> first operands -> items
// -> partial sums out -> arrival counter -> partial sums in), which a chain walks exactly like a launch does;
// and its co-resident workgroups keep 8 waves x 156 VGPRs on every CU for 0.8 ms per scan, which costs the
// scans running next to it more than the ~45 launches it saves them (5 scans in flight: 2.84 -> 3.17 ms/scan).
static bool chain_enabled() {
  static const int env = getenv("SG_CONV_CHAIN") ? atoi(getenv("SG_CONV_CHAIN")) : 0;
  const int ov = g_chain_override.load(std::memory_order_relaxed);
  return (ov >= 0 ? ov : env) != 0;
}

// has a barrier of an earlier chain launch on this device timed out?  (reported once)
int conv_chain_check_abort(const char *who) {
  int dev = 0;
  hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_chain_mu);
  auto it = g_chain_abort.find(dev);
  if (it != g_chain_abort.end() && *static_cast<volatile unsigned *>(it->second) != 0u) {
    *static_cast<volatile unsigned *>(it->second) = 0u;
    set_error("%s: a grid barrier of an earlier multi-layer conv launch timed out (its workgroups were not "
              "co-resident: another process on this GPU?); that forward's results are invalid. "
              "SG_CONV_CHAIN=0 launches the layers one by one", who);
    return SG_ERR_LAUNCH;
  }
  return SG_OK;
}

static int chain_flush() {
  ChainRec &c = t_chain;
  if (c.args.n == 0) return SG_OK;
  int dev = 0;
  hipGetDevice(&dev);
  static int num_cu = 0;
  static std::once_flag once;
  std::call_once(once, [&] {
    hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (num_cu <= 0) num_cu = 256;
  });
  c.args.bar = take_done(c.stream, kChainBarWords);
  if (c.args.bar == nullptr) {
    set_error("conv chain: no barrier block");
    return SG_ERR_LAUNCH;
  }
  const size_t lds = static_cast<size_t>(kChainWV) * 16 * 64 * sizeof(float) + 2 * kMetaInts * sizeof(int32_t) + 16 +
                     static_cast<size_t>(kChainWV) * 4096;
  // (developer knobs: SG_CHAIN_GRID = workgroups of a chain launch, at most one per CU; SG_CHAIN_SERIAL=0 lets chain
  // launches of different streams overlap -- only safe while all of them together fit the device)
  static const int grid_env = getenv("SG_CHAIN_GRID") ? atoi(getenv("SG_CHAIN_GRID")) : 0;
  static const bool serial = !(getenv("SG_CHAIN_SERIAL") && atoi(getenv("SG_CHAIN_SERIAL")) == 0);
  int grid = num_cu >= 8 ? num_cu - num_cu % 8 : num_cu;
  if (grid_env >= 8 && grid_env < grid) grid = grid_env - grid_env % 8;
  const bool prof = g_conv_prof.enabled && c.conv_steps > 0;
  {
    std::lock_guard<std::mutex> lock(g_chain_mu);
    unsigned *&ab = g_chain_abort[dev];
    if (ab == nullptr) {
      if (hipHostMalloc(reinterpret_cast<void **>(&ab), 64, hipHostMallocMapped) != hipSuccess) {
        ab = nullptr;
        set_error("conv chain: pinned allocation failed");
        return SG_ERR_LAUNCH;
      }
      *ab = 0u;
    }
    c.args.abort_host = ab;
    hipEvent_t &mine = g_chain_events[{dev, c.stream}];
    if (mine == nullptr && hipEventCreateWithFlags(&mine, hipEventDisableTiming) != hipSuccess) {
      mine = nullptr;
      set_error("conv chain: event creation failed");
      return SG_ERR_LAUNCH;
    }
    hipEvent_t &last = g_chain_last[dev];
    if (serial && last != nullptr && last != mine) hipStreamWaitEvent(c.stream, last, 0);    // one chain at a time per device
    if (prof) {
      hipEventRecord(g_conv_prof.take(), c.stream);
      for (int v : {-c.conv_steps, 0, 0, 0, 0}) g_conv_prof.dims.push_back(v);
    }
    static const char *trace_env = getenv("SG_CHAIN_TRACE");      // developer tool: per-step, per-workgroup stamps
    c.args.trace = nullptr;
    const size_t trace_bytes = static_cast<size_t>(c.args.n) * grid * 4 * sizeof(unsigned long long);
    if (trace_env) {
      hipMalloc(&c.args.trace, trace_bytes);
      hipMemsetAsync(c.args.trace, 0, trace_bytes, c.stream);
    }
    if (c.b16)
      conv_chain_kernel<2><<<grid, 64 * kChainWV, lds, c.stream>>>(c.args);
    else
      conv_chain_kernel<1><<<grid, 64 * kChainWV, lds, c.stream>>>(c.args);
    if (prof) hipEventRecord(g_conv_prof.take(), c.stream);
    hipEventRecord(mine, c.stream);
    last = mine;
    if (trace_env && c.args.trace) {
      hipStreamSynchronize(c.stream);
      std::vector<unsigned long long> h(trace_bytes / 8);
      hipMemcpy(h.data(), c.args.trace, trace_bytes, hipMemcpyDeviceToHost);
      hipFree(c.args.trace);
      if (FILE *f = fopen(trace_env, "ab")) {
        long long hdr[4] = {c.args.n, grid, 0, 0};
        fwrite(hdr, 8, 4, f);
        for (int i = 0; i < c.args.n; ++i) {
          const ChainStep &st = c.args.s[i];
          long long d[8] = {st.kind, st.M_out, st.K, st.Cin, st.Cout, st.num_units, st.ksplit, st.col_units};
          fwrite(d, 8, 8, f);
        }
        fwrite(h.data(), 8, h.size(), f);
        fclose(f);
      }
    }
  }
  g_chain_launches.fetch_add(1, std::memory_order_relaxed);
  g_chain_steps.fetch_add(c.args.n, std::memory_order_relaxed);
  c.args.n = 0;
  c.conv_steps = 0;
  c.arith_set = false;
  return check_launch("conv chain");
}

static int chain_push(const ChainStep &st, bool b16, bool counts) {
  ChainRec &c = t_chain;
  const bool conv = st.kind == 0;      // (an elementwise step runs under either instantiation)
  if (c.args.n > 0 && (c.args.n == kChainMax || (conv && c.arith_set && c.b16 != b16))) {
    const int rc = chain_flush();
    if (rc != SG_OK) return rc;
  }
  if (conv) {
    c.b16 = b16;
    c.arith_set = true;
  }
  c.args.s[c.args.n++] = st;
  if (counts) ++c.conv_steps;
  return SG_OK;
}

void conv_chain_begin(hipStream_t stream) {
  ChainRec &c = t_chain;
  c.mode = chain_enabled() ? 2 : 1;
  c.stream = stream;
  c.args.n = 0;
  c.conv_steps = 0;
  c.arith_set = false;
}
bool conv_chain_recording() { return t_chain.mode == 2; }
int conv_chain_end() {
  const int rc = t_chain.mode == 2 ? chain_flush() : SG_OK;
  t_chain.mode = 0;
  t_chain.args.n = 0;
  return rc;
}
void conv_chain_abort() {
  t_chain.mode = 0;
  t_chain.args.n = 0;
  t_chain.conv_steps = 0;
}
// the elementwise steps between the layers of a chain (call only while conv_chain_recording())
int conv_chain_concat(const float *a, const float *b, int64_t rows, int ca, int cb, const float *scale,
                      const float *shift, float *out, float *out_act) {
  ChainStep st = {};
  st.kind = 1;
  st.in = a; st.residual = b; st.act_scale = scale; st.act_shift = shift; st.out = out; st.out_act = out_act;
  st.M_out = static_cast<int>(rows); st.Cin = ca; st.Cout = cb;
  return chain_push(st, t_chain.b16, false);
}
int conv_chain_bn_relu(const float *x, const float *scale, const float *shift, int64_t rows, int c, float *out) {
  ChainStep st = {};
  st.kind = 2;
  st.in = x; st.post_scale = scale; st.post_shift = shift; st.out = out;
  st.M_out = static_cast<int>(rows); st.Cin = c; st.Cout = 0;
  return chain_push(st, t_chain.b16, false);
}

// ---------------------------------------------------------------------------------------------
// Launch of the split-precision persistent kernel.  Decomposition: unit = (32-row tile, one or
// two 32-column blocks, ALL offsets); the unit's items are split over the WV waves of one
// workgroup and meet in LDS -- no partial sums through memory, no reduce kernel.  (NBW, WV) is the
// first of (2,2) (1,2) -- >= 1.5 units per workgroup slot, K * Cin / 32 <= 108 -- then (2,4) (1,4)
// (1,8) (1,16) -- >= 2.5 units per slot, else the last: large layers keep the gathered rows in
// registers for two column blocks and split their few items per tile over two waves only, small
// layers put more waves on fewer, shorter units.
// ---------------------------------------------------------------------------------------------
typedef void (*PersistentFn)(ConvArgs, unsigned, unsigned);
struct SplitVariant {
  PersistentFn fn, fn_trace;
  PersistentFn fn_b16;      // the same decomposition with bf16 operands (one MFMA per product)
  PersistentFn fn_r16;      // ... and bf16 input rows (line-wise variants only, else null)
  int nbw, wv, ck, at;
  size_t lds;
  int occ, occ_b16, occ_r16;      // resident workgroups per CU
};
#ifndef SG_B16_DEPTH
#define SG_B16_DEPTH 2
#endif
#ifndef SG_B16_WPE_PLUS
#define SG_B16_WPE_PLUS 0
#endif
template <int CK, int WPE, int NBW, int WV, int AT>
constexpr PersistentFn r16_variant() {
  if constexpr (AT) return gather_conv_persistent_kernel<CK, SG_B16_DEPTH, 0, WPE + SG_B16_WPE_PLUS, NBW, 2, WV, 1, 1>;
  else return nullptr;
}
#define SG_SPLIT_VARIANT(CK, DEPTH, WPE, NBW, WV, AT)                                            \
  {gather_conv_persistent_kernel<CK, DEPTH, 0, WPE, NBW, 1, WV, AT>,                              \
   gather_conv_persistent_kernel<CK, DEPTH, 1, WPE, NBW, 1, WV, AT>,                              \
   gather_conv_persistent_kernel<CK, SG_B16_DEPTH, 0, WPE + SG_B16_WPE_PLUS, NBW, 2, WV, AT>,     \
   r16_variant<CK, WPE, NBW, WV, AT>(), NBW, WV, CK, AT, 0, 0, 0, 0}
constexpr int kSplitVariants = 9;
static SplitVariant g_split_variants[kSplitVariants] = {
    SG_SPLIT_VARIANT(32, 2, 2, 2, 2, 1),  SG_SPLIT_VARIANT(32, 2, 3, 1, 2, 1),
    SG_SPLIT_VARIANT(32, 2, 2, 2, 4, 1),  SG_SPLIT_VARIANT(32, 2, 3, 1, 4, 1), SG_SPLIT_VARIANT(32, 2, 1, 1, 8, 1),
    SG_SPLIT_VARIANT(16, 2, 3, 2, 4, 0),  SG_SPLIT_VARIANT(16, 2, 4, 1, 4, 0), SG_SPLIT_VARIANT(16, 2, 2, 1, 8, 0),
    SG_SPLIT_VARIANT(16, 2, 1, 1, 16, 0),
};
// (Tried for the offset-split layers, whose waves run alone on their SIMDs at 0.75 us per item: a
// 3-deep operand ring -- 2.31 instead of 2.28 ms of conv time per scan, not kept; what such a wave
// waits for is its own chain of dependent MFMAs and conversions, not its loads.)

static int launch_persistent_split(ConvArgs a, int num_tiles, long long in_bytes, long long w_bytes,
                                   hipStream_t stream, bool b16, bool in16) {
  static int num_cu = 0;
  static std::once_flag once;
  std::call_once(once, [] {
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (num_cu <= 0) num_cu = 256;
    auto prepare = [](SplitVariant &v) {
      v.lds = static_cast<size_t>(v.wv) * v.nbw * 16 * 64 * sizeof(float) + 2 * kMetaInts * sizeof(int32_t) + 16 +
              (v.at ? static_cast<size_t>(v.wv) * 4096 : 0);
      int o = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, v.fn, 64 * v.wv, v.lds);
      v.occ = o < 1 ? 1 : o;
      o = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, v.fn_b16, 64 * v.wv, v.lds);
      v.occ_b16 = o < 1 ? 1 : o;
      o = 0;
      if (v.fn_r16) hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, v.fn_r16, 64 * v.wv, v.lds);
      v.occ_r16 = o < 1 ? 1 : o;
    };
    for (SplitVariant &v : g_split_variants) prepare(v);
  });
  static const int nbw_env = getenv("SG_CONV_NBW") ? atoi(getenv("SG_CONV_NBW")) : 2;          // developer knobs
  static const int wv_env = getenv("SG_CONV_WV") ? atoi(getenv("SG_CONV_WV")) : 0;             // force 4 / 8 / 16
  static const float min_rounds = getenv("SG_CONV_ROUNDS") ? atof(getenv("SG_CONV_ROUNDS")) : 2.5f;
  static const int at_env = getenv("SG_CONV_AT") ? atoi(getenv("SG_CONV_AT")) : 1;             // line-wise gather
  const int NB = (a.Cout + 31) / 32;
  // (the line-wise gather forms its addresses with a 24-bit multiply: input rows < 2^24 - 1)
  // and an absent neighbour (-1 -> 0xffffff * row pitch, wrapped to 32 bits) must land past the end
  // of the buffer: row pitches that are not multiples of 256 B (Cin = 96, 160, 224) wrap to just
  // below 2^31, which is inside an input of a few million rows -- those take the fragment-shaped gather
  const unsigned long long row_bytes = (in16 ? 2ull : 4ull) * a.Cin;
  const unsigned absent_at = static_cast<unsigned>((0xFFFFFFull * row_bytes) & 0xFFFFFFFFull);
  const int use_at = ((at_env != 0 || in16) && a.Cin % 32 == 0 && in_bytes / static_cast<long long>(row_bytes) < (1LL << 24) - 1 &&
                      static_cast<long long>(absent_at) >= in_bytes) ? 1 : 0;
  if (in16 && (!use_at || !b16)) {
    set_error("sg_unet_forward(bf16 rows): a layer with %d input channels / %lld input bytes cannot take the line-wise bf16 gather",
              a.Cin, in_bytes);
    return SG_ERR_ARG;
  }
  int pick = -1;
  // tiny layers arrive with their offsets split over several units (ksplit > 1, partial sums to the
  // workspace, conv_reduce_kernel afterwards): measured faster than 16 waves on very few units
  // (141 rows x 192 columns: 22.6 us against 42.7 us).  They run as 4-wave, one-block units.
  // Two-wave units.  On the big levels a mask-sorted tile holds 3-5 of the 27 offsets: split four
  // ways that is ~1 item per wave inside a unit whose fixed cost (two barriers, LDS reduction, set-up
  // of the next unit) is 4 k ticks (profiles/r03_conv_trace_final.txt).  Two waves per unit halve
  // the waves that pay that cost and double the units in flight per CU: 32->32 x 124 k rows 30.8 ->
  // 25.6 us, 96->96 x 22.6 k 66 -> 52 us, 64->64 x 76.8 k 57 -> 53 us; layers with too few units to
  // fill the two-wave slots 1.5 times over keep four or eight waves (128->128 x 4.7 k rows: 35 -> 60 us
  // if forced).  SG_CONV_W2_MAX = largest K * Cin / 32 (items of a full tile) they are used for.
  static const int w2_max = getenv("SG_CONV_W2_MAX") ? atoi(getenv("SG_CONV_W2_MAX")) : 108;
  static const int w2_nbw = getenv("SG_CONV_W2_NBW") ? atoi(getenv("SG_CONV_W2_NBW")) : 2;      // A/B knob
  static const float w2_rounds = getenv("SG_CONV_W2_ROUNDS") ? atof(getenv("SG_CONV_W2_ROUNDS")) : 1.5f;
  const bool use_w2 = wv_env == 2 || (wv_env == 0 && a.K * (a.Cin / 32) <= w2_max);
  // (One wave per unit was measured too: the heaviest tiles -- 27 offsets in a single wave -- then
  // set the span of the big layers, 32->32 x 124 k rows 28.5 -> 32.1 us, 64->64 x 77 k 52 -> 69 us; on
  // the K = 8 strided / inverse convs alone it changes nothing, 2.22 ms per scan either way.)
  for (int i = 0; i < kSplitVariants; ++i) {
    const SplitVariant &v = g_split_variants[i];
    if (v.at != use_at) continue;
    if (v.wv == 2 && (!use_w2 || a.ksplit > 1 || v.nbw > w2_nbw)) continue;
    if (a.ksplit > 1) {
      // (8 waves per unit on these layers: 2.482 against 2.492 ms per backbone forward, no difference)
      if (v.nbw == 1 && v.wv == 4) { pick = i; break; }
      continue;
    }
    pick = i;                         // (the last candidate of the group stays if none qualifies)
    if (v.nbw == 2 && (a.Cout % 64 != 0 || nbw_env < 2)) continue;
    if (wv_env && v.wv != wv_env) continue;
    const long long units = static_cast<long long>(num_tiles) * (NB / v.nbw);
    if (wv_env || units >= static_cast<long long>((v.wv == 2 ? w2_rounds : min_rounds) * num_cu * (in16 ? v.occ_r16 : b16 ? v.occ_b16 : v.occ))) {
      pick = i;
      break;
    }
  }
  // a layer of an open chain: the chain's decomposition (8 waves per unit, one column block, line-wise
  // gather), whether it is recorded or launched by itself
  const bool chain_layer = t_chain.mode != 0 && use_at != 0;
  if (chain_layer) {
    for (int i = 0; i < kSplitVariants; ++i)
      if (g_split_variants[i].at == 1 && g_split_variants[i].nbw == 1 && g_split_variants[i].wv == kChainWV) pick = i;
  }
  const SplitVariant &v = g_split_variants[pick];
  a.col_units = NB / v.nbw;
  a.blocks_per_unit = v.nbw;
  const long long units = static_cast<long long>(num_tiles) * a.col_units * a.ksplit;
  a.num_units = static_cast<int>(units);
  auto magic = [](unsigned d) { return d <= 1 ? 0u : static_cast<unsigned>((1ULL << 32) / d) + 1u; };
  a.magic_upt = magic(static_cast<unsigned>(a.col_units * a.ksplit));
  a.magic_cu = magic(static_cast<unsigned>(a.col_units));
  a.magic_nsl = magic(static_cast<unsigned>(in16 ? (a.Cin + 63) / 64 : a.Cin / v.ck));
  if (t_chain.mode == 2) {
    if (chain_layer && !in16 && !a.out16 && !a.res16 && (a.ksplit == 1 || a.done != nullptr) && stream == t_chain.stream) {
      ChainStep st = {};
      st.kind = 0;
      st.in = a.in; st.w = a.w; st.post_scale = a.post_scale; st.post_shift = a.post_shift; st.residual = a.residual;
      st.act_scale = a.act_scale; st.act_shift = a.act_shift; st.out_act = a.out_act; st.order = a.order;
      st.tile_mask = a.tile_mask; st.nbr_tiles = a.nbr_tiles; st.out = a.out; st.done = a.done; st.out_final = a.out_final;
      st.M_out = a.M_out; st.K = a.K; st.Cin = a.Cin; st.Cout = a.Cout; st.col_units = a.col_units;
      st.ksplit = a.ksplit; st.k_per_split = a.k_per_split; st.num_units = a.num_units;
      st.magic_upt = a.magic_upt; st.magic_cu = a.magic_cu; st.magic_nsl = a.magic_nsl;
      st.in_bytes = static_cast<unsigned>(in_bytes); st.w_bytes = static_cast<unsigned>(w_bytes);
      return chain_push(st, b16, a.K > 1);
    }
    const int rc = chain_flush();      // not a layer the chain kernel takes: what was recorded runs first
    if (rc != SG_OK) return rc;
  }
  static const bool dyn_env = getenv("SG_CONV_STATIC") && atoi(getenv("SG_CONV_STATIC")) == 0;   // hand-out A/B
  // (SG_CONV_DYN_ROUNDS=1 with SG_CONV_STATIC=0: tickets from a workgroup's second unit on.  Measured slower
  //  than the static snake on every big layer -- 64->64 52.7 against 51.0 us, 32->32 27.8 against 24.8,
  //  profiles/r06_conv_dyn_ab.txt: the spread of workgroup end times is contention inside a CU, not assignment)
  static const int dyn_rounds_env = getenv("SG_CONV_DYN_ROUNDS") ? atoi(getenv("SG_CONV_DYN_ROUNDS")) : 2;
  a.queue = dyn_env ? take_tickets(stream) : nullptr;
  a.dyn_rounds = dyn_rounds_env == 1 ? 1 : 2;
  // grid: the resident workgroups, a multiple of 8 (unit u runs on XCD u % 8 in every round).  A layer
  // with fewer units than that gets one workgroup per unit, rounded UP to the next multiple of 8 --
  // the surplus workgroups leave at once.  (Rounded down, as until round 5, 98 units ran on 96
  // workgroups and two of them took a second unit: the 18-row layers spent 21.5 k instead of 10.5 k
  // ticks, the 141-row layers 24.5 k instead of 15 k -- profiles/r06_conv_trace_base.txt.)
  long long g = static_cast<long long>(num_cu) * (in16 ? v.occ_r16 : b16 ? v.occ_b16 : v.occ);
  if (g >= 8) g -= g % 8;
  if (units < g) g = units >= 8 ? (units + 7) / 8 * 8 : units;
  const unsigned ib = static_cast<unsigned>(in_bytes), wb = static_cast<unsigned>(w_bytes);
  if (in16) {
    v.fn_r16<<<static_cast<int>(g), 64 * v.wv, v.lds, stream>>>(a, ib, wb);
    return check_launch("sg_spconv_gather_conv(bf16 rows)");
  }
  if (b16) {
    v.fn_b16<<<static_cast<int>(g), 64 * v.wv, v.lds, stream>>>(a, ib, wb);
    return check_launch("sg_spconv_gather_conv_f32(bf16 operands)");
  }
  static const char *trace_env = getenv("SG_CONV_TRACE");     // developer tool: per-wave phase stamps
  if (trace_env) {
    const size_t nb = static_cast<size_t>(units) * kWavesPerWg * 8 * sizeof(unsigned long long);
    unsigned long long *dbuf = nullptr;
    hipMalloc(&dbuf, nb);
    hipMemsetAsync(dbuf, 0, nb, stream);
    a.trace = dbuf;
    v.fn_trace<<<static_cast<int>(g), 64 * v.wv, v.lds, stream>>>(a, ib, wb);
    hipStreamSynchronize(stream);
    std::vector<unsigned long long> h(nb / 8);
    hipMemcpy(h.data(), dbuf, nb, hipMemcpyDeviceToHost);
    hipFree(dbuf);
    if (FILE *f = fopen(trace_env, "ab")) {
      long long hdr[8] = {a.M_out, a.K, a.Cin, a.Cout, units, a.col_units, v.wv, g};
      fwrite(hdr, 8, 8, f);
      fwrite(h.data(), 8, h.size(), f);
      fclose(f);
    }
  } else {
    v.fn<<<static_cast<int>(g), 64 * v.wv, v.lds, stream>>>(a, ib, wb);
  }
  return check_launch("sg_spconv_gather_conv_f32(split)");
}


}  // namespace sg

using namespace sg;

extern "C" {

int sg_spconv_set_arithmetic(int mode) {
  SG_REQUIRE(mode >= -1 && mode <= 2, "sg_spconv_set_arithmetic: mode must be -1, 0, 1 or 2");
  g_arith_override = mode;
  return SG_OK;
}

int sg_spconv_set_combine(int mode) {
  SG_REQUIRE(mode >= -1 && mode <= 1, "sg_spconv_set_combine: mode must be -1, 0 or 1");
  g_combine_override = mode;
  return SG_OK;
}

int sg_spconv_set_chain(int mode) {
  SG_REQUIRE(mode >= -1 && mode <= 1, "sg_spconv_set_chain: mode must be -1, 0 or 1");
  g_chain_override = mode;
  return SG_OK;
}

int sg_spconv_chain_stats(int64_t *launches, int64_t *steps) {
  if (launches) *launches = g_chain_launches.load(std::memory_order_relaxed);
  if (steps) *steps = g_chain_steps.load(std::memory_order_relaxed);
  return SG_OK;
}

int sg_spconv_profile(int enable) {
  g_conv_prof.enabled = enable != 0;
  if (enable) {
    g_conv_prof.used = 0;
    g_conv_prof.dims.clear();
  }
  return SG_OK;
}

int sg_spconv_profile_detail(float *ms, int32_t *dims, int cap, int *calls) {
  const int n = static_cast<int>(g_conv_prof.used / 2);
  if (calls) *calls = n;
  for (int i = 0; i < n && i < cap; ++i) {
    if (hipEventSynchronize(g_conv_prof.pool[2 * i + 1]) != hipSuccess) {
      set_error("sg_spconv_profile_detail: event synchronisation failed");
      return SG_ERR_LAUNCH;
    }
    float t = 0.f;
    hipEventElapsedTime(&t, g_conv_prof.pool[2 * i], g_conv_prof.pool[2 * i + 1]);
    if (ms) ms[i] = t;
    if (dims)
      for (int j = 0; j < 5; ++j) dims[5 * i + j] = g_conv_prof.dims[5 * static_cast<size_t>(i) + j];
  }
  return SG_OK;
}

int sg_spconv_profile_read(double *total_ms, int *launches) {
  double sum = 0.0;
  for (size_t i = 0; i + 1 < g_conv_prof.used; i += 2) {
    if (hipEventSynchronize(g_conv_prof.pool[i + 1]) != hipSuccess) {
      set_error("sg_spconv_profile_read: event synchronisation failed");
      return SG_ERR_LAUNCH;
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, g_conv_prof.pool[i], g_conv_prof.pool[i + 1]);
    sum += ms;
  }
  if (total_ms) *total_ms = sum;
  if (launches) *launches = static_cast<int>(g_conv_prof.used / 2);
  return SG_OK;
}

// fp32 "k8" copy followed by its three bf16 split planes (h, m, l), 2 bytes per element each
static size_t packed_k8_elems(int kvol, int cin, int cout) {
  return static_cast<size_t>(kvol) * ((cin + 7) / 8) * cout * 8;
}
size_t sg_spconv_packed_weight_elems(int kvol, int cin, int cout) {
  const size_t n = packed_k8_elems(kvol, cin, cout);
  return n + 3 * n / 2;
}

int sg_spconv_pack_weight(const float *w, int cout, int kvol, int cin, int src_is_kio, float *w_k8,
                          sg_stream_t stream) {
  SG_REQUIRE(cout > 0 && kvol > 0 && cin > 0 && src_is_kio >= 0 && src_is_kio <= 3,
             "sg_spconv_pack_weight: bad arguments");
  const int64_t total = static_cast<int64_t>(packed_k8_elems(kvol, cin, cout));
  pack_weight_kernel<<<grid_for(total, 256), 256, 0, as_stream(stream)>>>(w, cout, kvol, cin,
                                                                         src_is_kio, w_k8);
  return check_launch("sg_spconv_pack_weight");
}

// workspace for the offset-split path: ksplit_max * M_out * Cout floats
// workspace for the offset-split path: ksplit_max * M_out * Cout floats
size_t sg_spconv_conv_workspace_bytes(int M_out, int Cout) {
  const int num_tiles = (M_out + kTileRows - 1) / kTileRows;
  if (num_tiles * ((Cout + 31) / 32) >= 1024) return 256;   // big layers never split
  return static_cast<size_t>(kMaxK) * M_out * Cout * sizeof(float) + 256;
}

// in16 / out16 / res16: `in` / `out` + `out_act` / `residual` are bf16 rows (the executor's arithmetic 3; the
// pointers are typed float for the fp32 case only).  Needs the bf16-operand arithmetic and, for in16, the
// line-wise gather (Cin % 32 == 0); an offset-split layer combines in the launch.
static int gather_conv_impl(const float *in, int num_in_rows, const int32_t *nbr, int M_out,
                            int K, int Cin, int Cout, const float *w_k8, const float *post_scale,
                            const float *post_shift, const float *residual, const float *act_scale,
                            const float *act_shift, float *out_act, const int32_t *order,
                            const uint32_t *tile_mask, const int32_t *nbr_tiles, float *out,
                            void *ws, size_t ws_bytes, sg_stream_t stream_, int in16, int out16, int res16) {
  SG_REQUIRE(M_out >= 0 && K >= 1 && K <= kMaxK && Cin >= 1 && Cout >= 1,
             "sg_spconv_gather_conv_f32: bad arguments (M_out=%d K=%d Cin=%d Cout=%d)", M_out, K,
             Cin, Cout);
  SG_REQUIRE((post_scale == nullptr) == (post_shift == nullptr),
             "sg_spconv_gather_conv_f32: post_scale and post_shift must come together");
  SG_REQUIRE((out_act == nullptr) == (act_scale == nullptr) && (out_act == nullptr) == (act_shift == nullptr),
             "sg_spconv_gather_conv_f32: act_scale, act_shift and out_act must come together");
  if (M_out == 0) return SG_OK;
  hipStream_t stream = as_stream(stream_);
  struct ProfScope {      // start event now, stop event when the call has enqueued its last kernel
    hipStream_t st;
    bool on;
    ProfScope(hipStream_t s, bool count, int m, int k, int ci, int co, int rows_in)
        : st(s), on(g_conv_prof.enabled && count && t_chain.mode != 2) {      // (a recorded layer is timed with its chain launch)
      if (on) {
        hipEventRecord(g_conv_prof.take(), st);
        for (int v : {m, k, ci, co, rows_in}) g_conv_prof.dims.push_back(v);
      }
    }
    ~ProfScope() {
      if (on) hipEventRecord(g_conv_prof.take(), st);
    }
  } prof_scope(stream, K > 1, M_out, K, Cin, Cout, num_in_rows);     // the 1x1 identity-branch convs are not part of the conv roofline
  if (Cout % 4 != 0) {
    if (t_chain.mode == 2) {
      const int rc = chain_flush();
      if (rc != SG_OK) return rc;
    }
    gather_conv_scalar_kernel<<<grid_for(static_cast<int64_t>(M_out) * Cout, 256, 256 * 32), 256, 0,
                                stream>>>(in, nbr, M_out, K, Cin, Cout, w_k8, post_scale, post_shift,
                                          residual, act_scale, act_shift, out_act, out);
    return check_launch("sg_spconv_gather_conv_f32(scalar)");
  }
  const int NB = (Cout + 31) / 32;
  const int num_tiles = (M_out + kTileRows - 1) / kTileRows;
  const long long in_bytes_ll = static_cast<long long>(num_in_rows) * Cin * (in16 ? 2 : 4);
  const long long w_bytes_ll = static_cast<long long>(sg_spconv_packed_weight_elems(K, Cin, Cout)) * 4;
  const bool persistent = Cin % 16 == 0 && in_bytes_ll < (1LL << 31) && num_in_rows > 0 &&
                          static_cast<long long>(M_out) * Cout * 4 * kMaxK < (1LL << 32) &&
                          static_cast<long long>(num_tiles) * kTileRows * K * 4 < (1LL << 31) &&
                          order != nullptr && tile_mask != nullptr && nbr_tiles != nullptr &&
                          reinterpret_cast<uintptr_t>(nbr_tiles) % 16 == 0;      // 16-byte LDS-DMA pieces
  // split-precision path (fp32 products as six bf16 MFMAs, see split3): the default for every layer
  // the persistent kernel takes with Cin >= SG_CONV_SPLIT_MIN_CIN (SG_CONV_SPLIT=0: the fp32-MFMA
  // kernel, kept for A/B)
  static const int split_env = getenv("SG_CONV_SPLIT") ? atoi(getenv("SG_CONV_SPLIT")) : 1;
  static const int split_min_cin = getenv("SG_CONV_SPLIT_MIN_CIN") ? atoi(getenv("SG_CONV_SPLIT_MIN_CIN")) : 16;
  const int arith_ov = t_conv_arith >= 0 ? t_conv_arith : g_arith_override.load(std::memory_order_relaxed);
  const int split_on = arith_ov >= 0 ? arith_ov : split_env;
  const bool split = persistent && split_on != 0 && Cin >= split_min_cin;
  // ---- decomposition: aim at >= ~2048 waves; the general kernel widens its column block while
  //      that still fills the chip, the persistent kernel always works on 32-column blocks
  static const int target_env = getenv("SG_CONV_TARGET") ? atoi(getenv("SG_CONV_TARGET")) : 0;   // developer knob
  // (the split-precision kernel's units are shorter: it wants half as many offset splits, measured)
  const int target = target_env > 0 ? target_env : split ? 1024 : 2048;
  const int waves_per_unit = persistent ? kWavesPerWg : 1;
  int bpu = 1;
  if (!persistent) {
    bpu = NB < 4 ? NB : 4;
    while (bpu > 1 && static_cast<long long>(num_tiles) * ((NB + bpu - 1) / bpu) < target) --bpu;
  }
  const int col_units = (NB + bpu - 1) / bpu;
  int ksplit = 1;
  const long long waves = static_cast<long long>(num_tiles) * col_units * waves_per_unit;
  if (waves < target / 2) {
    const long long want = (target / 2 + waves - 1) / waves;
    ksplit = static_cast<int>(want < K ? want : K);
    const size_t need = static_cast<size_t>(ksplit) * M_out * Cout * sizeof(float);
    if (ksplit > 1 && (ws == nullptr || ws_bytes < need)) ksplit = 1;   // no scratch: stay exact
  }
  const int k_per_split = (K + ksplit - 1) / ksplit;
  ksplit = (K + k_per_split - 1) / k_per_split;

  ConvArgs a;
  a.in = in; a.nbr = nbr; a.w = w_k8; a.post_scale = post_scale; a.post_shift = post_shift;
  a.residual = residual; a.act_scale = act_scale; a.act_shift = act_shift; a.out_act = out_act;
  a.order = order; a.tile_mask = tile_mask; a.nbr_tiles = nbr_tiles;
  a.out = ksplit > 1 ? static_cast<float *>(ws) : out;
  a.M_out = M_out; a.K = K; a.Cin = Cin; a.Cout = Cout;
  a.col_units = col_units; a.blocks_per_unit = bpu; a.ksplit = ksplit; a.k_per_split = k_per_split;
  a.trace = nullptr;
  a.queue = nullptr;
  a.dyn_rounds = 2;
  a.out16 = out16;
  a.res16 = res16;
  // offset-split layers of the persistent kernel: partial sums combined inside the launch by the last
  // workgroup of each (tile, column unit) instead of by conv_reduce_kernel (the default; SG_CONV_COMBINE=0
  // / sg_spconv_set_combine(0): the separate reduce kernel, same numbers)
  static const int combine_env = getenv("SG_CONV_COMBINE") ? atoi(getenv("SG_CONV_COMBINE")) : 1;
  a.done = nullptr;
  a.out_final = out;
  const int combine_ov = g_combine_override.load(std::memory_order_relaxed);
  if ((combine_ov >= 0 ? combine_ov : combine_env) != 0 && persistent && ksplit > 1 &&
      static_cast<long long>(num_tiles) * col_units <= kDoneCounters)
    a.done = take_done(stream, static_cast<size_t>(num_tiles) * col_units);
  const long long units = static_cast<long long>(num_tiles) * col_units * ksplit;
  a.num_units = static_cast<int>(units);
  auto magic = [](unsigned d) { return d <= 1 ? 0u : static_cast<unsigned>((1ULL << 32) / d) + 1u; };
  a.magic_upt = magic(static_cast<unsigned>(col_units * ksplit));
  a.magic_cu = magic(static_cast<unsigned>(col_units));
  a.magic_nsl = 0;

  if (!split && t_chain.mode == 2) {      // not a layer the chain kernel takes
    const int rc = chain_flush();
    if (rc != SG_OK) return rc;
  }
  if (in16 || out16 || res16) {
    SG_REQUIRE(split && split_on == 2, "sg_unet_forward(bf16 rows): layer K=%d Cin=%d Cout=%d does not run on the bf16-operand "
               "persistent kernel", K, Cin, Cout);
    SG_REQUIRE(ksplit == 1 || a.done != nullptr, "sg_unet_forward(bf16 rows): an offset-split layer needs the in-launch combine");
  }
  if (split) {
    const int rc = launch_persistent_split(a, num_tiles, in_bytes_ll, w_bytes_ll, stream, split_on == 2, in16 != 0);
    if (rc != SG_OK) return rc;
    if (t_chain.mode == 2 && t_chain.args.n > 0) return SG_OK;      // recorded (an offset-split layer combines in the launch)
  } else if (persistent) {
    // fp32-MFMA kernel (SG_CONV_SPLIT=0 / sg_spconv_set_arithmetic(0), and layers below
    // SG_CONV_SPLIT_MIN_CIN): 16-channel slices, 2-deep operand ring, 4 waves per unit, one column
    // block; as many workgroups as are resident at once (a multiple of 8 so that unit u always runs
    // on XCD u % 8).  (Rounds 1-3 also carried 4- and 8-deep rings, an 80-VGPR build and 64-column
    // units of this kernel; none of them was faster and they are gone.)
    const size_t lds = static_cast<size_t>(kWavesPerWg) * 16 * 64 * sizeof(float) +
                       2 * kMetaInts * sizeof(int32_t) + 16;
    constexpr int kSliceCh = 16;
    static int num_cu = 0, occ = 0;
    static std::once_flag once;
    std::call_once(once, [&] {
      int dev = 0;
      hipGetDevice(&dev);
      hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev);
      if (num_cu <= 0) num_cu = 256;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_conv_persistent_kernel<kSliceCh, 2, 0, 5>, 256, lds);
      if (occ < 1) occ = 1;
    });
    a.magic_nsl = magic(static_cast<unsigned>(Cin / kSliceCh));
    static const bool dyn_env = getenv("SG_CONV_STATIC") && atoi(getenv("SG_CONV_STATIC")) == 0;   // hand-out A/B
    a.queue = dyn_env ? take_tickets(stream) : nullptr;          // (null: static snake, the default)
    a.dyn_rounds = 2;
    long long g = static_cast<long long>(num_cu) * occ;
    if (g >= 8) g -= g % 8;
    if (units < g) g = units >= 8 ? (units + 7) / 8 * 8 : units;      // (see launch_persistent_split)
    const unsigned ib_ = static_cast<unsigned>(in_bytes_ll), wb_ = static_cast<unsigned>(w_bytes_ll);
    static const char *trace_env = getenv("SG_CONV_TRACE");     // developer tool: per-wave phase stamps
    if (trace_env) {
      const size_t nb = static_cast<size_t>(units) * kWavesPerWg * 8 * sizeof(unsigned long long);
      unsigned long long *dbuf = nullptr;
      hipMalloc(&dbuf, nb);
      hipMemsetAsync(dbuf, 0, nb, stream);
      a.trace = dbuf;
      gather_conv_persistent_kernel<kSliceCh, 2, 1, 5><<<static_cast<int>(g), 256, lds, stream>>>(a, ib_, wb_);
      hipStreamSynchronize(stream);
      std::vector<unsigned long long> h(nb / 8);
      hipMemcpy(h.data(), dbuf, nb, hipMemcpyDeviceToHost);
      hipFree(dbuf);
      if (FILE *f = fopen(trace_env, "ab")) {
        long long hdr[8] = {M_out, K, Cin, Cout, units, col_units, ksplit, g};
        fwrite(hdr, 8, 8, f);
        fwrite(h.data(), 8, h.size(), f);
        fclose(f);
      }
    } else {
      gather_conv_persistent_kernel<kSliceCh, 2, 0, 5><<<static_cast<int>(g), 256, lds, stream>>>(a, ib_, wb_);
    }
  } else {
    const size_t lds = kWavesPerWg * kTileRows * kMaxK * sizeof(int32_t);
    const int grid = static_cast<int>((units + kWavesPerWg - 1) / kWavesPerWg);
    const bool vec = (Cin % kCk) == 0;
    switch (bpu) {
      case 1: launch_tile<1>(a, grid, lds, vec, stream); break;
      case 2: launch_tile<2>(a, grid, lds, vec, stream); break;
      case 3: launch_tile<3>(a, grid, lds, vec, stream); break;
      default: launch_tile<4>(a, grid, lds, vec, stream); break;
    }
  }
  if (ksplit > 1 && a.done == nullptr) {
    const long long n4 = static_cast<long long>(M_out) * Cout / 4;
    conv_reduce_kernel<<<grid_for(n4, 256), 256, 0, stream>>>(
        reinterpret_cast<const float4 *>(ws), reinterpret_cast<const float4 *>(residual), post_scale,
        post_shift, act_scale, act_shift, reinterpret_cast<float4 *>(out_act), ksplit, n4, Cout / 4,
        reinterpret_cast<float4 *>(out));
  }
  return check_launch("sg_spconv_gather_conv_f32");
}

int sg_spconv_gather_conv_f32(const float *in, int num_in_rows, const int32_t *nbr, int M_out,
                              int K, int Cin, int Cout, const float *w_k8, const float *post_scale,
                              const float *post_shift, const float *residual, const float *act_scale,
                              const float *act_shift, float *out_act, const int32_t *order,
                              const uint32_t *tile_mask, const int32_t *nbr_tiles, float *out,
                              void *ws, size_t ws_bytes, sg_stream_t stream_) {
  return gather_conv_impl(in, num_in_rows, nbr, M_out, K, Cin, Cout, w_k8, post_scale, post_shift, residual, act_scale,
                          act_shift, out_act, order, tile_mask, nbr_tiles, out, ws, ws_bytes, stream_, 0, 0, 0);
}

}  // extern "C"

namespace sg {
// the executor's conv with bf16 rows on either side (unet_exec.hip, arithmetic 3)
int conv_gather_rows(const void *in, int num_in_rows, const int32_t *nbr, int M_out, int K, int Cin, int Cout,
                     const float *w_k8, const float *post_scale, const float *post_shift, const void *residual,
                     const float *act_scale, const float *act_shift, void *out_act, const int32_t *order,
                     const uint32_t *tile_mask, const int32_t *nbr_tiles, void *out, void *ws, size_t ws_bytes,
                     sg_stream_t stream, int in16, int out16, int res16) {
  return gather_conv_impl(static_cast<const float *>(in), num_in_rows, nbr, M_out, K, Cin, Cout, w_k8, post_scale, post_shift,
                          static_cast<const float *>(residual), act_scale, act_shift, static_cast<float *>(out_act), order,
                          tile_mask, nbr_tiles, static_cast<float *>(out), ws, ws_bytes, stream, in16, out16, res16);
}
}  // namespace sg
