// spconv_conv.hip -- sparse convolution as an output-stationary implicit GEMM on fp32 MFMA.
// Replaces the conv kernels of the un-vendored spconv 2.1 library for SubMConv3d /
// SparseConv3d(k2,s2) / SparseInverseConv3d (reference call sites: softgroup/model/softgroup.py:61,
// softgroup/model/blocks.py:57-70,101-119).  One kernel serves all three: it only sees a gather
// table nbr[M_out, K] (spconv_rulebook.hip) and weights re-laid out as [K][Cin][Cout].
//
//   out[j,:] = residual[j,:] + sum_k  act(in[nbr[j,k],:]) . W[k]        act = relu(x*s + b) | id
//
// MI355X mapping
//   * a wave owns a tile of 32 output rows (rows taken in neighbour-mask-sorted order, so the
//     tile skips kernel offsets none of its rows has) and ALL Cout columns: Cout/32 accumulators
//     of v_mfma_f32_32x32x2_f32 (exact fp32 = fmaf chain; 64 FLOP/clk/SIMD, the fp32 peak);
//   * a workgroup = 4 waves = 4 adjacent tiles shares each W[k] chunk through LDS;
//   * gathered rows are fetched as whole 128-B lines (8 lanes x 16 B per row, 8 rows per load
//     instruction), get the fused BatchNorm+ReLU on the way, and are staged in LDS with a +1
//     padded stride so the MFMA A-operand column reads are bank-conflict free;
//   * every output row is written exactly once, 128 B per half-wave (residual add fused).
// HBM traffic per layer ~ P*Cin*4 (gathered lines) + M*Cout*4 (stores) + index tables, i.e. the
// "gather/scatter" bytes B_gs of SURVEY 8(d); the weights (<= 8 MB) stay in L2/MALL.
#include "common.h"

namespace sg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileRows = 32;
constexpr int kWavesPerWg = 4;
constexpr int kChunk = 32;            // Cin slice per staging step
constexpr int kAStride = kChunk + 1;  // padded LDS row stride (floats)

template <int NB>
__global__ void __launch_bounds__(256) gather_conv_mfma_kernel(
    const float *__restrict__ in, const int32_t *__restrict__ nbr, int M_out, int K, int Cin,
    const float *__restrict__ w_kio, const float *__restrict__ bn_scale,
    const float *__restrict__ bn_shift, const float *__restrict__ residual,
    const int32_t *__restrict__ order, const uint32_t *__restrict__ tile_mask,
    float *__restrict__ out) {
  constexpr int Cout = NB * 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *w_lds = smem;                                  // [kChunk][Cout]
  float *a_lds_all = smem + kChunk * Cout;              // [4][32][kAStride]
  int32_t *rows_all = reinterpret_cast<int32_t *>(a_lds_all + kWavesPerWg * kTileRows * kAStride);
  int32_t *src_all = rows_all + kWavesPerWg * kTileRows;  // [4][32] gathered row for the current k

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *a_lds = a_lds_all + wave * kTileRows * kAStride;
  int32_t *rows = rows_all + wave * kTileRows;
  int32_t *src = src_all + wave * kTileRows;

  const int num_tiles = (M_out + kTileRows - 1) / kTileRows;
  const int tile = blockIdx.x * kWavesPerWg + wave;
  const bool tile_valid = tile < num_tiles;

  // output rows of this tile
  if (lane < kTileRows) {
    const int pos = tile * kTileRows + lane;
    rows[lane] = (tile_valid && pos < M_out) ? (order ? order[pos] : pos) : -1;
  }
  const uint32_t full = K >= 32 ? 0xffffffffu : ((1u << K) - 1u);
  uint32_t my_mask = tile_valid ? (tile_mask ? tile_mask[tile] : full) : 0u;
  // workgroup-wide union decides which W[k] chunks get staged
  __shared__ uint32_t wg_mask_s[kWavesPerWg];
  if (lane == 0) wg_mask_s[wave] = my_mask;
  __syncthreads();
  const uint32_t wg_mask = wg_mask_s[0] | wg_mask_s[1] | wg_mask_s[2] | wg_mask_s[3];

  f32x16 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  const int arow = lane & 31, ahalf = lane >> 5;
  const bool cin_vec = (Cin % kChunk) == 0;

  for (int k = 0; k < K; ++k) {
    if (!((wg_mask >> k) & 1u)) continue;
    const bool mine = (my_mask >> k) & 1u;
    if (mine && lane < kTileRows) {
      const int r = rows[lane];
      src[lane] = r >= 0 ? nbr[static_cast<int64_t>(r) * K + k] : -1;
    }
    for (int c0 = 0; c0 < Cin; c0 += kChunk) {
      __syncthreads();  // previous chunk fully consumed (W and A), src[] visible
      // ---- stage W[k][c0 .. c0+32) x Cout
      {
        const float *wsrc = w_kio + (static_cast<int64_t>(k) * Cin + c0) * Cout;
        const int valid_rows = min(kChunk, Cin - c0);
        for (int e = threadIdx.x * 4; e < kChunk * Cout; e += 256 * 4) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e / Cout < valid_rows) v = *reinterpret_cast<const float4 *>(wsrc + e);
          *reinterpret_cast<float4 *>(w_lds + e) = v;
        }
      }
      // ---- gather A: 32 rows x 32 channels of this wave's tile
      if (mine) {
        if (cin_vec) {
          const int q = lane & 7;  // 16-B piece of the 128-B line
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int r = it * 8 + (lane >> 3);
            const int s = src[r];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s >= 0) {
              v = *reinterpret_cast<const float4 *>(in + static_cast<int64_t>(s) * Cin + c0 + q * 4);
              if (bn_scale) {
                const float4 sc = *reinterpret_cast<const float4 *>(bn_scale + c0 + q * 4);
                const float4 sh = *reinterpret_cast<const float4 *>(bn_shift + c0 + q * 4);
                v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
                v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
                v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
              }
            }
            float *d = a_lds + r * kAStride + q * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
          }
        } else {
          const int c = lane & 31;
#pragma unroll 4
          for (int it = 0; it < 16; ++it) {
            const int r = it * 2 + (lane >> 5);
            const int s = src[r];
            float v = 0.f;
            if (s >= 0 && c0 + c < Cin) {
              v = in[static_cast<int64_t>(s) * Cin + c0 + c];
              if (bn_scale) v = fmaxf(fmaf(v, bn_scale[c0 + c], bn_shift[c0 + c]), 0.f);
            }
            a_lds[r * kAStride + c] = v;
          }
        }
      }
      __syncthreads();
      // ---- 16 MFMA steps (K=2 each) over the chunk
      if (mine) {
#pragma unroll 4
        for (int kk = 0; kk < kChunk; kk += 2) {
          const float a = a_lds[arow * kAStride + kk + ahalf];
          const float *wrow = w_lds + (kk + ahalf) * Cout + arow;
#pragma unroll
          for (int n = 0; n < NB; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wrow[n * 32], acc[n], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: acc[n][reg] -> out[row (reg&3)+8*(reg>>2)+4*half][n*32 + col]
  if (tile_valid) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = (reg & 3) + 8 * (reg >> 2) + 4 * ahalf;
      const int row = rows[r];
      if (row < 0) continue;
      float *o = out + static_cast<int64_t>(row) * Cout + arow;
      const float *res = residual ? residual + static_cast<int64_t>(row) * Cout + arow : nullptr;
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        float v = acc[n][reg];
        if (res) v += res[n * 32];
        o[n * 32] = v;
      }
    }
  }
}

// Scalar path of the same operator for channel counts the MFMA tiling does not cover
// (Cout % 32 != 0).  One thread per (row, cout).
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

template <int NB>
static int launch_mfma(const float *in, const int32_t *nbr, int M_out, int K, int Cin,
                       const float *w_kio, const float *bn_scale, const float *bn_shift,
                       const float *residual, const int32_t *order, const uint32_t *tile_mask,
                       float *out, hipStream_t stream) {
  constexpr int Cout = NB * 32;
  const size_t lds = (kChunk * Cout + kWavesPerWg * kTileRows * kAStride) * sizeof(float) +
                     2 * kWavesPerWg * kTileRows * sizeof(int32_t);
  const int num_tiles = (M_out + kTileRows - 1) / kTileRows;
  const int grid = (num_tiles + kWavesPerWg - 1) / kWavesPerWg;
  gather_conv_mfma_kernel<NB><<<grid, 256, lds, stream>>>(in, nbr, M_out, K, Cin, w_kio, bn_scale,
                                                          bn_shift, residual, order, tile_mask, out);
  return check_launch("sg_spconv_gather_conv_f32");
}

}  // namespace sg

using namespace sg;

extern "C" {

int sg_spconv_gather_conv_f32(const float *in, int num_in_rows, const int32_t *nbr, int M_out,
                              int K, int Cin, int Cout, const float *w_kio, const float *bn_scale,
                              const float *bn_shift, const float *residual, const int32_t *order,
                              const uint32_t *tile_mask, float *out, sg_stream_t stream_) {
  (void)num_in_rows;
  SG_REQUIRE(M_out >= 0 && K >= 1 && K <= 32 && Cin >= 1 && Cout >= 1,
             "sg_spconv_gather_conv_f32: bad arguments (M_out=%d K=%d Cin=%d Cout=%d)", M_out, K,
             Cin, Cout);
  SG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr),
             "sg_spconv_gather_conv_f32: bn_scale and bn_shift must come together");
  if (M_out == 0) return SG_OK;
  hipStream_t stream = as_stream(stream_);
  if (Cout % 32 == 0 && Cout <= 256) {
    switch (Cout / 32) {
#define SG_CASE(NB)                                                                              \
  case NB:                                                                                       \
    return launch_mfma<NB>(in, nbr, M_out, K, Cin, w_kio, bn_scale, bn_shift, residual, order,   \
                           tile_mask, out, stream);
      SG_CASE(1) SG_CASE(2) SG_CASE(3) SG_CASE(4) SG_CASE(5) SG_CASE(6) SG_CASE(7) SG_CASE(8)
#undef SG_CASE
    }
  }
  gather_conv_scalar_kernel<<<grid_for(static_cast<int64_t>(M_out) * Cout, 256, 256 * 32), 256, 0,
                              stream>>>(in, nbr, M_out, K, Cin, Cout, w_kio, bn_scale, bn_shift,
                                        residual, out);
  return check_launch("sg_spconv_gather_conv_f32(scalar)");
}

}  // extern "C"
