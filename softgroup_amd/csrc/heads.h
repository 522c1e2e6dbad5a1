// heads.h -- the two-layer MLP of softgroup/model/blocks.py:9-27 (Linear, BatchNorm1d, ReLU, Linear) in
// eval mode as a per-thread FMA chain: shared by the point-wise heads kernel (heads.hip) and the mask head
// of the one-call scan (scan_forward.hip).
#pragma once
#include "common.h"

namespace sg {

struct Mlp2 {
  const float *w1, *b1, *scale, *shift, *w2, *b2;   // w1 [C][C], w2 [out][C] (nn.Linear layout: [out, in])
  int out;
};

template <int C, int OUT_MAX>
__device__ __forceinline__ void mlp2(const float (&x)[C], const Mlp2 &m, float (&y)[OUT_MAX]) {
  float h[C];
#pragma unroll
  for (int o = 0; o < C; ++o) {
    float a = m.b1[o];
#pragma unroll
    for (int c = 0; c < C; ++c) a = fmaf(x[c], m.w1[o * C + c], a);
    h[o] = fmaxf(fmaf(a, m.scale[o], m.shift[o]), 0.f);
  }
#pragma unroll
  for (int o = 0; o < OUT_MAX; ++o) {
    if (o < m.out) {       // uniform
      float a = m.b2[o];
#pragma unroll
      for (int c = 0; c < C; ++c) a = fmaf(h[c], m.w2[o * C + c], a);
      y[o] = a;
    }
  }
}

}  // namespace sg
