/*
 * sg_oracle_conv.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (C + OpenMP) of the sparse-convolution subset of spconv 2.1 that
 * the reference model calls (/root/reference/softgroup/model/softgroup.py:60-62,
 * blocks.py:57-70,101-119).  spconv itself is an un-vendored, unpinned third-party
 * dependency (docs/installation.md:27, "pip install spconv-cu102"), so this file
 * restates its *published semantics* (SURVEY.md section 2.4):
 *   SubMConv3d k3 p1  : out[j] = sum_d W[:,d+1,:] . in[i],  coord_i = coord_j + d
 *   SparseConv3d k2 s2: out[c//2] += W[:,c%2,:] . in[c]      (c//2 >= D//2 dropped)
 *   SparseInverseConv3d k2: out[i] = W[:,c_i%2,:] . in[parent(i)]
 * Weight layout [Cout, kD, kH, kW, Cin] (tools/convert_checkpoint.py:17-19).
 * "Parity unpinned" against spconv itself; pinned instead against the dense
 * torch.nn.functional.conv3d / conv_transpose3d equivalence in
 * tests/test_oracle_golden.py::test_sparse_conv_matches_dense_torch (fp32, tolerance 1e-4).
 *
 * Also the CPU baseline timed by bench.py (kind "port").
 * Build: gcc -O3 -march=native -fopenmp -fPIC -shared.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- coordinate hash: key = linearised (b, d0, d1, d2) -> row ---------------- */
typedef struct { uint64_t *keys; int32_t *vals; size_t cap; } chash;
static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  return x ^ (x >> 33);
}
static chash chash_new(size_t n) {
  chash h; h.cap = 16;
  while (h.cap < n * 2 + 2) h.cap <<= 1;
  h.keys = (uint64_t *)malloc(h.cap * sizeof(uint64_t));
  h.vals = (int32_t *)malloc(h.cap * sizeof(int32_t));
  memset(h.keys, 0xff, h.cap * sizeof(uint64_t));
  return h;
}
static void chash_free(chash *h) { free(h->keys); free(h->vals); }
/* insert-if-absent; returns stored value */
static int32_t chash_put(chash *h, uint64_t k, int32_t v) {
  size_t s = mix64(k) & (h->cap - 1);
  for (;;) {
    if (h->keys[s] == ~0ULL) { h->keys[s] = k; h->vals[s] = v; return v; }
    if (h->keys[s] == k) return h->vals[s];
    s = (s + 1) & (h->cap - 1);
  }
}
static int32_t chash_get(const chash *h, uint64_t k) {
  size_t s = mix64(k) & (h->cap - 1);
  for (;;) {
    if (h->keys[s] == ~0ULL) return -1;
    if (h->keys[s] == k) return h->vals[s];
    s = (s + 1) & (h->cap - 1);
  }
}
static inline uint64_t lin(int64_t b, int64_t x, int64_t y, int64_t z, const int32_t *shape) {
  return (uint64_t)(((b * shape[0] + x) * shape[1] + y) * shape[2] + z);
}

/* SubM 3x3x3 neighbour table: nbr[j*27 + k] = row of coord_j + (k/9-1, k/3%3-1, k%3-1) or -1 */
void orc_subm_rulebook(const int32_t *indices, int M, const int32_t *shape, int32_t *nbr) {
  chash h = chash_new((size_t)M);
  for (int i = 0; i < M; i++)
    chash_put(&h, lin(indices[i * 4], indices[i * 4 + 1], indices[i * 4 + 2], indices[i * 4 + 3], shape), i);
#pragma omp parallel for schedule(static)
  for (int j = 0; j < M; j++) {
    const int32_t *c = indices + (size_t)j * 4;
    for (int k = 0; k < 27; k++) {
      int x = c[1] + k / 9 - 1, y = c[2] + (k / 3) % 3 - 1, z = c[3] + k % 3 - 1;
      int32_t r = -1;
      if (x >= 0 && y >= 0 && z >= 0 && x < shape[0] && y < shape[1] && z < shape[2])
        r = chash_get(&h, lin(c[0], x, y, z, shape));
      nbr[(size_t)j * 27 + k] = r;
    }
  }
  chash_free(&h);
}

/* strided k2 s2 p0: out coord = c//2 (dropped if >= shape//2); output rows in
 * first-seen order of their first child (row order is internal to spconv and not
 * observable through the reference model).  in2out[i] = out row or -1.
 * out_indices capacity [M,4].  Returns M_out. */
int orc_down_rulebook(const int32_t *indices, int M, const int32_t *shape, int32_t *out_indices,
                      int32_t *in2out) {
  int32_t oshape[3] = {shape[0] / 2, shape[1] / 2, shape[2] / 2};
  chash h = chash_new((size_t)M);
  int m_out = 0;
  for (int i = 0; i < M; i++) {
    const int32_t *c = indices + (size_t)i * 4;
    int x = c[1] / 2, y = c[2] / 2, z = c[3] / 2;
    if (x >= oshape[0] || y >= oshape[1] || z >= oshape[2]) { in2out[i] = -1; continue; }
    int32_t r = chash_put(&h, lin(c[0], x, y, z, oshape), m_out);
    if (r == m_out) {
      out_indices[m_out * 4] = c[0]; out_indices[m_out * 4 + 1] = x;
      out_indices[m_out * 4 + 2] = y; out_indices[m_out * 4 + 3] = z;
      m_out++;
    }
    in2out[i] = r;
  }
  chash_free(&h);
  return m_out;
}

/* generic output-stationary gather conv: out[j] = sum_k Wt[k] . in[nbr[j,k]]
 * W is [Cout, K, Cin] (OKKKI flattened); transposed once to [K][Cin][Cout]. */
static void gather_conv(const float *in, const int32_t *nbr, int M_out, int K, int Cin, int Cout,
                        const float *W, float *out) {
  float *Wt = (float *)malloc((size_t)K * Cin * Cout * sizeof(float));
  for (int co = 0; co < Cout; co++)
    for (int k = 0; k < K; k++)
      for (int ci = 0; ci < Cin; ci++)
        Wt[((size_t)k * Cin + ci) * Cout + co] = W[((size_t)co * K + k) * Cin + ci];
#pragma omp parallel for schedule(dynamic, 64)
  for (int j = 0; j < M_out; j++) {
    float *o = out + (size_t)j * Cout;
    for (int co = 0; co < Cout; co++) o[co] = 0.f;
    for (int k = 0; k < K; k++) {
      int32_t i = nbr[(size_t)j * K + k];
      if (i < 0) continue;
      const float *x = in + (size_t)i * Cin;
      const float *w = Wt + (size_t)k * Cin * Cout;
      for (int ci = 0; ci < Cin; ci++) {
        float a = x[ci];
        const float *wr = w + (size_t)ci * Cout;
#pragma omp simd
        for (int co = 0; co < Cout; co++) o[co] += a * wr[co];
      }
    }
  }
  free(Wt);
}

void orc_subm_conv3d(const float *in, const int32_t *nbr, int M, int Cin, int Cout,
                     const float *W, float *out) {
  gather_conv(in, nbr, M, 27, Cin, Cout, W, out);
}

/* children table of the strided conv: child[o*8 + k] = input row with c = 2*o + k, k=(kx,ky,kz) */
void orc_down_children(const int32_t *indices, const int32_t *in2out, int M, int M_out,
                       int32_t *child) {
  for (size_t i = 0; i < (size_t)M_out * 8; i++) child[i] = -1;
  for (int i = 0; i < M; i++) {
    if (in2out[i] < 0) continue;
    const int32_t *c = indices + (size_t)i * 4;
    int k = (c[1] & 1) * 4 + (c[2] & 1) * 2 + (c[3] & 1);
    child[(size_t)in2out[i] * 8 + k] = i;
  }
}

void orc_sparse_conv3d_k2s2(const float *in, const int32_t *child, int M_out, int Cin, int Cout,
                            const float *W, float *out) {
  gather_conv(in, child, M_out, 8, Cin, Cout, W, out);
}

/* inverse: out[i] = W[:, c_i%2, :] . in[parent(i)];  rows dropped by the down conv get 0 */
void orc_inverse_conv3d_k2(const float *in, const int32_t *indices_fine, const int32_t *in2out,
                           int M, int Cin, int Cout, const float *W, float *out) {
  int32_t *nbr = (int32_t *)malloc((size_t)M * 8 * sizeof(int32_t));
  for (int i = 0; i < M; i++) {
    const int32_t *c = indices_fine + (size_t)i * 4;
    int kk = (c[1] & 1) * 4 + (c[2] & 1) * 2 + (c[3] & 1);
    for (int k = 0; k < 8; k++) nbr[(size_t)i * 8 + k] = (k == kk) ? in2out[i] : -1;
  }
  gather_conv(in, nbr, M, 8, Cin, Cout, W, out);
  free(nbr);
}
