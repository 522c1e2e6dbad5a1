/*
 * sg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the SoftGroup grouping-head operators, written from
 * the reference's algorithm (file:line cited per function, all paths relative to
 * /root/reference/softgroup/ops/src).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (softgroup_amd/) never does.
 *
 * Parity pin: voxelize_idx / bfs_cluster / octree build are checked against the
 * reference's own C++ (compiled unmodified into oracle/_ref, see build_ref.py) in
 * tests/test_oracle_golden.py and against tests/golden/*.npz generated from it.
 * The CUDA-only kernels of the reference (voxelize_fp/bp, ballquery_batch_p,
 * octree_ball_query, sec_*, global_avg_pool, mask IoU/label) have no CPU build in
 * the reference and no golden vectors: their restatements below are "parity
 * unpinned" (followed line by line).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (fp contraction OFF: every
 * multiply/add below is a separate IEEE op unless fmaf() is written explicitly).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* voxelize_idx  (voxelize/voxelize.cpp:11-165)                               */
/* ------------------------------------------------------------------------- */
/* Point key = (batch, x, y, z) with coords cast long -> int32                 */
/* (voxelize.cpp:86-107: p[j] = coords[j]; one hash map per batch index).      */
/* Voxel id = first-seen order (sg.mp[p] = nActive++, voxelize.cpp:109-111).   */
typedef struct { int32_t b, x, y, z; } vkey_t;

static inline uint64_t vkey_hash(vkey_t k) {
  uint64_t h = 1469598103934665603ULL;
  uint32_t w[4] = {(uint32_t)k.b, (uint32_t)k.x, (uint32_t)k.y, (uint32_t)k.z};
  for (int i = 0; i < 4; i++) { h ^= w[i]; h *= 1099511628211ULL; }
  return h ^ (h >> 29);
}

/* phase 1: input_map[N], *M_out = #voxels, *maxActive_out (voxelize.cpp:70-163) */
int orc_voxelize_idx(const int64_t *coords, int n, int ncol, int mode, int32_t *input_map,
                     int32_t *M_out, int32_t *max_active_out) {
  size_t cap = 16;
  while (cap < (size_t)n * 2 + 2) cap <<= 1;
  int32_t *slot = (int32_t *)malloc(cap * sizeof(int32_t)); /* first point idx of the voxel */
  int32_t *vid_of_first = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  int32_t *count = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
  if (!slot || !vid_of_first || !count) return -1;
  for (size_t i = 0; i < cap; i++) slot[i] = -1;
  int32_t n_active = 0;
  for (int i = 0; i < n; i++) {
    const int64_t *c = coords + (size_t)i * ncol;
    vkey_t k;
    if (ncol == 3) { k.b = 0; k.x = (int32_t)c[0]; k.y = (int32_t)c[1]; k.z = (int32_t)c[2]; }
    else { k.b = (int32_t)c[0]; k.x = (int32_t)c[1]; k.y = (int32_t)c[2]; k.z = (int32_t)c[3]; }
    size_t h = vkey_hash(k) & (cap - 1);
    for (;;) {
      int32_t f = slot[h];
      if (f < 0) { slot[h] = i; vid_of_first[i] = n_active++; f = i; }
      const int64_t *d = coords + (size_t)f * ncol;
      vkey_t q;
      if (ncol == 3) { q.b = 0; q.x = (int32_t)d[0]; q.y = (int32_t)d[1]; q.z = (int32_t)d[2]; }
      else { q.b = (int32_t)d[0]; q.x = (int32_t)d[1]; q.y = (int32_t)d[2]; q.z = (int32_t)d[3]; }
      if (q.b == k.b && q.x == k.x && q.y == k.y && q.z == k.z) {
        int32_t v = vid_of_first[f];
        input_map[i] = v;
        count[v]++;
        break;
      }
      h = (h + 1) & (cap - 1);
    }
  }
  int32_t max_active = 1; /* voxelize.cpp:150 */
  if (mode == 3 || mode == 4)
    for (int32_t v = 0; v < n_active; v++)
      if (count[v] > max_active) max_active = count[v];
  *M_out = n_active;
  *max_active_out = max_active;
  free(slot); free(vid_of_first); free(count);
  return 0;
}

/* phase 2: output_map[M, maxActive+1] = [count, ascending point idx..., 0 pad]  */
/* (voxelize.cpp:151-163); modes 0/1/2 keep one entry (:127-149);               */
/* output_coords[v] = coords[rule[1]] (voxelize.cpp:41-56).                      */
void orc_voxelize_idx_fill(const int64_t *coords, int n, int ncol, int mode,
                           const int32_t *input_map, int M, int max_active,
                           int64_t *out_coords, int32_t *out_map) {
  size_t stride = (size_t)max_active + 1;
  memset(out_map, 0, (size_t)M * stride * sizeof(int32_t));
  if (mode == 3 || mode == 4) {
    for (int i = 0; i < n; i++) {
      int32_t *r = out_map + (size_t)input_map[i] * stride;
      r[++r[0]] = i;
    }
  } else {
    /* mode 0 (unique) and 1 keep outputRows.front(), mode 2 keeps .back() */
    for (int i = 0; i < n; i++) {
      int32_t *r = out_map + (size_t)input_map[i] * stride;
      if (r[0] == 0 || mode == 2) { r[0] = 1; r[1] = i; }
    }
  }
  for (int v = 0; v < M; v++) {
    int32_t first = out_map[(size_t)v * stride + 1];
    memcpy(out_coords + (size_t)v * ncol, coords + (size_t)first * ncol, ncol * sizeof(int64_t));
  }
}

/* ------------------------------------------------------------------------- */
/* voxelize_fp / voxelize_bp  (voxelize/voxelize.cu:10-54)                     */
/* ------------------------------------------------------------------------- */
/* out[row,p] = sum_{i=1..cnt} (m * feats[r[i],p]); separate mul then add      */
/* (atomicAdd blocks contraction, voxelize.cu:21); m = 1/cnt if average.        */
void orc_voxelize_fp(const float *feats, float *out, const int32_t *rules, int M, int max_active,
                     int C, int average) {
  for (int row = 0; row < M; row++) {
    const int32_t *r = rules + (size_t)row * (max_active + 1);
    int32_t cnt = r[0];
    float m = (average && cnt > 0) ? (float)1 / cnt : (float)1;
    float *o = out + (size_t)row * C;
    for (int p = 0; p < C; p++) o[p] = 0.f; /* pre-zeroed by functions.py:216 */
    for (int i = 1; i <= cnt; i++) {
      const float *in = feats + (size_t)r[i] * C;
      for (int p = 0; p < C; p++) { float t = m * in[p]; o[p] = o[p] + t; }
    }
  }
}

void orc_voxelize_bp(const float *d_out, float *d_feats, const int32_t *rules, int M,
                     int max_active, int C, int average) {
  /* d_feats pre-zeroed by the caller (functions.py:228) */
  for (int row = 0; row < M; row++) {
    const int32_t *r = rules + (size_t)row * (max_active + 1);
    int32_t cnt = r[0];
    float m = (average && cnt > 0) ? (float)1 / cnt : (float)1;
    const float *o = d_out + (size_t)row * C;
    for (int i = 1; i <= cnt; i++) {
      float *in = d_feats + (size_t)r[i] * C;
      for (int p = 0; p < C; p++) { float t = m * o[p]; in[p] = in[p] + t; }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* ballquery_batch_p  (bfs_cluster/bfs_cluster.cu:15-66)                       */
/* ------------------------------------------------------------------------- */
/* d2 = (dx*dx + dy*dy) + dz*dz under the reference build's default floating-point      */
/* contraction (nvcc -fmad=true, setup.py:15-23).  The fusion is done by LLVM's generic   */
/* DAG combiner, which NVVM (nvcc) and AMDGPU share: in fadd(fmul, fmul) the LEFT product  */
/* is fused and the right one rounded, i.e.  mul t = dy*dy; fma t = dx*dx + t;             */
/* fma d2 = dz*dz + t.  This is what the reference's own kernel compiles to here           */
/* (oracle/build_ref.py, SLP vectorisation off: v_mul dy,dy / v_fmac dx,dx / v_fmac dz,dz) */
/* and tests/test_ref_gpu_kernels.py pins it on pairs within 1 ulp of the radius.          */
/* (Round 1 assumed fma(dz,dz,fma(dy,dy,dx*dx)), SURVEY App. B-3: refuted by that test.)   */
/* The HIP kernel writes the same three operations explicitly.                             */
static inline float dist2_fma(float ox, float oy, float oz, float x, float y, float z) {
  float dx = ox - x, dy = oy - y, dz = oz - z;
  return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
}

/* start offsets follow ascending point order (one of the atomic orders the    */
/* reference can produce, bfs_cluster.cu:52).  Returns the un-truncated total. */
int64_t orc_ballquery_batch_p(const float *xyz, const int32_t *batch_idxs,
                              const int32_t *batch_offsets, int n, int mean_active, float radius,
                              int32_t *idx, int32_t *start_len) {
  float radius2 = radius * radius;
  int64_t cumsum = 0;
  int64_t thre = (int64_t)n * mean_active;
  int32_t *tmp = (int32_t *)malloc(1000 * sizeof(int32_t));
  for (int pt = 0; pt < n; pt++) {
    float ox = xyz[pt * 3 + 0], oy = xyz[pt * 3 + 1], oz = xyz[pt * 3 + 2];
    int b = batch_idxs[pt];
    int start = batch_offsets[b], end = batch_offsets[b + 1];
    int cnt = 0;
    for (int k = start; k < end; k++) {
      float d2 = dist2_fma(ox, oy, oz, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]);
      if (d2 < radius2) {
        if (cnt < 1000) tmp[cnt] = k; else break; /* bfs_cluster.cu:43-48 */
        ++cnt;
      }
    }
    start_len[pt * 2 + 0] = (int32_t)cumsum;
    start_len[pt * 2 + 1] = cnt;
    int64_t s = cumsum;
    cumsum += cnt;
    if (s >= thre) continue;
    int w = cnt;
    if (s + cnt >= thre) w = (int)(thre - s);
    for (int k = 0; k < w; k++) idx[s + k] = tmp[k];
  }
  free(tmp);
  return cumsum;
}

/* ------------------------------------------------------------------------- */
/* bfs_cluster  (bfs_cluster/bfs_cluster.cpp:33-126)                           */
/* ------------------------------------------------------------------------- */
/* seeds ascending, FIFO queue, neighbours in list order (find_cc :33-58);     */
/* keep iff (int)size >= thr, thr = threshold or threshold*mean (:73-81).       */
/* cluster_idxs capacity [N,2], cluster_offsets capacity [N+1].                */
void orc_bfs_cluster(const float *class_numpoint_mean, const int32_t *bq_idxs,
                     const int32_t *start_len, int N, float threshold, int class_id,
                     int32_t *cluster_idxs, int32_t *cluster_offsets, int32_t *n_cluster_out,
                     int32_t *sum_npoint_out) {
  int32_t *visited = (int32_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int32_t));
  int32_t *queue = (int32_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int32_t));
  int n_cluster = 0, sum_np = 0;
  cluster_offsets[0] = 0;
  for (int i = 0; i < N; i++) {
    if (visited[i]) continue;
    int head = 0, tail = 0;
    queue[tail++] = i;
    visited[i] = 1;
    while (head < tail) {
      int cur = queue[head++];
      int s = start_len[cur * 2], l = start_len[cur * 2 + 1];
      for (int e = s; e < s + l; e++) {
        int v = bq_idxs[e];
        if (visited[v]) continue;
        visited[v] = 1;
        queue[tail++] = v;
      }
    }
    float mean = class_numpoint_mean[class_id];
    float thr = (mean == -1) ? threshold : threshold * mean;
    if ((float)(int)tail >= thr) {
      for (int j = 0; j < tail; j++) {
        cluster_idxs[(size_t)(sum_np + j) * 2 + 0] = n_cluster;
        cluster_idxs[(size_t)(sum_np + j) * 2 + 1] = queue[j];
      }
      sum_np += tail;
      n_cluster++;
      cluster_offsets[n_cluster] = sum_np;
    }
  }
  *n_cluster_out = n_cluster;
  *sum_npoint_out = sum_np;
  free(visited); free(queue);
}

/* ------------------------------------------------------------------------- */
/* octree build + export (octree_ball_query/octree_ball_query.cpp:8-165)       */
/* ------------------------------------------------------------------------- */
typedef struct onode {
  float box[6]; /* x y z w h l : centre + size */
  int level, num_points, is_leaf;
  int32_t *pt_inds; int n_inds, cap_inds;
  struct onode *oct[8];
} onode;

static onode *onode_new(void) { return (onode *)calloc(1, sizeof(onode)); }
static void onode_push(onode *nd, int32_t v) {
  if (nd->n_inds == nd->cap_inds) {
    nd->cap_inds = nd->cap_inds ? nd->cap_inds * 2 : 16;
    nd->pt_inds = (int32_t *)realloc(nd->pt_inds, nd->cap_inds * sizeof(int32_t));
  }
  nd->pt_inds[nd->n_inds++] = v;
}
static void onode_free(onode *nd) {
  if (!nd->is_leaf) for (int i = 0; i < 8; i++) if (nd->oct[i]) onode_free(nd->oct[i]);
  free(nd->pt_inds); free(nd);
}
/* get_octant_box (octree_ball_query.cpp:60-82): child size = parent/2, centre +- size/2 */
static void octant_box(const float *pa, int ind, float *out) {
  float w = pa[3] / 2, h = pa[4] / 2, l = pa[5] / 2;
  out[0] = ((ind >> 0) & 1) ? pa[0] + w / 2 : pa[0] - w / 2;
  out[1] = ((ind >> 1) & 1) ? pa[1] + h / 2 : pa[1] - h / 2;
  out[2] = ((ind >> 2) & 1) ? pa[2] + l / 2 : pa[2] - l / 2;
  out[3] = w; out[4] = h; out[5] = l;
}
static void build_octants(onode *pa, const float *points, int num_levels) {
  int level = pa->level + 1;
  if (level > num_levels) return;
  for (int i = 0; i < 8; i++) {
    onode *o = onode_new();
    octant_box(pa->box, i, o->box);
    o->level = level;
    o->is_leaf = (level == num_levels);
    pa->oct[i] = o;
  }
  for (int i = 0; i < pa->num_points; i++) {
    int pt = (pa->n_inds == 0) ? i : pa->pt_inds[i]; /* root has no list (:101-103) */
    int ix = points[3 * pt] < pa->box[0] ? 0 : 1;     /* get_octant_ind (:52-57) */
    int iy = points[3 * pt + 1] < pa->box[1] ? 0 : 1;
    int iz = points[3 * pt + 2] < pa->box[2] ? 0 : 1;
    onode *o = pa->oct[(iz << 2) + (iy << 1) + ix];
    onode_push(o, pt);
    o->num_points++;
  }
  for (int i = 0; i < 8; i++) build_octants(pa->oct[i], points, num_levels);
}

/* boxes [num_nodes,6], pt_inds [n], pt_start_len [num_leaves,2]; breadth-first export (:115-148) */
void orc_build_and_export_octree(const float *points, const float *xyzwhl, int num_points,
                                 int num_levels, float *boxes, int32_t *pt_inds,
                                 int32_t *pt_start_len) {
  onode *root = onode_new();
  memcpy(root->box, xyzwhl, 6 * sizeof(float));
  root->num_points = num_points;
  if (num_levels == 0) root->is_leaf = 0; /* reference never marks the root a leaf */
  build_octants(root, points, num_levels);
  size_t total = 0, lvl = 1;
  for (int i = 0; i <= num_levels; i++) { total += lvl; lvl *= 8; }
  onode **queue = (onode **)malloc(total * sizeof(onode *));
  size_t head = 0, tail = 0;
  queue[tail++] = root;
  int node_ind = 0, leaf_ind = 0, pt_count = 0;
  while (head < tail) {
    onode *nd = queue[head++];
    memcpy(boxes + (size_t)node_ind * 6, nd->box, 6 * sizeof(float));
    node_ind++;
    if (nd->is_leaf) {
      pt_start_len[leaf_ind * 2] = pt_count;
      pt_start_len[leaf_ind * 2 + 1] = nd->n_inds;
      leaf_ind++;
      for (int i = 0; i < nd->n_inds; i++) pt_inds[pt_count++] = nd->pt_inds[i];
    } else if (num_levels > 0) {
      for (int i = 0; i < 8; i++) queue[tail++] = nd->oct[i];
    }
  }
  free(queue);
  if (num_levels > 0) onode_free(root); else { free(root->pt_inds); free(root); }
}

/* ------------------------------------------------------------------------- */
/* octree_ball_query (octree_ball_query/octree_ball_query.cu:14-126)           */
/* ------------------------------------------------------------------------- */
#define ORC_NUM_NODES 585
#define ORC_NUM_LEAVES 512
#define ORC_MAX_SAMPLES 1000
static inline int is_intersection(const float *box, const float *pt, float r) {
  float x = box[0], y = box[1], z = box[2], w = box[3], h = box[4], l = box[5];
  float dist_x = fabsf(x - pt[0]), dist_y = fabsf(y - pt[1]), dist_z = fabsf(z - pt[2]);
  if (dist_x > (w / 2 + r)) return 0;
  if (dist_y > (h / 2 + r)) return 0;
  if (dist_z > (l / 2 + r)) return 0;
  if (dist_x <= (w / 2)) return 1;
  if (dist_y <= (h / 2)) return 1;
  if (dist_z <= (l / 2)) return 1;
  float dx = dist_x - w / 2, dy = dist_y - h / 2, dz = dist_z - l / 2;
  return fmaf(dz, dz, fmaf(dx, dx, dy * dy)) <= r * r; /* same contraction note as dist2_fma */
}

int64_t orc_octree_ball_query(const float *points, const float *boxes, const int32_t *pt_inds,
                              const int32_t *pt_start_len, int n, int mean_active, float radius,
                              int32_t *out_inds, int32_t *out_start_len) {
  const int num_mids = ORC_NUM_NODES - ORC_NUM_LEAVES;
  int64_t ntotals = 0, thr = (int64_t)n * mean_active;
  int32_t nb[ORC_MAX_SAMPLES];
  int actives[ORC_NUM_NODES];
  for (int index = 0; index < n; index++) {
    int count = 0;
    for (int i = 0; i < ORC_NUM_NODES; i++) actives[i] = 1;
    const float *cur = points + (size_t)index * 3;
    for (int node = 0; node < num_mids; node++) {
      int cur_active = actives[node];
      for (int oo = 0; oo < 8; oo++) {
        int octant = node * 8 + oo + 1;
        if (!cur_active) { actives[octant] = 0; continue; }
        int flag = is_intersection(boxes + (size_t)octant * 6, cur, radius);
        actives[octant] = flag;
        if (flag && octant >= num_mids) {
          int leaf = octant - num_mids;
          int start = pt_start_len[leaf * 2], end = start + pt_start_len[leaf * 2 + 1];
          for (int i = start; i < end; i++) {
            int p = pt_inds[i];
            const float *q = points + (size_t)p * 3;
            if (dist2_fma(cur[0], cur[1], cur[2], q[0], q[1], q[2]) < radius * radius) {
              if (count < ORC_MAX_SAMPLES) nb[count++] = p; else break;
            }
          }
        }
      }
    }
    out_start_len[index * 2] = (int32_t)ntotals;
    out_start_len[index * 2 + 1] = count;
    int64_t s = ntotals;
    ntotals += count;
    if (s >= thr) continue;
    int w = count;
    if (s + count >= thr) w = (int)(thr - s);
    for (int i = 0; i < w; i++) out_inds[s + i] = nb[i];
  }
  return ntotals;
}

/* ------------------------------------------------------------------------- */
/* sec_mean / sec_min / sec_max  (sec_mean/sec_mean.cu:13-85)                  */
/* ------------------------------------------------------------------------- */
void orc_sec_mean(const float *inp, const int32_t *offsets, int nP, int C, float *out) {
  for (int p = 0; p < nP; p++) {
    int s = offsets[p], e = offsets[p + 1];
    float count = (float)(e - s);
    for (int c = 0; c < C; c++) {
      float mean = 0;
      for (int i = s; i < e; i++) mean += (inp[(size_t)i * C + c] / count); /* divides each term */
      out[(size_t)p * C + c] = mean;
    }
  }
}
void orc_sec_min(const float *inp, const int32_t *offsets, int nP, int C, float *out) {
  for (int p = 0; p < nP; p++) {
    int s = offsets[p], e = offsets[p + 1];
    for (int c = 0; c < C; c++) {
      float v = (float)1e50; /* overflows to +inf, sec_mean.cu:48 */
      for (int i = s; i < e; i++) if (inp[(size_t)i * C + c] < v) v = inp[(size_t)i * C + c];
      out[(size_t)p * C + c] = v;
    }
  }
}
void orc_sec_max(const float *inp, const int32_t *offsets, int nP, int C, float *out) {
  for (int p = 0; p < nP; p++) {
    int s = offsets[p], e = offsets[p + 1];
    for (int c = 0; c < C; c++) {
      float v = (float)-1e50;
      for (int i = s; i < e; i++) if (inp[(size_t)i * C + c] > v) v = inp[(size_t)i * C + c];
      out[(size_t)p * C + c] = v;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* global_avg_pool fp / bp  (roipool/roipool.cu:12-60)                         */
/* ------------------------------------------------------------------------- */
void orc_global_avg_pool_fp(const float *feats, const int32_t *offsets, int nP, int C, float *out) {
  for (int p = 0; p < nP; p++) {
    int s = offsets[p], e = offsets[p + 1];
    int np = e - s;
    for (int c = 0; c < C; c++) {
      float v = 0;
      for (int i = s; i < e; i++) v += feats[(size_t)i * C + c];
      out[(size_t)p * C + c] = v / (float)np; /* sum then one divide */
    }
  }
}
void orc_global_avg_pool_bp(float *d_feats, const int32_t *offsets, const float *d_out, int nP,
                            int C) {
  for (int p = 0; p < nP; p++) {
    int s = offsets[p], e = offsets[p + 1];
    int np = e - s;
    for (int c = 0; c < C; c++)
      for (int i = s; i < e; i++)
        d_feats[(size_t)i * C + c] += d_out[(size_t)p * C + c] / (float)np;
  }
}

/* ------------------------------------------------------------------------- */
/* mask IoU / mask label (cal_iou_and_masklabel/cal_iou_and_masklabel.cu:9-104)*/
/* ------------------------------------------------------------------------- */
/* iou = (float)inter / ((float)total + 1e-5)  evaluated in double (1e-5 is a  */
/* double literal, :29-31 / :63-65), rounded to float on store.                 */
void orc_get_mask_iou_on_cluster(const int32_t *proposals_idx, const int32_t *proposals_offset,
                                 const int64_t *instance_labels, const int32_t *instance_pointnum,
                                 int nInstance, int nProposal, float *iou) {
  for (int p = 0; p < nProposal; p++) {
    int s = proposals_offset[p], e = proposals_offset[p + 1];
    int ptotal = e - s;
    for (int g = 0; g < nInstance; g++) {
      int itotal = instance_pointnum[g], inter = 0;
      for (int i = s; i < e; i++) if ((int)instance_labels[proposals_idx[i]] == g) inter++;
      iou[(size_t)p * nInstance + g] =
          (float)((float)inter / ((float)(ptotal + itotal - inter) + 1e-5));
    }
  }
}
void orc_get_mask_iou_on_pred(const int32_t *proposals_idx, const int32_t *proposals_offset,
                              const int64_t *instance_labels, const int32_t *instance_pointnum,
                              const float *mask_scores_sigmoid, int nInstance, int nProposal,
                              float *iou) {
  for (int p = 0; p < nProposal; p++) {
    int s = proposals_offset[p], e = proposals_offset[p + 1];
    int ptotal = 0;
    for (int i = s; i < e; i++) if (mask_scores_sigmoid[i] > 0.5) ptotal++;
    for (int g = 0; g < nInstance; g++) {
      int itotal = instance_pointnum[g], inter = 0;
      for (int i = s; i < e; i++)
        if (mask_scores_sigmoid[i] > 0.5 && (int)instance_labels[proposals_idx[i]] == g) inter++;
      iou[(size_t)p * nInstance + g] =
          (float)((float)inter / ((float)(ptotal + itotal - inter) + 1e-5));
    }
  }
}
/* mask_label pre-filled with -1 by the caller (functions.py:147) */
void orc_get_mask_label(const int32_t *proposals_idx, const int32_t *proposals_offset,
                        const int64_t *instance_labels, const int64_t *instance_cls,
                        const float *proposals_iou, int nInstance, int nProposal, float iou_thr,
                        float *mask_label) {
  for (int p = 0; p < nProposal; p++) {
    int s = proposals_offset[p], e = proposals_offset[p + 1];
    float max_iou = 0.f;
    int max_ind = 0;
    for (int g = 0; g < nInstance; g++) {
      if (proposals_iou[(size_t)p * nInstance + g] > max_iou) {
        if (instance_cls[g] != -100) { max_iou = proposals_iou[(size_t)p * nInstance + g]; max_ind = g; }
      }
    }
    if (max_iou >= iou_thr)
      for (int i = s; i < e; i++)
        mask_label[i] = ((int)instance_labels[proposals_idx[i]] == max_ind) ? 1.f : 0.f;
  }
}
