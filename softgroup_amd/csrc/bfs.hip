// bfs.hip -- soft-grouping clustering on the GPU, bit-identical to the reference's sequential
// CPU BFS (bfs_cluster/bfs_cluster.cpp:33-126: seeds in ascending index order, FIFO queue,
// neighbours in list order, clusters kept iff (float)size >= thr).
//
// Why this is not a plain connected-components pass: neighbour lists are
// capped at 1000 entries, so the graph can be DIRECTED; the reference then yields, for every
// point v, the cluster of the smallest-index point that can reach v ("min ancestor").  That
// labelling is what we compute:
//   A. union-find over the symmetric edges (hook the larger root under the smaller, so every
//      root is the minimum index of its set).  For lists that come from a radius query
//      (SG_LISTS_RADIUS) an edge u->v is known symmetric when neither list is capped; otherwise
//      membership of u in list(v) is checked (binary search when SG_LISTS_SORTED, else a scan).
//   B. only if asymmetric edges exist: min-label propagation across them to the fixed point.
//   C. cluster sizes, threshold test per segment (class), ids = rank among kept seeds,
//      offsets = prefix sum of kept sizes.
//   D. member ORDER: one workgroup per kept cluster replays the BFS level-synchronously with
//      the output segment itself as the FIFO queue.  Inside a level, every edge e of the
//      frontier has a position pos(e) = (rank of its source in the queue, index in its list);
//      a node is claimed by atomicMin(pos) and the next frontier is the claimed nodes in pos
//      order (wave ballot + popcount prefix per source node, workgroup prefix over nodes) --
//      exactly the order the sequential queue produces.
#include "common.h"
#include "scan.h"

namespace sg {

constexpr int kCap = SG_BALLQUERY_MAX_NEIGHBORS;
constexpr int kEmitThreads = 512;
constexpr int kEmitWaves = kEmitThreads / 64;

#define SG_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SG_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

struct BfsWs {
  int32_t *parent, *lab, *label, *size, *cid, *coff, *owner, *seeds, *ebase, *wcnt, *asym_nodes;
  int32_t *counters;  // [0] #asym source nodes  [1] changed flag  [2] nCluster  [3] sumNPoint
  void *scan_ws;
  size_t scan_bytes;
};

static bool bfs_carve(void *ws, size_t ws_bytes, int n, BfsWs *w) {
  Workspace a(ws, ws_bytes);
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  w->parent = a.take<int32_t>(nn);
  w->lab = a.take<int32_t>(nn);
  w->label = a.take<int32_t>(nn);
  w->size = a.take<int32_t>(nn);
  w->cid = a.take<int32_t>(nn);
  w->coff = a.take<int32_t>(nn);
  w->owner = a.take<int32_t>(nn);
  w->seeds = a.take<int32_t>(nn);
  w->ebase = a.take<int32_t>(nn);
  w->wcnt = a.take<int32_t>(nn);
  w->asym_nodes = a.take<int32_t>(nn);
  w->counters = a.take<int32_t>(64);
  w->scan_bytes = scan_workspace_bytes(n);
  w->scan_ws = a.take<char>(w->scan_bytes);
  return w->scan_ws != nullptr;
}

// ---------------------------------------------------------------- A. union-find
__device__ __forceinline__ int uf_find(int32_t *parent, int x) {
  int cur = SG_LD(&parent[x]);
  if (cur != x) {
    int prev = x, next;
    while (cur > (next = SG_LD(&parent[cur]))) {
      SG_ST(&parent[prev], next);  // path halving; pointers only ever move to smaller ancestors
      prev = cur;
      cur = next;
    }
  }
  return cur;
}

__device__ __forceinline__ void uf_union(int32_t *parent, int a, int b) {
  while (a != b) {
    if (a < b) { const int t = a; a = b; b = t; }  // a > b: hook a under b (roots stay minimal)
    const int old = atomicCAS(&parent[a], a, b);
    if (old == a) return;
    a = uf_find(parent, old);
    b = uf_find(parent, b);
  }
}

// is `key` in the list [lst, lst+len)?  sorted: per-lane binary search
__device__ __forceinline__ bool list_has_sorted(const int32_t *__restrict__ lst, int len, int key) {
  int lo = 0, hi = len;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int v = lst[mid];
    if (v < key) lo = mid + 1; else hi = mid;
  }
  return lo < len && lst[lo] == key;
}

__global__ void __launch_bounds__(256) bfs_init_kernel(int n, int32_t *parent, int32_t *size,
                                                      int32_t *owner) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    parent[i] = i;
    size[i] = 0;
    owner[i] = 0x7fffffff;
  }
}

// one wave per source node u; lanes stride over list(u)
__global__ void __launch_bounds__(256) bfs_union_kernel(const int32_t *__restrict__ idx,
                                                       const int32_t *__restrict__ start_len, int n,
                                                       int lists_sorted, int radius_lists,
                                                       int32_t *parent, int32_t *asym_nodes,
                                                       int32_t *counters) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int u = blockIdx.x * 4 + wave; u < n; u += gridDim.x * 4) {
    const int st = start_len[2 * u], ln = start_len[2 * u + 1];
    const bool u_capped = ln >= kCap;
    bool any_asym = false;
    for (int p0 = 0; p0 < ln; p0 += 64) {
      const int p = p0 + lane;
      const int v = p < ln ? idx[st + p] : u;
      bool sym = true;
      if (v != u) {
        const int vst = start_len[2 * v], vln = start_len[2 * v + 1];
        if (!radius_lists || u_capped || vln >= kCap) {
          if (lists_sorted) {
            sym = list_has_sorted(idx + vst, vln, u);
          } else {
            sym = false;
            for (int j = 0; j < vln; ++j)
              if (idx[vst + j] == u) { sym = true; break; }
          }
        }
        if (sym) {
          if (v < u) uf_union(parent, uf_find(parent, u), uf_find(parent, v));
        }
      }
      any_asym |= !sym;
    }
    if (__any(any_asym)) {
      if (lane == 0) asym_nodes[atomicAdd(&counters[0], 1)] = u;
    }
  }
}

__global__ void __launch_bounds__(256) bfs_flatten_kernel(int n, int32_t *parent, int32_t *lab) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // all unions are done (kernel boundary): chase to the root, then point straight at it
  int r = i;
  while (true) {
    const int p = SG_LD(&parent[r]);
    if (p == r) break;
    r = p;
  }
  lab[i] = r;  // valid label only at roots; non-roots get the root id (used as "root of i")
}
// after the flatten kernel: lab[i] == root(i).  root_of = copy kept in `parent` for phase B.
__global__ void __launch_bounds__(256) bfs_store_root_kernel(int n, const int32_t *__restrict__ lab,
                                                            int32_t *__restrict__ parent) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) parent[i] = lab[i];
}

// ---------------------------------------------------------------- B. asymmetric propagation
// lab[] holds, at roots, the current min-ancestor label of the set; root_of[] is frozen.
__global__ void __launch_bounds__(256) bfs_propagate_kernel(const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ start_len,
                                                           const int32_t *__restrict__ asym_nodes,
                                                           int n_asym,
                                                           const int32_t *__restrict__ root_of,
                                                           int32_t *lab, int32_t *counters) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int a = blockIdx.x * 4 + wave; a < n_asym; a += gridDim.x * 4) {
    const int u = asym_nodes[a];
    const int ru = root_of[u];
    const int st = start_len[2 * u], ln = start_len[2 * u + 1];
    for (int p = lane; p < ln; p += 64) {
      const int rv = root_of[idx[st + p]];
      if (rv == ru) continue;
      const int lu = SG_LD(&lab[ru]);
      if (lu < SG_LD(&lab[rv])) {
        if (atomicMin(&lab[rv], lu) > lu) SG_ST(&counters[1], 1);
      }
    }
  }
}

// ---------------------------------------------------------------- C. sizes / kept clusters
__global__ void __launch_bounds__(256) bfs_label_kernel(int n, const int32_t *__restrict__ root_of,
                                                       const int32_t *__restrict__ lab,
                                                       int32_t *__restrict__ label, int32_t *size) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int l = lab[root_of[i]];
  label[i] = l;
  atomicAdd(&size[l], 1);
}

__global__ void __launch_bounds__(256) bfs_zero_size_kernel(int n, int32_t *size) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) size[i] = 0;
}

// emit: cluster_offsets[cid+1] and the seed list
__global__ void __launch_bounds__(256) bfs_seed_kernel(int n, const int32_t *__restrict__ label,
                                                      const int32_t *__restrict__ size,
                                                      const int32_t *__restrict__ cid,
                                                      const int32_t *__restrict__ coff,
                                                      const int32_t *__restrict__ seg_of_point,
                                                      const float *__restrict__ seg_thr,
                                                      int32_t *__restrict__ seeds,
                                                      int32_t *__restrict__ cluster_offsets) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) cluster_offsets[0] = 0;
  if (i >= n || label[i] != i) return;
  const float thr = seg_thr[seg_of_point ? seg_of_point[i] : 0];
  if (static_cast<float>(size[i]) >= thr) {
    seeds[cid[i]] = i;
    cluster_offsets[cid[i] + 1] = coff[i] + size[i];
  }
}

// ---------------------------------------------------------------- D. ordered emission
// workgroup-wide exclusive scan of one int per thread (kEmitThreads threads)
__device__ __forceinline__ int wg_excl_scan(int v, int *lds, int *total) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int incl = wave_incl_scan(v);
  if (l == 63) lds[w] = incl;
  __syncthreads();
  int carry = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kEmitWaves; ++i) {
    const int x = lds[i];
    if (i < w) carry += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return carry + incl - v;
}

__global__ void __launch_bounds__(kEmitThreads) bfs_emit_kernel(
    const int32_t *__restrict__ idx, const int32_t *__restrict__ start_len,
    const int32_t *__restrict__ label, const int32_t *__restrict__ seeds,
    const int32_t *__restrict__ cluster_offsets, int n_cluster, int32_t *owner, int32_t *ebase,
    int32_t *wcnt, int32_t *cluster_idxs) {
  __shared__ int lds[kEmitWaves];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = blockIdx.x; c < n_cluster; c += gridDim.x) {
    const int seed = seeds[c];
    const int off = cluster_offsets[c];
    int32_t *Q = cluster_idxs + 2LL * off;  // pairs (cluster id, point); queue = column 1
    int32_t *eb = ebase + off, *wc = wcnt + off;
    if (threadIdx.x == 0) {
      SG_ST(&Q[0], c);
      SG_ST(&Q[1], seed);
      SG_ST(&owner[seed], -1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int head = 0, tail = 1;
    while (head < tail) {
      const int L = tail - head;
      // (1) edge base of every frontier node = exclusive prefix of list lengths
      int carry = 0;
      for (int q0 = 0; q0 < L; q0 += kEmitThreads) {
        const int q = q0 + threadIdx.x;
        int ln = 0;
        if (q < L) ln = start_len[2 * SG_LD(&Q[2 * (head + q) + 1]) + 1];
        int tot;
        const int ex = wg_excl_scan(ln, lds, &tot);
        if (q < L) SG_ST(&eb[head + q], carry + ex);
        carry += tot;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // (2) claim: every edge proposes its position to its (still unvisited) target
      for (int q = head + wave; q < tail; q += kEmitWaves) {
        const int u = SG_LD(&Q[2 * q + 1]);
        const int st = start_len[2 * u], ln = start_len[2 * u + 1];
        const int base = SG_LD(&eb[q]);
        for (int p = lane; p < ln; p += 64) {
          const int v = idx[st + p];
          if (label[v] == seed && SG_LD(&owner[v]) > base + p) atomicMin(&owner[v], base + p);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // (3) winners per frontier node
      for (int q = head + wave; q < tail; q += kEmitWaves) {
        const int u = SG_LD(&Q[2 * q + 1]);
        const int st = start_len[2 * u], ln = start_len[2 * u + 1];
        const int base = SG_LD(&eb[q]);
        int cnt = 0;
        for (int p0 = 0; p0 < ln; p0 += 64) {
          const int p = p0 + lane;
          const bool win = p < ln && SG_LD(&owner[idx[st + p]]) == base + p;
          cnt += __popcll(__ballot(win));
        }
        if (lane == 0) SG_ST(&wc[q], cnt);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // (4) output base of every frontier node (prefix of winner counts); reuse eb for it
      carry = 0;
      int total_new = 0;
      for (int q0 = 0; q0 < L; q0 += kEmitThreads) {
        const int q = q0 + threadIdx.x;
        const int cnt = q < L ? SG_LD(&wc[head + q]) : 0;
        int tot;
        const int ex = wg_excl_scan(cnt, lds, &tot);
        if (q < L) SG_ST(&wc[head + q], carry + ex);
        carry += tot;
      }
      total_new = carry;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // (5) append the winners in edge order; mark them visited
      for (int q = head + wave; q < tail; q += kEmitWaves) {
        const int u = SG_LD(&Q[2 * q + 1]);
        const int st = start_len[2 * u], ln = start_len[2 * u + 1];
        const int base = SG_LD(&eb[q]);
        int obase = tail + SG_LD(&wc[q]);
        for (int p0 = 0; p0 < ln; p0 += 64) {
          const int p = p0 + lane;
          int v = 0;
          bool win = false;
          if (p < ln) {
            v = idx[st + p];
            win = SG_LD(&owner[v]) == base + p;
          }
          const uint64_t bal = __ballot(win);
          if (win) {
            const int o = obase + mask_prefix(bal);
            SG_ST(&Q[2 * o], c);
            SG_ST(&Q[2 * o + 1], v);
          }
          obase += __popcll(bal);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // winners become visited only after every edge of the level has been examined
      for (int o = tail + threadIdx.x; o < tail + total_new; o += kEmitThreads)
        SG_ST(&owner[SG_LD(&Q[2 * o + 1])], -1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      head = tail;
      tail += total_new;
    }
  }
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_bfs_workspace_bytes(int n, int64_t n_edges) {
  (void)n_edges;
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  return 11 * align_up(nn * 4) + align_up(64 * 4) + align_up(scan_workspace_bytes(n)) + 256;
}

// Synchronises `stream` (the cluster count decides the size of the outputs).
int sg_bfs_cluster_label(const int32_t *bq_idxs, const int32_t *start_len, int n, int64_t n_edges,
                         int list_flags, const int32_t *seg_of_point, const float *seg_thr,
                         int n_seg, int32_t *n_cluster_host, int32_t *sum_npoint_host, void *ws,
                         size_t ws_bytes, sg_stream_t stream_) {
  (void)n_edges;
  SG_REQUIRE(n >= 0 && n_seg >= 1 && seg_thr != nullptr, "sg_bfs_cluster_label: bad arguments");
  hipStream_t stream = as_stream(stream_);
  BfsWs w;
  if (!bfs_carve(ws, ws_bytes, n, &w)) {
    set_error("sg_bfs_cluster_label: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  *n_cluster_host = 0;
  *sum_npoint_host = 0;
  if (n == 0) return SG_OK;
  const int grid = (n + 255) / 256;
  hipMemsetAsync(w.counters, 0, 64 * 4, stream);
  bfs_init_kernel<<<grid, 256, 0, stream>>>(n, w.parent, w.size, w.owner);
  bfs_union_kernel<<<grid_for(n, 4, 256 * 16), 256, 0, stream>>>(bq_idxs, start_len, n,
                                                               list_flags & SG_LISTS_SORTED,
                                                               list_flags & SG_LISTS_RADIUS,
                                                               w.parent, w.asym_nodes, w.counters);
  bfs_flatten_kernel<<<grid, 256, 0, stream>>>(n, w.parent, w.lab);
  bfs_store_root_kernel<<<grid, 256, 0, stream>>>(n, w.lab, w.parent);  // parent := root_of

  int32_t host_counters[4] = {0, 0, 0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      // asymmetric edges exist: propagate min labels to the fixed point, then redo the sizes
      const int n_asym = host_counters[0];
      while (true) {
        hipMemsetAsync(w.counters + 1, 0, 4, stream);
        bfs_propagate_kernel<<<grid_for(n_asym, 4, 256 * 16), 256, 0, stream>>>(
            bq_idxs, start_len, w.asym_nodes, n_asym, w.parent, w.lab, w.counters);
        int32_t changed = 0;
        hipMemcpyAsync(&changed, w.counters + 1, 4, hipMemcpyDeviceToHost, stream);
        if (hipStreamSynchronize(stream) != hipSuccess) return check_launch("bfs propagate");
        if (!changed) break;
      }
      bfs_zero_size_kernel<<<grid, 256, 0, stream>>>(n, w.size);
    }
    bfs_label_kernel<<<grid, 256, 0, stream>>>(n, w.parent, w.lab, w.label, w.size);
    // kept-cluster ids and offsets: two prefix sums over the seeds in index order
    const int32_t *label = w.label, *size = w.size;
    int32_t *cid = w.cid, *coff = w.coff;
    auto keep = [label, size, seg_of_point, seg_thr] __device__(int64_t i) -> int {
      if (label[i] != static_cast<int32_t>(i)) return 0;
      const float thr = seg_thr[seg_of_point ? seg_of_point[i] : 0];
      return static_cast<float>(size[i]) >= thr ? 1 : 0;  // bfs_cluster.cpp:73-81
    };
    auto keep_size = [keep, size] __device__(int64_t i) -> int { return keep(i) ? size[i] : 0; };
    int rc = exclusive_scan(keep, [cid] __device__(int64_t i, int v) { cid[i] = v; }, n,
                            w.counters + 2, w.scan_ws, w.scan_bytes, stream);
    if (rc != SG_OK) return rc;
    rc = exclusive_scan(keep_size, [coff] __device__(int64_t i, int v) { coff[i] = v; }, n,
                        w.counters + 3, w.scan_ws, w.scan_bytes, stream);
    if (rc != SG_OK) return rc;
    hipMemcpyAsync(host_counters, w.counters, sizeof(host_counters), hipMemcpyDeviceToHost, stream);
    if (hipStreamSynchronize(stream) != hipSuccess) return check_launch("sg_bfs_cluster_label");
    if (host_counters[0] == 0) break;
  }
  *n_cluster_host = host_counters[2];
  *sum_npoint_host = host_counters[3];
  return check_launch("sg_bfs_cluster_label");
}

int sg_bfs_cluster_emit(const int32_t *bq_idxs, const int32_t *start_len, int n,
                        const int32_t *seg_of_point, const float *seg_thr, int n_cluster,
                        int sum_npoint, int32_t *cluster_idxs, int32_t *cluster_offsets, void *ws,
                        size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(n >= 0 && n_cluster >= 0 && sum_npoint >= 0, "sg_bfs_cluster_emit: bad arguments");
  hipStream_t stream = as_stream(stream_);
  BfsWs w;
  if (!bfs_carve(ws, ws_bytes, n, &w)) {
    set_error("sg_bfs_cluster_emit: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  if (n_cluster == 0 || n == 0) {
    hipMemsetAsync(cluster_offsets, 0, 4, stream);
    return SG_OK;
  }
  bfs_seed_kernel<<<(n + 255) / 256, 256, 0, stream>>>(n, w.label, w.size, w.cid, w.coff,
                                                       seg_of_point, seg_thr, w.seeds,
                                                       cluster_offsets);
  bfs_emit_kernel<<<min(n_cluster, 4096), kEmitThreads, 0, stream>>>(
      bq_idxs, start_len, w.label, w.seeds, cluster_offsets, n_cluster, w.owner, w.ebase, w.wcnt,
      cluster_idxs);
  return check_launch("sg_bfs_cluster_emit");
}

}  // extern "C"
