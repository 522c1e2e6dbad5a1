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
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"
#include "scan.h"

namespace sg {

constexpr int kCap = SG_BALLQUERY_MAX_NEIGHBORS;
constexpr int kEmitThreads = 512;
constexpr int kEmitWaves = kEmitThreads / 64;

#define SG_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SG_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

struct BfsWs {
  int32_t *parent, *lab, *size, *cid, *coff, *owner, *seeds, *ebase, *wcnt, *asym_nodes;
  int4 *label;  // per point: (label, local slot, list start, list len)
  int2 *erec;   // per edge of a kept cluster: (target slot | target list len << 16, target list start)
  int32_t *counters;  // [0] #asym source nodes  [1] changed flag  [2] nCluster  [3] sumNPoint
  void *scan_ws;
  size_t scan_bytes;
  // frontier staging of the multi-workgroup replay of giant clusters (bfs_emit_big_kernel<true>): per
  // level parity the shared pool (n entries) followed by the workgroups' private slices
  int32_t *big_stage[2];
  unsigned long long *big_rec[2];
};

constexpr int kBigFastWgs = 16;        // most workgroups the FAST replay is launched with
constexpr int kBigSlice = 8192;        // = kBigEdges: winners of one cached level of one workgroup
constexpr int kBigClusterMin = 16384;  // = kBigMin = kOwnCap: no giant cluster among fewer points
// SG_BFS_BIG_FAST=0 (developer knob, read per call so that one test process compares both forms): the
// round-4 form of the replay, without its staging arrays
inline bool big_fast_on() {
  const char *e = getenv("SG_BFS_BIG_FAST");
  return !(e && atoi(e) == 0);
}
// SG_BFS_THIN=1 (developer knob, read per call): single-wave replay of thin levels in bfs_emit_kernel.
// Off by default: measured SLOWER than the workgroup-wide FAST levels (bfs_emit_kernel 520 us against
// ~360 us on the bench scene, one scan 5.01 against 4.78 ms; profiles/r06_bfs_thin_ab.txt) -- one wave
// walking the frontier's nodes with readlane loops costs more than the seven barriers it saves.
inline int thin_levels_on() {
  const char *e = getenv("SG_BFS_THIN");
  return e ? atoi(e) : 0;
}
// SG_BFS_BIG_LOCAL=0 (developer knob, read per call): without the LOCAL form of the giant clusters' replay
// (bfs_emit_big_local_kernel), which runs in front of bfs_emit_big_kernel by default
inline bool big_local_on() {
  const char *e = getenv("SG_BFS_BIG_LOCAL");
  return !(e && atoi(e) == 0);
}
inline size_t big_stage_entries(int n) {
  return n > kBigClusterMin && big_fast_on() ? static_cast<size_t>(n) + static_cast<size_t>(kBigFastWgs) * kBigSlice : 0;
}

static bool bfs_carve(void *ws, size_t ws_bytes, int n, int64_t n_edges, BfsWs *w) {
  Workspace a(ws, ws_bytes);
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  w->parent = a.take<int32_t>(nn);
  w->lab = a.take<int32_t>(nn);
  w->label = a.take<int4>(nn);
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
  w->erec = a.take<int2>(static_cast<size_t>(n_edges > 0 ? n_edges : 1));
  bool ok = w->scan_ws != nullptr && w->erec != nullptr;
  const size_t be = big_stage_entries(n);
  for (int i = 0; i < 2; ++i) {
    w->big_stage[i] = be ? a.take<int32_t>(be) : nullptr;
    w->big_rec[i] = be ? a.take<unsigned long long>(be) : nullptr;
    ok = ok && (be == 0 || (w->big_stage[i] != nullptr && w->big_rec[i] != nullptr));
  }
  return ok;
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

// one wave per source node u; lanes stride over list(u).
// Two passes: a cheap first pass links every node to a neighbour (bfs_hook_min_kernel; SAMPLE = 1, the
// first kSample entries of the list through the union-find, is what unsorted lists take), the trees are
// flattened (bfs_compress_kernel: parent[i] = root); SAMPLE = 0 then
// walks the whole list, but an edge whose target already carries the source's root -- almost all of them:
// a point has ~36 neighbours and its cluster is connected through any two -- costs ONE load of
// parent[v] instead of two pointer chases and a CAS (the single pass took 0.16 ms on the bench scene and
// 1.19 ms on the KITTI-like sweep).  A stale root only sends the edge down the full union path.
constexpr int kSample = 2;
template <int SAMPLE>
__global__ void __launch_bounds__(256) bfs_union_kernel(const int32_t *__restrict__ idx,
                                                       const int32_t *__restrict__ start_len, int n,
                                                       int lists_sorted, int radius_lists,
                                                       int32_t *parent, int32_t *asym_nodes,
                                                       int32_t *counters) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int u = blockIdx.x * 4 + wave; u < n; u += gridDim.x * 4) {
    const int st = start_len[2 * u], ln_all = start_len[2 * u + 1];
    const int ln = SAMPLE ? min(ln_all, kSample) : ln_all;
    const bool u_capped = ln_all >= kCap;
    bool any_asym = false;
    const int ru = SAMPLE ? -1 : SG_LD(&parent[u]);      // (flattened: the root of u as of the sampling pass)
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
          if (v < u && (SAMPLE || SG_LD(&parent[v]) != ru)) uf_union(parent, uf_find(parent, u), uf_find(parent, v));
        }
      }
      any_asym |= !sym;
    }
    if (!SAMPLE && __any(any_asym)) {
      if (lane == 0) asym_nodes[atomicAdd(&counters[0], 1)] = u;
    }
  }
}

// Pass 1 of the clustering: every node hooks itself under its SMALLEST neighbour (lists are sorted: the
// first entry), a plain store -- pointers strictly decrease, so this is a forest; no atomics, no pointer
// chasing.  (Sampling the first two entries through the union-find itself, the first version of this pass,
// cost 88 / 309 us on the bench scene / the KITTI-like sweep: everybody hooks towards the same low roots.)
// Same edge predicate as the full pass: a capped list's edge counts only if it is symmetric.
__global__ void __launch_bounds__(256) bfs_hook_min_kernel(const int32_t *__restrict__ idx,
                                                          const int32_t *__restrict__ start_len, int n,
                                                          int lists_sorted, int radius_lists, int32_t *parent) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= n) return;
  const int st = start_len[2 * u], ln = start_len[2 * u + 1];
  if (ln == 0 || !lists_sorted) return;
  const int v = idx[st];
  if (v >= u) return;
  if (!radius_lists || ln >= kCap || start_len[2 * v + 1] >= kCap)
    if (!list_has_sorted(idx + start_len[2 * v], start_len[2 * v + 1], u)) return;
  parent[u] = v;
}
// between the passes: parent[i] = root(i) (path halving on the way: nothing else runs)
__global__ void __launch_bounds__(256) bfs_compress_kernel(int n, int32_t *parent) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) SG_ST(&parent[i], uf_find(parent, i));
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
                                                       const int32_t *__restrict__ start_len,
                                                       int4 *__restrict__ node_rec, int32_t *size) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < n;
  const int l = valid ? lab[root_of[i]] : -1;
  // slot = running count of the label.  Neighbouring points mostly share their label, so the
  // lanes of a wave with equal labels are counted together: one atomic per distinct label.
  int slot = 0;
  uint64_t todo = __ballot(valid);
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const int ll = __shfl(l, leader, 64);
    const uint64_t same = __ballot(valid && l == ll) & todo;
    int base = 0;
    if (lane == leader) base = atomicAdd(&size[ll], __popcll(same));
    base = __shfl(base, leader, 64);
    if ((same >> lane) & 1ull) slot = base + __popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  if (!valid) return;
  // (cluster label, slot of the point inside its cluster, list start, list length): one 16-B
  // record per point so that the ordered emission fetches everything about a node in one load.
  // The slot indexes the cluster's visited/claim array when that array lives in LDS.
  node_rec[i] = make_int4(l, slot, start_len[2 * i], start_len[2 * i + 1]);
}

__global__ void __launch_bounds__(256) bfs_zero_size_kernel(int n, int32_t *size) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) size[i] = 0;
}

// Recovery path of the multi-workgroup replay (bfs_emit_big_kernel): if its bounded grid barrier
// ever gave up (workgroups not co-resident: `fail` set), the claim words of the giant clusters are
// put back to "unvisited" and the per-cluster kernel replays those clusters alone.  Gated on the
// device: without a failure both launches return at once, the host never waits.
__global__ void __launch_bounds__(256) bfs_owner_reset_kernel(int n, const int4 *__restrict__ label,
                                                             const int32_t *__restrict__ size,
                                                             int min_size, const int32_t *gate,
                                                             int32_t *__restrict__ owner) {
  if (SG_LD(gate) == 0) return;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    if (size[label[i].x] > min_size) owner[i] = 0x7fffffff;
}

// emit: cluster_offsets[cid+1] and the seed list
__global__ void __launch_bounds__(256) bfs_seed_kernel(int n, const int4 *__restrict__ label,
                                                      const int32_t *__restrict__ size,
                                                      const int32_t *__restrict__ cid,
                                                      const int32_t *__restrict__ coff,
                                                      const int32_t *__restrict__ seg_of_point,
                                                      const float *__restrict__ seg_thr,
                                                      int32_t *__restrict__ seeds,
                                                      int32_t *__restrict__ cluster_offsets) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) cluster_offsets[0] = 0;
  if (i >= n || label[i].x != i) return;
  const float thr = seg_thr[seg_of_point ? seg_of_point[i] : 0];
  if (static_cast<float>(size[i]) >= thr) {
    seeds[cid[i]] = i;
    cluster_offsets[cid[i] + 1] = coff[i] + size[i];
  }
}

// Edge records for the ordered emission: for every edge e = (v -> t) whose source lies in a kept
// cluster, everything the BFS replay needs about the target in one 8-byte read: its slot in the
// cluster's claim array (0xffff = t belongs to another cluster, never claimed; saturates at 0xfffe
// for clusters too large for the LDS array, which claim by point id instead), and its own list
// (start, length) so that a claimed target becomes a frontier node without a second lookup.
// One wave per source node; this is the only place that gathers node_rec[] at random.
__global__ void __launch_bounds__(256) bfs_edge_rec_kernel(const int32_t *__restrict__ idx,
                                                          const int4 *__restrict__ node_rec,
                                                          const int32_t *__restrict__ cid,
                                                          const int32_t *__restrict__ seeds,
                                                          int n, int n_cluster,
                                                          int2 *__restrict__ erec) {
  const int lane = threadIdx.x & 63;
  for (int v = (blockIdx.x * 256 + threadIdx.x) >> 6; v < n; v += (gridDim.x * 256) >> 6) {
    const int4 rv = node_rec[v];
    const int c = cid[rv.x];
    if (c >= n_cluster || seeds[c] != rv.x) continue;          // source not in a kept cluster
    for (int p = lane; p < rv.w; p += 64) {
      const int4 rt = node_rec[idx[rv.z + p]];
      erec[rv.z + p] = rt.x == rv.x ? make_int2(min(rt.y, 0xfffe) | (rt.w << 16), rt.z)
                                    : make_int2(0xffff, 0);
    }
  }
}

// LDS-DMA of 4 bytes per active lane to LDS[lds_dst + lane * 4]: used as a PREFETCH (the data is never
// read).  asm on purpose, like the conv kernel's metadata DMA: outside the compiler's vmcnt bookkeeping
// an older untracked load can only make a later counted wait longer, never wrong.
__device__ __forceinline__ void lds_prefetch_b32(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
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

constexpr int kOwnCap = 16384;     // cluster sizes up to this keep their claim array in LDS (64 KB)
constexpr int kFrontChunk = 1024;  // frontier nodes staged per chunk (st, len, edge base, winners)
constexpr int kE2 = 4096;           // edges of a fast level (slot, list start/len, edge id cached in LDS: 48 KB)
constexpr int kThinIter = 8;        // edges per lane of a thin level (one wave)
constexpr int kThinEdges = 64 * kThinIter;
static_assert(kThinEdges <= kFrontChunk, "a thin level's winners must fit the LDS frontier");

// One workgroup per kept cluster.  The output segment doubles as the FIFO queue (column 1 of
// cluster_idxs).  Per BFS level:
//   claim   every edge (frontier node rank, list position) proposes pos to its unvisited target
//           with atomicMin -- on an LDS array indexed by the target's slot when the cluster has
//           <= kOwnCap points, else on the global owner[] array;
//   count   winners per frontier node (ballot + popcount), prefix over nodes;
//   append  winners in edge order to the queue and mark them visited (-1).
// FAST level (cluster claims in LDS, <= kFrontChunk frontier nodes, <= kE2 edges): the frontier's
// (start,len) stay in LDS from the previous level's append; the level's edges are processed FLAT
// (one thread per edge), every thread reads one 8-byte edge record (bfs_edge_rec_kernel) and
// caches the target's slot, list (start,len) and the edge id in LDS, so the level costs ONE memory
// round trip and counting / appending the winners touch no global memory on the critical path.
// Winners are written to the queue as EDGE indices (fire-and-forget stores); they are turned into
// point ids in one parallel sweep when the cluster is done (or before a GENERIC level, which reads
// the queue).  Anything larger takes the GENERIC path (chunked, all state re-read from global),
// which typically means few, fat levels.
__global__ void __launch_bounds__(kEmitThreads) bfs_emit_kernel(
    const int32_t *__restrict__ idx, const int32_t *__restrict__ start_len,
    const int4 *__restrict__ node_rec, const int2 *__restrict__ erec,
    const int32_t *__restrict__ seeds, const int32_t *__restrict__ cluster_offsets, int n_cluster,
    int32_t *owner_g, int32_t *cluster_idxs, int32_t *stats, int skip_above, int only_above,
    const int32_t *gate /* null, or a word that must be non-zero for the launch to do anything */,
    int thin_mode /* 0: FAST levels only (default); 1: runs of thin levels on one wave; 2: the same with the
                     next level's record lines prefetched (SG_BFS_THIN) */) {
  const bool thin_on = thin_mode != 0, thin_prefetch = thin_mode == 2;
  if (gate != nullptr && SG_LD(gate) == 0) return;
  __shared__ int lds_scan[kEmitWaves];
  __shared__ int own_lds[kOwnCap];
  __shared__ int f_st[2][kFrontChunk], f_ln[2][kFrontChunk];
  __shared__ int f_eb[kFrontChunk], f_wc[kFrontChunk];
  __shared__ unsigned short ebuf[kE2];     // fast level, per edge: target slot
  __shared__ int e_st[kE2], e_g[kE2];      //   target list start, global edge index
  __shared__ unsigned short e_ln[kE2];     //   target list length
  __shared__ int pf_sink[kEmitThreads];    // landing zone of the edge-record prefetches (never read)
  __shared__ unsigned char thin_map[kThinEdges];   // thin levels: frontier node of every edge
  __shared__ int thin_state[8];                    // what a thin run hands back to the workgroup
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned pf_dst = __builtin_amdgcn_readfirstlane(
      static_cast<unsigned>(reinterpret_cast<uintptr_t>(pf_sink + wave * 64)));
  static_assert(sizeof(int2) == 8, "edge records are 8 bytes: 16 per 128-byte line");
  for (int c = blockIdx.x; c < n_cluster; c += gridDim.x) {
    const int seed = seeds[c];
    const int off = cluster_offsets[c];
    const int size = cluster_offsets[c + 1] - off;
    if (size > skip_above || size <= only_above) continue;   // giant clusters: bfs_emit_big_kernel
    const bool own_in_lds = size <= kOwnCap;
    int32_t *Q = cluster_idxs + 2LL * off;  // pairs (cluster id, point); queue = column 1
    if (own_in_lds)
      for (int i = threadIdx.x; i < size; i += kEmitThreads) own_lds[i] = 0x7fffffff;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int4 rec = node_rec[seed];
      SG_ST(&Q[0], c);
      SG_ST(&Q[1], seed);
      if (own_in_lds) own_lds[rec.y] = -1; else SG_ST(&owner_g[seed], -1);
      f_st[0][0] = rec.z;
      f_ln[0][0] = rec.w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto rec2 = [&](int v) -> int2 { return *reinterpret_cast<const int2 *>(node_rec + v); };
    auto claim = [&](int v, int pos) {   // generic path: propose `pos` to an unvisited node
      const int2 ll = rec2(v);
      if (ll.x != seed) return;
      if (own_in_lds) { if (own_lds[ll.y] > pos) atomicMin(&own_lds[ll.y], pos); }
      else if (SG_LD(&owner_g[v]) > pos) atomicMin(&owner_g[v], pos);
    };
    auto owner_of = [&](int v) -> int {
      const int2 ll = rec2(v);
      if (ll.x != seed) return -2;
      return own_in_lds ? own_lds[ll.y] : SG_LD(&owner_g[v]);
    };
    auto mark_done = [&](int v) {
      if (own_in_lds) own_lds[rec2(v).y] = -1; else SG_ST(&owner_g[v], -1);
    };
    int head = 0, tail = 1;
    int n_fast = 0, n_gen = 0, sum_e = 0, max_l = 0;   // developer statistics (SG_BFS_STATS)
    int conv_lo = -1;       // queue entries [conv_lo, tail) hold edge indices, not point ids yet
    auto convert_pending = [&]() {
      if (conv_lo < 0) return;
      for (int o = conv_lo + threadIdx.x; o < tail; o += kEmitThreads)
        SG_ST(&Q[2 * o + 1], idx[SG_LD(&Q[2 * o + 1])]);
      conv_lo = -1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    };
    int cur = 0;            // which f_st/f_ln buffer holds the current frontier (if any)
    bool in_lds = true;     // frontier (start,len) of this level already in f_st/f_ln[cur]?
    bool thin_ok = thin_on; // (false right after a thin run that stopped at a level it does not take)
    while (head < tail) {
      const int L = tail - head;
      // ---------------- THIN levels: a run of them on ONE wave ----------------
      // The bench scene's clusters are ~1 000 points over ~100 levels: a level is ~10 frontier nodes and
      // a few hundred edges, and the FAST path below spends its 2.9 us per level on seven workgroup
      // barriers and three 8-wave scans around ONE memory round trip.  While the frontier has <= 64
      // nodes and <= kThinEdges edges, wave 0 replays level after level by itself -- scan by DPP, the
      // edge -> node map and the claims in LDS (operations of one wave reach the LDS in order: no
      // barrier), winners by ballot -- and the other seven waves wait at ONE barrier for the whole run.
      // Same claims (atomicMin of the edge rank on the target's slot), same winners, same order.
      if (thin_ok && own_in_lds && in_lds && L <= 64) {
        if (wave == 0) {
          int h = head, t = tail, cu = cur, cl = conv_lo, levels = 0, edges = 0;
          while (true) {
            const int Lw = t - h;
            if (Lw <= 0 || Lw > 64) break;
            const int ln = lane < Lw ? f_ln[cu][lane] : 0;
            const int incl = wave_incl_scan(ln);
            const int E = __builtin_amdgcn_readlane(incl, 63);
            if (E > kThinEdges) break;
            const int eb = incl - ln;
            if (lane < Lw) f_eb[lane] = eb;
            for (int q = 0; q < Lw; ++q) {                  // edge -> frontier node
              const int ln_q = __builtin_amdgcn_readlane(ln, q), eb_q = __builtin_amdgcn_readlane(eb, q);
              for (int p = lane; p < ln_q; p += 64) thin_map[eb_q + p] = static_cast<unsigned char>(q);
            }
            int2 r[kThinIter];
            int g[kThinIter];
#pragma unroll
            for (int j = 0; j < kThinIter; ++j) {           // every edge record of the level in flight at once
              const int e = j * 64 + lane;
              g[j] = 0;
              r[j] = make_int2(0xffff, 0);
              if (e < E) {
                const int q = thin_map[e];
                g[j] = f_st[cu][q] + (e - f_eb[q]);
                r[j] = erec[g[j]];
              }
            }
#pragma unroll
            for (int j = 0; j < kThinIter; ++j) {           // claims
              const int e = j * 64 + lane;
              const int slot = r[j].x & 0xffff;
              if (e < E && slot != 0xffff) {
                const int cur_owner = own_lds[slot];
                if (cur_owner > e) atomicMin(&own_lds[slot], e);
                if (thin_prefetch && cur_owner >= 0) {      // candidate of the next frontier: its records' lines
                  const int tl = r[j].x >> 16;
                  const char *first = reinterpret_cast<const char *>(erec + r[j].y);
                  const char *last = reinterpret_cast<const char *>(erec + r[j].y + max(tl, 1) - 1);
                  const uintptr_t l0 = reinterpret_cast<uintptr_t>(first) & ~static_cast<uintptr_t>(127);
                  const int nlines = static_cast<int>(((reinterpret_cast<uintptr_t>(last) & ~static_cast<uintptr_t>(127)) - l0) >> 7) + 1;
                  for (int k = 0; k < 4; ++k)
                    if (k < nlines) lds_prefetch_b32(reinterpret_cast<const void *>(l0 + 128u * k), pf_dst);
                }
              }
            }
            const int nxt = cu ^ 1;
            int t_new = 0;
#pragma unroll
            for (int j = 0; j < kThinIter; ++j) {           // winners in edge order: j major, lane minor
              const int e = j * 64 + lane;
              const int slot = r[j].x & 0xffff;
              const bool win = e < E && slot != 0xffff && own_lds[slot] == e;
              const uint64_t bal = __ballot(win);
              if (win) {
                const int oo = t_new + mask_prefix(bal);
                SG_ST(&Q[2 * (t + oo)], c);
                SG_ST(&Q[2 * (t + oo) + 1], g[j]);          // edge index; point id = idx[edge]
                f_st[nxt][oo] = r[j].y;                     // (t_new <= E <= kThinEdges <= kFrontChunk)
                f_ln[nxt][oo] = r[j].x >> 16;
                own_lds[slot] = -1;
              }
              t_new += __popcll(bal);
            }
            if (cl < 0 && t_new > 0) cl = t;
            h = t;
            t += t_new;
            cu = nxt;
            ++levels;
            edges += E;
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the queue entries are out before anyone reads them
          if (lane == 0) {
            thin_state[0] = h; thin_state[1] = t; thin_state[2] = cu; thin_state[3] = cl;
            thin_state[4] = levels; thin_state[5] = edges;
          }
        }
        __syncthreads();
        const int levels = thin_state[4];
        head = thin_state[0];
        tail = thin_state[1];
        cur = thin_state[2];
        conv_lo = thin_state[3];
        n_fast += levels;
        sum_e += thin_state[5];
        max_l = max(max_l, L);
        __syncthreads();            // (thin_state is rewritten by the next run)
        in_lds = true;              // a thin level's winners (<= kThinEdges) always fit the LDS frontier
        thin_ok = false;            // the level the run stopped at is for the paths below
        if (levels > 0) continue;
      }
      thin_ok = thin_on;
      // ---------------- FAST level ----------------
      // Flat over the level's edges: thread t owns edges t, t + 512, ... of the concatenated lists
      // (edge -> (frontier node, position) by a binary search over the nodes' edge bases in LDS),
      // so the edge records of a level of <= 512 edges -- the usual case: the bench scene's largest cluster has 122
      // levels of 477 edges on average -- are requested at once: ONE memory round trip per level, 3.3 us with the
      // search, the claims and the ranks.  (Requesting all of a thread's <= 8 records before the first is used, for
      // the levels above 512 edges: bfs_emit_kernel 432 against 380 us on that scene -- dropped.)
      int E = -1;
      if (in_lds && L <= kFrontChunk) {
        int carry = 0;
        for (int b0 = 0; b0 < L; b0 += kEmitThreads) {
          const int q = b0 + threadIdx.x;
          const int ln = q < L ? f_ln[cur][q] : 0;
          int tot;
          const int ex = wg_excl_scan(ln, lds_scan, &tot);
          if (q < L) f_eb[q] = carry + ex;
          carry += tot;
        }
        __syncthreads();
        E = carry;
      }
      if (E >= 0 && E <= kE2) {
        ++n_fast; sum_e += E; max_l = max(max_l, L);
        for (int e = threadIdx.x; e < E; e += kEmitThreads) {         // claim
          int lo = 0, hi = L;                                         // last q with f_eb[q] <= e
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (f_eb[mid] <= e) lo = mid; else hi = mid;
          }
          const int g = f_st[cur][lo] + (e - f_eb[lo]);
          const int2 r = erec[g];
          const unsigned short slot = static_cast<unsigned short>(r.x & 0xffff);
          ebuf[e] = slot;
          e_g[e] = g;
          e_st[e] = r.y;
          e_ln[e] = static_cast<unsigned short>(r.x >> 16);
          if (slot != 0xffffu) {
            if (own_in_lds) {
              const int cur_owner = own_lds[slot];
              if (cur_owner > e) atomicMin(&own_lds[slot], e);
              // A level is ONE dependent memory round trip (the edge records of its frontier), and
              // those records were written by another kernel on all 8 XCDs: ~1 us from the fabric.
              // Every still-unvisited target of this level is a candidate frontier node of the next:
              // request the lines of ITS edge records now (LDS-DMA into a sink: no register, nothing
              // waits for it), so that the next level finds them in this XCD's L2 / this CU's L1.
              if (cur_owner >= 0) {
                const int tl = r.x >> 16;
                const char *first = reinterpret_cast<const char *>(erec + r.y);
                const char *last = reinterpret_cast<const char *>(erec + r.y + max(tl, 1) - 1);
                const uintptr_t l0 = reinterpret_cast<uintptr_t>(first) & ~static_cast<uintptr_t>(127);
                const int nlines = static_cast<int>(((reinterpret_cast<uintptr_t>(last) & ~static_cast<uintptr_t>(127)) - l0) >> 7) + 1;
                for (int j = 0; j < 4; ++j)
                  if (j < nlines) lds_prefetch_b32(reinterpret_cast<const void *>(l0 + 128u * j), pf_dst);
              }
            } else {          // cluster larger than the LDS claim array: claims by point id in global
              const int v = idx[g];
              if (SG_LD(&owner_g[v]) > e) atomicMin(&owner_g[v], e);
            }
          }
        }
        if (!own_in_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nxt = cur ^ 1;
        int t_new = 0;
        for (int b0 = 0; b0 < E; b0 += kEmitThreads) {                // winners, in edge order
          const int e = b0 + threadIdx.x;
          unsigned short slot = 0xffffu;
          bool win = false;
          int v = 0;
          if (e < E) {
            slot = ebuf[e];
            if (slot != 0xffffu) {
              if (own_in_lds) {
                win = own_lds[slot] == e;
              } else {
                v = idx[e_g[e]];
                win = SG_LD(&owner_g[v]) == e;
              }
            }
          }
          int tot;
          const int oo = t_new + wg_excl_scan(win ? 1 : 0, lds_scan, &tot);
          if (win) {
            SG_ST(&Q[2 * (tail + oo)], c);
            SG_ST(&Q[2 * (tail + oo) + 1], e_g[e]);      // edge index; point id = idx[edge]
            if (oo < kFrontChunk) {
              f_st[nxt][oo] = e_st[e];
              f_ln[nxt][oo] = e_ln[e];
            }
            // visited; only this edge can match pos, see header
            if (own_in_lds) own_lds[slot] = -1; else SG_ST(&owner_g[v], -1);
          }
          t_new += tot;
        }
        if (conv_lo < 0 && t_new > 0) conv_lo = tail;
        if (!own_in_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        head = tail;
        tail += t_new;
        cur = nxt;
        in_lds = t_new <= kFrontChunk;
        continue;
      }
      // ---------------- GENERIC level ----------------
      ++n_gen; max_l = max(max_l, L);
      convert_pending();          // it reads point ids from the queue
      // stage chunk `ch` of the frontier: list start/len into LDS, edge base = carry + prefix
      auto stage = [&](int ch, int carry_in) -> int {
        const int q0 = ch * kFrontChunk, cnt = min(kFrontChunk, L - q0);
        int carry = carry_in;
        for (int b0 = 0; b0 < cnt; b0 += kEmitThreads) {
          const int q = b0 + threadIdx.x;
          int ln = 0;
          if (q < cnt) {
            const int4 rec = node_rec[SG_LD(&Q[2 * (head + q0 + q) + 1])];
            f_st[0][q] = rec.z;
            ln = rec.w;
            f_ln[0][q] = ln;
          }
          int tot;
          const int ex = wg_excl_scan(ln, lds_scan, &tot);
          if (q < cnt) f_eb[q] = carry + ex;
          carry += tot;
        }
        __syncthreads();
        return carry;
      };
      const int n_chunks = (L + kFrontChunk - 1) / kFrontChunk;
      int carry = 0;
      for (int ch = 0; ch < n_chunks; ++ch) {                         // pass 1: claims
        carry = stage(ch, carry);
        const int cnt = min(kFrontChunk, L - ch * kFrontChunk);
        for (int q = wave; q < cnt; q += kEmitWaves) {
          const int st = f_st[0][q], ln = f_ln[0][q], base = f_eb[q];
          for (int p = lane; p < ln; p += 64) claim(idx[st + p], base + p);
        }
        __syncthreads();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int appended = 0;
      carry = 0;
      for (int ch = 0; ch < n_chunks; ++ch) {                         // pass 2: winners -> queue
        carry = stage(ch, carry);
        const int cnt = min(kFrontChunk, L - ch * kFrontChunk);
        for (int q = wave; q < cnt; q += kEmitWaves) {
          const int st = f_st[0][q], ln = f_ln[0][q], base = f_eb[q];
          int wins = 0;
          for (int p0 = 0; p0 < ln; p0 += 64) {
            const int p = p0 + lane;
            wins += __popcll(__ballot(p < ln && owner_of(idx[st + p]) == base + p));
          }
          if (lane == 0) f_wc[q] = wins;
        }
        __syncthreads();
        int chunk_new = 0;
        for (int b0 = 0; b0 < cnt; b0 += kEmitThreads) {
          const int q = b0 + threadIdx.x;
          const int w_ = q < cnt ? f_wc[q] : 0;
          int tot;
          const int ex = wg_excl_scan(w_, lds_scan, &tot);
          if (q < cnt) f_wc[q] = chunk_new + ex;
          chunk_new += tot;
        }
        __syncthreads();
        for (int q = wave; q < cnt; q += kEmitWaves) {
          const int st = f_st[0][q], ln = f_ln[0][q], base = f_eb[q];
          int obase = tail + appended + f_wc[q];
          for (int p0 = 0; p0 < ln; p0 += 64) {
            const int p = p0 + lane;
            int v = 0;
            bool win = false;
            if (p < ln) {
              v = idx[st + p];
              win = owner_of(v) == base + p;
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
        appended += chunk_new;
        __syncthreads();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // winners become visited only after every edge of the level has been examined
      for (int o = tail + threadIdx.x; o < tail + appended; o += kEmitThreads)
        mark_done(SG_LD(&Q[2 * o + 1]));
      // hand the next frontier to the fast path when it fits
      const bool keep = own_in_lds && appended <= kFrontChunk;
      if (keep)
        for (int o = threadIdx.x; o < appended; o += kEmitThreads) {
          const int4 rec = node_rec[SG_LD(&Q[2 * (tail + o) + 1])];
          f_st[0][o] = rec.z;
          f_ln[0][o] = rec.w;
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      head = tail;
      tail += appended;
      cur = 0;
      in_lds = keep;
    }
    convert_pending();
    if (stats && threadIdx.x == 0 && c < 256) {
      stats[c * 8 + 0] = size; stats[c * 8 + 1] = n_fast; stats[c * 8 + 2] = n_gen;
      stats[c * 8 + 3] = sum_e; stats[c * 8 + 4] = max_l;
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------- D'. giant clusters
// A cluster with tens of thousands of points (a floor, a wall, a LiDAR ground plane) has BFS levels
// with 10^4..10^5 edges; replayed by ONE workgroup such a level is bound by a single CU's memory
// pipe (~45 us).  Here `big_wgs` workgroups replay the cluster together, level by level.
// Round 3 paid three grid barriers per level (claim | count | append) and tested every edge of a
// level three times; a 68 k-point floor has ~400 levels of ~17 k edges, so the barriers were the
// level.  Now TWO synchronisations per level and two edge sweeps:
//   claim   every edge (frontier rank q, list position p) proposes pos = q * 1024 + p (p < 1000) to
//           its unvisited target with a device-scope atomicMin -- the same total order as the
//           sequential queue, so the winner of a node is its BFS parent edge.            [grid barrier]
//   emit    every workgroup owns a contiguous range of the frontier: it counts the winners of its
//           nodes, takes a region of that size out of the level's staging pool with ONE atomicAdd
//           (regions land in arbitrary order; the pool holds at most the cluster's points), writes
//           its winners there in (node, position) order, marks them visited, and publishes (region
//           offset, length) tagged with the level number.                 [wait for all G records]
//   The next frontier is the concatenation of the regions in WORKGROUP order -- exactly the order
//   the sequential queue produces -- so no global prefix over winners is ever needed before the
//   write; ranks follow from the G lengths every workgroup reads anyway.  The output queue gets
//   each region as a copy once its position (tail + lengths of the workgroups before) is known.
// Every spin is bounded: a barrier / wait that does not complete sets `fail` and all workgroups
// leave (the giant clusters are then replayed by the per-cluster kernel, gated on the device).
constexpr int kBigWgsMax = 256;
constexpr int kBigMin = kOwnCap;        // clusters above this size take this path
constexpr int kBigNodes = 1024;         // frontier nodes / edges of a workgroup's range that the cached level holds
constexpr int kBigEdges = 8192;
constexpr int kBigDirectDegree = 32;    // FAST: mean list length of a level up to which claims are not filtered
static_assert(kBigSlice == kBigEdges && kBigClusterMin == kBigMin && kBfsGiantMin == kBigMin, "staging sizes follow the kernel's constants");
static_assert(kBigNodes <= 65535, "c_j holds node indices of a range as 16-bit");

// Everything the workgroups exchange (frontier regions, claims, records) is written with
// device-scope write-through stores / atomics and read with sc1 loads that bypass the CU's L1
// (SG_ST / SG_LD / atomicMin), so the barrier needs no L2 write-back or L1 invalidate
// (Guideline 16, form R1): every wave drains its stores, one lane arrives and polls.
__device__ __forceinline__ bool big_barrier(int32_t *bar, int &epoch, int32_t *fail, int *lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int target = (epoch + 1) * static_cast<int>(gridDim.x);
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0u) {
        if (spins > (1u << 24) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    *lds_flag = ok;
  }
  ++epoch;
  __syncthreads();
  return *lds_flag != 0;
}

// The same rendezvous without the arrival counter (SG_BFS_BIG_FLAGS=1; off by default, measured no faster): every workgroup
// stores the level's tag into ITS flag word once its claims have drained, and polls all G flags with one
// coalesced load -- store -> visible -> poll is ONE trip through the fabric where atomic arrive -> return ->
// poll is two.  Flags only grow (a workgroup that is already a level ahead still satisfies the wait).
__device__ __forceinline__ bool big_flags_barrier(unsigned *flags, unsigned tag, int32_t *fail, int *lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int G = gridDim.x;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (lane == 0) __hip_atomic_store(flags + blockIdx.x, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    for (int w0 = 0; w0 < G && ok; w0 += 64) {
      const int w = w0 + lane;
      unsigned spins = 0;
      while (true) {
        const bool ready = w >= G || __hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= tag;
        if (__all(ready)) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0u &&
            (spins > (1u << 24) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    if (lane == 0) *lds_flag = ok;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// sync words: [0] barrier counter, [1] fail, [2..3] staging pool heads (per parity),
// [64 ..] records: rec[parity][workgroup] = two self-validating 64-bit words (tag << 32 | value):
// region offset and region length
constexpr int kBigRecAt = 64;
constexpr int kBigFlagsAt = kBigRecAt + 8 * kBigWgsMax;      // claim-done flags, one word per workgroup
constexpr int kBigSyncWords = kBigFlagsAt + kBigWgsMax;

//
// FAST (the default; SG_BFS_BIG_FAST=0 selects the round-4 form): a level is a chain of dependent memory
// round trips between 16 workgroups on 8 XCDs (~0.8 us each through the fabric) and nothing else, so
// three of them are taken out of the cached level:
//   * a frontier entry carries its node record -- (list start, list length), which the edge record of the
//     winning edge already holds -- in a second staging array, so the next level's claim needs ONE load
//     per node (staged record) instead of two dependent ones (staged id -> node_rec[id]);
//   * the claim of a level with short lists issues its atomicMin straight away (no filtering load of the
//     claim word before it);
//   * a workgroup's region of the next frontier of a cached level (<= kBigEdges winners) is a fixed
//     private slice behind the shared pool, so no atomicAdd on the pool head sits between deciding the
//     winners and writing them, and ONE sweep decides, ranks and writes.
typedef unsigned long long u64;
__device__ __forceinline__ u64 pack_rec(int z, int w) {
  return static_cast<u64>(static_cast<unsigned>(z)) | (static_cast<u64>(static_cast<unsigned>(w)) << 32);
}

template <bool FAST>
__global__ void __launch_bounds__(kEmitThreads) bfs_emit_big_kernel(
    const int32_t *__restrict__ idx, const int4 *__restrict__ node_rec, const int2 *__restrict__ erec,
    const int32_t *__restrict__ seeds, const int32_t *__restrict__ cluster_offsets, int n_cluster,
    int32_t *owner_g, int32_t *wcnt, int32_t *stage0, int32_t *stage1, u64 *srec0, u64 *srec1, int priv_base,
    int32_t *cluster_idxs, int32_t *sync, bool flag_barrier, bool trace_on, const int32_t *gate) {
  if (gate != nullptr && *gate == 0) return;      // (the LOCAL form completed: nothing to redo)
  __shared__ int lds_scan[kEmitWaves];
  __shared__ int lds_flag, lds_off;
  __shared__ int node_off[kEmitThreads];
  __shared__ int pre[kBigWgsMax + 1], offs[kBigWgsMax];
  // cached level (this workgroup's range has <= kBigNodes nodes and <= kBigEdges edges, the usual
  // case): list starts and edge bases of its nodes and the target of every edge stay in LDS from
  // the claim sweep to the emit sweep, which then costs ONE memory round trip (the claim words)
  __shared__ int c_st[kBigNodes], c_eb[kBigNodes + 1];
  __shared__ int c_t[kBigEdges];
  __shared__ unsigned short c_j[FAST ? kBigEdges : 1];       // FAST: the node (index in the range) of every edge
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int G = gridDim.x, b = blockIdx.x;
  int32_t *bar = sync, *fail = sync + 1, *pool = sync + 2;
  unsigned long long *rec = reinterpret_cast<unsigned long long *>(sync + kBigRecAt);     // [2][G][2]
  int epoch = 0;
  unsigned tag = 0;            // level counter over the whole launch (never 0 in a record)
  // developer phase trace (SG_BFS_STATS; workgroup 0, thread 0; 100 MHz ticks summed over the cached levels):
  // [0] wait for the records  [1] ranks + own region to the queue + staged records  [2] claim sweep
  // [3] claim rendezvous  [4] emit sweep + drain  [5] cached levels
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
  const bool tracing = trace_on && b == 0 && threadIdx.x == 0;
  auto stamp = [&](int i) {
    if (tracing) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      ph[i] += now - t_prev;
      t_prev = now;
    }
  };
  for (int c = 0; c < n_cluster; ++c) {
    const int off = cluster_offsets[c];
    const int size = cluster_offsets[c + 1] - off;
    if (size <= kBigMin) continue;                               // uniform over the grid
    const int seed = seeds[c];
    int32_t *Q = cluster_idxs + 2LL * off;
    int par = 0;
    ++tag;
    // ---- level 0: the seed is workgroup 0's region of parity 0
    if (threadIdx.x == 0) {
      if (b == 0) {
        SG_ST(&stage0[0], seed);
        if constexpr (FAST) {
          const int4 r = node_rec[seed];
          SG_ST(&srec0[0], pack_rec(r.z, r.w));
        }
        SG_ST(&Q[0], c);
        SG_ST(&Q[1], seed);
        SG_ST(&owner_g[seed], -1);
        SG_ST(&pool[0], 1);
        SG_ST(&pool[1], 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&rec[(0 * G + b) * 2], (static_cast<unsigned long long>(tag) << 32), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&rec[(0 * G + b) * 2 + 1], (static_cast<unsigned long long>(tag) << 32) | (b == 0 ? 1u : 0u),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int tail = 0;       // queue entries written so far (the seed is counted with its level below)
    while (true) {
      if (tracing) t_prev = __builtin_amdgcn_s_memrealtime();
      // ---- wait for the G records of the current frontier, then ranks: pre[w] = nodes before workgroup w
      if (wave == 0) {
        int ok = 1;
        for (int w0 = 0; w0 < G; w0 += 64) {
          const int w = w0 + lane;
          unsigned long long a = 0, l = 0;
          unsigned spins = 0;
          while (true) {
            bool ready = true;
            if (w < G) {
              a = __hip_atomic_load(&rec[(par * G + w) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              l = __hip_atomic_load(&rec[(par * G + w) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ready = (a >> 32) == tag && (l >> 32) == tag;
            }
            if (__all(ready)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0u &&
                (spins > (1u << 24) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
              __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = 0;
              break;
            }
          }
          if (w < G) {
            offs[w] = static_cast<int>(a & 0xffffffffu);
            pre[w + 1] = static_cast<int>(l & 0xffffffffu);      // lengths first, prefix below
          }
          if (!ok) break;
        }
        if (lane == 0) lds_flag = ok;
      }
      __syncthreads();
      if (!lds_flag) return;
      stamp(0);
      if (threadIdx.x == 0) {
        pre[0] = 0;
        for (int w = 0; w < G; ++w) pre[w + 1] += pre[w];
      }
      __syncthreads();
      const int L = pre[G];
      const int my_pos = pre[b], my_len = pre[b + 1] - pre[b];
      int32_t *cur = par ? stage1 : stage0, *nxt = par ? stage0 : stage1;
      u64 *curr = par ? srec1 : srec0, *nxtr = par ? srec0 : srec1;      // FAST: node records of the entries
      // ---- this workgroup's region of the current frontier goes to the output queue
      // (the first kQueueRegs x 512 entries travel through registers: their loads are issued here, next to the
      //  staged-record loads of the claim below, and stored behind the claim sweep -- one fabric round trip
      //  less per level than load -> store -> load, profiles/r06_bfs_big_phases.txt)
      constexpr int kQueueRegs = 4;
      const int q_base = tail + my_pos;
      int qv[kQueueRegs];
#pragma unroll
      for (int k = 0; k < kQueueRegs; ++k) {
        const int i = threadIdx.x + k * kEmitThreads;
        qv[k] = i < my_len ? SG_LD(&cur[offs[b] + i]) : 0;
      }
      for (int i = threadIdx.x + kQueueRegs * kEmitThreads; i < my_len; i += kEmitThreads) {
        SG_ST(&Q[2 * (q_base + i)], c);
        SG_ST(&Q[2 * (q_base + i) + 1], SG_LD(&cur[offs[b] + i]));
      }
      tail += L;
      if (L == 0) {                                                // cluster complete (uniform)
        if (tracing)
          for (int i = 0; i < 6; ++i) reinterpret_cast<unsigned long long *>(sync + 16)[i] = ph[i];
        break;
      }
      const int lo = static_cast<int>(static_cast<long long>(L) * b / G);
      const int hi = static_cast<int>(static_cast<long long>(L) * (b + 1) / G);
      auto rec_at = [&](int q, int &st, int &ln) {                 // frontier rank -> (list start, list length)
        int w0 = 0, w1 = G;                                        // last w with pre[w] <= q
        while (w1 - w0 > 1) {
          const int mid = (w0 + w1) >> 1;
          if (pre[mid] <= q) w0 = mid; else w1 = mid;
        }
        const int at = offs[w0] + (q - pre[w0]);
        if constexpr (FAST) {
          const u64 r = SG_LD(&curr[at]);
          st = static_cast<int>(r & 0xffffffffu);
          ln = static_cast<int>(r >> 32);
        } else {
          const int4 r = node_rec[SG_LD(&cur[at])];
          st = r.z;
          ln = r.w;
        }
      };
      // ---- claim
      const int nn = hi - lo;
      int E = -1;                                                  // >= 0: cached level with E edges
      if (nn <= kBigNodes) {
        int carry = 0;
        for (int i0 = 0; i0 < nn; i0 += kEmitThreads) {
          const int i = i0 + threadIdx.x;
          int ln = 0;
          if (i < nn) {
            int st;
            rec_at(lo + i, st, ln);
            c_st[i] = st;
          }
          int tot;
          const int ex = wg_excl_scan(ln, lds_scan, &tot);
          if (i < nn) c_eb[i] = carry + ex;
          carry += tot;
        }
        if (threadIdx.x == 0) c_eb[nn] = carry;
        __syncthreads();
        if (carry <= kBigEdges) E = carry;
      }
      auto edge_node = [&](int e) {                                // last j with c_eb[j] <= e (lists may be empty)
        int j0 = 0, j1 = nn;
        while (j1 - j0 > 1) {
          const int mid = (j0 + j1) >> 1;
          if (c_eb[mid] <= e) j0 = mid; else j1 = mid;
        }
        return j0;
      };
      stamp(1);
      if (E >= 0) {
        const bool direct = E <= kBigDirectDegree * nn;
        for (int e = threadIdx.x; e < E; e += kEmitThreads) {      // flat over the range's edges
          const int jn = edge_node(e);
          const int p = e - c_eb[jn], g = c_st[jn] + p;
          int t = -1;
          const int ex = erec[g].x, tg = idx[g];                  // (both loads in flight together)
          if ((ex & 0xffff) != 0xffff) {                           // else: target in another cluster
            t = tg;
            const int pos = ((lo + jn) << 10) | p;
            // (sparse lists: the atomic goes out unfiltered, one round trip less; dense lists send most of
            //  their edges to visited nodes, ~100 per node and level: there the filtering load stays)
            if (FAST && direct) {
              atomicMin(&owner_g[t], pos);
            } else {
              if (SG_LD(&owner_g[t]) > pos) atomicMin(&owner_g[t], pos);
            }
          }
          c_t[e] = t;
          if constexpr (FAST) c_j[e] = static_cast<unsigned short>(jn);
        }
      } else {
        for (int q = lo + wave; q < hi; q += kEmitWaves) {
          int4 r;
          rec_at(q, r.z, r.w);
          for (int p = lane; p < r.w; p += 64) {
            const int g = r.z + p;
            if ((erec[g].x & 0xffff) == 0xffff) continue;          // target in another cluster
            const int t = idx[g];
            const int pos = (q << 10) | p;
            if (SG_LD(&owner_g[t]) > pos) atomicMin(&owner_g[t], pos);
          }
        }
      }
      if (b == 0 && threadIdx.x == 0) SG_ST(&pool[par ^ 1], 0);   // next level's pool (idle since two levels)
#pragma unroll
      for (int k = 0; k < kQueueRegs; ++k) {
        const int i = threadIdx.x + k * kEmitThreads;
        if (i < my_len) {
          SG_ST(&Q[2 * (q_base + i)], c);
          SG_ST(&Q[2 * (q_base + i) + 1], qv[k]);
        }
      }
      stamp(2);
      if (flag_barrier) {
        if (!big_flags_barrier(reinterpret_cast<unsigned *>(sync + kBigFlagsAt), tag, fail, &lds_flag)) return;
      } else if (!big_barrier(bar, epoch, fail, &lds_flag)) {
        return;
      }
      stamp(3);
      if (E >= 0) {
        // ---- emit of a cached level
        if constexpr (FAST) {
          // one sweep: the claim word and the winning edge's record (= the target's list) travel together,
          // winners are ranked in edge order = (node, position) order and written to the private slice
          const int base = priv_base + b * kBigEdges;
          int carry = 0;
          for (int e0 = 0; e0 < E; e0 += kEmitThreads) {
            const int e = e0 + threadIdx.x;
            bool win = false;
            int t = -1;
            int2 er = make_int2(0, 0);
            if (e < E) {
              t = c_t[e];
              if (t >= 0) {
                const int jn = c_j[e];
                const int p = e - c_eb[jn];
                er = erec[c_st[jn] + p];
                win = SG_LD(&owner_g[t]) == (((lo + jn) << 10) | p);
              }
            }
            int tot;
            const int ex = wg_excl_scan(win ? 1 : 0, lds_scan, &tot);
            if (win) {
              SG_ST(&nxt[base + carry + ex], t);
              SG_ST(&nxtr[base + carry + ex], pack_rec(er.y, static_cast<int>(static_cast<unsigned>(er.x) >> 16)));
              SG_ST(&owner_g[t], -1);                              // only this edge matches pos
            }
            carry += tot;
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          stamp(4);
          if (tracing) ++ph[5];
          ++tag;
          par ^= 1;
          if (threadIdx.x == 0) {
            __hip_atomic_store(&rec[(par * G + b) * 2], (static_cast<unsigned long long>(tag) << 32) | static_cast<unsigned>(base),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&rec[(par * G + b) * 2 + 1],
                               (static_cast<unsigned long long>(tag) << 32) | static_cast<unsigned>(carry),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          continue;
        }
        // round-4 form: one sweep over the claim words decides the winners (losers and foreign targets
        // become -1 in LDS), one atomicAdd takes the region out of the shared pool, one LDS-only sweep
        // writes the winners in edge order = (node, position) order
        int total = 0;
        for (int e0 = 0; e0 < E; e0 += kEmitThreads) {
          const int e = e0 + threadIdx.x;
          bool win = false;
          if (e < E) {
            const int t = c_t[e];
            if (t >= 0) {
              const int jn = edge_node(e);
              win = SG_LD(&owner_g[t]) == (((lo + jn) << 10) | (e - c_eb[jn]));
            }
            if (!win) c_t[e] = -1;
          }
          total += __popcll(__ballot(win));
        }
        if (lane == 0) lds_scan[wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) {
          int t = 0;
          for (int w = 0; w < kEmitWaves; ++w) t += lds_scan[w];
          lds_off = t > 0 ? __hip_atomic_fetch_add(&pool[par ^ 1], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
          lds_scan[0] = t;
        }
        __syncthreads();
        const int region = lds_off, region_len = lds_scan[0];
        __syncthreads();
        int carry = region;
        for (int e0 = 0; e0 < E; e0 += kEmitThreads) {
          const int e = e0 + threadIdx.x;
          const int t = e < E ? c_t[e] : -1;
          int tot;
          const int ex = wg_excl_scan(t >= 0 ? 1 : 0, lds_scan, &tot);
          if (t >= 0) {
            SG_ST(&nxt[carry + ex], t);
            SG_ST(&owner_g[t], -1);                                // only this edge matches pos
          }
          carry += tot;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++tag;
        par ^= 1;
        if (threadIdx.x == 0) {
          __hip_atomic_store(&rec[(par * G + b) * 2], (static_cast<unsigned long long>(tag) << 32) | static_cast<unsigned>(region),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&rec[(par * G + b) * 2 + 1],
                             (static_cast<unsigned long long>(tag) << 32) | static_cast<unsigned>(region_len),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        continue;
      }
      // ---- emit, pass 1: winners per node of this workgroup's range
      int my_total = 0;
      for (int q = lo + wave; q < hi; q += kEmitWaves) {
        int4 r;
        rec_at(q, r.z, r.w);
        int wins = 0;
        for (int p0 = 0; p0 < r.w; p0 += 64) {
          const int p = p0 + lane;
          bool win = false;
          if (p < r.w) {
            const int g = r.z + p;
            if ((erec[g].x & 0xffff) != 0xffff) win = SG_LD(&owner_g[idx[g]]) == ((q << 10) | p);
          }
          wins += __popcll(__ballot(win));
        }
        if (lane == 0) SG_ST(&wcnt[q], wins);
        my_total += wins;                                        // per wave (uniform over its lanes)
      }
      if (lane == 0) lds_scan[wave] = my_total;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kEmitWaves; ++w) t += lds_scan[w];
        lds_off = t > 0 ? __hip_atomic_fetch_add(&pool[par ^ 1], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        lds_scan[0] = t;
      }
      __syncthreads();
      const int region = lds_off, region_len = lds_scan[0];
      __syncthreads();
      // ---- emit, pass 2: winners in (node, position) order into the region; they become visited
      int carry = region;
      for (int c0 = lo; c0 < hi; c0 += kEmitThreads) {
        const int q_mine = c0 + threadIdx.x;
        const int cnt = q_mine < hi ? SG_LD(&wcnt[q_mine]) : 0;
        int chunk_total;
        const int ex = wg_excl_scan(cnt, lds_scan, &chunk_total);
        node_off[threadIdx.x] = carry + ex;
        __syncthreads();
        const int n_here = min(kEmitThreads, hi - c0);
        for (int jn = wave; jn < n_here; jn += kEmitWaves) {
          const int q = c0 + jn;
          int4 r;
          rec_at(q, r.z, r.w);
          int o = node_off[jn];
          for (int p0 = 0; p0 < r.w; p0 += 64) {
            const int p = p0 + lane;
            bool win = false;
            int t = 0;
            int2 er = make_int2(0, 0);
            if (p < r.w) {
              const int g = r.z + p;
              er = erec[g];
              if ((er.x & 0xffff) != 0xffff) {
                t = idx[g];
                win = SG_LD(&owner_g[t]) == ((q << 10) | p);
              }
            }
            const uint64_t bal = __ballot(win);
            if (win) {
              SG_ST(&nxt[o + mask_prefix(bal)], t);
              if constexpr (FAST)
                SG_ST(&nxtr[o + mask_prefix(bal)], pack_rec(er.y, static_cast<int>(static_cast<unsigned>(er.x) >> 16)));
              SG_ST(&owner_g[t], -1);                            // only this edge matches pos
            }
            o += __popcll(bal);
          }
        }
        carry += chunk_total;
        __syncthreads();
      }
      // ---- publish this workgroup's region of the next frontier
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ++tag;
      par ^= 1;
      if (threadIdx.x == 0) {
        __hip_atomic_store(&rec[(par * G + b) * 2], (static_cast<unsigned long long>(tag) << 32) | static_cast<unsigned>(region),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&rec[(par * G + b) * 2 + 1],
                           (static_cast<unsigned long long>(tag) << 32) | static_cast<unsigned>(region_len),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // (all workgroups leave a cluster together: the records of its last, empty level were seen by all)
    if (!big_barrier(bar, epoch, fail, &lds_flag)) return;
  }
}

// ---------------------------------------------------------------- D''. giant clusters, LOCAL form (round 6)
// bfs_emit_big_kernel pays ~9.5 us per level: five dependent trips through the fabric -- wait for the G records,
// fetch the staged records of a re-partitioned range, claim sweep, claim rendezvous, emit sweep + drain
// (profiles/r06_bfs_big_phases.txt), ~2 us each.  The first two exist only because the frontier is re-partitioned
// evenly EVERY level.  Here a workgroup keeps ITS OWN children: the frontier of level L+1 that workgroup w holds is
// the winners of its level-L nodes in (node, position) order, i.e. a CONTIGUOUS piece of the queue order, and the
// pieces follow each other in workgroup order (induction over the levels: children of an earlier piece come before
// children of a later one).  So the claim key of an edge is (w, local node index, position) -- no global ranks, no
// waiting for anybody's record, no remote fetch: a level is claim sweep | ONE grid rendezvous | emit sweep.
//   * claims of level L go to owner[L & 1][t] (two arrays): a workgroup already claiming for level L+1 cannot disturb
//     a slower one that still decides its level-L winners; a winner is marked visited (-1) in both.
//   * filter: a bit per point in LDS for the targets this workgroup ever sent a claim to (visited one level later,
//     whoever won): edges to them -- most edges of a dense cluster -- never leave the CU again.
//   * balance: every `every`-th level (and the first `warm`) is a RE-PARTITION level, done like the old kernel's
//     levels (records of all workgroups, even ranges over the global order, keys by global rank).  Measured on the
//     kitti cluster, 16 workgroups: every = 2 / 4 / 8 / 16 -> 2.61 / 2.56 / 2.80 / 3.36 ms (the rendezvous waits
//     1.97 / 2.45 / 3.12 / 4.61 us for the slowest workgroup); default 4.
//   * the queue: a workgroup's winners of a level go to a region of ONE pool (atomicAdd on its head; the answer is
//     first needed behind the next level's claim sweep); (offset, length) of every (level, workgroup) lands in a
//     table, and when the cluster is complete every workgroup copies ITS regions to their places in the output, all
//     entries at once (positions = prefix over the table in (level, workgroup) order).
//   * termination: frontier totals per level in a 4-slot ring of counters, read behind the rendezvous.
//   * limits: <= kLF frontier nodes per workgroup and level (kLE of its edges are cached between the sweeps, the
//     rest worked out twice), <= tab_levels levels; beyond them (or a wait that times out) `fail` is set, everybody
//     leaves, and bfs_emit_big_kernel replays the giant clusters (gated on that word).
// Measured (profiles/r06_bfs_local.txt): the kitti scene's cluster (299 levels) 2.97 -> 2.55 ms, the stpls3d one
// (423 levels) 3.07 -> 2.67 ms; per level 8.0 / 5.9 us = re-partition 0.8 / 0.7 (averaged over all levels), edge
// bases + claim sweep 2.0 / 1.3, drain of the claims 0.7 / 0.5, rendezvous 2.4 / 2.0, emit sweep 1.4 / 0.9, totals +
// publication 0.6 / 0.6.  What is left is the rendezvous (arrival skew between the workgroups, not the counter).
constexpr int kLF = 2048;
constexpr int kLE = 8192;
constexpr int kVisWords = 16384;     // the workgroup's visited filter: one bit per point, up to 524 288 points
constexpr int kLevSeg = 2048;        // levels whose output positions one pass of the final copy holds in LDS
constexpr int kLocalLdsInts = 5 * kLF + 8 + kLE + kLE / 2;
constexpr int kSweepU = 4;           // edges per thread whose loads are in flight together in the two sweeps

__global__ void __launch_bounds__(kEmitThreads) bfs_emit_big_local_kernel(
    const int32_t *__restrict__ idx, const int4 *__restrict__ node_rec, const int2 *__restrict__ erec,
    const int32_t *__restrict__ seeds, const int32_t *__restrict__ cluster_offsets, int n_cluster,
    int32_t *owner0, int32_t *owner1, int32_t *pool_ids, unsigned long long *pool_rec, unsigned long long *tab,
    int tab_levels, int32_t *cluster_idxs, int32_t *sync, int n, int every, int warm, bool trace_on) {
  typedef unsigned long long u64;
  __shared__ int lds_scan[kEmitWaves];
  __shared__ int lds_flag;
  __shared__ int pre[kBigWgsMax + 1], offs[kBigWgsMax];
  __shared__ int lds_raw[kLocalLdsInts];
  int (*f_st)[kLF] = reinterpret_cast<int (*)[kLF]>(lds_raw);                                            // [2][kLF]
  unsigned short (*f_ln)[kLF] = reinterpret_cast<unsigned short (*)[kLF]>(lds_raw + 2 * kLF);            // [2][kLF]
  int *f_eb = lds_raw + 3 * kLF;                  // [kLF + 1]
  int *w_t = f_eb + kLF + 8;                      // [kLF] ids of the winners of the last emit (until they are in the pool)
  int *c_t = w_t + kLF;                           // [kLE]
  unsigned short *c_j = reinterpret_cast<unsigned short *>(c_t + kLE);                                    // [kLE]
  // nodes this workgroup KNOWS to be visited (every target it ever sent a claim to is visited one level later):
  // edges to them -- most edges of a dense cluster -- are dropped in the claim sweep without touching memory.
  // Dynamic LDS, one bit per point of THIS call (19 KB at 150 k points: the workgroup fits next to others on a busy CU)
  extern __shared__ unsigned vis[];
  const bool use_vis = n <= kVisWords * 32;
  const int vis_words = use_vis ? (n + 31) >> 5 : 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int G = gridDim.x, b = blockIdx.x;
  int32_t *bar = sync, *fail = sync + 1, *pool_head = sync + 2, *tot = sync + 4;
  int epoch = 0;
  unsigned tagc = 0;          // tag of the current level's table row, unique over the launch, never 0
  // developer phase trace (SG_BFS_STATS; workgroup 0, thread 0; 100 MHz ticks): [0] re-partition (wait, ranks, records)
  // [1] edge bases + claim sweep  [2] pool write + rendezvous  [3] emit sweep  [4] totals, pool region, publication
  // [5] levels  [6] re-partition levels  [7] pool write + drain of the claims (then [2] is the rendezvous alone)
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  const bool tracing = trace_on && blockIdx.x == 0 && threadIdx.x == 0;
  auto stamp = [&](int i) {
    if (tracing) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      ph[i] += now - t_prev;
      t_prev = now;
    }
  };
  auto entry = [&](int L, int w) { return tab + (static_cast<size_t>(L) * G + w) * 2; };
  auto publish = [&](int L, unsigned tag, int off, int len) {      // (thread 0, stores drained by the caller)
    __hip_atomic_store(entry(L, b), (static_cast<u64>(tag) << 32) | static_cast<unsigned>(off), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(entry(L, b) + 1, (static_cast<u64>(tag) << 32) | static_cast<unsigned>(len), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  };
  auto give_up = [&](int why) {                   // (why: 2 levels, 3 frontier at a re-partition, 5 winners of a level)
    if (threadIdx.x == 0) __hip_atomic_store(fail, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (int c = 0; c < n_cluster; ++c) {
    const int off_c = cluster_offsets[c];
    const int size = cluster_offsets[c + 1] - off_c;
    if (size <= kBigMin) continue;                               // uniform over the grid
    const int seed = seeds[c];
    int32_t *Q = cluster_idxs + 2LL * off_c;
    const unsigned tag0 = tagc;                                   // level L of this cluster has tag tag0 + L + 1
    for (int i = threadIdx.x; i < vis_words; i += kEmitThreads) vis[i] = 0u;
    __syncthreads();
    // ---- level 0: the seed -- first entry of the queue, row 0 of the level table (workgroup 0's piece)
    // (measured and dropped: replaying the first, thin levels -- <= 4096 edges -- by workgroup 0 alone, claims in an
    //  LDS hash table, visited bits in LDS, no rendezvous: 6.8-7.1 us per level, no faster than a level of all
    //  workgroups, and only 12 / 58 of the 299 / 423 levels of the two benchmark clusters are that thin)
    constexpr int T0 = 1;
    if (threadIdx.x == 0) {
      if (b == 0) {
        const int4 r = node_rec[seed];
        __hip_atomic_store(&pool_rec[0], pack_rec(r.z, r.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SG_ST(&owner0[seed], -1);
        SG_ST(&owner1[seed], -1);
        Q[0] = c;
        Q[1] = seed;
        SG_ST(pool_head, 1);
        SG_ST(&tot[0], 0);
        SG_ST(&tot[1], 0);
        SG_ST(&tot[2], 0);
        SG_ST(&tot[3], 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      publish(0, tag0 + 1, 0, b == 0 ? 1 : 0);
    }
    int cur = 0, nf = 0, L = 0;
    bool pending = false;          // my current frontier's ids / records are not in the pool yet
    int pend_off = 0;              // (thread 0: the pool region of the pending piece, once the atomic has answered)
    bool dead = false;
    while (true) {
      if (L >= tab_levels) { give_up(2); dead = true; break; }
      const unsigned tag = tag0 + static_cast<unsigned>(L) + 1u;
      const bool reb = L < warm || (L % every) == 0;
      const bool next_reb = (L + 1) < warm || ((L + 1) % every) == 0;
      int keybase;
      if (tracing) t_prev = __builtin_amdgcn_s_memrealtime();
      if (reb) {
        // ---- re-partition: wait for the G entries of row L, ranks, my even range, its records from the pool
        if (wave == 0) {
          int ok = 1;
          for (int w0 = 0; w0 < G; w0 += 64) {
            const int w = w0 + lane;
            u64 a = 0, l = 0;
            unsigned spins = 0;
            while (true) {
              bool ready = true;
              if (w < G) {
                a = __hip_atomic_load(entry(L, w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                l = __hip_atomic_load(entry(L, w) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ready = (a >> 32) == tag && (l >> 32) == tag;
              }
              if (__all(ready)) break;
              __builtin_amdgcn_s_sleep(1);
              if ((++spins & 1023u) == 0u &&
                  (spins > (1u << 24) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
              }
            }
            if (w < G) {
              offs[w] = static_cast<int>(a & 0xffffffffu);
              pre[w + 1] = static_cast<int>(l & 0xffffffffu);
            }
            if (!ok) break;
          }
          if (lane == 0) lds_flag = ok;
        }
        __syncthreads();
        if (!lds_flag) return;
        if (threadIdx.x == 0) {
          pre[0] = 0;
          for (int w = 0; w < G; ++w) pre[w + 1] += pre[w];
        }
        __syncthreads();
        const int Ltot = pre[G];
        if (Ltot == 0) break;                                      // cluster complete (uniform)
        const int lo = static_cast<int>(static_cast<long long>(Ltot) * b / G);
        const int hi = static_cast<int>(static_cast<long long>(Ltot) * (b + 1) / G);
        nf = hi - lo;
        if (nf > kLF) { give_up(3); dead = true; break; }
        for (int i = threadIdx.x; i < nf; i += kEmitThreads) {
          const int q = lo + i;
          int w0 = 0, w1 = G;                                      // last w with pre[w] <= q
          while (w1 - w0 > 1) {
            const int mid = (w0 + w1) >> 1;
            if (pre[mid] <= q) w0 = mid; else w1 = mid;
          }
          const u64 r = __hip_atomic_load(&pool_rec[offs[w0] + (q - pre[w0])], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          f_st[cur][i] = static_cast<int>(r & 0xffffffffu);
          f_ln[cur][i] = static_cast<unsigned short>(r >> 32);
        }
        keybase = lo;
        __syncthreads();
      } else {
        keybase = b << 13;          // (kLF <= 8192 nodes per workgroup: disjoint key ranges, in workgroup order)
      }
      stamp(0);
      // ---- edge bases of my frontier
      int E = 0;
      {
        int carry = 0;
        for (int i0 = 0; i0 < nf; i0 += kEmitThreads) {
          const int i = i0 + threadIdx.x;
          const int ln = i < nf ? f_ln[cur][i] : 0;
          int t;
          const int ex = wg_excl_scan(ln, lds_scan, &t);
          if (i < nf) f_eb[i] = carry + ex;
          carry += t;
        }
        if (threadIdx.x == 0) f_eb[nf] = carry;
        __syncthreads();
        E = carry;
      }
      // (edges beyond the kLE the level cache holds are worked out again in the emit sweep)
      int32_t *own = (L & 1) ? owner1 : owner0;
      auto edge_node = [&](int e) {                                // last j with f_eb[j] <= e (lists may be empty)
        int j0 = 0, j1 = nf;
        while (j1 - j0 > 1) {
          const int mid = (j0 + j1) >> 1;
          if (f_eb[mid] <= e) j0 = mid; else j1 = mid;
        }
        return j0;
      };
      // ---- claim sweep (kSweepU edges per thread in flight: their loads go out together, one wait)
      {
        const bool direct = E <= kBigDirectDegree * nf;
        // (measured and dropped: the kSweepU searches stepping together branch-free, one body per number of chunks --
        //  claim sweep 2.3 against 2.0 us on the kitti cluster, 1.7 against 1.3 on the stpls3d one: the fixed trip
        //  count and the selects cost more than the overlap of the LDS reads gives)
        for (int e0 = threadIdx.x; e0 < E; e0 += kSweepU * kEmitThreads) {
          int jn[kSweepU], key[kSweepU], ex[kSweepU], tg[kSweepU], ow[kSweepU];
#pragma unroll
          for (int u = 0; u < kSweepU; ++u) {
            const int e = e0 + u * kEmitThreads;
            jn[u] = 0; key[u] = 0; ex[u] = 0xffff; tg[u] = 0;
            if (e < E) {
              jn[u] = edge_node(e);
              const int p = e - f_eb[jn[u]], g = f_st[cur][jn[u]] + p;
              key[u] = ((keybase + jn[u]) << 10) | p;
              ex[u] = erec[g].x;
              tg[u] = idx[g];
            }
          }
          bool mine_[kSweepU];
#pragma unroll
          for (int u = 0; u < kSweepU; ++u) {
            mine_[u] = (ex[u] & 0xffff) != 0xffff;                 // else: target in another cluster (or no edge)
            if (use_vis && mine_[u]) mine_[u] = ((vis[tg[u] >> 5] >> (tg[u] & 31)) & 1u) == 0u;
          }
          if (!direct && !use_vis) {
#pragma unroll
            for (int u = 0; u < kSweepU; ++u) ow[u] = mine_[u] ? SG_LD(&own[tg[u]]) : -1;
          }
#pragma unroll
          for (int u = 0; u < kSweepU; ++u) {
            const int e = e0 + u * kEmitThreads;
            const bool mine = mine_[u];
            if (mine && (direct || use_vis || ow[u] > key[u])) atomicMin(&own[tg[u]], key[u]);
            if (e < E && e < kLE) {
              c_t[e] = mine ? tg[u] : -1;
              c_j[e] = static_cast<unsigned short>(jn[u]);
            }
          }
        }
      }
      stamp(1);
      // ---- my current frontier (= the winners of my last emit) into the pool, if that was put off
      if (pending) {
        if (threadIdx.x == 0) lds_scan[0] = pend_off;             // (thread 0 holds the atomic's answer)
        __syncthreads();
        const int reg = lds_scan[0];
        __syncthreads();
        for (int i = threadIdx.x; i < nf; i += kEmitThreads) {
          SG_ST(&pool_ids[reg + i], w_t[i]);
          __hip_atomic_store(&pool_rec[reg + i], pack_rec(f_st[cur][i], f_ln[cur][i]), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (b == 0 && threadIdx.x == 0) SG_ST(&tot[(L + 2) & 3], 0);      // (idle since level L - 2)
      if (trace_on) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(7);
      }
      if (!big_barrier(bar, epoch, fail, &lds_flag)) return;
      if (pending) {
        if (threadIdx.x == 0) publish(L, tag, pend_off, nf);        // (its stores were drained by the rendezvous)
        pending = false;
      }
      if (!reb) {
        // frontier total of this level (added by everybody before the rendezvous): empty = cluster complete
        if (threadIdx.x == 0) lds_scan[0] = SG_LD(&tot[L & 3]);
        __syncthreads();
        const int Ltot = lds_scan[0];
        __syncthreads();
        if (Ltot == 0) break;
      }
      stamp(2);
      // ---- emit sweep: winners in edge order = (node, position) order -> my next frontier
      int cnt = 0;
      {
        const int nxt = cur ^ 1;
        bool over = false;
        for (int e0 = 0; e0 < E; e0 += kSweepU * kEmitThreads) {
          int t[kSweepU], key[kSweepU], ow[kSweepU];
          int2 er[kSweepU];
#pragma unroll
          for (int u = 0; u < kSweepU; ++u) {
            const int e = e0 + u * kEmitThreads + threadIdx.x;
            t[u] = -1; key[u] = 0; ow[u] = -1; er[u] = make_int2(0, 0);
            if (e < E) {
              int jn;
              if (e < kLE) {
                t[u] = c_t[e];
                jn = c_j[e];
              } else {
                jn = edge_node(e);
              }
              const int p = e - f_eb[jn], g = f_st[cur][jn] + p;
              key[u] = ((keybase + jn) << 10) | p;
              if (e >= kLE) {
                er[u] = erec[g];
                t[u] = (er[u].x & 0xffff) != 0xffff ? idx[g] : -1;
              } else if (t[u] >= 0) {
                er[u] = erec[g];
              }
              if (t[u] >= 0) {
                ow[u] = SG_LD(&own[t[u]]);
                if (use_vis) atomicOr(&vis[t[u] >> 5], 1u << (t[u] & 31));      // visited by the end of this level
              }
            }
          }
          // (measured and dropped, after reading the ISA: the compiler drains everything outstanding at the end of each
          //  chunk's data-dependent branch, so the four chunks' loads do not overlap.  Every load unconditional from
          //  clamped indices: emit sweep 1.72 against 1.43 us (kitti), 1.29 against 0.86 (stpls3d) -- most lanes have no
          //  edge or a filtered one, and skipping their loads is worth more than overlapping the others'.  All
          //  load-dependent decisions before the first atomic / store of a trip: no gain either.)
          // (ranks: one workgroup scan per chunk of 512 edges; a ballot table with one rendezvous for the
          //  kSweepU chunks measured the same on fat levels and slower on thin ones)
#pragma unroll
          for (int u = 0; u < kSweepU; ++u) {
            if (e0 + u * kEmitThreads >= E) break;                  // (uniform)
            const bool win = t[u] >= 0 && ow[u] == key[u];
            int tt;
            const int ex = wg_excl_scan(win ? 1 : 0, lds_scan, &tt);
            if (win) {
              const int o = cnt + ex;
              if (o < kLF) {
                f_st[nxt][o] = er[u].y;
                f_ln[nxt][o] = static_cast<unsigned short>(static_cast<unsigned>(er[u].x) >> 16);
                w_t[o] = t[u];
              }
              SG_ST(&owner0[t[u]], -1);
              SG_ST(&owner1[t[u]], -1);
            }
            cnt += tt;
            over = over || cnt > kLF;
          }
        }
        if (over) { give_up(5); dead = true; break; }
      }
      stamp(3);
      // ---- totals and my pool region (the region's offset is first needed behind the next claim sweep, or now
      //      if the next level re-partitions)
      if (threadIdx.x == 0) {
        if (cnt > 0) atomicAdd(&tot[(L + 1) & 3], cnt);
        pend_off = cnt > 0 ? __hip_atomic_fetch_add(pool_head, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      }
      cur ^= 1;
      nf = cnt;
      ++L;
      pending = true;
      if (next_reb) {
        if (threadIdx.x == 0) lds_scan[0] = pend_off;
        __syncthreads();
        const int reg = lds_scan[0];
        __syncthreads();
        for (int i = threadIdx.x; i < nf; i += kEmitThreads) {
          SG_ST(&pool_ids[reg + i], w_t[i]);
          __hip_atomic_store(&pool_rec[reg + i], pack_rec(f_st[cur][i], f_ln[cur][i]), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) publish(L, tag + 1u, reg, nf);
        pending = false;
      }
      stamp(4);
      ph[5] += 1;
      ph[6] += reb ? 1 : 0;
    }
    if (dead) {                      // a limit was hit: everybody learns it at the next wait; leave now
      return;
    }
    // ---- the cluster is complete: L levels (0 .. L-1) have entries.  (An entry put off at the last level was
    //      published behind its rendezvous; `pending` pieces of an EMPTY frontier need nothing.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!big_barrier(bar, epoch, fail, &lds_flag)) return;
    {
      int *lv_dst = c_t;                               // [kLevSeg] output position of my piece of the level
      int *lv_off = c_t + kLevSeg;                     // [kLevSeg] its pool region
      int *lv_cum = reinterpret_cast<int *>(f_st);     // [kLevSeg + 1] my pieces' cumulative lengths (2 * kLF ints)
      static_assert(2 * kLevSeg <= kLE && kLevSeg + 1 <= 2 * kLF, "the final copy's tables live in the level caches");
      int base = T0;                                   // output entries before the segment (row 0 is in the queue already)
      for (int l0 = 1; l0 < L; l0 += kLevSeg) {
        const int nl = min(kLevSeg, L - l0);
        // per level: row total, my prefix inside the row, my (offset, length)
        int seg_carry = 0, cum_carry = 0;
        for (int i0 = 0; i0 < nl; i0 += kEmitThreads) {
          const int i = i0 + threadIdx.x;
          int rowtot = 0, mypre = 0, myoff = 0, mylen = 0;
          if (i < nl) {
            for (int w = 0; w < G; ++w) {
              const int len = static_cast<int>(__hip_atomic_load(entry(l0 + i, w) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffffu);
              if (w < b) mypre += len;
              if (w == b) mylen = len;
              rowtot += len;
            }
            myoff = static_cast<int>(__hip_atomic_load(entry(l0 + i, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffffu);
          }
          int t1, t2;
          const int ex1 = wg_excl_scan(rowtot, lds_scan, &t1);
          const int ex2 = wg_excl_scan(mylen, lds_scan, &t2);
          if (i < nl) {
            lv_dst[i] = base + seg_carry + ex1 + mypre;
            lv_off[i] = myoff;
            lv_cum[i] = cum_carry + ex2;
          }
          seg_carry += t1;
          cum_carry += t2;
        }
        if (threadIdx.x == 0) lv_cum[nl] = cum_carry;
        __syncthreads();
        // flat copy of my pieces: entry k of my concatenated pieces -> its level by binary search
        for (int k = threadIdx.x; k < cum_carry; k += kEmitThreads) {
          int j0 = 0, j1 = nl;                                     // last j with lv_cum[j] <= k
          while (j1 - j0 > 1) {
            const int mid = (j0 + j1) >> 1;
            if (lv_cum[mid] <= k) j0 = mid; else j1 = mid;
          }
          const int r = k - lv_cum[j0];
          const int id = SG_LD(&pool_ids[lv_off[j0] + r]);
          Q[2LL * (lv_dst[j0] + r)] = c;
          Q[2LL * (lv_dst[j0] + r) + 1] = id;
        }
        base += seg_carry;
        __syncthreads();
      }
    }
    tagc = tag0 + static_cast<unsigned>(L) + 2u;
    if (b == 0 && threadIdx.x == 0) {                 // (developer counters, SG_BFS_STATS)
      sync[8] += L;
      sync[9] += 1;
      if (tracing)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long *>(sync + 16)[i] = ph[i];
    }
    // the next cluster re-uses pool, table rows and counters: everybody is past its reads before anybody re-writes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!big_barrier(bar, epoch, fail, &lds_flag)) return;
  }
}

// ---- the giant clusters' replay runs NEXT TO the per-cluster kernel, on a side stream of the caller's
//      stream (the two touch disjoint clusters): one side stream and two events per (device, caller stream)
struct BfsSide {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, back = nullptr;
};
// sg_scan_grouping_pp's request (bfs_emit_defer, thread-local, consumed by the next sg_bfs_cluster_emit call of
// the thread): when the call replays giant clusters on the side stream, do NOT join -- the side stream waits for
// the caller's stream instead (so it sees the small clusters too) and is handed to the caller, who queues the rest
// of this class behind the replay and goes on with the next class on its own stream (bfs_emit_join at the end)
static thread_local BfsDefer *t_bfs_defer = nullptr;
static std::mutex g_bfs_mu;
static std::map<std::pair<int, hipStream_t>, BfsSide> g_bfs_side;
static BfsSide *bfs_side(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> g(g_bfs_mu);
  BfsSide &b = g_bfs_side[{dev, stream}];
  if (b.side == nullptr) {
    if (hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&b.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b.join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b.back, hipEventDisableTiming) != hipSuccess) {
      b = BfsSide();
      return nullptr;
    }
  }
  return &b;
}
void bfs_release_stream(int dev, hipStream_t stream) {      // sg_stream_release: the caller's stream is idle
  std::lock_guard<std::mutex> g(g_bfs_mu);
  auto it = g_bfs_side.find({dev, stream});
  if (it == g_bfs_side.end()) return;
  if (it->second.side) {
    hipStreamSynchronize(it->second.side);
    hipStreamDestroy(it->second.side);
    hipEventDestroy(it->second.fork);
    hipEventDestroy(it->second.join);
    hipEventDestroy(it->second.back);
  }
  g_bfs_side.erase(it);
}
// what sg_bfs_cluster_label learned about the largest kept cluster, for the sg_bfs_cluster_emit call that
// follows on the same thread with the same workspace (else: unknown, the emit assumes a giant cluster may exist)
struct BfsLabelNote {
  const void *ws = nullptr;
  int max_kept = -1;
};
static thread_local BfsLabelNote t_bfs_note;

void bfs_emit_defer(BfsDefer *d) { t_bfs_defer = d; }
int bfs_emit_join(const BfsDefer &d, hipStream_t stream) {
  if (!d.deferred) return SG_OK;
  BfsSide *side = bfs_side(stream);
  if (side == nullptr || side->side != d.side || hipEventRecord(side->join, side->side) != hipSuccess ||
      hipStreamWaitEvent(stream, side->join, 0) != hipSuccess) {
    set_error("bfs_emit_join: joining the side stream failed");
    return SG_ERR_LAUNCH;
  }
  return SG_OK;
}
int bfs_label_max_kept(const void *ws) { return t_bfs_note.ws == ws ? t_bfs_note.max_kept : -1; }

// points of a class whose level voxel l2p[p] lies in a KEPT cluster of the labelling that just ran on `ws`
// (= the rows pyramid_inverse_map will produce for it): known before the clusters are emitted
__global__ void __launch_bounds__(256) bfs_kept_members_kernel(const int32_t *__restrict__ l2p, int n_pts,
                                                              const int4 *__restrict__ label,
                                                              const int32_t *__restrict__ size,
                                                              const float *__restrict__ thr_dev,
                                                              int32_t *__restrict__ out) {
  __shared__ int wsum[4];
  const float thr = thr_dev[0];
  int mine = 0;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < n_pts; p += gridDim.x * 256)
    mine += static_cast<float>(size[label[l2p[p]].x]) >= thr ? 1 : 0;      // (the labelling's own `keep` test)
  const int w = wave_sum(mine);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}
int bfs_count_kept_members(const void *ws, size_t ws_bytes, int n, int64_t n_edges, const int32_t *l2p, int n_pts,
                           const float *thr_dev, int32_t *out, hipStream_t stream) {
  BfsWs w;
  if (!bfs_carve(const_cast<void *>(ws), ws_bytes, n, n_edges, &w)) {
    set_error("bfs_count_kept_members: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  hipMemsetAsync(out, 0, 4, stream);
  bfs_kept_members_kernel<<<grid_for(n_pts, 256, 1024), 256, 0, stream>>>(l2p, n_pts, w.label, w.size, thr_dev, out);
  return check_launch("bfs_count_kept_members");
}

}  // namespace sg

using namespace sg;

extern "C" {

size_t sg_bfs_workspace_bytes(int n, int64_t n_edges) {
  const size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  const size_t ne = static_cast<size_t>(n_edges > 0 ? n_edges : 1);
  const size_t be = big_stage_entries(n);
  return 14 * align_up(nn * 4) + align_up(64 * 4) + align_up(scan_workspace_bytes(n)) +
         align_up(ne * sizeof(int2)) + 2 * (align_up(be * 4) + align_up(be * 8)) + 256;
}

// Synchronises `stream` (the cluster count decides the size of the outputs).
int sg_bfs_cluster_label(const int32_t *bq_idxs, const int32_t *start_len, int n, int64_t n_edges,
                         int list_flags, const int32_t *seg_of_point, const float *seg_thr,
                         int n_seg, int32_t *n_cluster_host, int32_t *sum_npoint_host, void *ws,
                         size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(n >= 0 && n_seg >= 1 && seg_thr != nullptr, "sg_bfs_cluster_label: bad arguments");
  hipStream_t stream = as_stream(stream_);
  BfsWs w;
  if (!bfs_carve(ws, ws_bytes, n, n_edges, &w)) {
    set_error("sg_bfs_cluster_label: workspace too small");
    return SG_ERR_WORKSPACE;
  }
  *n_cluster_host = 0;
  *sum_npoint_host = 0;
  if (n == 0) return SG_OK;
  const int grid = (n + 255) / 256;
  hipMemsetAsync(w.counters, 0, 64 * 4, stream);
  bfs_init_kernel<<<grid, 256, 0, stream>>>(n, w.parent, w.size, w.owner);
  static const bool one_pass = getenv("SG_BFS_ONE_PASS") != nullptr;      // developer A/B knob: round 4's single pass
  if (!one_pass) {
    static const bool sample_env = getenv("SG_BFS_SAMPLE") != nullptr;      // developer A/B knob: union-find sampling pass
    if (sample_env || !(list_flags & SG_LISTS_SORTED))
      bfs_union_kernel<1><<<grid_for(n, 4, 256 * 16), 256, 0, stream>>>(bq_idxs, start_len, n,
                                                                      list_flags & SG_LISTS_SORTED,
                                                                      list_flags & SG_LISTS_RADIUS,
                                                                      w.parent, w.asym_nodes, w.counters);
    else
      bfs_hook_min_kernel<<<grid, 256, 0, stream>>>(bq_idxs, start_len, n, list_flags & SG_LISTS_SORTED,
                                                   list_flags & SG_LISTS_RADIUS, w.parent);
    bfs_compress_kernel<<<grid, 256, 0, stream>>>(n, w.parent);
  }
  bfs_union_kernel<0><<<grid_for(n, 4, 256 * 16), 256, 0, stream>>>(bq_idxs, start_len, n,
                                                                  list_flags & SG_LISTS_SORTED,
                                                                  list_flags & SG_LISTS_RADIUS,
                                                                  w.parent, w.asym_nodes, w.counters);
  bfs_flatten_kernel<<<grid, 256, 0, stream>>>(n, w.parent, w.lab);
  bfs_store_root_kernel<<<grid, 256, 0, stream>>>(n, w.lab, w.parent);  // parent := root_of

  int32_t *host_counters = pinned_words();      // [0..3] counters, [8] propagation flag
  SG_REQUIRE(host_counters != nullptr, "sg_bfs_cluster_label: pinned allocation failed");
  for (int i = 0; i < 5; ++i) host_counters[i] = 0;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      // asymmetric edges exist: propagate min labels to the fixed point, then redo the sizes
      const int n_asym = host_counters[0];
      while (true) {
        hipMemsetAsync(w.counters + 1, 0, 4, stream);
        bfs_propagate_kernel<<<grid_for(n_asym, 4, 256 * 16), 256, 0, stream>>>(
            bq_idxs, start_len, w.asym_nodes, n_asym, w.parent, w.lab, w.counters);
        host_counters[8] = 0;
        hipMemcpyAsync(host_counters + 8, w.counters + 1, 4, hipMemcpyDeviceToHost, stream);
        if (hipStreamSynchronize(stream) != hipSuccess) return check_launch("bfs propagate");
        if (!host_counters[8]) break;
      }
      bfs_zero_size_kernel<<<grid, 256, 0, stream>>>(n, w.size);
    }
    bfs_label_kernel<<<grid, 256, 0, stream>>>(n, w.parent, w.lab, start_len, w.label, w.size);
    // kept-cluster ids and offsets: two prefix sums over the seeds in index order
    const int4 *label = w.label;
    const int32_t *size = w.size;
    int32_t *cid = w.cid, *coff = w.coff;
    auto keep = [label, size, seg_of_point, seg_thr] __device__(int64_t i) -> int {
      if (label[i].x != static_cast<int32_t>(i)) return 0;
      const float thr = seg_thr[seg_of_point ? seg_of_point[i] : 0];
      return static_cast<float>(size[i]) >= thr ? 1 : 0;  // bfs_cluster.cpp:73-81
    };
    int32_t *max_kept = w.counters + 4;      // largest kept cluster, if above kBigMin (else 0): decides the emit's launch set
    auto keep_size = [keep, size, max_kept] __device__(int64_t i) -> int {
      const int s = keep(i) ? size[i] : 0;
      if (s > kBigMin) atomicMax(max_kept, s);
      return s;
    };
    int rc = exclusive_scan(keep, [cid] __device__(int64_t i, int v) { cid[i] = v; }, n,
                            w.counters + 2, w.scan_ws, w.scan_bytes, stream);
    if (rc != SG_OK) return rc;
    rc = exclusive_scan(keep_size, [coff] __device__(int64_t i, int v) { coff[i] = v; }, n,
                        w.counters + 3, w.scan_ws, w.scan_bytes, stream);
    if (rc != SG_OK) return rc;
    hipMemcpyAsync(host_counters, w.counters, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, stream);
    if (hipStreamSynchronize(stream) != hipSuccess) return check_launch("sg_bfs_cluster_label");
    if (host_counters[0] == 0) break;
    if (pass == 0) hipMemsetAsync(w.counters + 4, 0, 4, stream);      // (sizes are redone after the propagation)
  }
  *n_cluster_host = host_counters[2];
  *sum_npoint_host = host_counters[3];
  t_bfs_note.ws = ws;
  t_bfs_note.max_kept = host_counters[4];
  return check_launch("sg_bfs_cluster_label");
}

int sg_bfs_cluster_emit(const int32_t *bq_idxs, const int32_t *start_len, int n, int64_t n_edges,
                        const int32_t *seg_of_point, const float *seg_thr, int n_cluster,
                        int sum_npoint, int32_t *cluster_idxs, int32_t *cluster_offsets, void *ws,
                        size_t ws_bytes, sg_stream_t stream_) {
  SG_REQUIRE(n >= 0 && n_cluster >= 0 && sum_npoint >= 0, "sg_bfs_cluster_emit: bad arguments");
  hipStream_t stream = as_stream(stream_);
  BfsWs w;
  if (!bfs_carve(ws, ws_bytes, n, n_edges, &w)) {
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
  bfs_edge_rec_kernel<<<grid_for(n, 4, 256 * 16), 256, 0, stream>>>(bq_idxs, w.label, w.cid, w.seeds, n,
                                                                     n_cluster, w.erec);
  static const bool want_stats = getenv("SG_BFS_STATS") != nullptr;     // developer tool
  int32_t *stats = nullptr;
  if (want_stats) {
    hipMalloc(&stats, 256 * 8 * 4);
    hipMemsetAsync(stats, 0, 256 * 8 * 4, stream);
  }
  // giant clusters (> kBigMin points) are replayed by many workgroups together; the per-cluster
  // kernel skips them.  SG_BFS_BIG=0 (developer knob) keeps everything on the per-cluster kernel.
  static const bool big_on = !(getenv("SG_BFS_BIG") && atoi(getenv("SG_BFS_BIG")) == 0);
  // is there a giant cluster?  Known from the labelling call when it was this thread's last one on this workspace
  const bool noted = t_bfs_note.ws == ws && t_bfs_note.max_kept >= 0;
  // (claim keys of the multi-workgroup replay are frontier rank * 1024 + list position in 31 bits: up to 2^21 points;
  //  beyond that the giant clusters stay on the per-cluster kernel, whose claims are edge indices)
  const bool has_big = big_on && n < (1 << 21) && sum_npoint > kBigMin && (!noted || t_bfs_note.max_kept > kBigMin);
  t_bfs_note.ws = nullptr;
  // the giant clusters' replay (16 workgroups, milliseconds) next to the per-cluster kernel (one workgroup per
  // cluster, the rest of the chip): fork a side stream here, join behind both (SG_BFS_BIG_SIDE=0: one after the other)
  static const bool side_env = !(getenv("SG_BFS_BIG_SIDE") && atoi(getenv("SG_BFS_BIG_SIDE")) == 0);
  BfsSide *side = has_big && side_env && !want_stats ? bfs_side(stream) : nullptr;
  hipStream_t main_stream = stream;
  BfsDefer *defer = t_bfs_defer;      // (one call's worth)
  t_bfs_defer = nullptr;
  if (defer != nullptr) defer->deferred = false;
  if (side != nullptr) {
    if (hipEventRecord(side->fork, stream) != hipSuccess || hipStreamWaitEvent(side->side, side->fork, 0) != hipSuccess) {
      set_error("sg_bfs_cluster_emit: forking the side stream failed");
      return SG_ERR_LAUNCH;
    }
  }
  bfs_emit_kernel<<<min(n_cluster, 4096), kEmitThreads, 0, stream>>>(
      bq_idxs, start_len, w.label, w.erec, w.seeds, cluster_offsets, n_cluster, w.owner, cluster_idxs,
      stats, big_on && n < (1 << 21) ? kBigMin : 0x7fffffff, -1, nullptr, thin_levels_on());
  if (side != nullptr) stream = side->side;      // everything of the giant clusters goes to the side stream
  if (has_big) {
    static const int big_wgs_env = getenv("SG_BFS_BIG_WGS") ? atoi(getenv("SG_BFS_BIG_WGS")) : 16;   // developer knob
    const int big_wgs = big_wgs_env < 8 ? 8 : big_wgs_env > kBigWgsMax ? kBigWgsMax : big_wgs_env;
    int32_t *sync = w.asym_nodes;               // free after labelling; >= 64 + 8 * kBigWgsMax ints
    // (developer knob; measured: no gain -- kitti 9.47 against 9.42 ms, stpls3d_pp 11.4-11.8 against 11.3-11.5,
    //  profiles/r06_bfs_big_flags_ab.txt -- the arrival atomic is not what a level waits for; off)
    static const bool flags_on = getenv("SG_BFS_BIG_FLAGS") && atoi(getenv("SG_BFS_BIG_FLAGS")) != 0;
    if (static_cast<size_t>(n) >= static_cast<size_t>(kBigSyncWords)) {
      hipMemsetAsync(sync, 0, kBigSyncWords * 4, stream);
      if (const char *e = getenv("SG_BFS_FORCE_FALLBACK"))      // test hook: pretend the barrier gave up
        if (atoi(e) & 1) hipMemsetAsync(sync + 1, 1, 1, stream);      // (bit 1: the LOCAL form's word, below)
      // frontier staging pools (one per level parity, at most a cluster's points each): the union-find
      // arrays of the labelling, idle by now
      // LOCAL form first (SG_BFS_BIG_LOCAL, read per call): its own sync words behind the old kernel's; if it gives
      // up (a level beyond its LDS limits, a wait that timed out) its fail word gates the old kernel in
      const int32_t *gate = nullptr;
      if (big_local_on() && big_fast_on() && w.big_stage[0] != nullptr &&
          static_cast<size_t>(n) >= static_cast<size_t>(kBigSyncWords) + 96) {
        const int lw_env = getenv("SG_BFS_BIG_LOCAL_WGS") ? atoi(getenv("SG_BFS_BIG_LOCAL_WGS")) : 16;   // developer knobs, read per call
        const int lw = lw_env < 8 ? 8 : lw_env > kBigWgsMax ? kBigWgsMax : lw_env;
        // a level re-partitions the frontier evenly when level % every == 0 (and during the first `warm` levels)
        const int every_env = getenv("SG_BFS_BIG_LOCAL_EVERY") ? atoi(getenv("SG_BFS_BIG_LOCAL_EVERY")) : 4;
        const int warm_env = getenv("SG_BFS_BIG_LOCAL_WARM") ? atoi(getenv("SG_BFS_BIG_LOCAL_WARM")) : 8;
        const int every = every_env < 1 ? 1 : every_env, warm = warm_env < 1 ? 1 : warm_env;
        // (test hook: the path of scenes above 524 288 points, whose visited filter does not fit the LDS)
        const bool novis = getenv("SG_BFS_BIG_LOCAL_NOVIS") && atoi(getenv("SG_BFS_BIG_LOCAL_NOVIS")) != 0;
        int32_t *sync3 = sync + kBigSyncWords;
        const size_t rows = big_stage_entries(n) / (2 * static_cast<size_t>(lw));
        hipMemsetAsync(sync3, 0, 96 * 4, stream);
        hipMemsetAsync(w.big_rec[1], 0, rows * 2 * lw * 8, stream);         // no stale tags in the level table
        hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(w.parent), 0x7fffffff, n, stream);    // second claim array
        if (const char *e = getenv("SG_BFS_FORCE_FALLBACK"))
          if (atoi(e) & 2) hipMemsetAsync(sync3 + 1, 1, 1, stream);      // test hook: the LOCAL form gave up
        // the visited filter: one bit per point in dynamic LDS (on top of ~92 KB static)
        const size_t vis_bytes = (!novis && n <= kVisWords * 32) ? static_cast<size_t>((n + 31) >> 5) * 4 : 0;
        {
          static std::mutex mu;
          static uint64_t done_mask = 0;      // bit d: device d configured (the attribute is per device)
          int dev = 0;
          hipGetDevice(&dev);
          std::lock_guard<std::mutex> g(mu);
          if (dev < 0 || dev >= 64 || !((done_mask >> dev) & 1)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(bfs_emit_big_local_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kVisWords * 4) != hipSuccess) {
              (void)hipGetLastError();
              set_error("sg_bfs_cluster_emit: %d bytes of dynamic LDS are not available on this device", kVisWords * 4);
              return SG_ERR_LAUNCH;
            }
            if (dev >= 0 && dev < 64) done_mask |= 1ull << dev;
          }
        }
        bfs_emit_big_local_kernel<<<lw, kEmitThreads, vis_bytes, stream>>>(
            bq_idxs, w.label, w.erec, w.seeds, cluster_offsets, n_cluster, w.owner, w.parent, w.big_stage[0],
            w.big_rec[0], w.big_rec[1], static_cast<int>(rows) - 1, cluster_idxs, sync3, novis ? 0x7fffffff : n, every, warm,
            want_stats);
        gate = sync3 + 1;
        bfs_owner_reset_kernel<<<grid_for(n, 256, 1024), 256, 0, stream>>>(n, w.label, w.size, kBigMin, gate, w.owner);
      }
      if (big_fast_on() && big_wgs <= kBigFastWgs && w.big_stage[0] != nullptr)
        bfs_emit_big_kernel<true><<<big_wgs, kEmitThreads, 0, stream>>>(
            bq_idxs, w.label, w.erec, w.seeds, cluster_offsets, n_cluster, w.owner, w.wcnt, w.big_stage[0],
            w.big_stage[1], w.big_rec[0], w.big_rec[1], n, cluster_idxs, sync, flags_on, want_stats, gate);
      else
        bfs_emit_big_kernel<false><<<big_wgs, kEmitThreads, 0, stream>>>(
            bq_idxs, w.label, w.erec, w.seeds, cluster_offsets, n_cluster, w.owner, w.wcnt, w.parent, w.lab,
            nullptr, nullptr, 0, cluster_idxs, sync, false, false, gate);
      // sync[1] != 0: the replay gave up somewhere (see big_barrier) -- redo the giant clusters on
      // the per-cluster kernel (same output, slower); both launches are no-ops otherwise
      bfs_owner_reset_kernel<<<grid_for(n, 256, 1024), 256, 0, stream>>>(n, w.label, w.size, kBigMin,
                                                                        sync + 1, w.owner);
      bfs_emit_kernel<<<min(n_cluster, 256), kEmitThreads, 0, stream>>>(
          bq_idxs, start_len, w.label, w.erec, w.seeds, cluster_offsets, n_cluster, w.owner,
          cluster_idxs, nullptr, 0x7fffffff, kBigMin, sync + 1, thin_levels_on());
    }
  }
  if (side != nullptr && defer != nullptr) {
    // deferred join: the side stream sees the per-cluster kernel's clusters and belongs to the caller from here
    stream = main_stream;
    if (hipEventRecord(side->back, stream) != hipSuccess || hipStreamWaitEvent(side->side, side->back, 0) != hipSuccess) {
      set_error("sg_bfs_cluster_emit: handing the side stream over failed");
      return SG_ERR_LAUNCH;
    }
    defer->deferred = true;
    defer->side = side->side;
  } else if (side != nullptr) {
    stream = main_stream;
    if (hipEventRecord(side->join, side->side) != hipSuccess || hipStreamWaitEvent(stream, side->join, 0) != hipSuccess) {
      set_error("sg_bfs_cluster_emit: joining the side stream failed");
      return SG_ERR_LAUNCH;
    }
  }
  if (want_stats && has_big && big_local_on()) {
    hipStreamSynchronize(stream);
    int32_t h[32];
    hipMemcpy(h, w.asym_nodes + kBigSyncWords, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long lp[8];
    hipMemcpy(lp, w.asym_nodes + kBigSyncWords + 16, sizeof(lp), hipMemcpyDeviceToHost);
    const double lv = lp[5] ? static_cast<double>(lp[5]) : 1.0;
    fprintf(stderr, "bfs giant clusters, local form: %s (%d); %d clusters, %d levels (%llu re-partition levels); us per level: "
            "re-partition %.2f, edge bases+claim sweep %.2f, pool+drain %.2f, rendezvous %.2f, emit sweep %.2f, totals+publication %.2f\n",
            h[1] ? "gave up" : "completed", h[1], h[9], h[8], lp[6], lp[0] / lv / 100.0, lp[1] / lv / 100.0, lp[7] / lv / 100.0,
            lp[2] / lv / 100.0, lp[3] / lv / 100.0, lp[4] / lv / 100.0);
  }
  if (want_stats && has_big) {
    unsigned long long ph[6];
    hipStreamSynchronize(stream);
    hipMemcpy(ph, w.asym_nodes + 16, sizeof(ph), hipMemcpyDeviceToHost);
    const double lv = ph[5] ? static_cast<double>(ph[5]) : 1.0;
    fprintf(stderr, "bfs giant clusters: %llu cached levels; us per level: wait records %.2f, ranks+queue+staged records %.2f, "
            "claim sweep %.2f, claim rendezvous %.2f, emit sweep+drain %.2f\n", ph[5], ph[0] / lv / 100.0, ph[1] / lv / 100.0,
            ph[2] / lv / 100.0, ph[3] / lv / 100.0, ph[4] / lv / 100.0);
  }
  if (want_stats) {
    int32_t h[256 * 8];
    hipMemcpy(h, stats, sizeof(h), hipMemcpyDeviceToHost);
    hipFree(stats);
    for (int c = 0; c < n_cluster && c < 256; ++c)
      fprintf(stderr, "bfs cluster %d: size %d fast levels %d generic levels %d edges %d max frontier %d\n", c,
              h[c * 8], h[c * 8 + 1], h[c * 8 + 2], h[c * 8 + 3], h[c * 8 + 4]);
  }
  return check_launch("sg_bfs_cluster_emit");
}

}  // extern "C"
