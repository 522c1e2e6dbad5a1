/*
 * softgroup_hip.h -- C ABI of libsoftgroup_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the SoftGroup hot path: plain pointers and sizes, no
 * torch/ATen types.  Each entry point replaces one symbol the reference binds
 * through pybind (softgroup/ops/src/softgroup_api.cpp:8-28) or one piece of the
 * un-vendored spconv 2.1 library the reference model calls
 * (softgroup/model/softgroup.py:60-62, softgroup/model/blocks.py:31-119).
 * The replaced reference interface is cited at every declaration (paths relative
 * to the reference repository root).
 *
 * Conventions
 *   - all `const T* x` / `T* x` arguments are DEVICE pointers unless the name ends in
 *     `_host` or the function name contains `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     kernel is launched on it, nothing synchronises unless documented;
 *   - no function allocates device memory: scratch comes from the caller through
 *     (`ws`, `ws_bytes`) sized by the matching `*_workspace_bytes` query;
 *   - return value: SG_OK (0) or a negative SG_ERR_* code; sg_last_error() gives text;
 *   - functions whose output size is data dependent are split into a sizing pass
 *     that writes small int32 "meta" scalars to device memory (the caller reads them
 *     back) and a fill pass.
 */
#ifndef SOFTGROUP_HIP_H
#define SOFTGROUP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *sg_stream_t;

#define SG_OK 0
#define SG_ERR_ARG (-1)
#define SG_ERR_WORKSPACE (-2)
#define SG_ERR_LAUNCH (-3)
#define SG_ERR_UNSUPPORTED (-4)

#define SG_BALLQUERY_MAX_NEIGHBORS 1000 /* bfs_cluster.cu:24 `int idx_temp[1000]` */
#define SG_OCTREE_NUM_NODES 585         /* octree_ball_query.cu:10 */
#define SG_OCTREE_NUM_LEAVES 512        /* octree_ball_query.cu:11 */
#define SG_LISTS_SORTED 1
#define SG_LISTS_RADIUS 2

int sg_version(void);
const char *sg_last_error(void);
/* name/CU count/clock of the current device, for bench records */
int sg_device_info(char *name_host, int name_cap, int *num_cu_host, int *clock_khz_host);

/* Runtime state the library keeps per (device, caller stream) -- the executors' events and pinned read-back
 * words, the conv launches' arrival-counter and ticket pools (8 MB of device
 * memory) -- is created on a stream's first use and normally lives as long as the process.  A caller
 * that retires a stream (a pool of scan threads being resized) calls this once the stream is idle and
 * before destroying it, on the stream's device.  No counterpart in the reference (spconv keeps its
 * workspaces in the torch allocator). */
int sg_stream_release(sg_stream_t stream);
/* A stream of the library's own (hipStreamCreateWithFlags, non-blocking) for a scan worker, and its end:
 * sg_stream_destroy synchronises it, releases its state (sg_stream_release) and destroys it.  Unlike the streams
 * a framework hands out from a fixed pool (torch.cuda.Stream() reuses 32 raw handles per device), such a handle
 * is never shared with a later owner, so releasing its state cannot pull buffers from under somebody else's
 * kernels.  The Python binding wraps it as torch.cuda.ExternalStream and, because PyTorch's caching allocator
 * aborts when a stream it has seen disappears, PARKS a retired worker's stream (sg_stream_release only) for the
 * next pool instead of destroying it; sg_stream_destroy is for hosts without such an allocator. */
int sg_stream_create(sg_stream_t *stream_out);
int sg_stream_destroy(sg_stream_t stream);
/* The same with a dispatch priority: level > 0 the device's highest, 0 its default, < 0 its lowest
 * (hipStreamCreateWithPriority over hipDeviceGetStreamPriorityRange).  For scan workers whose scans should not
 * all advance in lockstep: the scan on the high-priority stream finishes at its stand-alone latency. */
int sg_stream_create_priority(sg_stream_t *stream_out, int level);

/* ------------------------------------------------------------------------------------------
 * Voxelisation index build.  Replaces `voxelize_idx` (softgroup_api.cpp:12,
 * voxelize/voxelize.cpp:11-165; Python: ops/functions.py:168-197).
 * coords: int64 [n, ncol], ncol = 3 or 4 (column 0 = batch index when 4).
 * Voxel id = first-seen order; rule row = [count, ascending point idx..., 0 pad].
 * mode: 4 mean / 3 sum (same indices); 0,1 keep first point, 2 keeps last (voxelize.cpp:127-149).
 * ---------------------------------------------------------------------------------------- */
/* Host (CPU) variant -- what the reference runs inside DataLoader workers (data/custom.py:239).
 * Pass 1 fills input_map[n], returns M and maxActive; pass 2 fills the two outputs. */
int sg_voxelize_idx_host(const int64_t *coords_host, int n, int ncol, int mode,
                         int32_t *input_map_host, int32_t *num_voxels_host,
                         int32_t *max_active_host);
int sg_voxelize_idx_fill_host(const int64_t *coords_host, int n, int ncol, int mode,
                              const int32_t *input_map_host, int num_voxels, int max_active,
                              int64_t *out_coords_host, int32_t *out_map_host);
/* Device variant (used in-model: softgroup.py:494,703).  meta[0]=M, meta[1]=maxActive.
 * `fill` must get the same ws buffer, untouched since `build`. */
size_t sg_voxelize_idx_workspace_bytes(int n);
int sg_voxelize_idx_build(const int64_t *coords, int n, int ncol, int mode, int32_t *input_map,
                          int32_t *meta, void *ws, size_t ws_bytes, sg_stream_t stream);
int sg_voxelize_idx_fill(const int64_t *coords, int n, int ncol, int mode,
                         const int32_t *input_map, int num_voxels, int max_active,
                         int64_t *out_coords, int32_t *out_map, void *ws, size_t ws_bytes,
                         sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Voxel feature pooling.  Replaces `voxelize_fp` / `voxelize_bp` (softgroup_api.cpp:13-14,
 * voxelize/voxelize.cu:10-62; Python ops/functions.py:200-234).
 * out[row,:] = sum_{i=1..rules[row,0]} m * feats[rules[row,i],:], m = 1/count if average.
 * Bit-exact with the reference: sequential order, separate multiply and add.
 * ---------------------------------------------------------------------------------------- */
int sg_voxelize_fp(const float *feats, const int32_t *rules, int num_voxels, int max_active,
                   int channels, int average, float *out, sg_stream_t stream);
/* d_feats must be pre-zeroed by the caller (functions.py:228) */
int sg_voxelize_bp(const float *d_out, const int32_t *rules, int num_voxels, int max_active,
                   int channels, int average, float *d_feats, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Ball query.  Replaces `ballquery_batch_p` (softgroup_api.cpp:19, bfs_cluster/bfs_cluster.cpp:17-31,
 * bfs_cluster.cu:15-101; Python ops/functions.py:237-275).
 * Per point: ascending list of the (at most 1000 smallest) indices k of the same batch with
 * d2(k) < radius^2 (strict), self included.  Two passes give a deterministic CSR:
 *   count : start_len[i,1] = min(cnt,1000); meta[0] = total (int32, saturates at INT32_MAX)
 *   (caller exclusive-scans start_len[:,1] into start_len[:,0] -- sg_exclusive_scan_startlen)
 *   fill  : idx[start .. start+len) = neighbours.
 * All three calls of one query share ONE workspace: the grid lives there, and the count pass parks
 * the finished (sorted) lists of up to 64 entries there, which the fill pass only copies.
 * The reference's `nActive > n*meanActive` retry protocol (functions.py:258-266) is kept in
 * the Python facade; the kernels never truncate.
 * ---------------------------------------------------------------------------------------- */
size_t sg_ballquery_workspace_bytes(int n);
int sg_ballquery_build_grid(const float *xyz, const int32_t *batch_idxs, int n, float radius,
                            void *ws, size_t ws_bytes, sg_stream_t stream);
int sg_ballquery_count(const float *xyz, const int32_t *batch_idxs, int n, float radius,
                       int32_t *start_len, int32_t *meta, void *ws, size_t ws_bytes,
                       sg_stream_t stream);
int sg_ballquery_fill(const float *xyz, const int32_t *batch_idxs, int n, float radius,
                      const int32_t *start_len, int32_t *idx, void *ws, size_t ws_bytes,
                      sg_stream_t stream);
/* start_len[i,0] = sum_{j<i} start_len[j,1]; meta[0] = total.  ws >= sg_scan_workspace_bytes(n) */
size_t sg_scan_workspace_bytes(int n);
int sg_exclusive_scan_startlen(int32_t *start_len, int n, int32_t *meta, void *ws,
                               size_t ws_bytes, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Octree ball query (SoftGroup++).  Replaces `build_and_export_octree` + `octree_ball_query`
 * (softgroup_api.cpp:16-18, octree_ball_query/octree_ball_query.cpp:8-188, .cu:14-147;
 * Python ops/functions.py:14-44).  Fixed 3 levels / 585 nodes / 512 leaves.
 * Neighbour order = leaf export order, then within-leaf order; capped at the first 1000.
 * ---------------------------------------------------------------------------------------- */
int sg_octree_build_host(const float *points_host, const float *xyzwhl_host, int num_points,
                         int num_levels, float *boxes_host, int32_t *pt_inds_host,
                         int32_t *pt_start_len_host);
/* The same export built on the DEVICE (no .cpu() of the coordinates, no host thread): root box =
 * extent of `points` ((max + min) / 2, max - min), 3 levels, identical boxes / pt_inds / pt_start_len
 * (start, length per leaf) to sg_octree_build_host with that root.  boxes float [585, 6], pt_inds
 * int32 [n], pt_start_len int32 [512, 2].  Nothing synchronises. */
size_t sg_octree_build_workspace_bytes(int n);
int sg_octree_build(const float *points, int n, float *boxes, int32_t *pt_inds, int32_t *pt_start_len,
                    void *ws, size_t ws_bytes, sg_stream_t stream);
/* SoftGroup.pyramid_inverse_map (softgroup/model/softgroup.py:500-507): proposals over a class's level
 * voxels (proposals_idx int32 [num_pairs, 2] = (proposal, voxel), proposals of a class disjoint) ->
 * proposals over its points through l2p_map int32 [n_points] (voxel of every point): out_idx int32
 * [<= n_points, 2] = (proposal, point), proposal-major with points ascending (the row order of the
 * reference's dense nonzero), out_offsets int32 [n_prop + 1], *n_out_dev = rows written. */
size_t sg_pyramid_inverse_map_workspace_bytes(int n_points, int n_voxels, int n_prop);
int sg_pyramid_inverse_map(const int32_t *proposals_idx, int64_t num_pairs, int n_prop,
                           const int32_t *l2p_map, int n_points, int n_voxels, int32_t *out_idx,
                           int32_t *out_offsets, int32_t *n_out_dev, void *ws, size_t ws_bytes,
                           sg_stream_t stream);
int sg_octree_ballquery_count(const float *points, const float *boxes, const int32_t *pt_inds,
                              const int32_t *pt_start_len, int n, float radius,
                              int32_t *start_len, sg_stream_t stream);
int sg_octree_ballquery_fill(const float *points, const float *boxes, const int32_t *pt_inds,
                             const int32_t *pt_start_len, int n, float radius,
                             const int32_t *start_len, int32_t *idx, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BFS clustering (soft grouping).  Replaces `bfs_cluster` (softgroup_api.cpp:20,
 * bfs_cluster/bfs_cluster.cpp:33-126; Python ops/functions.py:278-308) -- CPU single thread in
 * the reference, wavefront-parallel here.  Output identical to the sequential algorithm:
 * seeds ascending, members in FIFO-BFS order, clusters with (float)size >= thr kept, where
 * thr = threshold if class_numpoint_mean == -1 else threshold * class_numpoint_mean.
 * `seg_thr[n_seg]` generalises the single (class_id) call to many classes per launch: point i
 * belongs to segment seg_of_point[i] (NULL = all segment 0) and thr = seg_thr[segment].
 *   label : min-ancestor labelling (directed reachability), sizes, kept clusters
 *   emit  : cluster_idxs int32 [sumNPoint,2] = (cluster_id, point_idx), cluster_offsets [nCluster+1]
 * ---------------------------------------------------------------------------------------- */
/* (14 n-word arrays, the scan workspace, one 8-byte record per edge and -- for n > 16 384, where a cluster
 * too large for one workgroup's replay can exist -- the frontier staging of the multi-workgroup replay:
 * 24 bytes x (n + 131 072)) */
size_t sg_bfs_workspace_bytes(int n, int64_t n_edges);
/* list_flags: SG_LISTS_SORTED  every list is strictly ascending (sg_ballquery_* output);
 *             SG_LISTS_RADIUS  lists come from a radius query, i.e. u in list(v) <=> v in list(u)
 *                              unless one of the two lists is capped at 1000 entries
 *                              (sg_ballquery_* and sg_octree_ballquery_* output).
 *             0 = arbitrary directed lists (every edge is checked for its reverse).
 * SYNCHRONISES `stream`: the number of kept clusters sizes the outputs, so it is returned to
 * host memory (*n_cluster_host, *sum_npoint_host). */
int sg_bfs_cluster_label(const int32_t *bq_idxs, const int32_t *start_len, int n,
                         int64_t n_edges, int list_flags, const int32_t *seg_of_point,
                         const float *seg_thr, int n_seg, int32_t *n_cluster_host,
                         int32_t *sum_npoint_host, void *ws, size_t ws_bytes, sg_stream_t stream);
/* same ws buffer, untouched since sg_bfs_cluster_label */
int sg_bfs_cluster_emit(const int32_t *bq_idxs, const int32_t *start_len, int n, int64_t n_edges,
                        const int32_t *seg_of_point, const float *seg_thr, int n_cluster,
                        int sum_npoint, int32_t *cluster_idxs, int32_t *cluster_offsets, void *ws,
                        size_t ws_bytes, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Segment reductions.  Replace `sec_mean` / `sec_min` / `sec_max` (softgroup_api.cpp:25-27,
 * sec_mean/sec_mean.cu:13-93; Python ops/functions.py:351-438) and `global_avg_pool_fp/bp`
 * (softgroup_api.cpp:22-23, roipool/roipool.cu:12-71; Python ops/functions.py:311-348).
 * inp float32 [S, C], offsets int32 [nP+1] -> out float32 [nP, C].
 * ---------------------------------------------------------------------------------------- */
int sg_sec_mean(const float *inp, const int32_t *offsets, int n_seg, int channels, float *out,
                sg_stream_t stream);
int sg_sec_min(const float *inp, const int32_t *offsets, int n_seg, int channels, float *out,
               sg_stream_t stream);
int sg_sec_max(const float *inp, const int32_t *offsets, int n_seg, int channels, float *out,
               sg_stream_t stream);
int sg_global_avg_pool_fp(const float *feats, const int32_t *offsets, int n_seg, int channels,
                          float *out, sg_stream_t stream);
/* d_feats[i,:] += d_out[p,:] / n_p  (d_feats pre-zeroed by the caller, functions.py:341) */
int sg_global_avg_pool_bp(float *d_feats, const int32_t *offsets, const float *d_out, int n_seg,
                          int channels, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Proposal/GT mask IoU and mask targets (training).  Replace `get_mask_iou_on_cluster`,
 * `get_mask_iou_on_pred`, `get_mask_label` (softgroup_api.cpp:8-10,
 * cal_iou_and_masklabel/cal_iou_and_masklabel.cu:9-164; Python ops/functions.py:47-165).
 * O(S) per-proposal label histogram in LDS instead of the reference's O(nP*nI*|P|);
 * the IoU quotient is evaluated in double and rounded to float like the reference.
 * ---------------------------------------------------------------------------------------- */
int sg_get_mask_iou_on_cluster(const int32_t *proposals_idx, const int32_t *proposals_offset,
                               const int64_t *instance_labels, const int32_t *instance_pointnum,
                               int n_instance, int n_proposal, float *proposals_iou,
                               sg_stream_t stream);
int sg_get_mask_iou_on_pred(const int32_t *proposals_idx, const int32_t *proposals_offset,
                            const int64_t *instance_labels, const int32_t *instance_pointnum,
                            const float *mask_scores_sigmoid, int n_instance, int n_proposal,
                            float *proposals_iou, sg_stream_t stream);
/* mask_label pre-filled with -1 by the caller (functions.py:147) */
int sg_get_mask_label(const int32_t *proposals_idx, const int32_t *proposals_offset,
                      const int64_t *instance_labels, const int64_t *instance_cls,
                      const float *proposals_iou, int n_instance, int n_proposal, float iou_thr,
                      float *mask_label, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse convolution (the spconv.pytorch subset of SURVEY.md 2.4).
 * indices int32 [M,4] = (batch, d0, d1, d2); spatial_shape int32[3] on the host.
 *
 * Rulebooks ("gather tables", output-stationary): nbr[j*K + k] = input row feeding output row
 * j through kernel offset k, or -1.
 *   SubM k3 p1 (SubMConv3d, softgroup.py:61-62, blocks.py:57-70): K=27, k=(dx+1)*9+(dy+1)*3+(dz+1)
 *   strided k2 s2 (SparseConv3d, blocks.py:101-107): output coords = c//2 in first-seen order,
 *       inputs with c//2 >= shape//2 dropped; K=8, k=(c0&1)*4+(c1&1)*2+(c2&1)
 *   inverse k2 (SparseInverseConv3d, blocks.py:114-119): K=8, one entry per row
 *       (k = parity of the fine coordinate, value = parent row) -- built from the saved pair.
 * ---------------------------------------------------------------------------------------- */
size_t sg_spconv_hash_workspace_bytes(int num_rows);
/* builds the coordinate hash in ws, then fills nbr[M,27] */
int sg_spconv_subm_rulebook(const int32_t *indices, int num_rows, const int32_t *spatial_shape_host,
                            int32_t *nbr, void *ws, size_t ws_bytes, sg_stream_t stream);
/* pass 1: in2out[M] (-1 = dropped), meta[0] = M_out.  pass 2: out_indices[M_out,4], child[M_out,8] */
int sg_spconv_down_build(const int32_t *indices, int num_rows, const int32_t *spatial_shape_host,
                         int32_t *in2out, int32_t *meta, void *ws, size_t ws_bytes,
                         sg_stream_t stream);
int sg_spconv_down_fill(const int32_t *indices, int num_rows, const int32_t *in2out,
                        int num_out_rows, int32_t *out_indices, int32_t *child, void *ws,
                        size_t ws_bytes, sg_stream_t stream);
/* inv_nbr[M,8] from (fine indices, in2out) */
int sg_spconv_inverse_rulebook(const int32_t *indices_fine, const int32_t *in2out, int num_rows,
                               int32_t *inv_nbr, sg_stream_t stream);

/* Tile plan for the implicit-GEMM kernel: rows are processed in mask-sorted order so that a
 * 32-row MFMA tile only visits kernel offsets some row of the tile really has.
 * T = ceil(M_out/32) tiles, emitted heaviest first (descending number of offsets) so that the
 * dispatcher spreads heavy and light tiles over the chip (longest-processing-time-first):
 *   order[T*32]       : row ids of every tile, -1 padding.  Rows are sorted by their neighbour mask
 *                       with the bits permuted by offset frequency in this layer (rarest offset =
 *                       most significant bit; ties: lower offset is the more common one)
 *   tile_mask[T + SG_PLAN_HIST_WORDS] : [0, T) OR of the masks of the tile's rows; behind them the
 *                       histogram hist[j] = number of tiles with j offsets (j = 0..32, rest 0): with the
 *                       tiles in descending order it gives a consumer the length of the heavy prefix
 *                       and the position of any (tile, offset) of the list without a search
 *   nbr_tiles[T*32*K] : the tile's gather-table rows copied contiguously (-1 for padding rows)
 * Row order behind the API is untouched: a tile computes rows order[32t .. 32t+31] and stores
 * them back at their own row index.  kvol <= 27.
 * (Developer A/B knob SG_PLAN_ORDER=1|2, read once: rows mask-sorted inside super-blocks of consecutive
 * rows and the tiles dealt to the 8 XCDs in contiguous ranges -- the spatially local plans measured and
 * rejected in round 5; the default 0 is the order described above.) */
#define SG_PLAN_HIST_WORDS 40
size_t sg_spconv_plan_workspace_bytes(int num_out_rows);
int sg_spconv_plan(const int32_t *nbr, int num_out_rows, int kvol, int32_t *order,
                   uint32_t *tile_mask, int32_t *nbr_tiles, void *ws, size_t ws_bytes,
                   sg_stream_t stream);

/* Whole-pyramid index build: all gather tables and tile plans of an n_levels-deep U-Net
 * (level l+1 = SparseConv3d k2 s2 of level l) in a handful of launches and ONE host read-back,
 * instead of one rulebook / plan call chain and one read-back per level (spconv sizes every strided
 * conv's output by its own device->host copy; softgroup/model/blocks.py:131-143 is the UBlock
 * recursion this serves).  Results per level are identical to the per-level entry points above
 * (same first-seen numbering of the coarse sites, same tables, same tile plans).
 *   1. sg_spconv_pyramid_rows : rows_dev[l] = number of sites of level l (rows_dev[0] = num_rows);
 *      the caller reads them back (its only synchronisation) and allocates the per-level outputs;
 *   2. sg_spconv_pyramid_build: fills every pointer of levels[0..n_levels) -- `ws` is the SAME
 *      workspace as in step 1 (it carries the hash tables), `ws2` is scratch of
 *      sg_spconv_pyramid_build_workspace_bytes(levels, n_levels).
 * levels[l]: rows (in); indices [rows,4]; nbr [rows,27] + plan `subm`; and for l < n_levels-1:
 * in2out [rows]; child [rows_{l+1},8] + plan `down` (strided conv l -> l+1); inv [rows,8] + plan
 * `up` (inverse conv l+1 -> l).  A plan of a table with R rows: order [T*32], tile_mask
 * [T + SG_PLAN_HIST_WORDS], nbr_tiles [T*32*K], T = ceil(R/32) (see sg_spconv_plan). */
#define SG_PYRAMID_MAX_LEVELS 10
typedef struct sg_plan_ptrs {
  int32_t *order;
  uint32_t *tile_mask;
  int32_t *nbr_tiles;
} sg_plan_ptrs;
typedef struct sg_pyramid_level {
  int rows;
  int32_t *indices;
  int32_t *nbr;
  sg_plan_ptrs subm;
  int32_t *in2out;
  int32_t *child;
  sg_plan_ptrs down;
  int32_t *inv;
  sg_plan_ptrs up;
} sg_pyramid_level;
size_t sg_spconv_pyramid_workspace_bytes(int num_rows, int n_levels);
int sg_spconv_pyramid_rows(const int32_t *indices, int num_rows, const int32_t *spatial_shape_host,
                           int n_levels, int32_t *rows_dev, void *ws, size_t ws_bytes,
                           sg_stream_t stream);
size_t sg_spconv_pyramid_build_workspace_bytes(const sg_pyramid_level *levels_host, int n_levels);
int sg_spconv_pyramid_build(const int32_t *indices, int num_rows, const int32_t *spatial_shape_host,
                            int n_levels, const sg_pyramid_level *levels_host, void *ws,
                            size_t ws_bytes, void *ws2, size_t ws2_bytes, sg_stream_t stream);

/* Weight packing for the conv kernel: src [Cout, K, Cin] (spconv "OKKKI", the checkpoint layout,
 * tools/convert_checkpoint.py:17-19; src_is_kio = 0) or [K, Cin, Cout] (src_is_kio = 1) ->
 * "k8" [K][ceil(Cin/8)][Cout][8] fp32, zero padded (the 8 reduction steps one lane feeds to 8
 * consecutive MFMAs are one 32-B read), FOLLOWED IN THE SAME BUFFER by the same elements as three
 * bf16 planes h, m, l (w = h + m + l, round to nearest even; 2 bytes per element each) for the
 * split-precision kernel.  The buffer is 2.5x the fp32 block: it MUST be sized with
 * sg_spconv_packed_weight_elems(kvol, cin, cout) (in floats) and filled by sg_spconv_pack_weight --
 * sg_spconv_gather_conv_f32 builds its weight descriptor over that full size and, on the default
 * split path, reads the planes.  Non-finite weights: h carries the Inf/NaN, m and l are NaN-free
 * only for finite values (x - Inf = NaN); a diverged model shows NaN where the fp32-MFMA kernel
 * (sg_spconv_set_arithmetic(0)) would show Inf.
 * src_is_kio = 2 / 3 pack the weights of the TRANSPOSED convolution (the input gradient of a layer)
 * straight from that layer's "OKKKI" tensor, read as src [cin][K][cout] (this call's cin = the
 * layer's Cout and vice versa); 3 also mirrors the kernel offsets (k -> K-1-k: SubMConv3d). */
size_t sg_spconv_packed_weight_elems(int kvol, int cin, int cout);
int sg_spconv_pack_weight(const float *w, int cout, int kvol, int cin, int src_is_kio, float *w_k8,
                          sg_stream_t stream);

/* out[j,:] = post( (residual ? residual[j,:] : 0) + sum_k W[k] . in[nbr[j,k],:] )
 *   post(x) = relu(x * post_scale + post_shift) when post_scale != NULL, identity otherwise: the
 *   eval-mode BatchNorm1d + ReLU that FOLLOWS this conv and precedes the next one in
 *   blocks.py:57-70 (conv_branch: BN, ReLU, conv, BN, ReLU, conv), fused into the epilogue.  `in`
 *   is taken as is (already activated by the producer's epilogue or by sg_bn_relu_f32).
 *   Optional second output out_act = relu(out * act_scale + act_shift): the BatchNorm1d + ReLU of
 *   the NEXT block, when `out` itself is still needed raw (residual identity, skip concat).
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32).  cout % 4 == 0 takes the MFMA
 * path; other shapes the scalar path of the same operator.  order/tile_mask/nbr_tiles from
 * sg_spconv_plan (all NULL = natural order, all offsets, slower general kernel).  cin % 16 == 0 with
 * a plan runs the persistent-workgroup kernel.  Layers too small to fill the chip split the kernel
 * offsets over several units and reduce partial sums from `ws` in a fixed order; pass
 * ws >= sg_spconv_conv_workspace_bytes(num_out_rows, cout) (ws = NULL disables the split). */
size_t sg_spconv_conv_workspace_bytes(int num_out_rows, int cout);
int sg_spconv_gather_conv_f32(const float *in, int num_in_rows, const int32_t *nbr,
                              int num_out_rows, int kvol, int cin, int cout, const float *w_k8,
                              const float *post_scale, const float *post_shift,
                              const float *residual, const float *act_scale,
                              const float *act_shift, float *out_act, const int32_t *order,
                              const uint32_t *tile_mask, const int32_t *nbr_tiles, float *out,
                              void *ws, size_t ws_bytes, sg_stream_t stream);

/* Arithmetic of the fp32 sparse convolution's products (process-wide, not thread-safe; meant for
 * tests and A/B measurements): 1 = on the bf16 matrix pipe at fp32 accuracy (operands split three
 * ways, six v_mfma_f32_32x32x16_bf16 per 16-channel slice, fp32 accumulation; dropped terms
 * <= 2^-24 |a b|) -- the default; 0 = v_mfma_f32_32x32x2_f32; 2 = bf16 OPERANDS (activations and
 * weights rounded to nearest-even bf16, one v_mfma_f32_32x32x16_bf16 per slice, fp32 accumulation,
 * fp32 in and out: the precision of gather_conv_bf16 without rounding the stored activations;
 * layers with cin % 32 != 0 keep mode 1); -1 = back to the environment (SG_CONV_SPLIT).  No counterpart in the reference: spconv 2.1 multiplies in fp32 (or fp16 under
 * autocast).  Edge behaviour of mode 1: an activation that is Inf, NaN or above the bf16 maximum
 * (3.39e38) yields NaN in every output it touches (x - bf16(x) is Inf - Inf), where fp32 products
 * would give Inf; finite inputs below that are unaffected. */
int sg_spconv_set_arithmetic(int mode);

/* Where the partial sums of a layer whose kernel offsets are split over several workgroups (layers
 * with few output rows) are added up (process-wide, not thread-safe; tests and A/B measurements):
 * 1 = inside the launch, by the last workgroup to arrive at each (tile, column unit), partial tiles
 * added in the fixed order 0 .. ksplit-1 -- the default; 0 = by a second kernel (conv_reduce_kernel),
 * same order, identical results; -1 = back to the environment (SG_CONV_COMBINE). */
int sg_spconv_set_combine(int mode);

/* Multi-layer conv launches ("conv chain", csrc/spconv_conv.hip): inside sg_unet_forward the layers of
 * the U-Net levels with at most SG_CONV_CHAIN_ROWS rows (default 6144) -- softgroup/model/blocks.py:82-143
 * from the strided conv into such a level down to the deepest level and back up, and a whole U-Net that
 * is that small (the tiny U-Net, softgroup/model/softgroup.py:93-95) -- run as ONE persistent launch per
 * <= 22 layers, with a grid barrier between two layers, instead of one launch per layer.  mode 1 = on,
 * 0 = every layer its own launch (same decomposition, bit-identical results; the default: measured
 * neutral for one scan at a time and slower with several scans in flight, profiles/r06_conv_chain.txt),
 * -1 = back to the environment (SG_CONV_CHAIN).  Process-wide, not thread-safe: tests and A/B measurements.
 * sg_spconv_chain_stats: chain launches and the steps (layers, concats) they carried since process start. */
int sg_spconv_set_chain(int mode);
int sg_spconv_chain_stats(int64_t *launches, int64_t *steps);

/* Measurement hook (bench.py roofline): while enabled, every sg_spconv_gather_conv_f32 call -- from
 * Python or from inside sg_unet_forward -- is bracketed by a HIP event pair on its launch stream.
 * sg_spconv_profile_read waits for the events and returns the summed kernel time and the number of
 * calls since the last sg_spconv_profile(1). */
int sg_spconv_profile(int enable);
int sg_spconv_profile_read(double *total_ms, int *launches);
/* Per call since the last sg_spconv_profile(1), in call order: ms[i] and dims[5*i .. 5*i+4] = num_out_rows,
 * kvol, cin, cout, num_in_rows; at most `cap` entries are written, *calls = number recorded (the profiler is
 * a single-threaded developer hook: one stream at a time). */
int sg_spconv_profile_detail(float *ms, int32_t *dims, int cap, int *calls);

/* ---- point-wise heads (csrc/heads.hip) -------------------------------------------------------
 * Replaces, in SoftGroup.forward_backbone / forward_test (softgroup/model/softgroup.py:374-376,320), the
 * devoxelize gather `output.features[input_map.long()]`, the two MLP heads `semantic_linear` /
 * `offset_linear` (softgroup/model/blocks.py:9-27: Linear, BatchNorm1d, ReLU, Linear) in eval mode and
 * `semantic_scores.max(1)[1]` by one kernel.  sg_mlp2: w1 [C][C] and w2 [out][C] in nn.Linear's [out, in]
 * layout, b1 [C], b2 [out], the eval-mode BatchNorm1d folded to y = x * bn_scale + bn_shift.
 * v2p_map [n_points] int32 or int64 (NULL: identity), channels 16 or 32, semantic->out <= 32,
 * offset->out <= 4.  output_feats [n_points, channels] and semantic_preds [n_points] (int64, first
 * maximum) may be NULL.  fp32 FMA chains in ascending channel order (deterministic). */
typedef struct sg_mlp2 {
  const float *w1, *b1, *bn_scale, *bn_shift, *w2, *b2;
  int out;
} sg_mlp2;
int sg_pointwise_heads(const float *voxel_feats, const void *v2p_map, int v2p_is_int64, int n_points,
                       int channels, const sg_mlp2 *semantic, const sg_mlp2 *offset, float *output_feats,
                       float *semantic_scores, float *pt_offsets, int64_t *semantic_preds,
                       sg_stream_t stream);

/* ---- training side of the sparse convolution (csrc/spconv_train.hip; spconv's autograd reached from
 * tools/train.py:47-58 under autocast) ---------------------------------------------------------
 * bf16 operands, fp32 accumulation (v_mfma_f32_32x32x16_bf16), bf16 result rounded to nearest even:
 *   out[j,:] = bf16( sum_k in[nbr[j,k],:] . W[k] )
 * Weights are packed from the fp32 master copy by sg_spconv_pack_weight_bf16 (layout
 * [K][ceil16(Cin)/8][Cout][8] bf16, zero padded); order / tile_mask / nbr_tiles are the plan of
 * sg_spconv_plan (all three, or none of them and `nbr` [M_out][K] instead).  ws: deep levels split the
 * offsets and need sg_spconv_conv_bf16_workspace_bytes (may be NULL: no split).  The input gradient
 * is the same entry point on the transposed gather table with transposed weights. */
size_t sg_spconv_packed_weight_elems_bf16(int kvol, int cin, int cout);
int sg_spconv_pack_weight_bf16(const float *w, int cout, int kvol, int cin, int src_is_kio,
                               uint16_t *w_k8_bf16, sg_stream_t stream);
size_t sg_spconv_conv_bf16_workspace_bytes(int num_out_rows, int cout);
int sg_spconv_gather_conv_bf16(const uint16_t *in, int num_in_rows, const int32_t *nbr, int num_out_rows,
                               int kvol, int cin, int cout, const uint16_t *w_k8_bf16,
                               const int32_t *order, const uint32_t *tile_mask, const int32_t *nbr_tiles,
                               uint16_t *out, void *ws, size_t ws_bytes, sg_stream_t stream);

/* Weight gradient: dw_kio[k][ci][co] = sum_j in[nbr[j,k]][ci] * g_out[j][co]  ([K][Cin][Cout] fp32,
 * overwritten).  in / g_out are fp32, or bf16 when the matching flag is set (widened on load; products
 * and sums are fp32).  Row chunks are accumulated separately and added in chunk order: the result is
 * bit-identical from run to run.  ws: sg_spconv_wgrad_workspace_bytes. */
/* nbr_t is the gather table transposed to [K][M_out] (sg_spconv_transpose_table; one per rulebook,
 * shared by all convs that use it): the kernel reads one offset's column at a time. */
int sg_spconv_transpose_table(const int32_t *nbr, int num_out_rows, int kvol, int32_t *nbr_t,
                              sg_stream_t stream);
size_t sg_spconv_wgrad_workspace_bytes(int num_out_rows, int kvol, int cin, int cout);
int sg_spconv_wgrad(const void *in, int in_is_bf16, const void *g_out, int g_is_bf16, const int32_t *nbr_t,
                    int num_out_rows, int kvol, int cin, int cout, float *dw_kio, void *ws,
                    size_t ws_bytes, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Native executor of the sparse U-Net (inference): the whole of
 *   [input_conv] -> UBlock -> [output BatchNorm1d + ReLU]
 * (softgroup/model/softgroup.py:60-65 backbone, :93-95 tiny U-Net; modules of
 * softgroup/model/blocks.py:44-143) in one call: rulebooks, plans and convolutions are launched
 * back to back from C++, device memory comes from the caller's arena.  All pointers are device
 * pointers; weights are packed with sg_spconv_pack_weight, BatchNorm1d is passed in eval form
 * y = x*scale + shift.
 * ---------------------------------------------------------------------------------------- */
typedef struct sg_unet_block {     /* ResidualBlock, blocks.py:44-79 */
  int cin, cout;
  const float *bn1_scale, *bn1_shift;   /* [cin]  conv_branch.0 */
  const float *w1;                      /* SubMConv3d(cin, cout)   conv_branch.2 */
  const float *bn2_scale, *bn2_shift;   /* [cout] conv_branch.3 */
  const float *w2;                      /* SubMConv3d(cout, cout)  conv_branch.5 */
  const float *w_i;                     /* i_branch 1x1 conv (cin != cout) or NULL */
} sg_unet_block;
typedef struct sg_unet_level {     /* UBlock, blocks.py:82-143 */
  int planes, n_blocks;
  const sg_unet_block *blocks;                                /* [n_blocks] */
  const sg_unet_block *tail;                                  /* [n_blocks] blocks_tail, NULL on the deepest level */
  const float *down_bn_scale, *down_bn_shift, *down_w;        /* conv:   BN, ReLU, SparseConv3d(planes, next, k2 s2) */
  const float *up_bn_scale, *up_bn_shift, *up_w;              /* deconv: BN, ReLU, SparseInverseConv3d(next, planes, k2) */
} sg_unet_level;
typedef struct sg_unet_desc {
  int n_levels;
  const sg_unet_level *levels;       /* host array, outermost first */
  int input_cin;
  const float *input_w;              /* SubMConv3d(input_cin, planes[0]) before the UBlock, or NULL */
  const float *out_bn_scale, *out_bn_shift;   /* BatchNorm1d + ReLU after the UBlock, or NULL */
  int input_cin_packed;              /* Cin `input_w` was packed for: 0 or input_cin = as is; a larger
                                        multiple of 16 (weights zero-padded along Cin before packing)
                                        makes the executor convolve a zero-padded copy of the features
                                        on the persistent MFMA kernel instead of the general one */
  int arithmetic;                    /* 0 = the process-wide conv arithmetic (fp32 products); 2 = bf16
                                        operands for this call's convolutions (sg_spconv_set_arithmetic's
                                        mode 2, scoped to the call and the calling thread), activations
                                        fp32 in memory; 3 = bf16 operands AND bf16 activations between the
                                        layers (needs input_w and planes % 32 == 0 on every level, else
                                        it runs as 2): what spconv does with a frozen backbone under the
                                        reference's autocast (tools/train.py:47).  `feats` and `out` stay fp32 */
} sg_unet_desc;
/* upper bound of the arena sg_unet_forward needs for num_rows input voxels */
size_t sg_unet_arena_bytes(const sg_unet_desc *desc, int num_rows);
/* feats [num_rows, input_cin or planes[0]], indices int32 [num_rows,4], out [num_rows, planes[0]].
 * One host synchronisation per call (the row counts of all levels, read back from an internal
 * stream right at the start). */
int sg_unet_forward(const sg_unet_desc *desc, const float *feats, const int32_t *indices,
                    int num_rows, const int32_t *spatial_shape_host, float *out, void *arena,
                    size_t arena_bytes, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Native TRAINING executor of the same U-Net (the DDP training step, tools/train.py:44-62 ->
 * SoftGroup.forward_train, softgroup.py:113-150: tiny U-Net of the refinement head always, the
 * backbone when it is not in `fixed_modules`).  The reference trains these modules through spconv's
 * autograd functions and torch.nn.BatchNorm1d in train() mode, one interpreter round trip and
 * several launches per layer; here
 *   sg_unet_train_forward   runs the whole forward -- BatchNorm1d with BATCH statistics (biased
 *                           variance for the normalisation, running_mean / running_var updated with
 *                           `momentum`, unbiased variance, as torch does), ReLU, the convolutions --
 *                           and records what the backward needs (activations, statistics, tables) in
 *                           the caller's arena and in a host-side tape;
 *   sg_unet_train_backward  consumes the tape: input gradients by the forward kernel on the
 *                           transposed rulebooks, weight gradients by sg_spconv_wgrad, BatchNorm1d
 *                           gradients from fp64 column sums.  Deterministic: no floating-point atomics.
 * Weights and their gradients are the module's own tensors in the checkpoint layout [Cout][K][Cin]
 * (packing happens inside); a NULL gradient pointer skips that gradient (frozen parameter).
 * ---------------------------------------------------------------------------------------- */
typedef struct sg_train_bn {          /* BatchNorm1d(c) in train() mode; weight == NULL: absent */
  const float *weight, *bias;         /* [c] */
  float *running_mean, *running_var;  /* [c], updated in place by the forward */
  float momentum, eps;
  float *g_weight, *g_bias;           /* [c] written by the backward, or NULL */
} sg_train_bn;
typedef struct sg_train_conv {        /* bias-free sparse conv; w == NULL: absent */
  const float *w;                     /* [Cout][K][Cin] */
  float *g_w;                         /* same layout, written by the backward, or NULL */
} sg_train_conv;
typedef struct sg_unet_train_block {  /* ResidualBlock, blocks.py:44-79 */
  int cin, cout;
  sg_train_bn bn1, bn2;
  sg_train_conv c1, c2, ci;           /* ci: 1x1 conv of the identity branch (cin != cout) */
} sg_unet_train_block;
typedef struct sg_unet_train_level {  /* UBlock, blocks.py:82-143 */
  int planes, n_blocks;
  const sg_unet_train_block *blocks, *tail;      /* [n_blocks]; tail NULL on the deepest level */
  sg_train_bn down_bn, up_bn;
  sg_train_conv down, up;
} sg_unet_train_level;
typedef struct sg_unet_train_desc {
  int n_levels;
  const sg_unet_train_level *levels;  /* host array, outermost first */
  int input_cin;
  sg_train_conv input;                /* SubMConv3d(input_cin, planes[0]) before the UBlock, or absent */
  sg_train_bn out_bn;                 /* BatchNorm1d + ReLU after the UBlock, or absent */
  int arithmetic;                     /* as sg_unet_desc.arithmetic (0 | 2), forward and input gradients */
} sg_unet_train_desc;
/* Arena: index tables + every activation of the forward + the gradients of the backward, bump
 * allocated and alive until the backward has run.  *arena_needed (if not NULL) receives the exact
 * size for this input once the level row counts are known; SG_ERR_WORKSPACE = call again with at
 * least that much.  sg_unet_train_arena_hint: a first guess from the row count alone. */
size_t sg_unet_train_arena_hint(const sg_unet_train_desc *desc, int num_rows);
int sg_unet_train_forward(const sg_unet_train_desc *desc, const float *feats, const int32_t *indices,
                          int num_rows, const int32_t *spatial_shape_host, float *out, void *arena,
                          size_t arena_bytes, size_t *arena_needed, void **tape, sg_stream_t stream);
/* g_out [num_rows, planes[0]]; g_feats [num_rows, input_cin or planes[0]] or NULL.  Parameter
 * gradients go to the pointers of the descriptor the forward was given (they must still be valid).
 * Frees the tape, whatever it returns. */
int sg_unet_train_backward(void *tape, const float *g_out, float *g_feats, sg_stream_t stream);
/* a forward whose backward will never run */
void sg_unet_train_release(void *tape);

/* Fused eval-mode BatchNorm1d + ReLU over [M, C] rows (output_layer, softgroup.py:65):
 * out = relu(x*scale + shift) (relu optional). */
int sg_bn_relu_f32(const float *x, const float *scale, const float *shift, int64_t num_rows,
                   int channels, int relu, float *out, sg_stream_t stream);

/* Row gather: out[i,:] = in[index[i],:]  (devoxelize, softgroup.py:374; feats[c_idxs], :677) */
int sg_gather_rows_f32(const float *in, const int32_t *index, int64_t num_out_rows, int channels,
                       float *out, sg_stream_t stream);
int sg_gather_rows_i64idx_f32(const float *in, const int64_t *index, int64_t num_out_rows,
                              int channels, float *out, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Instance extraction for the inference result (SoftGroup.get_instances, softgroup.py:537-604),
 * all instance classes at once, without the reference's dense int32 [nProposal, N] mask per class.
 *   proposals_idx int32 [S,2] = (proposal, point), mask_scores f32 [S, stride] (stride >= n_classes)
 *   sg_instance_npoint: npoint[p*n_classes + i] = #{pairs of proposal p with mask_scores[e,i] >
 *       mask_thr}  (softgroup.py:569-570,584: the row sums of the dense mask)
 *   sg_instance_runs: inst_of[i*n_prop + p] = output index of the kept (class i, proposal p)
 *       instance or -1 (the caller applies cls_score_thr / min_npoint, softgroup.py:573-588, and
 *       numbers the survivors in the reference's order: class-major, proposal-ascending).
 *       Produces the runs of consecutive points of every kept instance's mask in ascending point
 *       order: runs of instance k = [bounds[k], bounds[k+1]) of starts[] / ends[] (0-based,
 *       end exclusive) -- the input of sg_rle_format_host (lens = ends - starts).  Runs past
 *       runs_capacity are dropped (bounds still count them: check bounds[n_kept] <= capacity;
 *       sum of the kept instances' npoint is always enough).
 * ---------------------------------------------------------------------------------------- */
int sg_instance_npoint(const int32_t *proposals_idx, const float *mask_scores, int64_t num_pairs,
                       int stride, int n_classes, float mask_thr, int n_prop, int32_t *npoint,
                       sg_stream_t stream);
size_t sg_instance_runs_workspace_bytes(int n_kept, int n_points);
int sg_instance_runs(const int32_t *proposals_idx, const float *mask_scores, int64_t num_pairs,
                     int stride, int n_classes, float mask_thr, const int32_t *inst_of, int n_prop,
                     int n_kept, int n_points, int32_t *starts, int32_t *ends, int64_t *bounds,
                     int64_t runs_capacity, void *ws, size_t ws_bytes, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Instance-mask run-length strings in the reference's wire format (softgroup/util/rle.py:5-19,
 * called per instance from softgroup.py:595-603): "start len start len ..." with 1-based starts.
 * runs of instance g = [bounds[g], bounds[g+1]) of (starts, lens), all host int64.
 * Text of instance g = out[out_offsets[g] .. out_offsets[g+1]) (no terminators);
 * out_capacity >= sg_rle_format_bound(total_runs, digits of the largest number).
 * ---------------------------------------------------------------------------------------- */
int64_t sg_rle_format_bound(int64_t total_runs, int digits);
int sg_rle_format_host(const int64_t *starts_host, const int64_t *lens_host,
                       const int64_t *bounds_host, int n_groups, char *out_host,
                       int64_t out_capacity, int64_t *out_offsets_host);
/* The same text produced on the device from the output of sg_instance_runs, without a host round trip
 * in between (the run count is read from bounds[n_inst] on the device; run_capacity = the capacity
 * given to sg_instance_runs, length = mask length): text (uint8, >= sg_rle_format_device_text_bytes)
 * holds "start len " for every run in order; instance g's string is
 * text[text_off[g] .. text_off[g+1] - 1) (the space after its last run excluded; empty when it has
 * no run).  Only text[0 .. text_off[n_inst]) and the n_inst+1 offsets have to travel to the host. */
size_t sg_rle_format_device_workspace_bytes(int64_t run_capacity);
int64_t sg_rle_format_device_text_bytes(int64_t run_capacity, int64_t length);
int sg_rle_format_device(const int32_t *starts, const int32_t *ends, const int64_t *bounds, int n_inst,
                         int64_t run_capacity, int64_t length, uint8_t *text, int64_t text_capacity,
                         int64_t *text_off, void *ws, size_t ws_bytes, sg_stream_t stream);
/* same text from the output of sg_instance_runs (int32 starts and exclusive ends, copied to the host) */
int sg_rle_format_runs_host(const int32_t *starts_host, const int32_t *ends_host,
                            const int64_t *bounds_host, int n_groups, char *out_host,
                            int64_t out_capacity, int64_t *out_offsets_host);

/* ------------------------------------------------------------------------------------------
 * Panoptic fusion (SoftGroup.panoptic_fusion, softgroup/model/softgroup.py:606-639) over the bit rows
 * of the kept instances (row k = N-bit mask of instance k, ceil(N/32) words per row; the rows
 * sg_instance_runs builds, see sg_instances_result.bits): instances are visited in `order`
 * (descending confidence -- the caller sorts, like the reference's np.argsort(scores)[::-1]); one
 * whose overlap with already pasted points exceeds skip_iou (intersect / (npoint + 1e-5) > skip_iou,
 * evaluated in double) is skipped, otherwise its free points take the next panoptic id (from 1) and
 * class label_id[k] + cls_offset.  out[i] = (class & 0xFFFF) | (id << 16); points of thing classes
 * (class >= thing_class_min; the reference hard-codes 11) without an id become semantic_classes.
 * semantic_preds int64 [N].  ws >= sg_panoptic_fusion_workspace_bytes. */
size_t sg_panoptic_fusion_workspace_bytes(int n_inst, int n_points);
int sg_panoptic_fusion(const uint32_t *bits, int n_inst, int n_points, const int32_t *order,
                       const int32_t *label_id, const int64_t *semantic_preds, int cls_offset,
                       double skip_iou, int semantic_classes, int thing_class_min, uint32_t *out,
                       void *ws, size_t ws_bytes, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Native host driver of the grouping head and of the result extraction (csrc/scan_exec.hip): what
 * SoftGroup.forward_grouping + clusters_voxelization (softgroup/model/softgroup.py:411-480,655-709)
 * and get_instances (:537-604) do between the network's dense heads, as one C call each --
 * class selection, ball query, BFS clustering, proposal voxel index and pooled features in the
 * first; kept-instance table, mask runs, RLE text and the copy to the host in the second.  Same
 * kernels as the per-operator entry points above plus fused replacements of the torch glue; same
 * results, bit for bit (every float operation of the glue is the separate IEEE operation torch
 * performs).  Covers the plain SoftGroup path (no pyramid / octree grouping, no lvl_fusion, no
 * sem2ins classes); anything else stays on the per-operator path.
 * Device memory: ONE caller-provided arena per call, carved in call order; results are returned as
 * BYTE OFFSETS into it.  SG_ERR_WORKSPACE + arena_needed (an estimate; retry may ask again) when it
 * is too small.  Both calls synchronise `stream` several times (data-dependent sizes).
 * ---------------------------------------------------------------------------------------- */
typedef struct sg_grouping_cfg {
  int n_points;              /* N */
  int n_sem_classes;         /* columns of `scores` */
  int n_seg;                 /* grouped classes, <= 32 (semantic classes minus ignore_classes) */
  const int32_t *seg_class;  /* device [n_seg], ascending: semantic class of segment s */
  const float *seg_thr;      /* device [n_seg]: npoint_thr, or npoint_thr * class_numpoint_mean (fp32 product) */
  float score_thr;           /* grouping_cfg.score_thr (strict >) */
  int min_npoint;            /* test_cfg.min_npoint: classes with fewer selected points are skipped */
  float radius;              /* grouping_cfg.radius */
  int batch_size;
  float voxel_scale;         /* instance_voxel_cfg.scale */
  int voxel_shape;           /* instance_voxel_cfg.spatial_shape; 0 = stop after the proposals (the
                                training step voxelises them itself, with rand_quantize) */
  int feat_channels;         /* C of point_feats */
} sg_grouping_cfg;
typedef struct sg_grouping_result {   /* host */
  int n_selected, n_neighbours, n_proposals, sum_npoint, n_voxels, max_active;
  size_t proposals_idx;      /* int32 [sum_npoint, 2] = (proposal, scene point) */
  size_t proposals_offset;   /* int32 [n_proposals + 1] */
  size_t voxel_coords;       /* int32 [n_voxels, 4] = (proposal, x, y, z) */
  size_t voxel_offsets;      /* int32 [n_proposals + 1]: first voxel of every proposal */
  size_t voxel_feats;        /* float [n_voxels, C] */
  size_t point_to_voxel;     /* int32 [sum_npoint] */
  size_t arena_used, arena_needed;
  int deferred_classes;      /* sg_scan_grouping_pp: classes whose tail ran behind their giant clusters' replay (0 | 1) */
  int reserved_;
} sg_grouping_result;
/* scores f32 [N, n_sem_classes] = softmax of the semantic logits; pt_offsets, coords_float f32 [N,3];
 * batch_idxs int32 [N]; point_feats f32 [N, C] = backbone features per point. */
int sg_scan_grouping(const sg_grouping_cfg *cfg, const float *scores, const float *pt_offsets,
                     const float *coords_float, const int32_t *batch_idxs, const float *point_feats,
                     void *arena, size_t arena_bytes, sg_grouping_result *result_host,
                     sg_stream_t stream);

/* SoftGroup++ grouping (softgroup/model/softgroup.py:433-466 with with_pyramid / with_octree; per-class level
 * get_level :485-489, pyramid_map :491-498, octree query softgroup/ops/functions.py:14-44, bfs_cluster :278-308,
 * pyramid_inverse_map :500-507) as ONE C call: class selection of all classes at once (one read-back of the
 * per-class counts), then class by class on the operators above -- level voxels and their pooled coordinates /
 * offsets, octree or hashed-grid neighbour lists at radius * level, clusters, inverse map -- and the same
 * proposal voxelisation as sg_scan_grouping.  base.radius is ignored: `radius` and `base_size` are the
 * configuration's Python floats (the level's radius and voxel size are their double products rounded once,
 * as the reference's Python computes them).  Same results as the per-class loop over the operator surface,
 * bit for bit.  2 + 4 host read-backs per grouped class (voxel count, neighbour total, cluster count, rows).
 * A class with a cluster above 16 384 points (its ordered emission is a multi-workgroup replay of milliseconds
 * on a side stream) does not hold up the classes behind it: the rest of that class is queued behind the replay,
 * the next classes run next to it, one join before the proposals are voxelised (result->deferred_classes). */
typedef struct sg_grouping_pp_cfg {
  sg_grouping_cfg base;
  int with_pyramid, with_octree, lvl_fusion;
  double radius;             /* grouping_cfg.radius */
  double base_size;          /* grouping_cfg.pyramid_base_size */
} sg_grouping_pp_cfg;
int sg_scan_grouping_pp(const sg_grouping_pp_cfg *cfg, const float *scores, const float *pt_offsets,
                        const float *coords_float, const int32_t *batch_idxs, const float *point_feats,
                        void *arena, size_t arena_bytes, sg_grouping_result *result_host,
                        sg_stream_t stream);

/* Host-side staging copies of the device-side collate (the reference's collate_fn, data/custom.py:196-256, does
 * them with torch.cat on the CPU): dense / strided row copies, the float64 -> float32 cast of an item's labels and
 * the batch-index column of the collated coordinates, as plain C so that a binding can run them without the
 * interpreter lock (a loader thread preparing the next scan next to the threads that drive the scans in flight). */
int sg_host_copy_2d(void *dst, int64_t dst_pitch_bytes, const void *src, int64_t src_pitch_bytes, int64_t rows,
                    int64_t row_bytes);
int sg_host_cast_f64_f32(float *dst, const double *src, int64_t n);
int sg_host_fill_i64_strided(int64_t *dst, int64_t pitch_elems, int64_t rows, int64_t value);
/* out_max[c] = max over the rows of src[r * cols + c] (INT64_MIN for no rows), cols <= 8: the collate's
 * spatial_shape (data/custom.py:246) without numpy's 1.3 ms axis-0 reduction under the interpreter lock */
int sg_host_colmax_i64(const int64_t *src, int64_t rows, int64_t cols, int64_t *out_max);

typedef struct sg_instances_cfg {
  int n_proposals, n_classes;   /* instance classes (without the background column) */
  int score_stride;             /* columns of cls_prob / iou_scores / mask_scores (n_classes + 1) */
  int64_t sum_npoint;           /* rows of proposals_idx / mask_scores */
  int n_points;                 /* mask length N */
  float cls_score_thr, mask_score_thr;
  int min_npoint;
} sg_instances_cfg;
typedef struct sg_instances_result {   /* host */
  int n_kept;
  size_t off_class, off_score, off_text, text_bytes;   /* byte offsets into host_out */
  size_t host_needed, arena_used, arena_needed;
  size_t bits;        /* arena offset: uint32 [n_kept, ceil(n_points / 32)] bit rows of the kept masks */
  size_t label_id;    /* arena offset: int32 [n_kept] (device copy of the labels) */
} sg_instances_result;
/* proposals_idx int32 [S,2]; mask_scores f32 [S, stride] (per pair); cls_prob f32 [nP, stride] =
 * softmax of the class logits; iou_scores f32 [nP, stride].  host_out (pinned host memory) receives
 *   int64 text_off[n_kept + 1] | int32 label_id[n_kept] at off_class | f32 conf[n_kept] at off_score
 *   | the RLE text at off_text: instance k's "start len ..." string is
 *   text[text_off[k] .. text_off[k+1] - 1) (sg_rle_format_device's convention).
 * Instances come in the reference's order (class-major, proposal-ascending, softgroup.py:566-603). */
int sg_scan_instances(const sg_instances_cfg *cfg, const int32_t *proposals_idx, const float *mask_scores,
                      const float *cls_prob, const float *iou_scores, void *arena, size_t arena_bytes,
                      void *host_out, size_t host_bytes, sg_instances_result *result_host,
                      sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One scan = one call (csrc/scan_forward.hip): the whole of SoftGroup.forward_test
 * (softgroup/model/softgroup.py:299-361) for the instance-segmentation configurations (SoftGroup, and SoftGroup++
 * through desc->with_pyramid / with_octree; not panoptic fusion, lvl_fusion or x4_split) -- voxel feature pooling
 * (:305), backbone (:307-309 -> forward_backbone :363-378), point-wise heads + arg-max, softmax of the
 * semantic scores, grouping head and proposal voxelisation (:411-480, :655-709), tiny U-Net (:671-675),
 * mask / class / IoU heads (:676-686), instance extraction + RLE text (:537-604) and the dense per-point
 * results of get_point_wise_results / get_gt_instances (:641-653) -- chained on the caller's stream with
 * nothing but kernel launches and the data-dependent read-backs in between.  The reference runs one scan
 * per process at a time from its test loop (tools/test.py:145-150); a host thread that makes this ONE call
 * holds no interpreter lock while the scan runs.  Same kernels and entry points as the calls above
 * (sg_voxelize_fp's arithmetic, sg_unet_forward, sg_pointwise_heads, sg_scan_grouping, sg_scan_instances);
 * the dense heads of the refinement (nn.Linear / MLP on <= a few thousand rows) are fp32 FMA chains in
 * ascending channel order like sg_pointwise_heads.
 *   arena: device scratch, carved in call order; SG_ERR_WORKSPACE + result->arena_needed when too small.
 *   host_dense / host_inst: PINNED host memory for the dense results and for sg_scan_instances' image.
 * ---------------------------------------------------------------------------------------- */
typedef struct sg_linear {            /* nn.Linear: w [out][in], b [out] */
  const float *w, *b;
  int out, in;
} sg_linear;
/* row-wise softmax with torch's arithmetic for rows of <= 32 columns (softmax_warp_forward: maximum,
 * exp(x - max) by expf, the sum as a 32-lane butterfly -- offsets 16, 8, 4, 2, 1 -- and one division per
 * element): bit-identical to tensor.softmax(-1) on this build (tests/test_scan_forward_gpu.py) */
int sg_softmax_rows(const float *x, int64_t rows, int cols, float *out, sg_stream_t stream);
/* out[r, :] = mlp(feats[idx[r], :]) (idx NULL: identity); channels 16 or 32, mlp->out <= 32 */
int sg_mlp_rows(const float *feats, const int32_t *idx, int64_t rows, int channels, const sg_mlp2 *mlp,
                float *out, sg_stream_t stream);
/* out[r, j] = b[j] + sum_c x[r, c] * w[j, c], ascending c (fmaf chain) */
int sg_linear_rows(const float *x, int64_t rows, const sg_linear *lin, float *out, sg_stream_t stream);

#define SG_SCAN_DENSE_MAX 16
typedef struct sg_scan_dense_item {
  int kind;            /* 0 = `ptr`/`bytes` (a device buffer passed through to the host block),
                          1 = semantic_preds int64 [N], 2 = offset_preds f32 [N,3],
                          3 = gt_instances int64 [N] (softgroup.py:641-653 from semantic_labels / instance_labels),
                          4 = semantic_scores f32 [N, n_sem] (the logits) */
  const void *ptr;
  size_t bytes;
} sg_scan_dense_item;
typedef struct sg_scan_desc {
  const sg_unet_desc *backbone;       /* input_conv + unet + output_layer */
  const sg_unet_desc *tiny;           /* tiny_unet + tiny_unet_outputlayer */
  sg_mlp2 semantic, offset, mask;     /* semantic_linear, offset_linear, mask_linear */
  sg_linear cls, iou;                 /* cls_linear, iou_score_linear */
  int channels;                       /* backbone output channels: 16 or 32 */
  int with_coords;                    /* voxel features = [feats | coords_float] */
  int semantic_classes, instance_classes;
  sg_grouping_cfg grouping;           /* n_points / batch_size / feat_channels are filled per call */
  float cls_score_thr, mask_score_thr;
  int min_npoint;
  int want_instances;                 /* 0: stop after the point-wise heads */
  /* SoftGroup++ grouping (sg_scan_grouping_pp instead of sg_scan_grouping) when either switch is set; pp_radius /
   * pp_base_size are the configuration's Python floats (grouping_cfg.radius, .pyramid_base_size) */
  int with_pyramid, with_octree;
  double pp_radius, pp_base_size;
} sg_scan_desc;
typedef struct sg_scan_input {        /* one collated batch, device pointers (data/custom.py:240-256) */
  int n_points, n_voxels, max_active; /* p2v_map is int32 [n_voxels, 1 + max_active] */
  int batch_size;
  int spatial_shape[3];
  const float *feats;                 /* f32 [N, feat_dim] */
  int feat_dim;
  const float *coords_float;          /* f32 [N, 3] */
  const int32_t *p2v_map;
  const void *v2p_map;                /* [N] int32 or int64 */
  int v2p_is_int64;
  const void *voxel_coords;           /* [M, 4] int32 or int64 */
  int voxel_coords_is_int64;
  const int32_t *batch_idxs;          /* int32 [N] */
  const int64_t *semantic_labels, *instance_labels;   /* for dense kind 3 (may be NULL) */
  int n_dense;
  sg_scan_dense_item dense[SG_SCAN_DENSE_MAX];
} sg_scan_input;
typedef struct sg_scan_result {       /* host */
  sg_grouping_result grouping;        /* counts; its offsets are relative to `grouping_base` */
  sg_instances_result instances;      /* offsets into host_inst / relative to `instances_base` */
  size_t grouping_base, instances_base;               /* arena offsets of the two stages' sub-arenas */
  size_t dense_offset[SG_SCAN_DENSE_MAX];             /* byte offsets into host_dense */
  size_t dense_bytes;                                 /* bytes of host_dense in use */
  /* arena offsets of the stage outputs (device; valid until the next call on this arena) */
  size_t voxel_feats_in, backbone_out, output_feats, semantic_scores, semantic_prob, pt_offsets,
      semantic_preds, tiny_out, mask_scores, cls_scores, cls_prob, iou_scores;
  size_t arena_used, arena_needed, host_dense_needed, host_inst_needed;
  int stage;                          /* how far the call got: 1 heads, 2 grouping, 3 refinement, 4 instances */
} sg_scan_result;
size_t sg_scan_arena_bytes(const sg_scan_desc *desc, int n_points, int n_voxels);
int sg_scan_forward(const sg_scan_desc *desc, const sg_scan_input *in, void *arena, size_t arena_bytes,
                    void *host_dense, size_t host_dense_bytes, void *host_inst, size_t host_inst_bytes,
                    sg_scan_result *result_host, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation (ScanNetEval.assign_instances_for_scan, softgroup/evaluation/instance_eval.py:228-309):
 * counts[p*n_slots + s] = number of points of prediction p's mask whose ground-truth slot is s.
 * Masks come as runs: run r covers points run_start[r] .. run_start[r] + len(r) - 1 of prediction
 * run_pred[r]; run_off[r] = sum of the lengths of runs < r (run_off[n_runs] = total_points).
 * gt_slot[point] in [0, n_slots): index of the point's GT instance, n_slots-1 = void.
 * ---------------------------------------------------------------------------------------- */
int sg_eval_intersections(const int32_t *run_start, const int64_t *run_off, const int32_t *run_pred,
                          int n_runs, int64_t total_points, const int32_t *gt_slot, int n_pred,
                          int n_slots, int32_t *counts, sg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SOFTGROUP_HIP_H */
