// TEST-ONLY translation unit: the reference's CUDA kernels, compiled UNMODIFIED from where they
// lie (/root/reference/softgroup/ops/src/cuda.cu = unity TU of bfs_cluster.cu, cal_iou_and_masklabel.cu,
// octree_ball_query.cu, roipool.cu, sec_mean.cu, voxelize.cu) with hipcc --offload-arch=gfx950 into
// oracle/_ref/sg_ref_gpu_ops.so.  Nothing is copied into this repo.  The extern "C" entry points
// below call the reference's own host launchers (`*_cuda`), i.e. exactly what the pybind
// functions of softgroup/ops/src/softgroup_api.cpp:8-28 reach after unwrapping their at::Tensor
// arguments (the .cpp wrappers only do `tensor.data_ptr<T>()`, e.g. sec_mean/sec_mean.cpp:9-16,
// bfs_cluster/bfs_cluster.cpp:17-31, octree_ball_query/octree_ball_query.cpp:167-188).
// Used by tests/test_ref_gpu_kernels.py to pin (i) oracle/sg_oracle.c and (ii) the HIP kernels
// against the reference's kernels on the GPU box.  Recipe: oracle/build_ref.py.
#include "sg_cuda_on_hip.h"

#include "cuda.cu"

extern "C" {

void sgref_set_stream(void *stream) { at::cuda::sg_ref_stream() = static_cast<hipStream_t>(stream); }

// voxelize/voxelize.cu:30-37,56-63 (feats/out device pointers; out zero-filled by the caller as
// functions.py:216,228 does)
void sgref_voxelize_fp(int nOutputRows, int maxActive, int nPlanes, float *feats, float *output_feats,
                       int *rules, int average) {
  voxelize_fp_cuda<float>(nOutputRows, maxActive, nPlanes, feats, output_feats, rules, average != 0);
}
void sgref_voxelize_bp(int nOutputRows, int maxActive, int nPlanes, float *d_output_feats,
                       float *d_feats, int *rules, int average) {
  voxelize_bp_cuda<float>(nOutputRows, maxActive, nPlanes, d_output_feats, d_feats, rules, average != 0);
}

// bfs_cluster/bfs_cluster.cu:68-101
int sgref_ballquery_batch_p(int n, int meanActive, float radius, const float *xyz,
                            const int *batch_idxs, const int *batch_offsets, int *idx,
                            int *start_len) {
  return ballquery_batch_p_cuda(n, meanActive, radius, xyz, batch_idxs, batch_offsets, idx,
                                start_len, at::cuda::getCurrentCUDAStream());
}

// octree_ball_query/octree_ball_query.cu:128-147
int sgref_octree_ball_query(const float *points, const float *boxes, const int *pt_inds,
                            const int *pt_start_len, int *out_inds, int *out_start_len,
                            int mean_active, float radius, int num_points, int num_nodes,
                            int num_leaves) {
  return octree_ball_query_cuda_launcher(points, boxes, pt_inds, pt_start_len, out_inds,
                                         out_start_len, mean_active, radius, num_points,
                                         num_nodes, num_leaves);
}

// sec_mean/sec_mean.cu:34-37,62-65,90-93
void sgref_sec_mean(int nProposal, int C, float *inp, int *offsets, float *out) {
  sec_mean_cuda(nProposal, C, inp, offsets, out);
}
void sgref_sec_min(int nProposal, int C, float *inp, int *offsets, float *out) {
  sec_min_cuda(nProposal, C, inp, offsets, out);
}
void sgref_sec_max(int nProposal, int C, float *inp, int *offsets, float *out) {
  sec_max_cuda(nProposal, C, inp, offsets, out);
}

// roipool/roipool.cu:38-43,66-71
void sgref_global_avg_pool_fp(int nProposal, int C, float *feats, int *proposals_offset,
                              float *output_feats) {
  global_avg_pool_fp_cuda(nProposal, C, feats, proposals_offset, output_feats);
}
void sgref_global_avg_pool_bp(int nProposal, int C, float *d_feats, int *proposals_offset,
                              float *d_output_feats) {
  global_avg_pool_bp_cuda(nProposal, C, d_feats, proposals_offset, d_output_feats);
}

// cal_iou_and_masklabel/cal_iou_and_masklabel.cu:131-164
void sgref_get_mask_iou_on_cluster(int nInstance, int nProposal, int *proposals_idx,
                                   int *proposals_offset, long *instance_labels,
                                   int *instance_pointnum, float *proposals_iou) {
  get_mask_iou_on_cluster_cuda(nInstance, nProposal, proposals_idx, proposals_offset,
                               instance_labels, instance_pointnum, proposals_iou);
}
void sgref_get_mask_iou_on_pred(int nInstance, int nProposal, int *proposals_idx,
                                int *proposals_offset, long *instance_labels,
                                int *instance_pointnum, float *proposals_iou,
                                float *mask_scores_sigmoid) {
  get_mask_iou_on_pred_cuda(nInstance, nProposal, proposals_idx, proposals_offset,
                            instance_labels, instance_pointnum, proposals_iou, mask_scores_sigmoid);
}
void sgref_get_mask_label(int nInstance, int nProposal, float iou_thr, int *proposals_idx,
                          int *proposals_offset, long *instance_labels, long *instance_cls,
                          float *proposals_iou, float *mask_label) {
  get_mask_label_cuda(nInstance, nProposal, iou_thr, proposals_idx, proposals_offset,
                      instance_labels, instance_cls, proposals_iou, mask_label);
}

int sgref_sync(void) { return static_cast<int>(hipDeviceSynchronize()); }

}  // extern "C"
