// TEST-ONLY translation unit: compiles the reference's CPU-side ops *from where
// they lie* under /root/reference (nothing is copied into this repo) into
// oracle/_ref/sg_ref_ops*.so.  Used to (1) pin oracle/sg_oracle.c and (2) generate
// tests/golden/*.npz.  Recipe: oracle/build_ref.py.  See SURVEY.md App. C.
#include <torch/extension.h>
#include <limits>
#include <stdexcept>
#include "bfs_cluster/bfs_cluster.cpp"
#include "datatype/datatype.cpp"
#include "octree_ball_query/octree_ball_query.cpp"
#include "voxelize/voxelize.cpp"

// GPU-only entry points of the reference: never called from the CPU oracle.
int ballquery_batch_p_cuda(int, int, float, const float *, const int *, const int *, int *, int *,
                           cudaStream_t) {
  throw std::runtime_error("oracle/_ref: ballquery_batch_p_cuda is CUDA-only in the reference");
}
int octree_ball_query_cuda_launcher(const float *, const float *, const int *, const int *, int *,
                                    int *, const int, const float, const int, const int,
                                    const int) {
  throw std::runtime_error("oracle/_ref: octree_ball_query_cuda is CUDA-only in the reference");
}
template <typename T>
void voxelize_fp_cuda(Int, Int, Int, T *, T *, Int *, bool) {
  throw std::runtime_error("oracle/_ref: voxelize_fp_cuda is CUDA-only in the reference");
}
template <typename T>
void voxelize_bp_cuda(Int, Int, Int, T *, T *, Int *, bool) {
  throw std::runtime_error("oracle/_ref: voxelize_bp_cuda is CUDA-only in the reference");
}

static void voxelize_idx_3d(at::Tensor coords, at::Tensor output_coords, at::Tensor input_map,
                            at::Tensor output_map, Int batchSize, Int mode) {
  voxelize_idx<3>(coords, output_coords, input_map, output_map, batchSize, mode);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("voxelize_idx", &voxelize_idx_3d);
  m.def("bfs_cluster", &bfs_cluster);
  m.def("build_and_export_octree", &build_and_export_octree);
}
