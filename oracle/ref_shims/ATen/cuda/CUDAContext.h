// TEST-ONLY shim (oracle/_ref build): the reference headers include
// <ATen/cuda/CUDAContext.h> for cudaStream_t and at::cuda::getCurrentCUDAStream()
// (/root/reference/softgroup/ops/src/bfs_cluster/bfs_cluster.h).  Only the CPU
// functions of the reference are ever called through oracle/_ref, so the stream
// type just has to exist.
#pragma once
typedef void *cudaStream_t;
namespace at { namespace cuda {
inline cudaStream_t getCurrentCUDAStream() { return nullptr; }
}}  // namespace at::cuda
