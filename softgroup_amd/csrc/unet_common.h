// unet_common.h -- what the inference executor (unet_exec.hip) and the training executor
// (unet_train.hip) of the sparse U-Net share: the arena, the per-level gather tables / tile plans and
// the whole-pyramid index build that fills them.
#pragma once
#include <mutex>

#include "common.h"

namespace sg {

// bump allocator with stack discipline
struct Arena {
  char *base;
  size_t cap, off, peak;
  Arena(void *p, size_t n) : base(static_cast<char *>(p)), cap(n), off(0), peak(0) {}
  template <typename T>
  T *take(size_t count) {
    const size_t bytes = align_up(count * sizeof(T));
    if (off + bytes > cap) return nullptr;
    T *r = reinterpret_cast<T *>(base + off);
    off += bytes;
    if (off > peak) peak = off;
    return r;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

struct Plan {
  const int32_t *nbr = nullptr;
  int32_t *order = nullptr;
  uint32_t *tile_mask = nullptr;
  int32_t *nbr_tiles = nullptr;
  int rows = 0, kvol = 0;
};

struct LevelIdx {
  int rows = 0;
  int32_t shape[3] = {0, 0, 0};
  Plan subm, down, up, ident;
};


#define SG_TRY(expr)              \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != SG_OK) return rc_; \
  } while (0)

// Tables and plans of all `n_levels` levels of the voxel pyramid over `indices`, laid out from the
// start of `arena` (-> *used bytes), built on an internal index stream that the caller's stream then
// waits for; one host synchronisation (the level row counts).  Per-(device, stream) runtime state;
// `*lock` holds that state's mutex until the caller lets go of it (calls on one stream are serial).
int unet_build_index(const char *who, int n_levels, const int32_t *indices, int num_rows,
                     const int32_t *spatial_shape_host, void *arena, size_t arena_bytes, sg_stream_t stream,
                     LevelIdx *li, size_t *used, std::unique_lock<std::mutex> *lock);
// upper bound of *used for num_rows input voxels
size_t unet_index_bytes(int n_levels, int num_rows);

}  // namespace sg
