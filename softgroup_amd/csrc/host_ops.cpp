// host_ops.cpp -- the two operators the reference itself runs on the CPU and that callers use
// without a GPU context (DataLoader worker processes): voxel index build and octree export.
// Native C++ (no torch, no HIP); the device variants live in voxelize_idx.hip / octree.hip.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/softgroup_hip.h"

namespace sg {
void set_error(const char *fmt, ...);
}

namespace {

struct CellKey {
  int32_t b, x, y, z;
  bool operator==(const CellKey &o) const { return b == o.b && x == o.x && y == o.y && z == o.z; }
};

inline uint64_t hash_key(const CellKey &k) {
  uint64_t h = (static_cast<uint64_t>(static_cast<uint32_t>(k.b)) << 32) ^ static_cast<uint32_t>(k.x);
  h *= 0x9E3779B97F4A7C15ULL;
  h ^= (static_cast<uint64_t>(static_cast<uint32_t>(k.y)) << 32) | static_cast<uint32_t>(k.z);
  h ^= h >> 31;
  h *= 0xD6E8FEB86659FD93ULL;
  return h ^ (h >> 32);
}

inline CellKey load_key(const int64_t *row, int ncol) {
  // coordinates are narrowed to int32 like the reference (voxelize.cpp:86-102)
  if (ncol == 3) return {0, static_cast<int32_t>(row[0]), static_cast<int32_t>(row[1]), static_cast<int32_t>(row[2])};
  return {static_cast<int32_t>(row[0]), static_cast<int32_t>(row[1]), static_cast<int32_t>(row[2]),
          static_cast<int32_t>(row[3])};
}

}  // namespace

extern "C" {

// voxel id = order of first appearance; reference: voxelize/voxelize.cpp:70-163
int sg_voxelize_idx_host(const int64_t *coords, int n, int ncol, int mode, int32_t *input_map,
                         int32_t *num_voxels, int32_t *max_active) {
  if (n < 0 || (ncol != 3 && ncol != 4) || mode < 0 || mode > 4) {
    sg::set_error("sg_voxelize_idx_host: bad arguments (n=%d ncol=%d mode=%d)", n, ncol, mode);
    return SG_ERR_ARG;
  }
  size_t cap = 64;
  while (cap < static_cast<size_t>(n) * 2) cap <<= 1;
  struct Slot { int32_t first; int32_t voxel; };
  std::vector<Slot> table(cap, Slot{-1, -1});
  std::vector<int32_t> counts;
  counts.reserve(static_cast<size_t>(n) / 2 + 1);
  for (int i = 0; i < n; ++i) {
    const CellKey k = load_key(coords + static_cast<size_t>(i) * ncol, ncol);
    size_t s = hash_key(k) & (cap - 1);
    while (true) {
      Slot &sl = table[s];
      if (sl.first < 0) {
        sl.first = i;
        sl.voxel = static_cast<int32_t>(counts.size());
        counts.push_back(0);
      } else if (!(load_key(coords + static_cast<size_t>(sl.first) * ncol, ncol) == k)) {
        s = (s + 1) & (cap - 1);
        continue;
      }
      input_map[i] = sl.voxel;
      ++counts[sl.voxel];
      break;
    }
  }
  int32_t ma = 1;
  if (mode == 3 || mode == 4)
    for (int32_t c : counts) ma = std::max(ma, c);
  *num_voxels = static_cast<int32_t>(counts.size());
  *max_active = ma;
  return SG_OK;
}

int sg_voxelize_idx_fill_host(const int64_t *coords, int n, int ncol, int mode,
                              const int32_t *input_map, int num_voxels, int max_active,
                              int64_t *out_coords, int32_t *out_map) {
  if (n < 0 || num_voxels < 0 || max_active < 1 || (ncol != 3 && ncol != 4)) {
    sg::set_error("sg_voxelize_idx_fill_host: bad arguments");
    return SG_ERR_ARG;
  }
  const size_t stride = static_cast<size_t>(max_active) + 1;
  std::fill(out_map, out_map + static_cast<size_t>(num_voxels) * stride, 0);
  const bool pooled = (mode == 3 || mode == 4);
  for (int i = 0; i < n; ++i) {
    int32_t *row = out_map + static_cast<size_t>(input_map[i]) * stride;
    if (pooled) {
      row[1 + row[0]] = i;  // points arrive in ascending index order
      ++row[0];
    } else if (row[0] == 0 || mode == 2) {  // modes 0/1 keep the first point, 2 the last
      row[0] = 1;
      row[1] = i;
    }
  }
  for (int v = 0; v < num_voxels; ++v)
    std::memcpy(out_coords + static_cast<size_t>(v) * ncol,
                coords + static_cast<size_t>(out_map[v * stride + 1]) * ncol, sizeof(int64_t) * ncol);
  return SG_OK;
}

// Octree export without building a pointer tree.  Reference: octree_ball_query.cpp:8-165.
// The tree is complete, so node `path` at level L sits at BFS slot first[L] + path where
// path = base-8 digits (octant per level) and the box of a child follows from its parent's
// with the reference's float expressions (cpp:60-82).  A point's leaf is found by walking the
// boxes with the reference's `<` tests (cpp:52-57); leaves hold their points in ascending
// index order because every level of the reference splits an ascending list in order.
int sg_octree_build_host(const float *points, const float *xyzwhl, int num_points, int num_levels,
                         float *boxes, int32_t *pt_inds, int32_t *pt_start_len) {
  if (num_points < 0 || num_levels < 1 || num_levels > 7) {
    sg::set_error("sg_octree_build_host: bad arguments (n=%d levels=%d)", num_points, num_levels);
    return SG_ERR_ARG;
  }
  std::vector<size_t> first(num_levels + 2, 0);
  size_t width = 1;
  for (int l = 0; l <= num_levels; ++l) {
    first[l + 1] = first[l] + width;
    width *= 8;
  }
  std::memcpy(boxes, xyzwhl, 6 * sizeof(float));
  width = 1;
  for (int l = 0; l < num_levels; ++l, width *= 8) {
    for (size_t path = 0; path < width; ++path) {
      const float *pa = boxes + (first[l] + path) * 6;
      const float w = pa[3] / 2, h = pa[4] / 2, d = pa[5] / 2;
      for (int oct = 0; oct < 8; ++oct) {
        float *c = boxes + (first[l + 1] + path * 8 + oct) * 6;
        c[0] = (oct & 1) ? pa[0] + w / 2 : pa[0] - w / 2;
        c[1] = (oct & 2) ? pa[1] + h / 2 : pa[1] - h / 2;
        c[2] = (oct & 4) ? pa[2] + d / 2 : pa[2] - d / 2;
        c[3] = w; c[4] = h; c[5] = d;
      }
    }
  }
  const size_t num_leaves = width;
  std::vector<int32_t> leaf_of(static_cast<size_t>(num_points));
  std::vector<int32_t> fill(num_leaves + 1, 0);
  for (int i = 0; i < num_points; ++i) {
    const float *p = points + static_cast<size_t>(i) * 3;
    size_t path = 0;
    for (int l = 0; l < num_levels; ++l) {
      const float *b = boxes + (first[l] + path) * 6;
      const int oct = (p[0] < b[0] ? 0 : 1) | (p[1] < b[1] ? 0 : 2) | (p[2] < b[2] ? 0 : 4);
      path = path * 8 + oct;
    }
    leaf_of[i] = static_cast<int32_t>(path);
    ++fill[path + 1];
  }
  for (size_t l = 0; l < num_leaves; ++l) {
    pt_start_len[2 * l] = fill[l];
    pt_start_len[2 * l + 1] = fill[l + 1];
    fill[l + 1] += fill[l];
  }
  for (int i = 0; i < num_points; ++i) pt_inds[fill[leaf_of[i]]++] = i;
  return SG_OK;
}


// Run-length strings of instance masks in the reference's wire format (util/rle.py:5-19:
// "start len start len ..." with 1-based starts), for all instances of a scan at once.
// runs of instance g are [bounds[g], bounds[g+1]) in (starts, lens); string g occupies
// out[out_offsets[g] .. out_offsets[g+1]) (no terminators).  Multi-threaded: the reference does
// this with a Python loop per instance (softgroup.py:595-603).
static inline int dec_len(int64_t v) {
  if (v < 10) return 1;
  if (v < 100) return 2;
  if (v < 1000) return 3;
  if (v < 10000) return 4;
  if (v < 100000) return 5;
  if (v < 1000000) return 6;
  if (v < 10000000) return 7;
  int n = 8;
  v /= 100000000;
  while (v) { v /= 10; ++n; }
  return n;
}
static const char kDigitPairs[] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
    "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
    "8081828384858687888990919293949596979899";
static inline char *put_dec(char *p, int64_t v) {
  const int n = dec_len(v);
  char *e = p + n;
  char *q = e;
  uint64_t u = static_cast<uint64_t>(v);
  while (u >= 100) {
    const unsigned r = static_cast<unsigned>(u % 100);
    u /= 100;
    q -= 2;
    q[0] = kDigitPairs[2 * r];
    q[1] = kDigitPairs[2 * r + 1];
  }
  if (u >= 10) {
    q -= 2;
    q[0] = kDigitPairs[2 * u];
    q[1] = kDigitPairs[2 * u + 1];
  } else {
    *--q = static_cast<char>('0' + u);
  }
  return e;
}
}  // extern "C"

template <typename StartFn, typename LenFn>
static int rle_format_impl(StartFn start_of, LenFn len_of, const int64_t *bounds, int n_groups, char *out,
                           int64_t out_capacity, int64_t *out_offsets) {
  // One parallel region: every thread sizes its contiguous range of groups, the ranges' byte
  // totals are prefix-summed after a rendezvous, then every thread writes its groups.
  // out_capacity must be >= sg_rle_format_bound(total_runs); out_offsets[n_groups+1].
  if (n_groups < 0 || !bounds || !out_offsets || !out) {
    sg::set_error("sg_rle_format_host: bad arguments");
    return SG_ERR_ARG;
  }
  const int hw = static_cast<int>(std::thread::hardware_concurrency());
  const int n_thr = std::max(1, std::min({hw > 0 ? hw : 1, 16, n_groups / 8 + 1}));
  std::vector<int64_t> part(n_thr + 1, 0);
  std::atomic<int> arrived{0};
  std::atomic<bool> overflow{false};
  auto body = [&](int t) {
    const int lo = static_cast<int>(static_cast<int64_t>(n_groups) * t / n_thr);
    const int hi = static_cast<int>(static_cast<int64_t>(n_groups) * (t + 1) / n_thr);
    int64_t total = 0;
    for (int g = lo; g < hi; ++g) {
      int64_t bytes = 0;
      for (int64_t r = bounds[g]; r < bounds[g + 1]; ++r)
        bytes += dec_len(start_of(r) + 1) + dec_len(len_of(r)) + 2;
      bytes = bytes > 0 ? bytes - 1 : 0;  // no trailing space
      out_offsets[g + 1] = bytes;          // local size, turned into an offset below
      total += bytes;
    }
    part[t + 1] = total;
    arrived.fetch_add(1, std::memory_order_acq_rel);
    while (arrived.load(std::memory_order_acquire) < n_thr) std::this_thread::yield();
    int64_t base = 0;
    for (int i = 0; i < t; ++i) base += part[i + 1];
    for (int g = lo; g < hi; ++g) {
      const int64_t bytes = out_offsets[g + 1];
      if (base + bytes > out_capacity) { overflow.store(true); return; }
      char *p = out + base;
      for (int64_t r = bounds[g]; r < bounds[g + 1]; ++r) {
        if (r != bounds[g]) *p++ = ' ';
        p = put_dec(p, start_of(r) + 1);
        *p++ = ' ';
        p = put_dec(p, len_of(r));
      }
      base += bytes;
      out_offsets[g + 1] = base;           // end offset of group g
    }
  };
  if (n_thr == 1) {
    body(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 1; t < n_thr; ++t) pool.emplace_back(body, t);
    body(0);
    for (auto &th : pool) th.join();
  }
  out_offsets[0] = 0;
  if (overflow.load()) {
    sg::set_error("sg_rle_format_host: output buffer too small");
    return SG_ERR_WORKSPACE;
  }
  return SG_OK;
}

extern "C" {

int sg_rle_format_host(const int64_t *starts, const int64_t *lens, const int64_t *bounds,
                       int n_groups, char *out, int64_t out_capacity, int64_t *out_offsets) {
  return rle_format_impl([starts](int64_t r) { return starts[r]; }, [lens](int64_t r) { return lens[r]; },
                         bounds, n_groups, out, out_capacity, out_offsets);
}

// the same from what sg_instance_runs produces: int32 run starts and exclusive ends
int sg_rle_format_runs_host(const int32_t *starts, const int32_t *ends, const int64_t *bounds,
                            int n_groups, char *out, int64_t out_capacity, int64_t *out_offsets) {
  return rle_format_impl([starts](int64_t r) { return static_cast<int64_t>(starts[r]); },
                         [starts, ends](int64_t r) { return static_cast<int64_t>(ends[r]) - starts[r]; },
                         bounds, n_groups, out, out_capacity, out_offsets);
}

// upper bound of the text size for `total_runs` runs whose numbers are < 10^digits
int64_t sg_rle_format_bound(int64_t total_runs, int digits) {
  return total_runs * 2 * (static_cast<int64_t>(digits) + 1) + 16;
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------
// Staging copies of the device-side collate (softgroup_amd/data: the reference's collate_fn,
// data/custom.py:196-256, concatenates its items with torch.cat on the CPU): plain C loops that a ctypes
// binding runs WITHOUT the interpreter lock, so that a loader thread filling the next scan's pinned
// staging buffers does not stall the threads that drive the scans in flight.
// ---------------------------------------------------------------------------------------------
extern "C" {

// rows x row_bytes from src (pitch src_pitch) to dst (pitch dst_pitch); one memcpy when both are dense
int sg_host_copy_2d(void *dst, int64_t dst_pitch, const void *src, int64_t src_pitch, int64_t rows, int64_t row_bytes) {
  if (rows <= 0 || row_bytes <= 0) return SG_OK;
  if (dst == nullptr || src == nullptr || dst_pitch < row_bytes || src_pitch < row_bytes) {
    sg::set_error("sg_host_copy_2d: bad arguments");
    return SG_ERR_ARG;
  }
  if (dst_pitch == row_bytes && src_pitch == row_bytes) {
    memcpy(dst, src, static_cast<size_t>(rows) * row_bytes);
    return SG_OK;
  }
  char *d = static_cast<char *>(dst);
  const char *s = static_cast<const char *>(src);
  for (int64_t r = 0; r < rows; ++r) memcpy(d + r * dst_pitch, s + r * src_pitch, static_cast<size_t>(row_bytes));
  return SG_OK;
}

// dst[i] = (float)src[i] (float64 labels of an item into a float32 batch tensor)
int sg_host_cast_f64_f32(float *dst, const double *src, int64_t n) {
  if (n > 0 && (dst == nullptr || src == nullptr)) {
    sg::set_error("sg_host_cast_f64_f32: null pointer");
    return SG_ERR_ARG;
  }
  for (int64_t i = 0; i < n; ++i) dst[i] = static_cast<float>(src[i]);
  return SG_OK;
}

// the batch-index column of the collated coordinates: dst[r * pitch_elems] = value for r < rows
int sg_host_fill_i64_strided(int64_t *dst, int64_t pitch_elems, int64_t rows, int64_t value) {
  if (rows > 0 && (dst == nullptr || pitch_elems < 1)) {
    sg::set_error("sg_host_fill_i64_strided: bad arguments");
    return SG_ERR_ARG;
  }
  for (int64_t r = 0; r < rows; ++r) dst[r * pitch_elems] = value;
  return SG_OK;
}

// column-wise maximum of a row-major int64 matrix (the collate's spatial_shape = max voxel coordinate + 1,
// data/custom.py:246: numpy's axis-0 reduction of a [150 000, 3] array takes 1.3 ms under the interpreter lock)
int sg_host_colmax_i64(const int64_t *src, int64_t rows, int64_t cols, int64_t *out_max) {
  if (cols < 1 || cols > 8 || out_max == nullptr || (rows > 0 && src == nullptr)) {
    sg::set_error("sg_host_colmax_i64: bad arguments (1..8 columns)");
    return SG_ERR_ARG;
  }
  int64_t m[8];
  for (int64_t c = 0; c < cols; ++c) m[c] = INT64_MIN;
  if (cols == 3) {      // (the usual case; fixed trip counts let the compiler keep the three maxima in registers)
    int64_t a = INT64_MIN, b = INT64_MIN, c3 = INT64_MIN;
    for (int64_t r = 0; r < rows; ++r) {
      const int64_t x = src[3 * r], y = src[3 * r + 1], z = src[3 * r + 2];
      a = x > a ? x : a;
      b = y > b ? y : b;
      c3 = z > c3 ? z : c3;
    }
    m[0] = a; m[1] = b; m[2] = c3;
  } else {
    for (int64_t r = 0; r < rows; ++r)
      for (int64_t c = 0; c < cols; ++c) {
        const int64_t v = src[r * cols + c];
        m[c] = v > m[c] ? v : m[c];
      }
  }
  for (int64_t c = 0; c < cols; ++c) out_max[c] = m[c];
  return SG_OK;
}

}  // extern "C"
