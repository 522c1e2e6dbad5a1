// core.hip -- library plumbing: error text, device info, the shared scan tail kernel.
#include <stdarg.h>
#include <string.h>

#include "common.h"
#include "scan.h"

namespace sg {

static thread_local char g_err[512] = "";

// per host thread: a small pinned buffer for device->host read-backs (a pageable destination makes
// the runtime lock user pages for every copy, which was measured to cost milliseconds per call in
// processes that also hold large pinned staging areas)
int32_t *pinned_words() {
  static thread_local int32_t *p = nullptr;
  if (p == nullptr && hipHostMalloc(reinterpret_cast<void **>(&p), 256) != hipSuccess) p = nullptr;
  return p;
}

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__global__ void __launch_bounds__(kScanBlock) scan_block_sums_kernel(int32_t *block_sums,
                                                                    int num_blocks,
                                                                    int32_t *total_out) {
  __shared__ int lds4[4];
  int carry = 0;
  for (int base = 0; base < num_blocks; base += kScanBlock) {
    int idx = base + threadIdx.x;
    int v = idx < num_blocks ? block_sums[idx] : 0;
    int tot;
    int incl = block_incl_scan_256(v, lds4, &tot);
    if (idx < num_blocks) block_sums[idx] = carry + incl - v;
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

void launch_scan_block_sums(int32_t *block_sums, int num_blocks, int32_t *total_out,
                            hipStream_t stream) {
  scan_block_sums_kernel<<<1, kScanBlock, 0, stream>>>(block_sums, num_blocks, total_out);
}

}  // namespace sg

extern "C" {

int sg_version(void) { return 1; }

const char *sg_last_error(void) { return sg::g_err; }

int sg_device_info(char *name_host, int name_cap, int *num_cu_host, int *clock_khz_host) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    sg::set_error("sg_device_info: no HIP device");
    return SG_ERR_LAUNCH;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return SG_ERR_LAUNCH;
  if (name_host && name_cap > 0) {
    snprintf(name_host, name_cap, "%s (%s)", p.name, p.gcnArchName);
  }
  if (num_cu_host) *num_cu_host = p.multiProcessorCount;
  if (clock_khz_host) *clock_khz_host = p.clockRate;
  return SG_OK;
}

int sg_stream_release(sg_stream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    sg::set_error("sg_stream_release: no HIP device");
    return SG_ERR_LAUNCH;
  }
  sg::conv_release_stream(dev, sg::as_stream(stream));
  sg::unet_release_stream(dev, sg::as_stream(stream));
  sg::scan_release_stream(dev, sg::as_stream(stream));
  sg::bfs_release_stream(dev, sg::as_stream(stream));
  return SG_OK;
}

int sg_stream_create(sg_stream_t *stream_out) {
  SG_REQUIRE(stream_out != nullptr, "sg_stream_create: null argument");
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
    sg::set_error("sg_stream_create: hipStreamCreateWithFlags failed");
    return SG_ERR_LAUNCH;
  }
  *stream_out = reinterpret_cast<sg_stream_t>(s);
  return SG_OK;
}

int sg_stream_create_priority(sg_stream_t *stream_out, int level) {
  SG_REQUIRE(stream_out != nullptr, "sg_stream_create_priority: null argument");
  int least = 0, greatest = 0;      // (numerically: greatest priority = the smaller number)
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
    sg::set_error("sg_stream_create_priority: hipDeviceGetStreamPriorityRange failed");
    return SG_ERR_LAUNCH;
  }
  const int prio = level > 0 ? greatest : level < 0 ? least : (least + greatest) / 2;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio) != hipSuccess) {
    sg::set_error("sg_stream_create_priority: hipStreamCreateWithPriority failed");
    return SG_ERR_LAUNCH;
  }
  *stream_out = reinterpret_cast<sg_stream_t>(s);
  return SG_OK;
}

int sg_stream_destroy(sg_stream_t stream) {
  SG_REQUIRE(stream != nullptr, "sg_stream_destroy: the null stream is not the library's");
  hipStream_t s = sg::as_stream(stream);
  if (hipStreamSynchronize(s) != hipSuccess) {
    sg::set_error("sg_stream_destroy: synchronising the stream failed");
    return SG_ERR_LAUNCH;
  }
  const int rc = sg_stream_release(stream);
  if (rc != SG_OK) return rc;
  if (hipStreamDestroy(s) != hipSuccess) {
    sg::set_error("sg_stream_destroy: hipStreamDestroy failed");
    return SG_ERR_LAUNCH;
  }
  return SG_OK;
}

size_t sg_scan_workspace_bytes(int n) { return sg::scan_workspace_bytes(n); }

// start_len[i,0] = exclusive prefix of start_len[:,1]
int sg_exclusive_scan_startlen(int32_t *start_len, int n, int32_t *meta, void *ws, size_t ws_bytes,
                               sg_stream_t stream) {
  SG_REQUIRE(n >= 0, "sg_exclusive_scan_startlen: n < 0");
  auto in = [start_len] __device__(int64_t i) { return start_len[2 * i + 1]; };
  auto out = [start_len] __device__(int64_t i, int v) { start_len[2 * i] = v; };
  return sg::exclusive_scan(in, out, n, meta, ws, ws_bytes, sg::as_stream(stream));
}

}  // extern "C"
