// Process-wide context of libxmodal_hip.so: last-error text, stream-ordered scratch buffer and a
// content-addressed cache of small read-only device tables (convolution tap tables).
// Mirrors MatConvNet's persistent per-process context (workspace + handles), SURVEY.md 8b.
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "xm_common.h"

namespace xm {
bool path_on(Path p) {
  static const char *const kEnv[kPathCount] = {"XM_NO_HYBRID",      "XM_NO_HALO",        "XM_NO_SKINNY",     "XM_NO_SKINNY4",
                                               "XM_NO_STEM",        "XM_NO_STEM_WGRAD",  "XM_NO_DMA",        "XM_NO_FUSED_STATS",
                                               "XM_DGRAD_MERGE",    "XM_NO_FAST_TRANSPOSE", "XM_NO_POOL_LDS", "XM_NO_POOL_PATCH",
                                               "XM_NO_POOL_POOLED", "XM_NO_W8", "XM_NO_WGRAD_PATCH", "XM_NO_WGRAD_PATCH_S2", "XM_NO_DGRAD_S2", "XM_NO_STEM3"};
  static bool on[kPathCount];
  static bool read = false;
  if (!read) {
    for (int i = 0; i < kPathCount; ++i) on[i] = getenv(kEnv[i]) == nullptr;
    read = true;
  }
  return on[p];
}

long long env_int(const char *name, long long dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoll(e) : dflt;
}
double env_double(const char *name, double dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atof(e) : dflt;
}

static char g_err[512] = "";
char *err_buf() { return g_err; }

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

struct WsBuf {
  void *ptr = nullptr;
  size_t cap = 0;
};
static std::map<hipStream_t, WsBuf> g_ws;
static unsigned long long g_ws_generation = 0;  // bumped whenever a scratch buffer moves (see xm_workspace_generation)

int ws_get(size_t bytes, void **ptr, hipStream_t stream) {
  WsBuf &w = g_ws[stream];
  if (bytes > w.cap) {
    // grow geometrically; a grow happens only while shapes are first seen (warm-up)
    size_t want = bytes + bytes / 4 + (1u << 20);
    static const bool verbose = getenv("XM_WS_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "[xm ws] stream %p grow %zu -> %zu\n", (void *)stream, w.cap, want);
    ++g_ws_generation;
    if (w.ptr) {
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) return fail(XM_EHIP, "hipDeviceSynchronize -> %s", hipGetErrorString(e));
      (void)hipFree(w.ptr);
      w.ptr = nullptr;
      w.cap = 0;
    }
    hipError_t e = hipMalloc(&w.ptr, want);
    if (e != hipSuccess) {
      w.ptr = nullptr;
      return fail(XM_ENOMEM, "workspace hipMalloc(%zu) -> %s", want, hipGetErrorString(e));
    }
    w.cap = want;
  }
  *ptr = w.ptr;
  return XM_OK;
}

static std::map<std::string, void *> g_tables;

const void *cached_device_table(const void *host, size_t bytes) {
  std::string key((const char *)host, bytes);
  auto it = g_tables.find(key);
  if (it != g_tables.end()) return it->second;
  void *d = nullptr;
  static const bool verbose = getenv("XM_WS_VERBOSE") != nullptr;
  if (verbose) fprintf(stderr, "[xm ws] new device table %zu bytes (#%zu)\n", bytes, g_tables.size());
  if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
  if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    return nullptr;
  }
  g_tables.emplace(std::move(key), d);
  return d;
}

}  // namespace xm

extern "C" {

int xm_version(void) { return 106; }   // 101: xm_nnbnorm_relu_pool_backward takes y_pool (round 2 signature)
const char *xm_last_error(void) { return xm::err_buf(); }

int xm_workspace_reserve(size_t bytes) {
  void *p;
  return xm::ws_get(bytes, &p, nullptr);
}
int xm_workspace_reserve_stream(size_t bytes, void *stream) {
  void *p;
  return xm::ws_get(bytes, &p, (hipStream_t)stream);
}
unsigned long long xm_workspace_generation(void) { return xm::g_ws_generation; }

int xm_device_alloc(void **ptr, size_t bytes) {
  if (!ptr) return xm::fail(XM_EINVAL, "device_alloc: NULL output");
  *ptr = nullptr;
  if (bytes == 0) return XM_OK;
  hipError_t e = hipMalloc(ptr, bytes);
  if (e != hipSuccess) return xm::fail(XM_ENOMEM, "hipMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
  return XM_OK;
}
int xm_device_free(void *ptr) {
  if (!ptr) return XM_OK;
  hipError_t e = hipFree(ptr);
  if (e != hipSuccess) return xm::fail(XM_EHIP, "hipFree -> %s", hipGetErrorString(e));
  return XM_OK;
}
int xm_device_upload(void *dst_device, const void *src_host, size_t bytes) {
  if (bytes == 0) return XM_OK;
  if (!dst_device || !src_host) return xm::fail(XM_EINVAL, "device_upload: NULL pointer");
  hipError_t e = hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) return xm::fail(XM_EHIP, "hipMemcpy(H2D, %zu) -> %s", bytes, hipGetErrorString(e));
  return XM_OK;
}
int xm_device_download(void *dst_host, const void *src_device, size_t bytes) {
  if (bytes == 0) return XM_OK;
  if (!dst_host || !src_device) return xm::fail(XM_EINVAL, "device_download: NULL pointer");
  hipError_t e = hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return xm::fail(XM_EHIP, "hipMemcpy(D2H, %zu) -> %s", bytes, hipGetErrorString(e));
  return XM_OK;
}
int xm_device_synchronize(void) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return xm::fail(XM_EHIP, "hipDeviceSynchronize -> %s", hipGetErrorString(e));
  return XM_OK;
}
size_t xm_workspace_bytes(void) {
  size_t t = 0;
  for (auto &kv : xm::g_ws) t += kv.second.cap;
  return t;
}

int xm_out_size(int in, int pad_a, int pad_b, int f, int dilate, int stride) {
  return xm::out_size(in, pad_a, pad_b, f, dilate, stride);
}
}
