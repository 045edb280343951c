// Shared host-side plumbing for libxmodal_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/xmodal.h"

namespace xm {

// ---- error reporting (C ABI never throws) -------------------------------------------------
char *err_buf();
int fail(int code, const char *fmt, ...);

#define XM_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess)                                                            \
      return xm::fail(XM_EHIP, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

#define XM_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess)                                                            \
      return xm::fail(XM_EHIP, "kernel launch -> %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

// ---- stream-ordered scratch ----------------------------------------------------------------
// One growable device buffer PER STREAM.  Every API call carves what it needs from offset 0 of its
// stream's buffer; calls are stream-ordered, so reuse across consecutive calls on one stream is
// safe, and calls on different streams (teacher forward overlapping the student step) never share
// scratch.  Growing synchronises the device (only while shapes are first seen).
int ws_get(size_t bytes, void **ptr, hipStream_t stream);
struct WsCarver {
  char *base = nullptr;
  size_t off = 0, cap = 0;
  int init(size_t bytes, hipStream_t stream) {
    void *p = nullptr;
    int rc = ws_get(bytes, &p, stream);
    base = (char *)p;
    cap = bytes;
    off = 0;
    return rc;
  }
  template <class T>
  T *take(size_t count) {
    size_t b = (count * sizeof(T) + 255) & ~size_t(255);
    T *p = (T *)(base + off);
    off += b;
    return p;
  }
  static size_t need(size_t count, size_t elt) { return (count * elt + 255) & ~size_t(255); }
};

// ---- kernel-path selectors ----------------------------------------------------------------------------------------
// Several operators have more than one implementation (an LDS-DMA kernel next to the register-staged one, a fused next
// to a composed path ...).  Every arm is a complete, parity-tested implementation of the same operator -- tests cover the
// fallback arms through these switches, profiles/ uses them for A/B lines -- and none of them changes what is computed,
// only by which kernel.  They are read from the environment ONCE, in one place (context.cpp); nothing else in the
// library looks at the environment except the tuning-table settings (XM_TUNE_*, XM_AUTOTUNE, XM_HALO_MARGIN) and the
// workspace log (XM_WS_VERBOSE), all listed in include/xmodal_prof.h.
enum Path {
  kPathHybrid,          // XM_NO_HYBRID          partly filled last round split along the reduction (conv.hip hybrid_plan)
  kPathHalo,            // XM_NO_HALO            halo-patch kernels for <= 3 x 3 unit-stride gathers
  kPathSkinny,          // XM_NO_SKINNY          fc_skinny_kernel for 1 x 1 layers over few outputs
  kPathSkinny4,         // XM_NO_SKINNY4         its four-rows-per-block variant
  kPathStem,            // XM_NO_STEM            single-channel stem kernels (forward and filter derivative)
  kPathStemWgrad,       // XM_NO_STEM_WGRAD      ... the filter derivative only
  kPathDma,             // XM_NO_DMA             LDS-DMA kernel for 1 x 1 unit-stride layers
  kPathFusedStats,      // XM_NO_FUSED_STATS     batch moments from the convolution epilogue
  kPathDgradMerge,      // XM_DGRAD_MERGE (set = off)  all stride-parity classes of a strided dgrad in one launch
  kPathFastTranspose,   // XM_NO_FAST_TRANSPOSE  FC-shaped dgrad operands as plain LDS-tiled transposes
  kPathPoolLds,         // XM_NO_POOL_LDS        LDS-staged fused bnorm + relu + pool forward
  kPathPoolPatch,       // XM_NO_POOL_PATCH      stride-cell variant of its backward apply
  kPathPoolPooled,      // XM_NO_POOL_POOLED     backward sums from the pooled tensors
  kPathW8,              // XM_NO_W8              128 x 128 tiles by eight waves of 128 VGPRs (conv.hip kCfgs[7]); off = four waves of 222
  kPathWgradPatch,      // XM_NO_WGRAD_PATCH     filter derivative of 3 x 3 / stride 1 layers from an input patch (conv_wgrad_patch_kernel)
  kPathWgradPatchS2,    // XM_NO_WGRAD_PATCH_S2  filter derivative of 5 x 5 / stride 2 layers from an input patch (conv_wgrad_patch_s2_kernel)
  kPathDgradS2,         // XM_NO_DGRAD_S2        dgrad of 5 x 5 / stride 2 layers with both row parities per wave (conv_dgrad_s2_kernel)
  kPathStem3,           // XM_NO_STEM3           three-channel 7 x 7 / stride 2 stem kernel (conv_stem3_kernel: the teachers' conv1)
  kPathCount
};
bool path_on(Path p);
// numeric settings from the environment (tuning-table knobs listed in include/xmodal_prof.h): read through context.cpp
long long env_int(const char *name, long long dflt);
double env_double(const char *name, double dflt);

// comm.cpp: test switch behind xm_debug_set("comm_single", v)
int comm_force_single(int on);

// persistent small device objects keyed by content (tap tables)
const void *cached_device_table(const void *host, size_t bytes);

// batch moments of vl_nnbnorm [mean, sqrt(var + eps)] (C x 2) by a pass over x (norm_pool.hip) -- what the fused
// convolution epilogue of conv.hip replaces when its tile configuration cannot carry the partial sums
int bn_batch_moments(const float *x, int H, int W, int C, int N, float eps, float *moments_out, hipStream_t st);

// sums of the fused bnorm + relu + max-pool backward for a consumer that rebuilds DZDX itself (norm_pool.hip): dg, db
// and the per-channel constants [C][6] = {g/sigma, mu, b, k1 hi, k1 lo, k2}; scratch comes from the caller's carver
size_t bnpool_sums_need(int C, int N);
int bnpool_backward_sums(WsCarver &ws, const float *x, int H, int W, int C, int N, const float *g, const float *b,
                         const float *moments, int train, int ph, int pw, int sy, int sx, int pt, int pb, int pl, int pr,
                         const unsigned char *amax, const float *y_pool, const float *dzdy_pool, float *dg_out,
                         float *db_out, float *rowc_out, hipStream_t st);

static inline int out_size(int in, int pa, int pb, int f, int d, int s) {
  int feff = (f - 1) * d + 1;
  int t = in + pa + pb - feff;
  return t < 0 ? 0 : t / s + 1;
}

static inline bool too_big(long long a, long long b = 1, long long c = 1, long long d = 1) {
  return a * b * c * d >= (1LL << 31);
}

// magic division of a 31-bit numerator by a runtime divisor: q = (n * m) >> s
struct FastDiv {
  uint32_t m;
  uint32_t s;
  uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.s = 31 + l;
  f.m = (uint32_t)(((1ull << f.s) + d - 1) / d);
  return f;
}

}  // namespace xm

#ifdef __HIPCC__
__device__ __forceinline__ uint32_t xm_div(uint32_t n, const xm::FastDiv f) {
  return (uint32_t)(((uint64_t)n * f.m) >> f.s);
}
__device__ __forceinline__ float xm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Reductions that decide a mean (bnorm moments, the two sums of its backward pass, bias derivatives) are
// accumulated in fp64: these kernels are HBM-bound, so the fp64 adds are free, and the derivative sums of
// a train-mode vl_nnbnorm cancel almost completely (sum of dx over a channel is exactly 0 in exact
// arithmetic) -- an fp32 mean that is off by one ulp leaves a bias that the layers below amplify.
__device__ __forceinline__ double xm_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float xm_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif
