// vl_nnbnorm and vl_nnpool on gfx950 -- HBM-bound kernels: coalesced runs along H (the MATLAB
// fastest axis), float4 where the plane size allows, wave-shuffle + LDS tree reductions, fixed
// reduction order (no atomics) so results are run-to-run identical.
// Replaces MatConvNet's vl_nnbnorm / vl_nnpool MEX (bits/nnbnorm.cu, bits/nnpooling.cu).
#include <algorithm>

#include "xm_common.h"

namespace xm {

// ============================== batch normalisation =========================================
// Per-channel moments over H*W*N.  One pass, shifted sums (shift = first element of the channel), fp64
// accumulators: the student's first layer reduces 1.2 M values per channel at 32 samples.
// grid = (C, S): block (c, s) reduces samples s, s+S, ... of channel c.
__device__ __forceinline__ void block_reduce2(double &a, double &b, double *red /*8 doubles*/) {
  a = xm_wave_sum_d(a);
  b = xm_wave_sum_d(b);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[w] = a;
    red[4 + w] = b;
  }
  __syncthreads();
  a = red[0] + red[1] + red[2] + red[3];
  b = red[4] + red[5] + red[6] + red[7];
  __syncthreads();
}

// The (sample, position) pairs a block owns are walked as ONE flat index (magic division by the run length):
// the FC-shaped layers have H*W = 8, where a loop over positions inside a loop over samples kept 2 of 256
// threads busy and cost 30 us per call.
__global__ void __launch_bounds__(256)
bn_stats_partial_kernel(const float *__restrict__ x, double *__restrict__ part, int HW, int C, int N,
                        int S, FastDiv divRun) {
  const int c = blockIdx.x, s = blockIdx.y;
  const float shift = x[(size_t)HW * c];
  double a = 0.0, b = 0.0;
  const bool vec = (HW & 3) == 0;
  const int nper = (N - s + S - 1) / S;  // samples s, s + S, ...
  if (vec) {
    const int run = HW >> 2;
    for (int j = threadIdx.x; j < nper * run; j += 256) {
      const int nn = (int)xm_div((uint32_t)j, divRun), i = j - nn * run;
      const float4 v = reinterpret_cast<const float4 *>(x + (size_t)HW * (c + (size_t)C * (s + nn * S)))[i];
      double d0 = (double)v.x - shift, d1 = (double)v.y - shift, d2 = (double)v.z - shift, d3 = (double)v.w - shift;
      a += (d0 + d1) + (d2 + d3);
      b += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  } else {
    for (int j = threadIdx.x; j < nper * HW; j += 256) {
      const int nn = (int)xm_div((uint32_t)j, divRun), i = j - nn * HW;
      double d = (double)x[(size_t)HW * (c + (size_t)C * (s + nn * S)) + i] - shift;
      a += d;
      b += d * d;
    }
  }
  __shared__ double red[8];
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) {
    part[2 * ((size_t)c * S + s)] = a;
    part[2 * ((size_t)c * S + s) + 1] = b;
  }
}

// moments(c) = [mean, sqrt(var + eps)]
// (finalize kernels: one WAVE per channel -- lane l adds partials l, l + 64, ..., then a shuffle reduction; a
// thread per channel walking its S partials one dependent load at a time cost 5-10 us per call)
__global__ void __launch_bounds__(256)
bn_finalize_kernel(const float *__restrict__ x, const double *__restrict__ part, float *__restrict__ mom, int HW,
                   int C, int S, double m, float eps) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = lane; s < S; s += 64) {
    a += part[2 * ((size_t)c * S + s)];
    b += part[2 * ((size_t)c * S + s) + 1];
  }
  a = xm_wave_sum_d(a);
  b = xm_wave_sum_d(b);
  if (lane) return;
  double shift = x[(size_t)HW * c];
  double d = a / m;
  double var = b / m - d * d;
  var = var < 0.0 ? 0.0 : var;
  mom[c] = (float)(shift + d);
  mom[C + c] = (float)sqrt(var + (double)eps);
}

// y = g/sigma * (x - mu) + b   [relu]
template <bool VEC>
__global__ void __launch_bounds__(256)
bn_apply_kernel(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ g,
                const float *__restrict__ b, const float *__restrict__ mom, FastDiv divHW, int C,
                size_t total, int relu) {
  size_t stride = (size_t)gridDim.x * 256;
  if (VEC) {
    size_t n4 = total >> 2;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += stride) {
      uint32_t plane = xm_div((uint32_t)(i << 2), divHW);
      int c = plane % C;
      float sc = g[c] / mom[C + c], mu = mom[c], bb = b[c];
      float4 v = reinterpret_cast<const float4 *>(x)[i];
      v.x = sc * (v.x - mu) + bb;
      v.y = sc * (v.y - mu) + bb;
      v.z = sc * (v.z - mu) + bb;
      v.w = sc * (v.w - mu) + bb;
      if (relu) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
      }
      reinterpret_cast<float4 *>(y)[i] = v;
    }
  } else {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += stride) {
      uint32_t plane = xm_div((uint32_t)i, divHW);
      int c = plane % C;
      float v = g[c] / mom[C + c] * (x[i] - mom[c]) + b[c];
      if (relu) v = fmaxf(v, 0.f);
      y[i] = v;
    }
  }
}

// backward sums: part[c][s] = (sum dy, sum dy*(x-mu)); dy masked by (yfwd > 0) when yfwd != NULL
__global__ void __launch_bounds__(256)
bn_bwd_partial_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                      const float *__restrict__ yfwd, const float *__restrict__ mom,
                      double *__restrict__ part, int HW, int C, int N, int S, FastDiv divRun) {
  const int c = blockIdx.x, s = blockIdx.y;
  const double mu = mom[c];
  double a = 0.0, b = 0.0;
  const bool vec = (HW & 3) == 0;
  const int nper = (N - s + S - 1) / S;
  if (vec) {
    const int run = HW >> 2;
    for (int j = threadIdx.x; j < nper * run; j += 256) {
      const int nn = (int)xm_div((uint32_t)j, divRun), i = j - nn * run;
      const size_t off = (size_t)HW * (c + (size_t)C * (s + nn * S));
      float4 xv = reinterpret_cast<const float4 *>(x + off)[i], dv = reinterpret_cast<const float4 *>(dy + off)[i];
      if (yfwd) {
        float4 yv = reinterpret_cast<const float4 *>(yfwd + off)[i];
        dv.x = yv.x > 0.f ? dv.x : 0.f;
        dv.y = yv.y > 0.f ? dv.y : 0.f;
        dv.z = yv.z > 0.f ? dv.z : 0.f;
        dv.w = yv.w > 0.f ? dv.w : 0.f;
      }
      a += ((double)dv.x + (double)dv.y) + ((double)dv.z + (double)dv.w);
      b += ((double)dv.x * ((double)xv.x - mu) + (double)dv.y * ((double)xv.y - mu)) +
           ((double)dv.z * ((double)xv.z - mu) + (double)dv.w * ((double)xv.w - mu));
    }
  } else {
    for (int j = threadIdx.x; j < nper * HW; j += 256) {
      const int nn = (int)xm_div((uint32_t)j, divRun), i = j - nn * HW;
      const size_t off = (size_t)HW * (c + (size_t)C * (s + nn * S)) + i;
      float d = dy[off];
      if (yfwd && !(yfwd[off] > 0.f)) d = 0.f;
      a += (double)d;
      b += (double)d * ((double)x[off] - mu);
    }
  }
  __shared__ double red[8];
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) {
    part[2 * ((size_t)c * S + s)] = a;
    part[2 * ((size_t)c * S + s) + 1] = b;
  }
}

// sums(c) = [sum dy, sum dy*(x-mu)];  dg = sums1 / sigma, db = sums0
__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(const double *__restrict__ part, const float *__restrict__ mom, double *__restrict__ sums,
                       float *__restrict__ dg, float *__restrict__ db, int C, int S) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = lane; s < S; s += 64) {
    a += part[2 * ((size_t)c * S + s)];
    b += part[2 * ((size_t)c * S + s) + 1];
  }
  a = xm_wave_sum_d(a);
  b = xm_wave_sum_d(b);
  if (lane) return;
  sums[c] = a;
  sums[C + c] = b;
  if (dg) dg[c] = (float)(b / (double)mom[C + c]);
  if (db) db[c] = (float)a;
}

// train: dx = g/sigma * (dy - sum_dy/m - (x-mu) * sum_dyx / (m sigma^2));  test: dx = g/sigma * dy
template <bool VEC>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                    const float *__restrict__ yfwd, float *__restrict__ dx,
                    const float *__restrict__ g, const float *__restrict__ mom,
                    const double *__restrict__ sums, FastDiv divHW, int C, size_t total, double m,
                    int train) {
  size_t stride = (size_t)gridDim.x * 256;
  const size_t cnt = VEC ? (total >> 2) : total;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < cnt; i += stride) {
    uint32_t plane = xm_div((uint32_t)(VEC ? (i << 2) : i), divHW);
    int c = plane % C;
    // coefficients and the per-element expression in fp64, ONE rounding per output: the means c1 / c2 are
    // then exact to ~1e-16 and sum(dx) over a channel stays at the sqrt(m) * ulp(dx) level
    const double sg = mom[C + c], mu = mom[c];
    const double gs = (double)g[c] / sg;
    const double c1 = train ? sums[c] / m : 0.0;
    const double c2 = train ? sums[C + c] / (m * sg * sg) : 0.0;
    if (VEC) {
      float4 xv = reinterpret_cast<const float4 *>(x)[i];
      float4 dv = reinterpret_cast<const float4 *>(dy)[i];
      if (yfwd) {
        float4 yv = reinterpret_cast<const float4 *>(yfwd)[i];
        dv.x = yv.x > 0.f ? dv.x : 0.f;
        dv.y = yv.y > 0.f ? dv.y : 0.f;
        dv.z = yv.z > 0.f ? dv.z : 0.f;
        dv.w = yv.w > 0.f ? dv.w : 0.f;
      }
      float4 o;
      o.x = (float)(gs * ((double)dv.x - c1 - ((double)xv.x - mu) * c2));
      o.y = (float)(gs * ((double)dv.y - c1 - ((double)xv.y - mu) * c2));
      o.z = (float)(gs * ((double)dv.z - c1 - ((double)xv.z - mu) * c2));
      o.w = (float)(gs * ((double)dv.w - c1 - ((double)xv.w - mu) * c2));
      reinterpret_cast<float4 *>(dx)[i] = o;
    } else {
      float d = dy[i];
      if (yfwd && !(yfwd[i] > 0.f)) d = 0.f;
      dx[i] = (float)(gs * ((double)d - c1 - ((double)x[i] - mu) * c2));
    }
  }
}

// Per-channel variant of bn_bwd_apply_kernel (grid (C, S2), the layout of the reduction kernels): the block first adds
// the Sp partial sums of ITS channel (bn_bwd_partial_kernel's output -- no separate finalize launch; block s == 0
// also writes dg / db), applies the derivative to its share of the channel, and leaves sum(dx) of that share in
// part2[c][s]: summed over s this is dzdb of the convolution that produced x (vl_nnconv: dzdb = sum of dzdy), so the
// bias derivative costs no pass of its own over dx.
template <bool VEC>
__global__ void __launch_bounds__(256)
bn_bwd_apply_ch_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ yfwd,
                       float *__restrict__ dx, const float *__restrict__ g, const float *__restrict__ mom,
                       const double *__restrict__ part, int Sp, float *__restrict__ dg, float *__restrict__ db,
                       double *__restrict__ part2, int HW, int C, int N, int S2, FastDiv divRun, double m, int train) {
  const int c = blockIdx.x, s = blockIdx.y;
  double sa = 0.0, sb = 0.0;
  for (int i = 0; i < Sp; ++i) {      // fixed order, the same in every block of the channel
    sa += part[2 * ((size_t)c * Sp + i)];
    sb += part[2 * ((size_t)c * Sp + i) + 1];
  }
  const double sg = mom[C + c], mu = mom[c];
  if (s == 0 && threadIdx.x == 0) {
    if (dg) dg[c] = (float)(sb / sg);
    if (db) db[c] = (float)sa;
  }
  const double gs = (double)g[c] / sg;
  const double c1 = train ? sa / m : 0.0;
  const double c2 = train ? sb / (m * sg * sg) : 0.0;
  double acc = 0.0;
  const int nper = (N - s + S2 - 1) / S2;
  const int run = VEC ? HW >> 2 : HW;
  for (int j = threadIdx.x; j < nper * run; j += 256) {
    const int nn = (int)xm_div((uint32_t)j, divRun), i = j - nn * run;
    const size_t off = (size_t)HW * (c + (size_t)C * (s + nn * S2));
    if (VEC) {
      float4 xv = reinterpret_cast<const float4 *>(x + off)[i], dv = reinterpret_cast<const float4 *>(dy + off)[i];
      if (yfwd) {
        float4 yv = reinterpret_cast<const float4 *>(yfwd + off)[i];
        dv.x = yv.x > 0.f ? dv.x : 0.f;
        dv.y = yv.y > 0.f ? dv.y : 0.f;
        dv.z = yv.z > 0.f ? dv.z : 0.f;
        dv.w = yv.w > 0.f ? dv.w : 0.f;
      }
      float4 o;
      o.x = (float)(gs * ((double)dv.x - c1 - ((double)xv.x - mu) * c2));
      o.y = (float)(gs * ((double)dv.y - c1 - ((double)xv.y - mu) * c2));
      o.z = (float)(gs * ((double)dv.z - c1 - ((double)xv.z - mu) * c2));
      o.w = (float)(gs * ((double)dv.w - c1 - ((double)xv.w - mu) * c2));
      reinterpret_cast<float4 *>(dx + off)[i] = o;
      acc += ((double)o.x + (double)o.y) + ((double)o.z + (double)o.w);
    } else {
      float d = dy[off + i];
      if (yfwd && !(yfwd[off + i] > 0.f)) d = 0.f;
      const float o = (float)(gs * ((double)d - c1 - ((double)x[off + i] - mu) * c2));
      dx[off + i] = o;
      acc += (double)o;
    }
  }
  acc = xm_wave_sum_d(acc);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part2[(size_t)c * S2 + s] = (red[0] + red[1]) + (red[2] + red[3]);
}

// run length of one sample inside the flat (sample, position) index of the reduction kernels
static FastDiv bn_run_div(int HW) { return make_fastdiv((uint32_t)((HW & 3) == 0 ? HW >> 2 : HW)); }
static int bn_splits(int C, int N) {
  int s = 2048 / (C > 0 ? C : 1);
  if (s < 1) s = 1;
  if (s > N) s = N;
  return s;
}

static unsigned ew_grid(size_t work_items) {
  size_t b = (work_items + 255) / 256;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (unsigned)b;
}

static int bn_check(int H, int W, int C, int N) {
  if (H <= 0 || W <= 0 || C <= 0 || N <= 0) return fail(XM_EINVAL, "vl_nnbnorm: empty tensor");
  if (too_big(H, W, C, N)) return fail(XM_ETOOBIG, "vl_nnbnorm: tensor with >= 2^31 elements");
  return XM_OK;
}

int bn_batch_moments(const float *x, int H, int W, int C, int N, float eps, float *moments_out, hipStream_t st) {
  int rc = bn_check(H, W, C, N);
  if (rc) return rc;
  const int HW = H * W, S = bn_splits(C, N);
  WsCarver ws;
  rc = ws.init(WsCarver::need((size_t)2 * C * S, 8), st);
  if (rc) return rc;
  double *part = ws.take<double>((size_t)2 * C * S);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, S), dim3(256), 0, st, x, part, HW, C, N, S, bn_run_div(HW));
  XM_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, x, part, moments_out, HW, C, S,
                     (double)HW * N, eps);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

static int bnorm_forward(const float *x, int H, int W, int C, int N, const float *g, const float *b,
                         float eps, const float *moments_in, float *y, float *moments_out, int relu,
                         hipStream_t st) {
  int rc = bn_check(H, W, C, N);
  if (rc) return rc;
  if (!x || !g || !b || !y) return fail(XM_EINVAL, "vl_nnbnorm: NULL tensor");
  const int HW = H * W;
  const int S = bn_splits(C, N);
  const float *mom = moments_in;
  WsCarver ws;
  if (!moments_in) {
    rc = ws.init(WsCarver::need((size_t)2 * C * S, 8) + WsCarver::need((size_t)2 * C, 4), st);
    if (rc) return rc;
    double *part = ws.take<double>((size_t)2 * C * S);
    float *momw = moments_out ? moments_out : ws.take<float>((size_t)2 * C);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, S), dim3(256), 0, st, x, part, HW, C, N, S, bn_run_div(HW));
    XM_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, x, part, momw, HW,
                       C, S, (double)HW * N, eps);
    XM_LAUNCH_CHECK();
    mom = momw;
  } else if (moments_out && moments_out != moments_in) {
    XM_HIP(hipMemcpyAsync(moments_out, moments_in, sizeof(float) * 2 * C, hipMemcpyDeviceToDevice, st));
  }
  size_t total = (size_t)HW * C * N;
  FastDiv d = make_fastdiv((uint32_t)HW);
  if ((HW & 3) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(ew_grid(total / 4)), dim3(256), 0, st, x, y, g, b,
                       mom, d, C, total, relu);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(ew_grid(total)), dim3(256), 0, st, x, y, g, b,
                       mom, d, C, total, relu);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

__global__ void sum_partials_kernel(const double *__restrict__ part, float *__restrict__ out, int C, int S);

// dxsum_out != NULL (needs dx_out): also sum(dx) per channel = the bias derivative of the producing convolution
static int bnorm_backward(const float *x, const float *yfwd, int H, int W, int C, int N,
                          const float *g, const float *dzdy, float eps, const float *moments_in,
                          float *dx_out, float *dg_out, float *db_out, float *moments_out,
                          hipStream_t st, bool batch_moments = false, float *dxsum_out = nullptr) {
  int rc = bn_check(H, W, C, N);
  if (rc) return rc;
  if (!x || !g || !dzdy) return fail(XM_EINVAL, "vl_nnbnorm: NULL tensor");
  if (batch_moments && !moments_in) return fail(XM_EINVAL, "vl_nnbnorm: XM_BN_BATCH_MOMENTS needs moments");
  if (dxsum_out && !dx_out) return fail(XM_EINVAL, "vl_nnbnorm: dxsum needs dx");
  const int HW = H * W;
  const int S = bn_splits(C, N);
  // apply blocks per channel of the dxsum path: enough blocks to fill the chip, at most one per sample
  const int S2 = std::max(1, std::min(N, 4096 / std::max(1, C)));
  WsCarver ws;
  rc = ws.init(WsCarver::need((size_t)2 * C * S, 8) + WsCarver::need((size_t)2 * C, 8) +
                   WsCarver::need((size_t)2 * C, 4) + WsCarver::need(dxsum_out ? (size_t)C * S2 : 0, 8), st);
  if (rc) return rc;
  double *part = ws.take<double>((size_t)2 * C * S);
  double *sums = ws.take<double>((size_t)2 * C);
  double *part2 = dxsum_out ? ws.take<double>((size_t)C * S2) : nullptr;
  const float *mom = moments_in;
  if (!moments_in) {
    float *momw = moments_out ? moments_out : ws.take<float>((size_t)2 * C);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, S), dim3(256), 0, st, x, part, HW, C, N, S, bn_run_div(HW));
    XM_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, x, part, momw, HW,
                       C, S, (double)HW * N, eps);
    XM_LAUNCH_CHECK();
    mom = momw;
  } else if (moments_out && moments_out != moments_in) {
    XM_HIP(hipMemcpyAsync(moments_out, moments_in, sizeof(float) * 2 * C, hipMemcpyDeviceToDevice, st));
  }
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(C, S), dim3(256), 0, st, x, dzdy, yfwd, mom, part, HW,
                     C, N, S, bn_run_div(HW));
  XM_LAUNCH_CHECK();
  if (dxsum_out) {
    // partial sums -> [finalize + apply + sum(dx) partials] -> dzdb of the producing convolution: three launches
    const double m = (double)HW * N;
    const int train = (moments_in && !batch_moments) ? 0 : 1;
    const bool al = ((((uintptr_t)x | (uintptr_t)dzdy | (uintptr_t)dx_out | (uintptr_t)yfwd) & 15) == 0);
    if ((HW & 3) == 0 && al)
      hipLaunchKernelGGL(bn_bwd_apply_ch_kernel<true>, dim3(C, S2), dim3(256), 0, st, x, dzdy, yfwd, dx_out, g, mom,
                         part, S, dg_out, db_out, part2, HW, C, N, S2, bn_run_div(HW), m, train);
    else
      hipLaunchKernelGGL(bn_bwd_apply_ch_kernel<false>, dim3(C, S2), dim3(256), 0, st, x, dzdy, yfwd, dx_out, g, mom,
                         part, S, dg_out, db_out, part2, HW, C, N, S2, make_fastdiv((uint32_t)HW), m, train);
    XM_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part2, dxsum_out, C, S2);
    XM_LAUNCH_CHECK();
    return XM_OK;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, mom, sums,
                     dg_out, db_out, C, S);
  XM_LAUNCH_CHECK();
  if (dx_out) {
    size_t total = (size_t)HW * C * N;
    FastDiv d = make_fastdiv((uint32_t)HW);
    double m = (double)HW * N;
    int train = (moments_in && !batch_moments) ? 0 : 1;
    bool al = ((((uintptr_t)x | (uintptr_t)dzdy | (uintptr_t)dx_out | (uintptr_t)yfwd) & 15) == 0);
    if ((HW & 3) == 0 && al)
      hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(ew_grid(total / 4)), dim3(256), 0, st, x,
                         dzdy, yfwd, dx_out, g, mom, sums, d, C, total, m, train);
    else
      hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(ew_grid(total)), dim3(256), 0, st, x, dzdy,
                         yfwd, dx_out, g, mom, sums, d, C, total, m, train);
    XM_LAUNCH_CHECK();
  }
  return XM_OK;
}

// ===================================== pooling ===============================================
struct PoolGeo {
  int H, W, Ho, Wo, ph, pw, sy, sx, pt, pl;
};

// 2-D indexing shared by the pooling kernels: threadIdx.x runs along the contiguous H axis,
// (blockIdx.x * blockDim.y + threadIdx.y) along W, blockIdx.y (+ grid-stride) over C*N planes --
// no per-element integer division.
struct PoolLaunch {
  dim3 grid, block;
};
static int pow2_ge(int v, int cap) {
  int p = 1;
  while (p < v && p < cap) p <<= 1;
  return p;
}
static PoolLaunch pool_launch(int rows, int cols, long long planes) {
  // 256 threads = bx (rows, contiguous) x by (cols) x bz (planes): small planes pack several
  // planes into one block instead of idling lanes
  int bx = pow2_ge(rows, 256);
  int by = pow2_ge(cols, 256 / bx);
  int bz = 256 / (bx * by);
  PoolLaunch l;
  l.block = dim3(bx, by, bz);
  // ~8k blocks in total: each block then strides over several planes (amortises block start-up)
  int gx = (cols + by - 1) / by, gz = (rows + bx - 1) / bx;
  long long pg = (planes + bz - 1) / bz;
  long long gy = std::max<long long>(1, 8192 / ((long long)gx * gz));
  gy = std::min<long long>(std::min<long long>(gy, pg), 65535);
  l.grid = dim3(gx, (unsigned)gy, gz);
  return l;
}

// one thread per output element; window scan is column-major.
// `amax` (optional, max pooling): position of the FIRST maximum inside the un-clipped window,
// code = dh + ph * dw -- the routing table of the backward pass.
// PH, PW > 0: compile-time window -> all PH*PW loads are independent (clamped address + validity
// select) and in flight together; PH == 0: generic runtime window.
// Optional fused producer (bn_g != NULL): every loaded value first goes through
// relu(g/sigma * (v - mu) + b) of its channel -- vl_nnbnorm + vl_nnrelu without materialising them.
template <int PH, int PW>
__global__ void __launch_bounds__(256)
pool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned char *__restrict__ amax,
                PoolGeo g, int planes, int method, const float *__restrict__ bn_g,
                const float *__restrict__ bn_b, const float *__restrict__ bn_mom, int C) {
  const int ho = blockIdx.z * blockDim.x + threadIdx.x;
  const int wo = blockIdx.x * blockDim.y + threadIdx.y;
  if (ho >= g.Ho || wo >= g.Wo) return;
  const int w0 = wo * g.sx - g.pl, h0 = ho * g.sy - g.pt;
  const int w2 = min(w0 + g.pw, g.W), h2 = min(h0 + g.ph, g.H);
  const int w1 = max(w0, 0), h1 = max(h0, 0);
#pragma unroll 2
  for (int plane = blockIdx.y * blockDim.z + threadIdx.z; plane < planes; plane += gridDim.y * blockDim.z) {
    const float *p = x + (size_t)plane * g.H * g.W;
    const size_t o = (size_t)plane * g.Ho * g.Wo + ho + (size_t)g.Ho * wo;
    float bsc = 1.f, bmu = 0.f, bbb = 0.f;
    if (bn_g) {
      int c = plane % C;
      bsc = bn_g[c] / bn_mom[C + c];
      bmu = bn_mom[c];
      bbb = bn_b[c];
    }
    float r;
    if (PH > 0) {
      float v[PH * PW > 0 ? PH * PW : 1];
#pragma unroll
      for (int dw = 0; dw < PW; ++dw)
#pragma unroll
        for (int dh = 0; dh < PH; ++dh) {
          int h = h0 + dh, w = w0 + dw;
          bool ok = ((unsigned)h < (unsigned)g.H) & ((unsigned)w < (unsigned)g.W);
          float ld = p[ok ? h + g.H * w : 0];
          if (bn_g) ld = fmaxf(bsc * (ld - bmu) + bbb, 0.f);
          v[dh + PH * dw] = ok ? ld : (method == XM_POOL_MAX ? -INFINITY : 0.f);
        }
      if (method == XM_POOL_MAX) {
        r = -INFINITY;
        int code = 0;
#pragma unroll
        for (int i = 0; i < PH * PW; ++i)
          if (v[i] > r) {
            r = v[i];
            code = i;
          }
        if (amax) amax[o] = (unsigned char)code;
      } else {
        r = 0.f;
#pragma unroll
        for (int i = 0; i < PH * PW; ++i) r += v[i];
        r *= 1.0f / (float)((h2 - h1) * (w2 - w1));
      }
    } else if (method == XM_POOL_MAX) {
      r = -INFINITY;
      int code = 0;
      for (int w = w1; w < w2; ++w)
        for (int h = h1; h < h2; ++h) {
          float v = p[h + g.H * w];
          if (bn_g) v = fmaxf(bsc * (v - bmu) + bbb, 0.f);
          if (v > r) {
            r = v;
            code = (h - h0) + g.ph * (w - w0);
          }
        }
      if (amax) amax[o] = (unsigned char)code;
    } else {
      r = 0.f;
      for (int w = w1; w < w2; ++w)
        for (int h = h1; h < h2; ++h) {
          float v = p[h + g.H * w];
          if (bn_g) v = fmaxf(bsc * (v - bmu) + bbb, 0.f);
          r += v;
        }
      r *= 1.0f / (float)((h2 - h1) * (w2 - w1));
    }
    if (y) y[o] = r;
  }
}

// LDS-staged variant for max pooling with a compile-time window: one block owns `wob` output columns of one
// plane.  The input columns under them are ONE contiguous run of the plane (H is the fastest axis): it is
// copied into LDS with 16-byte loads (each element fetched once, the fused bnorm+relu applied once per element
// instead of once per window tap), then every output reads its PH*PW taps from LDS.  The one-thread-per-output
// kernel above issues PH*PW strided 4-byte loads per output and runs at ~2.7 TB/s on the student's first
// pooling layer; this one is bound by the HBM stream.
template <int PH, int PW>
__global__ void __launch_bounds__(256)
pool_fwd_lds_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned char *__restrict__ amax,
                    PoolGeo g, FastDiv divHo, int wob, size_t total, const float *__restrict__ bn_g,
                    const float *__restrict__ bn_b, const float *__restrict__ bn_mom, int C) {
  extern __shared__ float tile[];
  // (planes on grid.x: grid.y stops at 65535, and 256 filters x 256 spectrograms -- the student's pool2 at the north_star
  // batch -- are 65536 planes: that one launch fell back to the one-thread-per-output kernel, 290 us instead of ~150)
  const int plane = blockIdx.x;
  const int wo0 = blockIdx.y * wob;
  const int nwo = min(wob, g.Wo - wo0);
  const int wlo = max(wo0 * g.sx - g.pl, 0);
  const int whi = min((wo0 + nwo - 1) * g.sx - g.pl + PW, g.W);
  const size_t base = (size_t)plane * g.H * g.W + (size_t)wlo * g.H;
  const int lead = (int)(base & 3);
  const size_t start = base - lead;
  const int cnt = (whi - wlo) * g.H + lead;
  float bsc = 1.f, bmu = 0.f, bbb = 0.f;
  if (bn_g) {
    int c = plane % C;
    bsc = bn_g[c] / bn_mom[C + c];
    bmu = bn_mom[c];
    bbb = bn_b[c];
  }
  for (int i = threadIdx.x * 4; i < cnt; i += 1024) {
    float4 v;
    if (start + i + 3 < total) {
      v = *reinterpret_cast<const float4 *>(x + start + i);
    } else {
      v.x = x[start + i];
      v.y = start + i + 1 < total ? x[start + i + 1] : 0.f;
      v.z = start + i + 2 < total ? x[start + i + 2] : 0.f;
      v.w = 0.f;
    }
    if (bn_g) {
      v.x = fmaxf(bsc * (v.x - bmu) + bbb, 0.f);
      v.y = fmaxf(bsc * (v.y - bmu) + bbb, 0.f);
      v.z = fmaxf(bsc * (v.z - bmu) + bbb, 0.f);
      v.w = fmaxf(bsc * (v.w - bmu) + bbb, 0.f);
    }
    *reinterpret_cast<float4 *>(tile + i) = v;
  }
  __syncthreads();
  const float *t = tile + lead - wlo * g.H;  // t[h + H * w] for w in [wlo, whi)
  for (int o = threadIdx.x; o < nwo * g.Ho; o += 256) {
    const int wl = (int)xm_div((uint32_t)o, divHo);
    const int ho = o - wl * g.Ho, wo = wo0 + wl;
    const int w0 = wo * g.sx - g.pl, h0 = ho * g.sy - g.pt;
    float v[PH * PW];
#pragma unroll
    for (int dw = 0; dw < PW; ++dw)
#pragma unroll
      for (int dh = 0; dh < PH; ++dh) {
        int h = h0 + dh, w = w0 + dw;
        bool ok = ((unsigned)h < (unsigned)g.H) & ((unsigned)w < (unsigned)g.W);
        float ld = t[ok ? h + g.H * w : wlo * g.H];
        v[dh + PH * dw] = ok ? ld : -INFINITY;
      }
    float r = -INFINITY;
    int code = 0;
#pragma unroll
    for (int i = 0; i < PH * PW; ++i)
      if (v[i] > r) {
        r = v[i];
        code = i;
      }
    const size_t oo = (size_t)plane * g.Ho * g.Wo + ho + (size_t)g.Ho * wo;
    if (amax) amax[oo] = (unsigned char)code;
    if (y) y[oo] = r;
  }
}

// global pooling (window = whole plane, no padding): one wave per plane, shuffle reduction
__global__ void __launch_bounds__(256)
pool_global_kernel(const float *__restrict__ x, float *__restrict__ y, int HW, int planes,
                   int method) {
  int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (plane >= planes) return;
  const float *p = x + (size_t)plane * HW;
  int lane = threadIdx.x & 63;
  // 16-byte loads when the plane is a whole number of aligned quads (the SE squeeze: 56^2 ... 14^2 maps)
  const bool vec = (HW & 3) == 0 && (((uintptr_t)p) & 15) == 0;
  if (method == XM_POOL_MAX) {
    float r = -INFINITY;
    if (vec) {
      const float4 *p4 = reinterpret_cast<const float4 *>(p);
#pragma unroll 4
      for (int i = lane; i < (HW >> 2); i += 64) {
        const float4 v = p4[i];
        r = fmaxf(fmaxf(r, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
      }
    } else {
      for (int i = lane; i < HW; i += 64) r = fmaxf(r, p[i]);
    }
    r = xm_wave_max(r);
    if (lane == 0) y[plane] = r;
  } else {
    float r = 0.f;
    if (vec) {
      const float4 *p4 = reinterpret_cast<const float4 *>(p);
#pragma unroll 4
      for (int i = lane; i < (HW >> 2); i += 64) {
        const float4 v = p4[i];
        r += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int i = lane; i < HW; i += 64) r += p[i];
    }
    r = xm_wave_sum(r);
    if (lane == 0) y[plane] = r * (1.0f / (float)HW);
  }
}

// backward as a gather: one thread per INPUT element visits the <= ceil(ph/sy)*ceil(pw/sx)
// windows that contain it and takes dzdy(window) iff the window's recorded first-maximum position
// (amax, written by the forward kernel; MatConvNet's CPU tie rule) is this element.  avg: every
// covering window contributes dzdy / clipped area.  No atomics, no dependent load chains.
// FAST: at most 2 x 2 windows cover an input element (ceil(ph/sy) <= 2, ceil(pw/sx) <= 2): the 4
// candidate table bytes and dzdy values are loaded unconditionally (clamped) and selected -- all
// loads independent; otherwise a runtime loop.
template <bool FAST>
__global__ void __launch_bounds__(256)
pool_bwd_kernel(const unsigned char *__restrict__ amax, const float *__restrict__ dy,
                float *__restrict__ dx, PoolGeo g, FastDiv divSy, FastDiv divSx, int planes,
                int method) {
  const int h = blockIdx.z * blockDim.x + threadIdx.x;
  const int w = blockIdx.x * blockDim.y + threadIdx.y;
  if (h >= g.H || w >= g.W) return;
  // windows ho with ho*sy - pt <= h < ho*sy - pt + ph   (all dividends are >= 0)
  int ho_lo = h + g.pt - g.ph + 1;
  ho_lo = ho_lo <= 0 ? 0 : (int)xm_div((uint32_t)(ho_lo + g.sy - 1), divSy);
  const int ho_hi = min((int)xm_div((uint32_t)(h + g.pt), divSy), g.Ho - 1);
  int wo_lo = w + g.pl - g.pw + 1;
  wo_lo = wo_lo <= 0 ? 0 : (int)xm_div((uint32_t)(wo_lo + g.sx - 1), divSx);
  const int wo_hi = min((int)xm_div((uint32_t)(w + g.pl), divSx), g.Wo - 1);
  if (FAST) {
    int off[4], code[4];
    float scale[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int ho = ho_lo + i, wo = wo_lo + j;
        ok[i + 2 * j] = (ho <= ho_hi) & (wo <= wo_hi);
        int hoc = min(ho, g.Ho - 1), woc = min(wo, g.Wo - 1);
        off[i + 2 * j] = hoc + g.Ho * woc;
        int w0 = woc * g.sx - g.pl, h0 = hoc * g.sy - g.pt;
        code[i + 2 * j] = (h - h0) + g.ph * (w - w0);
        int w2 = min(w0 + g.pw, g.W), h2 = min(h0 + g.ph, g.H);
        scale[i + 2 * j] = 1.0f / (float)((h2 - max(h0, 0)) * (w2 - max(w0, 0)));
      }
    for (int plane = blockIdx.y * blockDim.z + threadIdx.z; plane < planes; plane += gridDim.y * blockDim.z) {
      const size_t ob = (size_t)plane * g.Ho * g.Wo;
      float d[4];
      int a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d[k] = dy[ob + off[k]];
        a[k] = method == XM_POOL_MAX ? (int)amax[ob + off[k]] : 0;
      }
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (method == XM_POOL_MAX)
          acc += (ok[k] & (a[k] == code[k])) ? d[k] : 0.f;
        else
          acc += ok[k] ? d[k] * scale[k] : 0.f;
      }
      dx[(size_t)plane * g.H * g.W + h + (size_t)g.H * w] = acc;
    }
    return;
  }
  for (int plane = blockIdx.y * blockDim.z + threadIdx.z; plane < planes; plane += gridDim.y * blockDim.z) {
    const size_t ob = (size_t)plane * g.Ho * g.Wo;
    float acc = 0.f;
    for (int wo = wo_lo; wo <= wo_hi; ++wo)
      for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        const int w0 = wo * g.sx - g.pl, h0 = ho * g.sy - g.pt;
        const size_t o = ob + ho + (size_t)g.Ho * wo;
        if (method == XM_POOL_MAX) {
          const int code = (h - h0) + g.ph * (w - w0);
          if ((int)amax[o] == code) acc += dy[o];
        } else {
          const int w2 = min(w0 + g.pw, g.W), h2 = min(h0 + g.ph, g.H);
          acc += dy[o] * (1.0f / (float)((h2 - max(h0, 0)) * (w2 - max(w0, 0))));
        }
      }
    dx[(size_t)plane * g.H * g.W + h + (size_t)g.H * w] = acc;
  }
}

static int pool_geo(PoolGeo &g, int H, int W, int C, int N, int ph, int pw, int sy, int sx, int pt,
                    int pb, int pl, int pr, int method) {
  if (H <= 0 || W <= 0 || C <= 0 || N <= 0) return fail(XM_EINVAL, "vl_nnpool: empty tensor");
  if (ph < 1 || pw < 1 || sy < 1 || sx < 1 || pt < 0 || pb < 0 || pl < 0 || pr < 0)
    return fail(XM_EINVAL, "vl_nnpool: bad pool/stride/pad");
  if (method != XM_POOL_MAX && method != XM_POOL_AVG)
    return fail(XM_EINVAL, "vl_nnpool: unknown method %d", method);
  // MatConvNet: padding must be smaller than the window
  if (pt >= ph || pb >= ph || pl >= pw || pr >= pw)
    return fail(XM_EINVAL, "vl_nnpool: pad must be smaller than the pooling window");
  int Ho = out_size(H, pt, pb, ph, 1, sy), Wo = out_size(W, pl, pr, pw, 1, sx);
  if (Ho <= 0 || Wo <= 0) return fail(XM_EINVAL, "vl_nnpool: window larger than padded input");
  if (too_big(H, W, C, N)) return fail(XM_ETOOBIG, "vl_nnpool: tensor with >= 2^31 elements");
  g = PoolGeo{H, W, Ho, Wo, ph, pw, sy, sx, pt, pl};
  return XM_OK;
}

static int pool_forward(const float *x, int H, int W, int C, int N, int ph, int pw, int sy, int sx,
                        int pt, int pb, int pl, int pr, int method, float *y, unsigned char *amax,
                        hipStream_t st, const float *bn_g = nullptr, const float *bn_b = nullptr,
                        const float *bn_mom = nullptr) {
  PoolGeo g;
  int rc = pool_geo(g, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, method);
  if (rc) return rc;
  if (!x || (!y && !amax)) return fail(XM_EINVAL, "vl_nnpool: NULL tensor");
  if (amax && ph * pw > 255) return fail(XM_ENOTSUP, "vl_nnpool: windows above 255 elements are not built");
  if (!amax && !bn_g && g.Ho == 1 && g.Wo == 1 && ph >= H && pw >= W && !(pt | pl)) {
    int planes = C * N;
    hipLaunchKernelGGL(pool_global_kernel, dim3((planes + 3) / 4), dim3(256), 0, st, x, y, H * W,
                       planes, method);
    XM_LAUNCH_CHECK();
    return XM_OK;
  }
  // LDS-staged kernel: max pooling, 3x3 window, planes large enough to fill a block, 16-byte aligned tensor
  if (method == XM_POOL_MAX && ph == 3 && pw == 3 && ((uintptr_t)x & 15) == 0 && g.Ho * g.Wo >= 256 && path_on(kPathPoolLds)) {
    const int maxcols = 16 * 256 / H;  // 16 KB of LDS per block: 10 blocks per CU, measured best of 8 / 16 / 32 / 64
    int wob = maxcols >= pw ? (maxcols - pw) / sx + 1 : 0;
    wob = std::min(wob, g.Wo);
    if (wob >= 4 || (wob >= 1 && wob == g.Wo)) {
      // even out the column groups (the last block would otherwise hold a sliver)
      const int groups = (g.Wo + wob - 1) / wob;
      wob = (g.Wo + groups - 1) / groups;
      const int ncols = (wob - 1) * sx + pw;
      const size_t lds = ((size_t)ncols * H + 8) * sizeof(float);
      hipLaunchKernelGGL((pool_fwd_lds_kernel<3, 3>), dim3(C * N, groups), dim3(256), lds, st, x, y, amax, g,
                         make_fastdiv((uint32_t)g.Ho), wob, (size_t)H * W * C * N, bn_g, bn_b, bn_mom, C);
      XM_LAUNCH_CHECK();
      return XM_OK;
    }
  }
  PoolLaunch pl_ = pool_launch(g.Ho, g.Wo, (long long)C * N);
  if (ph == 3 && pw == 3)
    hipLaunchKernelGGL((pool_fwd_kernel<3, 3>), pl_.grid, pl_.block, 0, st, x, y, amax, g, C * N, method, bn_g, bn_b,
                       bn_mom, C);
  else if (ph == 5 && pw == 3)
    hipLaunchKernelGGL((pool_fwd_kernel<5, 3>), pl_.grid, pl_.block, 0, st, x, y, amax, g, C * N, method, bn_g, bn_b,
                       bn_mom, C);
  else if (ph == 2 && pw == 2)
    hipLaunchKernelGGL((pool_fwd_kernel<2, 2>), pl_.grid, pl_.block, 0, st, x, y, amax, g, C * N, method, bn_g, bn_b,
                       bn_mom, C);
  else
    hipLaunchKernelGGL((pool_fwd_kernel<0, 0>), pl_.grid, pl_.block, 0, st, x, y, amax, g, C * N, method, bn_g, bn_b,
                       bn_mom, C);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

// ---- fused backward of vl_nnpool('max') o vl_nnrelu o vl_nnbnorm ------------------------------
// dz(h,w) = [relu(bn(x)) > 0] * sum over covering windows [argmax(window) == (h,w)] dzdy_pool(window)
// is formed on the fly from the 1-byte routing table; it is never written to HBM.  Threads own a
// fixed (h, w) and walk the N samples of one channel, so all window arithmetic is done once.
struct Route4 {
  int off[4], code[4];
  bool ok[4];
};
__device__ __forceinline__ Route4 make_route(const PoolGeo &g, int h, int w, FastDiv divSy, FastDiv divSx) {
  int ho_lo = h + g.pt - g.ph + 1;
  ho_lo = ho_lo <= 0 ? 0 : (int)xm_div((uint32_t)(ho_lo + g.sy - 1), divSy);
  const int ho_hi = min((int)xm_div((uint32_t)(h + g.pt), divSy), g.Ho - 1);
  int wo_lo = w + g.pl - g.pw + 1;
  wo_lo = wo_lo <= 0 ? 0 : (int)xm_div((uint32_t)(wo_lo + g.sx - 1), divSx);
  const int wo_hi = min((int)xm_div((uint32_t)(w + g.pl), divSx), g.Wo - 1);
  Route4 r;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int ho = ho_lo + i, wo = wo_lo + j;
      r.ok[i + 2 * j] = (ho <= ho_hi) & (wo <= wo_hi);
      int hoc = min(ho, g.Ho - 1), woc = min(wo, g.Wo - 1);
      r.off[i + 2 * j] = hoc + g.Ho * woc;
      r.code[i + 2 * j] = (h - (hoc * g.sy - g.pt)) + g.ph * (w - (woc * g.sx - g.pl));
    }
  return r;
}
__device__ __forceinline__ float routed(const unsigned char *__restrict__ amax,
                                        const float *__restrict__ dp, size_t ob, const Route4 &r) {
  float d[4];
  int a[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    d[k] = dp[ob + r.off[k]];
    a[k] = (int)amax[ob + r.off[k]];
  }
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc += (r.ok[k] & (a[k] == r.code[k])) ? d[k] : 0.f;
  return acc;
}

// grid (gx, C, gz*S), block (bx, by): partial (sum dz, sum dz*(x-mu)) per block
__global__ void __launch_bounds__(256)
bnpool_bwd_partial_kernel(const float *__restrict__ x, const float *__restrict__ gg,
                          const float *__restrict__ bb, const float *__restrict__ mom,
                          const unsigned char *__restrict__ amax, const float *__restrict__ dp,
                          double *__restrict__ part, PoolGeo g, FastDiv divSy, FastDiv divSx, int C, int N,
                          int S, int gz) {
  const int c = blockIdx.y;
  const int zz = blockIdx.z / S, sp = blockIdx.z % S;
  const int h = zz * blockDim.x + threadIdx.x;
  const int w = blockIdx.x * blockDim.y + threadIdx.y;
  double a = 0.0, b = 0.0;
  if (h < g.H && w < g.W) {
    const Route4 r = make_route(g, h, w, divSy, divSx);
    const float mu = mom[c], sc = gg[c] / mom[C + c], bc = bb[c];
#pragma unroll 4
    for (int n = sp; n < N; n += S) {
      const size_t plane = (size_t)c + (size_t)C * n;
      float xv = x[plane * g.H * g.W + h + (size_t)g.H * w];
      float d = routed(amax, dp, plane * g.Ho * g.Wo, r);
      d = (sc * (xv - mu) + bc > 0.f) ? d : 0.f;
      a += (double)d;
      b += (double)d * ((double)xv - (double)mu);
    }
  }
  __shared__ double red[8];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if (tid < 8) red[tid] = 0.0;  // blocks may hold fewer than 4 waves (small planes)
  __syncthreads();
  a = xm_wave_sum_d(a);
  b = xm_wave_sum_d(b);
  if ((tid & 63) == 0) {
    red[tid >> 6] = a;
    red[4 + (tid >> 6)] = b;
  }
  __syncthreads();
  if (tid == 0) {
    size_t nb = (size_t)gridDim.x * gridDim.z;
    size_t slot = ((size_t)c * nb + (size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2;
    part[slot] = red[0] + red[1] + red[2] + red[3];
    part[slot + 1] = red[4] + red[5] + red[6] + red[7];
  }
  (void)gz;
}

// `part2` (optional): per-block partial sums of the produced dx -- the derivative of a bias that was
// added right before the normalisation (dzdb of the producing vl_nnconv = sum over pixels of its dzdy),
// so that tensor does not have to be read a second time just to be summed.
__global__ void __launch_bounds__(256)
bnpool_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ gg,
                        const float *__restrict__ bb, const float *__restrict__ mom,
                        const double *__restrict__ sums, const unsigned char *__restrict__ amax,
                        const float *__restrict__ dp, float *__restrict__ dx, double *__restrict__ part2,
                        PoolGeo g, FastDiv divSy, FastDiv divSx, int C, int N, int S, double m,
                        int train) {
  const int c = blockIdx.y;
  const int zz = blockIdx.z / S, sp = blockIdx.z % S;
  const int h = zz * blockDim.x + threadIdx.x;
  const int w = blockIdx.x * blockDim.y + threadIdx.y;
  double acc = 0.0;
  if (h < g.H && w < g.W) {
    const Route4 r = make_route(g, h, w, divSy, divSx);
    const float mu = mom[c], sg = mom[C + c];
    const float gs = gg[c] / sg, bc = bb[c];
    const double gsd = (double)gg[c] / (double)sg;
    const double c1 = train ? sums[c] / m : 0.0;
    const double c2 = train ? sums[C + c] / (m * (double)sg * (double)sg) : 0.0;
#pragma unroll 4
    for (int n = sp; n < N; n += S) {
      const size_t plane = (size_t)c + (size_t)C * n;
      const size_t xi = plane * g.H * g.W + h + (size_t)g.H * w;
      float xv = x[xi];
      float d = routed(amax, dp, plane * g.Ho * g.Wo, r);
      d = (gs * (xv - mu) + bc > 0.f) ? d : 0.f;
      float o = (float)(gsd * ((double)d - c1 - ((double)xv - (double)mu) * c2));
      dx[xi] = o;
      acc += (double)o;
    }
  }
  if (part2) {
    __shared__ double red[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (tid < 4) red[tid] = 0.0;
    __syncthreads();
    acc = xm_wave_sum_d(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      size_t nb = (size_t)gridDim.x * gridDim.z;
      part2[(size_t)c * nb + (size_t)blockIdx.z * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
  }
}

// Pooled-domain sums.  The routed derivative is non-zero only at window maxima that passed the relu, and there
//   y_p = g/sigma (x_argmax - mu) + b > 0    =>    x_argmax - mu = (y_p - b) sigma / g,
// so   sum dz = sum_p [y_p > 0] dzdy_p   and   sum dz (x - mu) = sigma/g * sum_p [y_p > 0] dzdy_p (y_p - b):
// two POOLED tensors are read instead of x + table + dzdy (3.7x fewer bytes on a 3x3 / stride-2 layer).
// Channels whose gain is too small for the inversion (g == 0 or |b| > 100 |g|: rounding of y_p - b would be amplified)
// gather x at the recorded argmax instead.  grid (C, S), part[c][s] = (sum dz, sum dz (x - mu)).
__global__ void __launch_bounds__(256)
bnpool_bwd_partial_pooled_kernel(const float *__restrict__ x, const float *__restrict__ yp,
                                 const float *__restrict__ dp, const unsigned char *__restrict__ amax,
                                 const float *__restrict__ gg, const float *__restrict__ bb,
                                 const float *__restrict__ mom, double *__restrict__ part, PoolGeo g,
                                 FastDiv divHo, int C, int N, int S) {
  const int c = blockIdx.x, s = blockIdx.y;
  const int HWo = g.Ho * g.Wo;
  const float gc = gg[c], bc = bb[c];
  const double mu = mom[c];
  const bool invert = gc != 0.f && fabsf(bc) <= 100.f * fabsf(gc);   // g == 0 (zero-initialised gain): 0 * inf
  double a = 0.0, b = 0.0;
  for (int n = s; n < N; n += S) {
    const size_t plane = (size_t)c + (size_t)C * n;
    const size_t off = plane * HWo;
    if (invert) {
#pragma unroll 4
      for (int i = threadIdx.x; i < HWo; i += 256) {
        const float yv = yp[off + i];
        const float d = yv > 0.f ? dp[off + i] : 0.f;
        a += (double)d;
        b += (double)d * ((double)yv - (double)bc);
      }
    } else {
      for (int i = threadIdx.x; i < HWo; i += 256) {
        const float yv = yp[off + i];
        const float d = yv > 0.f ? dp[off + i] : 0.f;
        const int code = amax[off + i];
        const int wo = (int)xm_div((uint32_t)i, divHo), ho = i - wo * g.Ho;
        const int dw = code / g.ph, dh = code - dw * g.ph;
        const int h = min(max(ho * g.sy - g.pt + dh, 0), g.H - 1), w = min(max(wo * g.sx - g.pl + dw, 0), g.W - 1);
        a += (double)d;
        b += (double)d * ((double)x[plane * g.H * g.W + h + (size_t)g.H * w] - mu);
      }
    }
  }
  if (invert) b *= (double)mom[C + c] / (double)gc;
  __shared__ double red[8];
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) {
    part[2 * ((size_t)c * S + s)] = a;
    part[2 * ((size_t)c * S + s) + 1] = b;
  }
}

// Patch variant of bnpool_bwd_apply_kernel: a thread owns the SY x SX input elements whose (h + pt, w + pl)
// fall in one stride cell.  With ph <= 2 SY and pw <= 2 SX they are covered by the same <= 2 x 2 windows, so
// the 4 (dzdy, table byte) pairs are loaded once per PATCH instead of once per element (10 loads per four
// elements instead of 36 on the 3x3 / stride-2 layers); contributions are added in the same window order.
// grid (gx, C, gz * S), block (bx, by) over patches.  VEC2 (SY == 2, H and pt even): the two rows of a patch
// column are one aligned 8-byte access.
template <int SY, int SX, bool VEC2>
__global__ void __launch_bounds__(256)
bnpool_bwd_apply_patch_kernel(const float *__restrict__ x, const float *__restrict__ gg,
                              const float *__restrict__ bb, const float *__restrict__ mom,
                              const double *__restrict__ sums, const unsigned char *__restrict__ amax,
                              const float *__restrict__ dp, float *__restrict__ dx, double *__restrict__ part2,
                              PoolGeo g, int KH, int KW, int C, int N, int S, double m, int train) {
  const int c = blockIdx.y;
  const int zz = blockIdx.z / S, sp = blockIdx.z % S;
  const int kh = zz * blockDim.x + threadIdx.x;
  const int kw = blockIdx.x * blockDim.y + threadIdx.y;
  double acc = 0.0;
  if (kh < KH && kw < KW) {
    int off[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ho = kh - 1 + i, wo = kw - 1 + j;
        ok[i + 2 * j] = ((unsigned)ho < (unsigned)g.Ho) & ((unsigned)wo < (unsigned)g.Wo);
        off[i + 2 * j] = min(max(ho, 0), g.Ho - 1) + g.Ho * min(max(wo, 0), g.Wo - 1);
      }
    const int h0 = kh * SY - g.pt, w0 = kw * SX - g.pl;
    const float mu = mom[c], sg = mom[C + c];
    const float gs = gg[c] / sg, bc = bb[c];
    const double gsd = (double)gg[c] / (double)sg;
    const double c1 = train ? sums[c] / m : 0.0;
    const double c2 = train ? sums[C + c] / (m * (double)sg * (double)sg) : 0.0;
#pragma unroll 2
    for (int n = sp; n < N; n += S) {
      const size_t plane = (size_t)c + (size_t)C * n;
      const size_t ob = plane * g.Ho * g.Wo;
      float d[4];
      int a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d[k] = dp[ob + off[k]];
        a[k] = ok[k] ? (int)amax[ob + off[k]] : -1;
      }
#pragma unroll
      for (int rw = 0; rw < SX; ++rw) {
        const int w = w0 + rw;
        if ((unsigned)w >= (unsigned)g.W) continue;
        const size_t xc = plane * g.H * g.W + (size_t)g.H * w;
        float xv[SY], o[SY];
        if (VEC2) {
          if ((unsigned)h0 >= (unsigned)g.H) continue;  // H, pt even: both rows in or both out
          const float2 t = *reinterpret_cast<const float2 *>(x + xc + h0);
          xv[0] = t.x;
          xv[SY - 1] = t.y;
        } else {
#pragma unroll
          for (int rh = 0; rh < SY; ++rh) xv[rh] = (unsigned)(h0 + rh) < (unsigned)g.H ? x[xc + h0 + rh] : 0.f;
        }
#pragma unroll
        for (int rh = 0; rh < SY; ++rh) {
          float dz = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int dh = rh + (1 - i) * SY, dw = rw + (1 - j) * SX;
              const bool hit = (dh < g.ph) & (dw < g.pw) & (a[i + 2 * j] == dh + g.ph * dw);
              dz += hit ? d[i + 2 * j] : 0.f;
            }
          dz = (gs * (xv[rh] - mu) + bc > 0.f) ? dz : 0.f;
          o[rh] = (float)(gsd * ((double)dz - c1 - ((double)xv[rh] - (double)mu) * c2));
        }
        if (VEC2) {
          *reinterpret_cast<float2 *>(dx + xc + h0) = float2{o[0], o[SY - 1]};
          acc += (double)o[0];
          acc += (double)o[SY - 1];
        } else {
#pragma unroll
          for (int rh = 0; rh < SY; ++rh)
            if ((unsigned)(h0 + rh) < (unsigned)g.H) {
              dx[xc + h0 + rh] = o[rh];
              acc += (double)o[rh];
            }
        }
      }
    }
  }
  if (part2) {
    __shared__ double red[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (tid < 4) red[tid] = 0.0;
    __syncthreads();
    acc = xm_wave_sum_d(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      size_t nb = (size_t)gridDim.x * gridDim.z;
      part2[(size_t)c * nb + (size_t)blockIdx.z * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
  }
}

__global__ void __launch_bounds__(256)
sum_partials_kernel(const double *__restrict__ part, float *__restrict__ out, int C, int S) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0.0;
  for (int i = lane; i < S; i += 64) s += part[(size_t)c * S + i];
  s = xm_wave_sum_d(s);
  if (lane == 0) out[c] = (float)s;
}

static int pool_backward(const float *x, const unsigned char *amax, int H, int W, int C, int N, int ph,
                         int pw, int sy, int sx, int pt, int pb, int pl, int pr, int method,
                         const float *dzdy, float *dx_out, hipStream_t st) {
  PoolGeo g;
  int rc = pool_geo(g, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, method);
  if (rc) return rc;
  if (!dzdy || !dx_out) return fail(XM_EINVAL, "vl_nnpool: NULL tensor");
  if (method == XM_POOL_MAX && !amax) {
    // plain MatConvNet signature: recompute the routing table from X into scratch first
    if (!x) return fail(XM_EINVAL, "vl_nnpool: X is NULL");
    size_t ny = (size_t)g.Ho * g.Wo * C * N;
    WsCarver ws;
    rc = ws.init(WsCarver::need(ny, 1), st);
    if (rc) return rc;
    unsigned char *aw = ws.take<unsigned char>(ny);
    rc = pool_forward(x, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, method, nullptr, aw, st);
    if (rc) return rc;
    amax = aw;
  }
  PoolLaunch pl_ = pool_launch(H, W, (long long)C * N);
  const bool fast = (ph + sy - 1) / sy <= 2 && (pw + sx - 1) / sx <= 2;
  if (fast)
    hipLaunchKernelGGL(pool_bwd_kernel<true>, pl_.grid, pl_.block, 0, st, amax, dzdy, dx_out, g,
                       make_fastdiv((uint32_t)sy), make_fastdiv((uint32_t)sx), C * N, method);
  else
    hipLaunchKernelGGL(pool_bwd_kernel<false>, pl_.grid, pl_.block, 0, st, amax, dzdy, dx_out, g,
                       make_fastdiv((uint32_t)sy), make_fastdiv((uint32_t)sx), C * N, method);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

static int bnrelupool_forward(const float *x, int H, int W, int C, int N, const float *g,
                              const float *b, float eps, const float *moments_in, int ph, int pw,
                              int sy, int sx, int pt, int pb, int pl, int pr, float *y_pool,
                              unsigned char *amax, float *moments_out, hipStream_t st) {
  int rc = bn_check(H, W, C, N);
  if (rc) return rc;
  if (!x || !g || !b || !y_pool || !amax || !moments_out)
    return fail(XM_EINVAL, "bnorm+relu+pool: NULL tensor");
  const int HW = H * W;
  if (!moments_in) {
    const int S = bn_splits(C, N);
    WsCarver ws;
    rc = ws.init(WsCarver::need((size_t)2 * C * S, 8), st);
    if (rc) return rc;
    double *part = ws.take<double>((size_t)2 * C * S);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, S), dim3(256), 0, st, x, part, HW, C, N, S, bn_run_div(HW));
    XM_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, x, part, moments_out,
                       HW, C, S, (double)HW * N, eps);
    XM_LAUNCH_CHECK();
  } else if (moments_out != moments_in) {
    XM_HIP(hipMemcpyAsync(moments_out, moments_in, sizeof(float) * 2 * C, hipMemcpyDeviceToDevice, st));
  }
  return pool_forward(x, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, XM_POOL_MAX, y_pool, amax, st, g, b,
                      moments_out);
}

static int bnrelupool_backward(const float *x, int H, int W, int C, int N, const float *g,
                               const float *b, const float *moments, int train, int ph, int pw, int sy,
                               int sx, int pt, int pb, int pl, int pr, const unsigned char *amax,
                               const float *y_pool, const float *dzdy_pool, float *dx_out, float *dg_out,
                               float *db_out, float *dxsum_out, hipStream_t st) {
  PoolGeo pg;
  int rc = pool_geo(pg, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, XM_POOL_MAX);
  if (rc) return rc;
  if (!x || !g || !b || !moments || !amax || !dzdy_pool)
    return fail(XM_EINVAL, "bnorm+relu+pool: NULL tensor");
  if ((ph + sy - 1) / sy > 2 || (pw + sx - 1) / sx > 2)
    return fail(XM_ENOTSUP, "bnorm+relu+pool backward: more than 2x2 windows cover an element");
  if (dxsum_out && !dx_out) return fail(XM_EINVAL, "bnorm+relu+pool backward: dxsum needs dx");
  const bool no_patch = !path_on(kPathPoolPatch);
  const bool no_pooled = !path_on(kPathPoolPooled);
  // element kernels: (bx, by) threads own (h, w); grid.y = channel; the N samples are split S ways
  int bx = pow2_ge(H, 256), by = 256 / bx;
  by = pow2_ge(W, by);
  int gx = (W + by - 1) / by, gz = (H + bx - 1) / bx;
  int S = std::max(1, std::min(N, 4096 / std::max(1, C * gx * gz)));
  const size_t nb = (size_t)gx * gz * S;
  // patch kernel (apply): threads own stride cells
  const bool patch = !no_patch && dx_out && ((sy == 2 && sx == 2) || (sy == 3 && sx == 2));
  const int KH = (H + pt + sy - 1) / sy, KW = (W + pl + sx - 1) / sx;
  int pbx = pow2_ge(KH, 256), pby = pow2_ge(KW, 256 / pbx);
  int pgx = (KW + pby - 1) / pby, pgz = (KH + pbx - 1) / pbx;
  int pS = std::max(1, std::min(N, 16384 / std::max(1, C * pgx * pgz)));   // >= 16 k blocks (4 k and 64 k measured worse)
  const size_t pnb = (size_t)pgx * pgz * pS;
  // pooled-domain sums
  const bool pooled = !no_pooled && y_pool != nullptr;
  const int S2 = bn_splits(C, N);
  const size_t npart = pooled ? (size_t)S2 : nb;
  const size_t npart2 = patch ? pnb : nb;
  WsCarver ws;
  rc = ws.init(WsCarver::need((size_t)2 * C * npart, 8) + WsCarver::need((size_t)2 * C, 8) +
                   WsCarver::need((size_t)C * npart2, 8), st);
  if (rc) return rc;
  double *part = ws.take<double>((size_t)2 * C * npart);
  double *sums = ws.take<double>((size_t)2 * C);
  double *part2 = dxsum_out ? ws.take<double>((size_t)C * npart2) : nullptr;
  dim3 grid(gx, C, gz * S), block(bx, by);
  FastDiv dsy = make_fastdiv((uint32_t)sy), dsx = make_fastdiv((uint32_t)sx);
  if (pooled)
    hipLaunchKernelGGL(bnpool_bwd_partial_pooled_kernel, dim3(C, S2), dim3(256), 0, st, x, y_pool, dzdy_pool, amax,
                       g, b, moments, part, pg, make_fastdiv((uint32_t)pg.Ho), C, N, S2);
  else
    hipLaunchKernelGGL(bnpool_bwd_partial_kernel, grid, block, 0, st, x, g, b, moments, amax, dzdy_pool,
                       part, pg, dsy, dsx, C, N, S, gz);
  XM_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, moments, sums,
                     dg_out, db_out, C, (int)npart);
  XM_LAUNCH_CHECK();
  if (dx_out) {
    const double m = (double)H * W * N;
    if (patch) {
      dim3 pgrid(pgx, C, pgz * pS), pblock(pbx, pby);
      const bool vec2 = sy == 2 && (H & 1) == 0 && (pt & 1) == 0 && (((uintptr_t)x | (uintptr_t)dx_out) & 7) == 0;
      if (sy == 2 && vec2)
        hipLaunchKernelGGL((bnpool_bwd_apply_patch_kernel<2, 2, true>), pgrid, pblock, 0, st, x, g, b, moments, sums,
                           amax, dzdy_pool, dx_out, part2, pg, KH, KW, C, N, pS, m, train);
      else if (sy == 2)
        hipLaunchKernelGGL((bnpool_bwd_apply_patch_kernel<2, 2, false>), pgrid, pblock, 0, st, x, g, b, moments,
                           sums, amax, dzdy_pool, dx_out, part2, pg, KH, KW, C, N, pS, m, train);
      else
        hipLaunchKernelGGL((bnpool_bwd_apply_patch_kernel<3, 2, false>), pgrid, pblock, 0, st, x, g, b, moments,
                           sums, amax, dzdy_pool, dx_out, part2, pg, KH, KW, C, N, pS, m, train);
    } else {
      hipLaunchKernelGGL(bnpool_bwd_apply_kernel, grid, block, 0, st, x, g, b, moments, sums, amax,
                         dzdy_pool, dx_out, part2, pg, dsy, dsx, C, N, S, m, train);
    }
    XM_LAUNCH_CHECK();
    if (part2) {
      hipLaunchKernelGGL(sum_partials_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part2, dxsum_out, C,
                         (int)npart2);
      XM_LAUNCH_CHECK();
    }
  }
  return XM_OK;
}

// ---- the sums of the fused backward, for a consumer that rebuilds the bnorm's DZDX itself ----------------------------
// conv.hip's xm_nnconv_backward_filter_bnrelupool never materialises DZDX of the bnorm: the filter-derivative kernel of the
// producing convolution recomputes it per element from the pooled derivative and the routing table.  What it needs from
// here: dg / db, and per channel the constants of  dx = g/sigma dz - k1 - k2 (x - mu)  in fp32, k1 = g/sigma mean(dz)
// split in two floats (a rounded k1 would shift all elements of a channel the same way).
__global__ void __launch_bounds__(256)
bnpool_rowconst_kernel(const float *__restrict__ gg, const float *__restrict__ bb, const float *__restrict__ mom,
                       const double *__restrict__ sums, float *__restrict__ rowc, int C, double m, int train) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double sg = (double)mom[C + c], gsd = (double)gg[c] / sg;
  const double k1 = train ? gsd * (sums[c] / m) : 0.0;
  const double k2 = train ? gsd * (sums[C + c] / (m * sg * sg)) : 0.0;
  const float k1h = (float)k1;
  float *o = rowc + 6 * c;
  o[0] = gg[c] / mom[C + c];      // the gate uses the forward pass's own fp32 expression g/sigma (x - mu) + b > 0
  o[1] = mom[c];
  o[2] = bb[c];
  o[3] = k1h;
  o[4] = (float)(k1 - (double)k1h);
  o[5] = (float)k2;
}

size_t bnpool_sums_need(int C, int N) {
  return WsCarver::need((size_t)2 * C * bn_splits(C, N), 8) + WsCarver::need((size_t)2 * C, 8);
}

int bnpool_backward_sums(WsCarver &ws, const float *x, int H, int W, int C, int N, const float *g, const float *b,
                         const float *moments, int train, int ph, int pw, int sy, int sx, int pt, int pb, int pl, int pr,
                         const unsigned char *amax, const float *y_pool, const float *dzdy_pool, float *dg_out,
                         float *db_out, float *rowc_out, hipStream_t st) {
  PoolGeo pg;
  int rc = pool_geo(pg, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, XM_POOL_MAX);
  if (rc) return rc;
  if (!x || !g || !b || !moments || !amax || !y_pool || !dzdy_pool || !rowc_out)
    return fail(XM_EINVAL, "bnorm+relu+pool sums: NULL tensor");
  const int S2 = bn_splits(C, N);
  double *part = ws.take<double>((size_t)2 * C * S2);
  double *sums = ws.take<double>((size_t)2 * C);
  hipLaunchKernelGGL(bnpool_bwd_partial_pooled_kernel, dim3(C, S2), dim3(256), 0, st, x, y_pool, dzdy_pool, amax, g, b,
                     moments, part, pg, make_fastdiv((uint32_t)pg.Ho), C, N, S2);
  XM_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, moments, sums, dg_out, db_out, C,
                     S2);
  XM_LAUNCH_CHECK();
  hipLaunchKernelGGL(bnpool_rowconst_kernel, dim3((C + 255) / 256), dim3(256), 0, st, g, b, moments, sums, rowc_out, C,
                     (double)H * W * N, train);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

}  // namespace xm

using namespace xm;

// Global average pooling (the SE squeeze) backward at a FORK: dx = accum + dzdy(plane) / (H W) -- the derivative the
// other consumer of X (the SE excite) already left is added in the same pass instead of a broadcast pass + a sum pass.
// Same two roundings as the separate passes (multiply, then add: no fused multiply-add).
template <bool VEC>
__global__ void __launch_bounds__(256)
pool_global_avg_bwd_accum_kernel(const float *__restrict__ dy, const float *__restrict__ accum, float *__restrict__ dx,
                                 size_t n, xm::FastDiv divHW, float scale) {
#pragma clang fp contract(off)   // round(accum + round(dy * scale)): the two roundings of the separate passes
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (VEC) {
    if (idx * 4 >= n) return;
    const float t = dy[xm_div((uint32_t)(idx * 4), divHW)] * scale;
    const float4 a = reinterpret_cast<const float4 *>(accum)[idx];
    reinterpret_cast<float4 *>(dx)[idx] = make_float4(a.x + t, a.y + t, a.z + t, a.w + t);
  } else {
    if (idx >= n) return;
    const float t = dy[xm_div((uint32_t)idx, divHW)] * scale;
    dx[idx] = accum[idx] + t;
  }
}

extern "C" {

int xm_nnbnorm_forward_fused(const float *x, int H, int W, int C, int N, const float *g,
                             const float *b, float epsilon, const float *moments_in, float *y,
                             float *moments_out, int flags, void *stream) {
  return bnorm_forward(x, H, W, C, N, g, b, epsilon, moments_in, y, moments_out,
                       (flags & XM_FUSE_RELU) ? 1 : 0, (hipStream_t)stream);
}

int xm_nnbnorm_forward(const float *x, int H, int W, int C, int N, const float *g, const float *b,
                       float epsilon, const float *moments_in, float *y, float *moments_out,
                       void *stream) {
  return bnorm_forward(x, H, W, C, N, g, b, epsilon, moments_in, y, moments_out, 0,
                       (hipStream_t)stream);
}

int xm_nnbnorm_backward(const float *x, int H, int W, int C, int N, const float *g, const float *b,
                        const float *dzdy, float epsilon, const float *moments_in, float *dx_out,
                        float *dg_out, float *db_out, float *moments_out, void *stream) {
  (void)b;
  return bnorm_backward(x, nullptr, H, W, C, N, g, dzdy, epsilon, moments_in, dx_out, dg_out, db_out,
                        moments_out, (hipStream_t)stream);
}

int xm_nnbnorm_backward_fused(const float *x, const float *y, int H, int W, int C, int N,
                              const float *g, const float *b, const float *dzdy, float epsilon,
                              const float *moments_in, float *dx_out, float *dg_out, float *db_out,
                              float *moments_out, int flags, void *stream) {
  (void)b;
  if ((flags & XM_FUSE_RELU) && !y) return fail(XM_EINVAL, "vl_nnbnorm(fused bwd): y is NULL");
  return bnorm_backward(x, (flags & XM_FUSE_RELU) ? y : nullptr, H, W, C, N, g, dzdy, epsilon,
                        moments_in, dx_out, dg_out, db_out, moments_out, (hipStream_t)stream,
                        (flags & XM_BN_BATCH_MOMENTS) != 0);
}

int xm_nnbnorm_backward_dxsum(const float *x, const float *y, int H, int W, int C, int N,
                              const float *g, const float *b, const float *dzdy, float epsilon,
                              const float *moments_in, float *dx_out, float *dg_out, float *db_out,
                              float *moments_out, float *dxsum_out, int flags, void *stream) {
  (void)b;
  if ((flags & XM_FUSE_RELU) && !y) return fail(XM_EINVAL, "vl_nnbnorm(fused bwd): y is NULL");
  return bnorm_backward(x, (flags & XM_FUSE_RELU) ? y : nullptr, H, W, C, N, g, dzdy, epsilon,
                        moments_in, dx_out, dg_out, db_out, moments_out, (hipStream_t)stream,
                        (flags & XM_BN_BATCH_MOMENTS) != 0, dxsum_out);
}

int xm_nnpool_forward(const float *x, int H, int W, int C, int N, int ph, int pw, int sy, int sx,
                      int pt, int pb, int pl, int pr, int method, float *y, void *stream) {
  return pool_forward(x, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, method, y, nullptr,
                      (hipStream_t)stream);
}

int xm_nnpool_forward_argmax(const float *x, int H, int W, int C, int N, int ph, int pw, int sy,
                             int sx, int pt, int pb, int pl, int pr, float *y, unsigned char *argmax,
                             void *stream) {
  if (!y || !argmax) return fail(XM_EINVAL, "vl_nnpool: NULL tensor");
  return pool_forward(x, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, XM_POOL_MAX, y, argmax,
                      (hipStream_t)stream);
}

int xm_nnpool_global_avg_backward_accum(const float *dzdy, const float *accum, float *dx_out, int H, int W, int C, int N,
                                        void *stream) {
  if (H <= 0 || W <= 0 || C <= 0 || N <= 0) return xm::fail(XM_EINVAL, "vl_nnpool: empty tensor");
  if (!dzdy || !accum || !dx_out) return xm::fail(XM_EINVAL, "vl_nnpool: NULL tensor");
  if (xm::too_big(H, W, C, N)) return xm::fail(XM_ETOOBIG, "vl_nnpool: tensor with >= 2^31 elements");
  const size_t n = (size_t)H * W * C * N;
  const xm::FastDiv d = xm::make_fastdiv((uint32_t)(H * W));
  const float scale = 1.0f / (float)(H * W);   // correctly rounded, as pool_bwd_kernel's
  const bool vec = (H * W) % 4 == 0 && ((((uintptr_t)accum | (uintptr_t)dx_out)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(pool_global_avg_bwd_accum_kernel<true>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dzdy, accum, dx_out, n, d, scale);
  else
    hipLaunchKernelGGL(pool_global_avg_bwd_accum_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dzdy, accum, dx_out, n, d, scale);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

int xm_nnpool_backward(const float *x, int H, int W, int C, int N, int ph, int pw, int sy, int sx,
                       int pt, int pb, int pl, int pr, int method, const float *dzdy, float *dx_out,
                       void *stream) {
  return pool_backward(x, nullptr, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, method, dzdy, dx_out,
                       (hipStream_t)stream);
}

int xm_nnpool_backward_argmax(const unsigned char *argmax, int H, int W, int C, int N, int ph, int pw,
                              int sy, int sx, int pt, int pb, int pl, int pr, const float *dzdy,
                              float *dx_out, void *stream) {
  if (!argmax) return fail(XM_EINVAL, "vl_nnpool: argmax table is NULL");
  return pool_backward(nullptr, argmax, H, W, C, N, ph, pw, sy, sx, pt, pb, pl, pr, XM_POOL_MAX, dzdy,
                       dx_out, (hipStream_t)stream);
}

int xm_nnbnorm_relu_pool_forward(const float *x, int H, int W, int C, int N, const float *g,
                                 const float *b, float epsilon, const float *moments_in, int ph, int pw,
                                 int sy, int sx, int pt, int pb, int pl, int pr, float *y_pool,
                                 unsigned char *argmax, float *moments_out, void *stream) {
  return bnrelupool_forward(x, H, W, C, N, g, b, epsilon, moments_in, ph, pw, sy, sx, pt, pb, pl, pr,
                            y_pool, argmax, moments_out, (hipStream_t)stream);
}

int xm_nnbnorm_relu_pool_backward(const float *x, int H, int W, int C, int N, const float *g,
                                  const float *b, const float *moments, int train, int ph, int pw,
                                  int sy, int sx, int pt, int pb, int pl, int pr,
                                  const unsigned char *argmax, const float *y_pool, const float *dzdy_pool,
                                  float *dx_out, float *dg_out, float *db_out, float *dxsum_out, void *stream) {
  return bnrelupool_backward(x, H, W, C, N, g, b, moments, train, ph, pw, sy, sx, pt, pb, pl, pr, argmax, y_pool,
                             dzdy_pool, dx_out, dg_out, db_out, dxsum_out, (hipStream_t)stream);
}
}
