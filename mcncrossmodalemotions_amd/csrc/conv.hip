// vl_nnconv forward / backward on gfx950 -- host-side planning + C ABI.
// Replaces MatConvNet's vl_nnconv MEX (matlab/src/vl_nnconv.cu, bits/nnconv.cu: im2row + SGEMM
// per image) behind the same operator contract; see include/xmodal.h and conv_kernels.h.
#include <algorithm>
#include <vector>

#include "conv_kernels.h"

namespace xm {

// ---- small helper kernels -------------------------------------------------------------------

// zero-pad the filter bank [M][R] to [M][Rp] (only needed when R % 16 != 0: the two conv1 layers)
__global__ void pad_filter_kernel(const float *__restrict__ f, float *__restrict__ o, int M, int R,
                                  int Rp) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)M * Rp) return;
  int r = (int)(i % Rp);
  size_t m = i / Rp;
  o[i] = r < R ? f[m * R + r] : 0.f;
}

// dgrad operand for one stride-parity class: At[c][iu + nU*(iv + nV*k)] = F[u(iu), v(iv), c, k]
// with u(iu) = u0 + iu*ustep (the taps whose u*dil == a mod sy), zero padded to lda columns.
__global__ void prep_dgrad_filter_kernel(const float *__restrict__ f, float *__restrict__ o, int FH,
                                         int FW, int FC, int Kg, int u0, int ustep, int nU, int v0,
                                         int vstep, int nV, int lda) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)FC * lda) return;
  int rr = (int)(i % lda);
  int c = (int)(i / lda);
  float val = 0.f;
  if (rr < nU * nV * Kg) {
    int iu = rr % nU, iv = (rr / nU) % nV, k = rr / (nU * nV);
    int u = u0 + iu * ustep, v = v0 + iv * vstep;
    val = f[(size_t)u + FH * ((size_t)v + FW * ((size_t)c + (size_t)FC * k))];
  }
  o[i] = val;
}

__global__ void reduce_splits_kernel(const float *__restrict__ part, float *__restrict__ out,
                                     int M, int R, int ldo, int splits, size_t splitStride) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)M * R) return;
  int r = (int)(i % R);
  size_t m = i / R;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += part[z * splitStride + m * ldo + r];
  out[i] = s;
}

// dzdb(k) = sum over pixels and samples of dzdy(:,:,k,:) : one block per channel
__global__ void __launch_bounds__(256)
bias_grad_kernel(const float *__restrict__ dy, float *__restrict__ db, int HW, int K, int N) {
  int k = blockIdx.x;
  float s = 0.f;
  for (int n = 0; n < N; ++n) {
    const float *p = dy + (size_t)HW * (k + (size_t)K * n);
    for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
  }
  __shared__ float red[4];
  s = xm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) db[k] = red[0] + red[1] + red[2] + red[3];
}

// ---- tile configuration ---------------------------------------------------------------------
struct Cfg {
  int tm, tn, wgm, wgn;
  int bm() const { return 32 * tm * wgm; }
  int bn() const { return 32 * tn * wgn; }
};
static const Cfg kCfgs[] = {
    {2, 2, 2, 2},  // 128 x 128
    {2, 2, 1, 4},  //  64 x 256
    {3, 1, 1, 4},  //  96 x 128
    {1, 2, 2, 2},  //  64 x 128
    {1, 1, 2, 2},  //  64 x  64
    {1, 1, 4, 1},  // 128 x  32
    {1, 1, 1, 4},  //  32 x 128
};
constexpr int kNumCfg = sizeof(kCfgs) / sizeof(kCfgs[0]);

// crude cost model: padded MACs per CU-wave, with a mild bonus for bigger wave tiles
static int pick_cfg(long long M, long long NP) {
  double best = 1e300;
  int bi = 0;
  for (int i = 0; i < kNumCfg; ++i) {
    const Cfg &c = kCfgs[i];
    long long tiles = ((M + c.bm() - 1) / c.bm()) * ((NP + c.bn() - 1) / c.bn());
    int per_cu = c.bm() * c.bn() >= 128 * 128 ? 2 : 3;  // co-resident blocks that overlap
    long long slots = 256LL * per_cu;
    long long rounds = (tiles + slots - 1) / slots;
    double eff = (c.tm * c.tn >= 4) ? 1.0 : (c.tm * c.tn >= 2 ? 1.15 : 1.35);
    double cost = (double)rounds * slots / per_cu * c.bm() * c.bn() * eff;
    // when everything fits in one round the padded tile work itself is what matters
    if (rounds == 1) cost = (double)tiles * c.bm() * c.bn() * eff / std::min<long long>(tiles, 256) * 1.0;
    if (cost < best) {
      best = cost;
      bi = i;
    }
  }
  return bi;
}

template <bool CHECK>
static void launch_gemm_cfg(int ci, const ConvGemmArgs &a, int nblk, hipStream_t st) {
  dim3 grid(nblk), block(256);
  switch (ci) {
    case 0: hipLaunchKernelGGL((conv_gemm_kernel<2, 2, 2, 2, CHECK>), grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL((conv_gemm_kernel<2, 2, 1, 4, CHECK>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((conv_gemm_kernel<3, 1, 1, 4, CHECK>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((conv_gemm_kernel<1, 2, 2, 2, CHECK>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((conv_gemm_kernel<1, 1, 2, 2, CHECK>), grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL((conv_gemm_kernel<1, 1, 4, 1, CHECK>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((conv_gemm_kernel<1, 1, 1, 4, CHECK>), grid, block, 0, st, a); break;
  }
}

static int g_force_cfg = -1;  // test hook (xm_debug_force_conv_cfg)

static int launch_gemm(ConvGemmArgs &a, bool check, hipStream_t st) {
  int ci = g_force_cfg >= 0 ? g_force_cfg : pick_cfg(a.M, a.NP);
  const Cfg &c = kCfgs[ci];
  a.nbm = (a.M + c.bm() - 1) / c.bm();
  a.nbn = (a.NP + c.bn() - 1) / c.bn();
  int nblk = a.nbm * a.nbn;
  if (check)
    launch_gemm_cfg<true>(ci, a, nblk, st);
  else
    launch_gemm_cfg<false>(ci, a, nblk, st);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

static void launch_wgrad_cfg(int ci, const WgradArgs &a, dim3 grid, hipStream_t st) {
  dim3 block(256);
  switch (ci) {
    case 0: hipLaunchKernelGGL((conv_wgrad_kernel<2, 2, 2, 2>), grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL((conv_wgrad_kernel<2, 2, 1, 4>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((conv_wgrad_kernel<3, 1, 1, 4>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((conv_wgrad_kernel<1, 2, 2, 2>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 2, 2>), grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 4, 1>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 1, 4>), grid, block, 0, st, a); break;
  }
}

// ---- geometry shared by the three directions ------------------------------------------------
struct Geo {
  int H, W, C, N, FH, FW, FC, K, G, Kg, Ho, Wo, R;
  int sy, sx, pt, pb, pl, pr, dy, dx;
};

static int make_geo(Geo &g, int H, int W, int C, int N, int FH, int FW, int FC, int K, int sy,
                    int sx, int pt, int pb, int pl, int pr, int dy, int dx) {
  if (H <= 0 || W <= 0 || C <= 0 || N <= 0 || FH <= 0 || FW <= 0 || FC <= 0 || K <= 0)
    return fail(XM_EINVAL, "vl_nnconv: empty tensor (X %dx%dx%dx%d, F %dx%dx%dx%d)", H, W, C, N, FH,
                FW, FC, K);
  if (sy < 1 || sx < 1 || dy < 1 || dx < 1 || pt < 0 || pb < 0 || pl < 0 || pr < 0)
    return fail(XM_EINVAL, "vl_nnconv: stride/dilate must be >= 1 and pad >= 0");
  if (C % FC) return fail(XM_EINVAL, "vl_nnconv: size(F,3)=%d does not divide size(X,3)=%d", FC, C);
  int G = C / FC;
  if (K % G) return fail(XM_EINVAL, "vl_nnconv: %d filters not divisible into %d groups", K, G);
  int Ho = out_size(H, pt, pb, FH, dy, sy), Wo = out_size(W, pl, pr, FW, dx, sx);
  if (Ho <= 0 || Wo <= 0)
    return fail(XM_EINVAL, "vl_nnconv: filter (%dx%d, dilate %dx%d) larger than padded input (%dx%d)",
                FH, FW, dy, dx, H + pt + pb, W + pl + pr);
  if (too_big(H, W, C, N) || too_big(Ho, Wo, K, N) || too_big(FH, FW, FC, K))
    return fail(XM_ETOOBIG, "vl_nnconv: tensor with >= 2^31 elements");
  g = Geo{H, W, C, N, FH, FW, FC, K, G, K / G, Ho, Wo, FH * FW * FC, sy, sx, pt, pb, pl, pr, dy, dx};
  return XM_OK;
}

// forward tap table: r = u + FH*(v + FW*c)  ->  {offset in X, u*dy, v*dx}
static const int4 *fwd_taps(const Geo &g, int count) {
  std::vector<int4> t(count);
  for (int r = 0; r < count; ++r) {
    if (r < g.R) {
      int u = r % g.FH, v = (r / g.FH) % g.FW, c = r / (g.FH * g.FW);
      t[r] = make_int4(u * g.dy + g.H * (v * g.dx) + g.H * g.W * c, u * g.dy, v * g.dx, 0);
    } else {
      t[r] = make_int4(0, -(1 << 28), 0, 0);
    }
  }
  return (const int4 *)cached_device_table(t.data(), t.size() * sizeof(int4));
}

static int conv_forward(const float *x, const float *f, const float *b, float *y, const Geo &g,
                        const float *scale, const float *shift, const float *resid, int relu,
                        hipStream_t st) {
  const int Rp = (g.R + kBK - 1) / kBK * kBK;
  const bool need_pad = (g.R % kBK) != 0 || ((uintptr_t)f & 15);
  WsCarver ws;
  if (need_pad) {
    int rc = ws.init(WsCarver::need((size_t)g.K * Rp, 4));
    if (rc) return rc;
  }
  const int4 *taps = fwd_taps(g, Rp);
  if (!taps) return fail(XM_ENOMEM, "vl_nnconv: tap table allocation failed");
  const float *A = f;
  int lda = g.R;
  if (need_pad) {
    float *Ap = ws.take<float>((size_t)g.K * Rp);
    size_t n = (size_t)g.K * Rp;
    hipLaunchKernelGGL(pad_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f, Ap,
                       g.K, g.R, Rp);
    XM_LAUNCH_CHECK();
    A = Ap;
    lda = Rp;
  }
  const bool check = (g.pt | g.pb | g.pl | g.pr) != 0 || Rp != g.R;
  for (int grp = 0; grp < g.G; ++grp) {
    ConvGemmArgs a{};
    a.A = A + (size_t)grp * g.Kg * lda;
    a.X = x + (size_t)grp * g.FC * g.H * g.W;
    a.Y = y + (size_t)grp * g.Kg * g.Ho * g.Wo;
    a.taps = taps;
    a.bias = b ? b + grp * g.Kg : nullptr;
    a.scale = scale ? scale + grp * g.Kg : nullptr;
    a.shift = shift ? shift + grp * g.Kg : nullptr;
    a.resid = resid ? resid + (size_t)grp * g.Kg * g.Ho * g.Wo : nullptr;
    a.relu = relu;
    a.lda = lda;
    a.M = g.Kg;
    a.Rp = Rp;
    a.PI = g.Ho;
    a.PJ = g.Wo;
    a.NP = g.Ho * g.Wo * g.N;
    a.divPIJ = make_fastdiv((uint32_t)(g.Ho * g.Wo));
    a.divPI = make_fastdiv((uint32_t)g.Ho);
    a.gsy = g.sy;
    a.gsx = g.sx;
    a.gh0 = -g.pt;
    a.gw0 = -g.pl;
    a.LimH = g.H;
    a.LimW = g.W;
    a.xSampleStride = g.H * g.W * g.C;
    a.osy = 1;
    a.osx = 1;
    a.oh0 = 0;
    a.ow0 = 0;
    a.OH = g.Ho;
    a.oChanStride = g.Ho * g.Wo;
    a.oSampleStride = g.Ho * g.Wo * g.K;
    int rc = launch_gemm(a, check, st);
    if (rc) return rc;
  }
  return XM_OK;
}

// dX: one implicit GEMM per stride-parity class (a, b) of the input pixels.
static int conv_dgrad(const float *f, const float *dzdy, float *dxo, const Geo &g, hipStream_t st) {
  struct Cls {
    int a, b, u0, ustep, nU, v0, vstep, nV, Rc, Rp, i0, hi0, PI, j0, wi0, PJ;
    size_t aoff;
  };
  std::vector<Cls> cls;
  size_t abytes = 0;
  bool covers_all = true;
  for (int b = 0; b < g.sx; ++b)
    for (int a = 0; a < g.sy; ++a) {
      Cls c{};
      c.a = a;
      c.b = b;
      // taps with (u*dy) % sy == a form an arithmetic progression u0, u0+ustep, ...
      c.u0 = -1;
      c.nU = 0;
      for (int u = 0; u < g.FH; ++u)
        if ((u * g.dy) % g.sy == a) {
          if (c.u0 < 0) c.u0 = u;
          else if (c.nU == 1) c.ustep = u - c.u0;
          ++c.nU;
        }
      c.v0 = -1;
      c.nV = 0;
      for (int v = 0; v < g.FW; ++v)
        if ((v * g.dx) % g.sx == b) {
          if (c.v0 < 0) c.v0 = v;
          else if (c.nV == 1) c.vstep = v - c.v0;
          ++c.nV;
        }
      if (c.nU <= 1) c.ustep = 1;
      if (c.nV <= 1) c.vstep = 1;
      c.i0 = g.pt > a ? (g.pt - a + g.sy - 1) / g.sy : 0;
      c.hi0 = g.sy * c.i0 + a - g.pt;
      c.PI = c.hi0 < g.H ? (g.H - c.hi0 + g.sy - 1) / g.sy : 0;
      c.j0 = g.pl > b ? (g.pl - b + g.sx - 1) / g.sx : 0;
      c.wi0 = g.sx * c.j0 + b - g.pl;
      c.PJ = c.wi0 < g.W ? (g.W - c.wi0 + g.sx - 1) / g.sx : 0;
      if (c.PI <= 0 || c.PJ <= 0) continue;
      if (c.nU == 0 || c.nV == 0) {
        covers_all = false;
        continue;
      }
      c.Rc = c.nU * c.nV * g.Kg;
      c.Rp = (c.Rc + kBK - 1) / kBK * kBK;
      c.aoff = abytes;
      abytes += WsCarver::need((size_t)g.FC * c.Rp * g.G, 4);
      cls.push_back(c);
    }
  // pixels whose class has no tap (e.g. 1x1 stride 2) receive no gradient
  if (!covers_all || cls.empty())
    XM_HIP(hipMemsetAsync(dxo, 0, sizeof(float) * (size_t)g.H * g.W * g.C * g.N, st));
  if (cls.empty()) return XM_OK;
  WsCarver ws;
  int rc = ws.init(abytes);
  if (rc) return rc;
  for (const Cls &c : cls) {
    // tap table in dY space: r' = iu + nU*(iv + nV*k); u' = (u*dy - a)/sy; ho = i' - u'
    std::vector<int4> t(c.Rp);
    for (int r = 0; r < c.Rp; ++r) {
      if (r < c.Rc) {
        int iu = r % c.nU, iv = (r / c.nU) % c.nV, k = r / (c.nU * c.nV);
        int up = ((c.u0 + iu * c.ustep) * g.dy - c.a) / g.sy;
        int vp = ((c.v0 + iv * c.vstep) * g.dx - c.b) / g.sx;
        t[r] = make_int4(-up - g.Ho * vp + g.Ho * g.Wo * k, -up, -vp, 0);
      } else {
        t[r] = make_int4(0, -(1 << 28), 0, 0);
      }
    }
    const int4 *taps = (const int4 *)cached_device_table(t.data(), t.size() * sizeof(int4));
    if (!taps) return fail(XM_ENOMEM, "vl_nnconv: tap table allocation failed");
    float *At = (float *)(ws.base + c.aoff);
    for (int grp = 0; grp < g.G; ++grp) {
      float *Ag = At + (size_t)grp * g.FC * c.Rp;
      size_t n = (size_t)g.FC * c.Rp;
      hipLaunchKernelGGL(prep_dgrad_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                         st, f + (size_t)g.R * g.Kg * grp, Ag, g.FH, g.FW, g.FC, g.Kg, c.u0, c.ustep,
                         c.nU, c.v0, c.vstep, c.nV, c.Rp);
      XM_LAUNCH_CHECK();
      ConvGemmArgs a{};
      a.A = Ag;
      a.lda = c.Rp;
      a.X = dzdy + (size_t)grp * g.Kg * g.Ho * g.Wo;
      a.Y = dxo + (size_t)grp * g.FC * g.H * g.W;
      a.taps = taps;
      a.M = g.FC;
      a.Rp = c.Rp;
      a.PI = c.PI;
      a.PJ = c.PJ;
      a.NP = c.PI * c.PJ * g.N;
      a.divPIJ = make_fastdiv((uint32_t)(c.PI * c.PJ));
      a.divPI = make_fastdiv((uint32_t)c.PI);
      a.gsy = 1;
      a.gsx = 1;
      a.gh0 = c.i0;
      a.gw0 = c.j0;
      a.LimH = g.Ho;
      a.LimW = g.Wo;
      a.xSampleStride = g.Ho * g.Wo * g.K;
      a.osy = g.sy;
      a.osx = g.sx;
      a.oh0 = c.hi0;
      a.ow0 = c.wi0;
      a.OH = g.H;
      a.oChanStride = g.H * g.W;
      a.oSampleStride = g.H * g.W * g.C;
      rc = launch_gemm(a, true, st);
      if (rc) return rc;
    }
  }
  return XM_OK;
}

static int conv_wgrad(const float *x, const float *dzdy, float *dfo, const Geo &g, hipStream_t st) {
  int ci = g_force_cfg >= 0 ? g_force_cfg : pick_cfg(g.Kg, g.R);
  const Cfg &c = kCfgs[ci];
  const int NP = g.Ho * g.Wo * g.N;
  const int nkt = (NP + kBK - 1) / kBK;
  const int nbm = (g.Kg + c.bm() - 1) / c.bm(), nbn = (g.R + c.bn() - 1) / c.bn();
  const int Rn = nbn * c.bn();
  // split the pixel reduction until the grid fills the chip about twice
  int tiles = nbm * nbn;
  int splits = std::max(1, std::min(nkt / 8, (512 + tiles - 1) / tiles));
  splits = std::min(splits, 64);
  int tps = (nkt + splits - 1) / splits;
  splits = (nkt + tps - 1) / tps;
  const int4 *taps = fwd_taps(g, Rn);
  if (!taps) return fail(XM_ENOMEM, "vl_nnconv: tap table allocation failed");
  size_t slab = (size_t)g.Kg * g.R;
  WsCarver ws;
  float *part = nullptr;
  if (splits > 1) {
    int rc = ws.init(WsCarver::need(slab * splits, 4));
    if (rc) return rc;
    part = ws.take<float>(slab * splits);
  }
  for (int grp = 0; grp < g.G; ++grp) {
    WgradArgs a{};
    a.dY = dzdy + (size_t)grp * g.Kg * g.Ho * g.Wo;
    a.X = x + (size_t)grp * g.FC * g.H * g.W;
    float *dst = dfo + (size_t)grp * g.Kg * g.R;
    a.out = splits > 1 ? part : dst;
    a.taps = taps;
    a.M = g.Kg;
    a.R = g.R;
    a.Rn = Rn;
    a.ldo = g.R;
    a.Ho = g.Ho;
    a.Wo = g.Wo;
    a.NP = NP;
    a.divHW = make_fastdiv((uint32_t)(g.Ho * g.Wo));
    a.divHo = make_fastdiv((uint32_t)g.Ho);
    a.sy = g.sy;
    a.sx = g.sx;
    a.pt = g.pt;
    a.pl_ = g.pl;
    a.H = g.H;
    a.W = g.W;
    a.xSampleStride = g.H * g.W * g.C;
    a.dyChanStride = g.Ho * g.Wo;
    a.dySampleStride = g.Ho * g.Wo * g.K;
    a.nbm = nbm;
    a.nbn = nbn;
    a.tilesPerSplit = tps;
    a.nkt = nkt;
    a.splitStride = slab;
    launch_wgrad_cfg(ci, a, dim3(nbm * nbn, splits), st);
    XM_LAUNCH_CHECK();
    if (splits > 1) {
      hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st,
                         part, dst, g.Kg, g.R, g.R, splits, slab);
      XM_LAUNCH_CHECK();
    }
  }
  return XM_OK;
}

}  // namespace xm

using namespace xm;

extern "C" {

int xm_debug_force_conv_cfg(int cfg) {
  int old = g_force_cfg;
  g_force_cfg = (cfg >= 0 && cfg < kNumCfg) ? cfg : -1;
  return old;
}
int xm_debug_num_conv_cfgs(void) { return kNumCfg; }

int xm_nnconv_forward_fused(const float *x, int H, int W, int C, int N, const float *f, int FH,
                            int FW, int FC, int K, const float *b, float *y, int sy, int sx,
                            int pt, int pb, int pl, int pr, int dy, int dx, const float *scale,
                            const float *shift, const float *residual, int flags, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, C, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx);
  if (rc) return rc;
  if (!x || !f || !y) return fail(XM_EINVAL, "vl_nnconv: NULL tensor");
  if ((scale == nullptr) != (shift == nullptr))
    return fail(XM_EINVAL, "vl_nnconv(fused): scale and shift must be given together");
  return conv_forward(x, f, b, y, g, scale, shift, residual, (flags & XM_FUSE_RELU) ? 1 : 0,
                      (hipStream_t)stream);
}

int xm_nnconv_forward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                      int FC, int K, const float *b, float *y, int sy, int sx, int pt, int pb,
                      int pl, int pr, int dy, int dx, void *stream) {
  return xm_nnconv_forward_fused(x, H, W, C, N, f, FH, FW, FC, K, b, y, sy, sx, pt, pb, pl, pr, dy,
                                 dx, nullptr, nullptr, nullptr, 0, stream);
}

int xm_nnconv_backward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                       int FC, int K, const float *dzdy, float *dx_out, float *df_out,
                       float *db_out, int sy, int sx, int pt, int pb, int pl, int pr, int dy,
                       int dx, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, C, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx);
  if (rc) return rc;
  if (!dzdy) return fail(XM_EINVAL, "vl_nnconv: DZDY is NULL in backward mode");
  hipStream_t st = (hipStream_t)stream;
  if (db_out) {
    hipLaunchKernelGGL(bias_grad_kernel, dim3(K), dim3(256), 0, st, dzdy, db_out, g.Ho * g.Wo, K, N);
    XM_LAUNCH_CHECK();
  }
  if (df_out) {
    if (!x) return fail(XM_EINVAL, "vl_nnconv: X is NULL but DZDF requested");
    rc = conv_wgrad(x, dzdy, df_out, g, st);
    if (rc) return rc;
  }
  if (dx_out) {
    if (!f) return fail(XM_EINVAL, "vl_nnconv: F is NULL but DZDX requested");
    rc = conv_dgrad(f, dzdy, dx_out, g, st);
    if (rc) return rc;
  }
  return XM_OK;
}
}
