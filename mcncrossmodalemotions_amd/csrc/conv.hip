// vl_nnconv forward / backward on gfx950 -- host-side planning + C ABI.
// Replaces MatConvNet's vl_nnconv MEX (matlab/src/vl_nnconv.cu, bits/nnconv.cu: im2row + SGEMM
// per image) behind the same operator contract; see include/xmodal.h and conv_kernels.h.
#include <dlfcn.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "conv_kernels.h"
#include "stem_pool_kernels.h"

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
// A (c <-> k) transpose of T-float elements through LDS: reads are contiguous runs of TS*T floats
// per output channel, writes contiguous runs of TS*nT floats per input channel.
// index arithmetic through magic-number division: with runtime `/` and `%` the kernel was bound by the integer
// division sequences (5 per element), 20 us per launch instead of the few us the bytes take
struct PrepDiv {
  FastDiv tsT, T, tsNT, nT, nU, padc, rows;
};
__global__ void __launch_bounds__(256)
prep_dgrad_filter_kernel(const float *__restrict__ f, float *__restrict__ o, int FH, int FW, int FC,
                         int Kg, int u0, int ustep, int nU, int v0, int vstep, int nV, int lda,
                         int TS, int fold, PrepDiv dv) {
  extern __shared__ float tile[];  // [kl][cl][t], kl pitch TS*T + 1
  const int T = FH * FW, nT = nU * nV;
  const int c0 = blockIdx.x * TS, k0 = blockIdx.y * TS;
  const int pitch = TS * T + 1;
  for (int i = threadIdx.x; i < TS * TS * T; i += 256) {
    int kl = (int)xm_div((uint32_t)i, dv.tsT), rem = i - kl * (TS * T);
    int cl = (int)xm_div((uint32_t)rem, dv.T);
    int c = c0 + cl, k = k0 + kl;
    tile[kl * pitch + rem] = (c < FC && k < Kg) ? f[(size_t)rem + (size_t)T * (c0 + (size_t)FC * k)] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TS * TS * nT; i += 256) {
    int cl = (int)xm_div((uint32_t)i, dv.tsNT), rem = i - cl * (TS * nT);
    int kl = (int)xm_div((uint32_t)rem, dv.nT), tt = rem - kl * nT;
    int iv = (int)xm_div((uint32_t)tt, dv.nU), iu = tt - iv * nU;
    int t = (u0 + iu * ustep) + FH * (v0 + iv * vstep);
    int c = c0 + cl, k = k0 + kl;
    if (c < FC && k < Kg) {
      // fold: one GEMM row per (filter row u, channel c); the reduction keeps only (iv, k)
      size_t dst = fold ? (size_t)(iu + nU * c) * lda + (size_t)nV * k + iv
                        : (size_t)c * lda + (size_t)nT * k + tt;
      o[dst] = tile[kl * pitch + cl * T + t];
    }
  }
  if (blockIdx.y == 0) {  // zero the K padding columns
    const int used = (fold ? nV : nT) * Kg, rows = fold ? nU : 1;
    int padc = lda - used;
    for (int i = threadIdx.x; i < TS * rows * padc; i += 256) {
      int rl = (int)xm_div((uint32_t)i, dv.padc), rr = used + (i - rl * padc);
      int cl = (int)xm_div((uint32_t)rl, dv.rows), ul = rl - cl * rows;
      if (c0 + cl < FC) o[((size_t)(c0 + cl) * rows + ul) * lda + rr] = 0.f;
    }
  }
}

// The FC-shaped layers' dgrad operand is a plain matrix transpose: At[r][k] = F[k][r] for a 1 x 1 filter (r = c) and for
// an H-collapsing FH x 1 filter folded into the GEMM rows (r = u + FH c) -- 110 of the 130 MB the student's filter
// preparation moves per step (fc6: 4096 x 2304, fc7: 1024 x 4096).  64 x 64 tiles through LDS, 16-byte accesses on both
// sides (the generic kernel above walks T-float runs element by element: ~1 TB/s).
__global__ void __launch_bounds__(256)
transpose_filter_kernel(const float *__restrict__ f, float *__restrict__ o, int K, int R, int ldo) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 float4 columns x 16 rows per pass
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int k = k0 + ty + 16 * p, r = r0 + 4 * tx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K && r < R) v = *reinterpret_cast<const float4 *>(f + (size_t)k * R + r);   // R % 4 == 0
    tile[ty + 16 * p][4 * tx + 0] = v.x;
    tile[ty + 16 * p][4 * tx + 1] = v.y;
    tile[ty + 16 * p][4 * tx + 2] = v.z;
    tile[ty + 16 * p][4 * tx + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = r0 + ty + 16 * p, k = k0 + 4 * tx;
    if (r < R && k < K) {                                                                // K % 4 == 0
      float4 v = make_float4(tile[4 * tx + 0][ty + 16 * p], tile[4 * tx + 1][ty + 16 * p],
                             tile[4 * tx + 2][ty + 16 * p], tile[4 * tx + 3][ty + 16 * p]);
      *reinterpret_cast<float4 *>(o + (size_t)r * ldo + k) = v;
    }
  }
}

// out[i] = sum_z part[z][i] in a FIXED order (deterministic): ZL "z-lanes" per output each sum a
// strided subset of the splits with four independent accumulators (loads in flight instead of one
// dependent chain), then the lanes are combined through LDS in lane order.
template <int ZL>
__global__ void __launch_bounds__(256)
reduce_splits_kernel(const float *__restrict__ part, float *__restrict__ out, size_t total, int splits,
                     size_t splitStride) {
  constexpr int OPB = 256 / ZL;
  const int ol = threadIdx.x % OPB, zl = threadIdx.x / OPB;
  const size_t i = blockIdx.x * (size_t)OPB + ol;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < total) {
    const float *p = part + i;
    int z = zl;
    for (; z + 3 * ZL < splits; z += 4 * ZL) {
      s0 += p[(size_t)z * splitStride];
      s1 += p[(size_t)(z + ZL) * splitStride];
      s2 += p[(size_t)(z + 2 * ZL) * splitStride];
      s3 += p[(size_t)(z + 3 * ZL) * splitStride];
    }
    for (; z < splits; z += ZL) s0 += p[(size_t)z * splitStride];
  }
  float s = (s0 + s1) + (s2 + s3);
  if (ZL == 1) {
    if (i < total) out[i] = s;
    return;
  }
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (zl == 0 && i < total) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < ZL; ++k) t += red[k * OPB + ol];
    out[i] = t;
  }
}

static void launch_reduce_splits(const float *part, float *out, size_t total, int splits,
                                 size_t splitStride, hipStream_t st) {
  if (splits >= 64)
    hipLaunchKernelGGL(reduce_splits_kernel<16>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st,
                       part, out, total, splits, splitStride);
  else if (splits >= 8)
    hipLaunchKernelGGL(reduce_splits_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st,
                       part, out, total, splits, splitStride);
  else
    hipLaunchKernelGGL(reduce_splits_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       part, out, total, splits, splitStride);
}

// dzdb(k) = sum over pixels and samples of dzdy(:,:,k,:): grid (K, S) partial sums, then a
// fixed-order finalize (deterministic)
__global__ void __launch_bounds__(256)
bias_grad_partial_kernel(const float *__restrict__ dy, double *__restrict__ part, int HW, int K, int N,
                         int S, FastDiv divRun) {
  int k = blockIdx.x, sp = blockIdx.y;
  double s = 0.0;
  const bool vec = (HW & 3) == 0;
  // flat (sample, position) index: FC-shaped layers (H*W = 8) keep all lanes busy
  const int nper = (N - sp + S - 1) / S;
  const int run = vec ? HW >> 2 : HW;
  for (int j = threadIdx.x; j < nper * run; j += 256) {
    const int nn = (int)xm_div((uint32_t)j, divRun), i = j - nn * run;
    const float *p = dy + (size_t)HW * (k + (size_t)K * (sp + nn * S));
    if (vec) {
      float4 v = reinterpret_cast<const float4 *>(p)[i];
      s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    } else {
      s += (double)p[i];
    }
  }
  __shared__ double red[4];
  s = xm_wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)k * S + sp] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256)
bias_grad_finalize_kernel(const double *__restrict__ part, float *__restrict__ db, int K, int S) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // one wave per channel
  if (k >= K) return;
  double s = 0.0;
  for (int i = lane; i < S; i += 64) s += part[(size_t)k * S + i];
  s = xm_wave_sum_d(s);
  if (lane == 0) db[k] = (float)s;
}

// batch moments from the convolution epilogue's partial sums (ConvGemmArgs::statPart, [pixel tile][row] pairs of fp32
// {sum, sum of squares} over <= 256 values each): fp64 accumulation over the pixel tiles in a fixed order.
// mom = [mean, sqrt(var + eps)] (M x 2).  Block = 8 channels x 32 tile lanes (a wave reads eight runs of 8 consecutive
// pairs): with 32 channels per block a 96-channel layer ran on THREE blocks whose lanes each walked ncg / 8 dependent
// loads -- 21-34 us for 0.5-2 MB.  grid (ceil(M / 8), S): S > 1 splits the tiles, the partial fp64 sums go to
// part2[s][M][2] and conv_stats_finalize2_kernel adds them.
__global__ void __launch_bounds__(256)
conv_stats_reduce_kernel(const float *__restrict__ part, float *__restrict__ mom, double *__restrict__ part2, int M,
                         int ncg, int S, double m, float eps) {
  const int cl = threadIdx.x & 7, j = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl, s = blockIdx.y;
  const int chunk = (ncg + S - 1) / S, g0 = s * chunk, g1 = min(ncg, g0 + chunk);
  double a = 0.0, b = 0.0;
  if (c < M)
    for (int g = g0 + j; g < g1; g += 32) {
      const float2 v = *reinterpret_cast<const float2 *>(part + ((size_t)g * M + c) * 2);
      a += (double)v.x;
      b += (double)v.y;
    }
  __shared__ double ra[32][8], rb[32][8];
  ra[j][cl] = a;
  rb[j][cl] = b;
  __syncthreads();
  if (j == 0 && c < M) {
#pragma unroll
    for (int k = 1; k < 32; ++k) a += ra[k][cl], b += rb[k][cl];
    if (S == 1) {
      const double mu = a / m;
      double var = b / m - mu * mu;
      var = var < 0.0 ? 0.0 : var;
      mom[c] = (float)mu;
      mom[M + c] = (float)sqrt(var + (double)eps);
    } else {
      part2[2 * ((size_t)s * M + c)] = a;
      part2[2 * ((size_t)s * M + c) + 1] = b;
    }
  }
}
__global__ void __launch_bounds__(256)
conv_stats_finalize2_kernel(const double *__restrict__ part2, float *__restrict__ mom, int M, int S, double m, float eps) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= M) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < S; ++s) {
    a += part2[2 * ((size_t)s * M + c)];
    b += part2[2 * ((size_t)s * M + c) + 1];
  }
  const double mu = a / m;
  double var = b / m - mu * mu;
  var = var < 0.0 ? 0.0 : var;
  mom[c] = (float)mu;
  mom[M + c] = (float)sqrt(var + (double)eps);
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
    // EIGHT waves per block: the 128 x 128 tile by waves of 32 x 64 -- 128 VGPRs, so two blocks per CU are four waves per
    // SIMD instead of two: +3 ... 4 % on the long-reduction forward / dgrad layers (the filter derivative is
    // 2 % slower with it, 64 x 256 by eight waves is no faster than 64 x 128 by four, 96 x 256 by waves of 96 x 32 spills,
    // 96 x 128 by six waves stages 512 gather units with 384 threads: 75 against 117 TFLOP/s)
    {1, 2, 4, 2},  //  7
    // LDS-DMA variants (conv_gemm_dma_kernel): forward / dgrad GEMMs of 1x1 unit-stride layers only
    {2, 2, 2, 2},  //  8: 128 x 128, 3 LDS stages
    {2, 2, 1, 4},  //  9:  64 x 256, 3 stages
    {1, 2, 2, 2},  // 10:  64 x 128, 4 stages
    {1, 1, 2, 2},  // 11:  64 x  64, 4 stages
};
constexpr int kNumCfg = sizeof(kCfgs) / sizeof(kCfgs[0]);
constexpr int kNumBaseCfg = 8;                       // configurations every forward / dgrad geometry can run
constexpr int kCfgW8 = 7;                            // the eight-wave configuration (XM_NO_W8 runs configuration 0 in its place)
constexpr int kNumWgradCfg = 7;                      // ... and the filter derivative (four-wave configurations only)
static const int kDmaBase[] = {0, 1, 3, 4};          // the register-staged configuration of the same tile shape
static inline bool is_dma_cfg(int ci) { return ci >= kNumBaseCfg; }
static inline int base_cfg(int ci) { return is_dma_cfg(ci) ? kDmaBase[ci - kNumBaseCfg] : ci; }
static inline int wgrad_cfg(int ci) { ci = base_cfg(ci); return ci >= kNumWgradCfg ? 0 : ci; }
// The eight-wave configuration is a candidate for launches of at least XM_W8_MIN_TILES 128 x 128 tiles (default: two rounds
// of the chip).  Measured (profiles/r04/schedule_experiments.txt): alone it is 3 ... 4 % faster at every size, but next to the
// filter derivatives of the side stream the shorter launches lose more than that (student step at 64 spectrograms - 3 %).
static inline bool w8_ok(long long M, long long NP) {
  static const long long min_tiles = env_int("XM_W8_MIN_TILES", 1024);
  return path_on(kPathW8) && ((M + 127) / 128) * ((NP + 127) / 128) >= min_tiles;
}
static inline unsigned w8_skip(long long M, long long NP) { return w8_ok(M, NP) ? 0u : 1u << 7; }
static inline unsigned cfg_threads(int ci) { return 64u * kCfgs[ci].wgm * kCfgs[ci].wgn; }
// analytic model: relative cost per multiply-add of a configuration (32 x 32 MFMA tiles per block; measured ordering)
static inline double cfg_eff(const Cfg &c) {
  const int t = c.tm * c.tn * c.wgm * c.wgn;
  return c.wgm * c.wgn == 8 ? 0.97 : (t >= 16 ? 1.0 : (t >= 8 ? 1.12 : 1.3));
}

static int g_force_splits = 0;  // test hook

// With few output tiles the reduction is split over grid.y; the split count fills ONE round of
// 2 co-resident blocks per CU (no tail round).
static int pick_splits(int tiles, int nkt) {
  if (g_force_splits > 0) return std::max(1, std::min(g_force_splits, nkt));
  if (tiles >= 384 || nkt < 16) return 1;
  int s = 512 / tiles;
  s = std::min(s, nkt / 8);  // keep >= 8 stages per split
  return std::max(1, std::min(s, 64));
}

// Time model per tile configuration (seconds, coarse):
//   block time  = 2*BM*BN*16*stages*eff / (157.3 TF / 256 CUs)    (a CU's MFMA pipes are shared by
//                 its co-resident blocks, so >256 blocks cost proportionally more)
//   rounds      = 1 if blocks <= 256 else 2*ceil(blocks/512)
//   split-K adds the slab write + combine pass at HBM speed
static int pick_cfg(long long M, long long NP, int nkt) {
  double best = 1e300;
  int bi = 0;
  for (int i = 0; i < kNumBaseCfg; ++i) {
    if (i == kCfgW8 && !w8_ok(M, NP)) continue;
    const Cfg &c = kCfgs[i];
    long long tiles = ((M + c.bm() - 1) / c.bm()) * ((NP + c.bn() - 1) / c.bn());
    int s = pick_splits((int)std::min<long long>(tiles, 1 << 20), nkt);
    double eff = cfg_eff(c);
    double stages = (double)((nkt + s - 1) / s) + 3.0;  // + prologue / epilogue
    double t_block = 2.0 * c.bm() * c.bn() * 16.0 * stages * eff / (157.3e12 / 256.0);
    double blocks = (double)tiles * s;
    double rounds = blocks <= 256 ? 1.0 : 2.0 * std::ceil(blocks / 512.0);
    double t = rounds * t_block;
    if (s > 1) t += 5e-6 + (double)(s + 1) * M * NP * 4.0 / 4e12;
    if (t < best) {
      best = t;
      bi = i;
    }
  }
  return bi;
}

#ifndef XM_DMA_NST7
#define XM_DMA_NST7 4
#endif
#ifndef XM_DMA_NST9
#define XM_DMA_NST9 4
#endif
#ifndef XM_DMA_NST10
#define XM_DMA_NST10 4
#endif
#ifndef XM_DMA_PERCU
#define XM_DMA_PERCU {2, 2, 4, 4}
#endif
static void launch_gemm_dma(int ci, const ConvGemmArgs &a, dim3 grid, hipStream_t st) {
  dim3 block(256);
  switch (ci) {
    case kNumBaseCfg + 0: hipLaunchKernelGGL((conv_gemm_dma_kernel<2, 2, 2, 2, XM_DMA_NST7>), grid, block, 0, st, a); break;
    case kNumBaseCfg + 1: hipLaunchKernelGGL((conv_gemm_dma_kernel<2, 2, 1, 4, 3>), grid, block, 0, st, a); break;
    case kNumBaseCfg + 2: hipLaunchKernelGGL((conv_gemm_dma_kernel<1, 2, 2, 2, XM_DMA_NST9>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((conv_gemm_dma_kernel<1, 1, 2, 2, XM_DMA_NST10>), grid, block, 0, st, a); break;
  }
}

template <int MODE>
static void launch_gemm_cfg(int ci, const ConvGemmArgs &a, dim3 grid, hipStream_t st) {
  dim3 block(cfg_threads(ci));
  switch (ci) {
    case 7: hipLaunchKernelGGL((conv_gemm_kernel<1, 2, 4, 2, MODE>), grid, block, 0, st, a); break;
    case 0: hipLaunchKernelGGL((conv_gemm_kernel<2, 2, 2, 2, MODE>), grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL((conv_gemm_kernel<2, 2, 1, 4, MODE>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((conv_gemm_kernel<3, 1, 1, 4, MODE>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((conv_gemm_kernel<1, 2, 2, 2, MODE>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((conv_gemm_kernel<1, 1, 2, 2, MODE>), grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL((conv_gemm_kernel<1, 1, 4, 1, MODE>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((conv_gemm_kernel<1, 1, 1, 4, MODE>), grid, block, 0, st, a); break;
  }
}

static int g_force_cfg = -1;  // test hook (xm_debug_force_conv_cfg)
static int g_force_stem = -1; // test hook (xm_debug_force_conv_stem): 1 = conv_stem_kernel wherever it can run, 0 = never
static int g_force_wgrad_patch = -1; // test hook (xm_debug_force_wgrad_patch)
// Execution hint of the HOST (xm_set_exec_hint, include/xmodal.h): XM_EXEC_SINGLE_STREAM = every operator call of the
// process arrives on one stream, so a kernel has the chip to itself -- conv_wgrad_patch_kernel (48 KB of LDS per block,
// three blocks per CU for its whole life) is a candidate only then (next to another stream's kernels it
// loses).  An explicit statement of the caller, never inferred: the kernel a shape gets is a function of (shape, table,
// hint) and of nothing the process did before.
static unsigned g_exec_hint = 0;
static int g_force_halo = -1; // test hook (xm_debug_force_conv_halo): 1 = halo-patch kernel wherever it can run, 0 = never
static unsigned long long *g_dbg_cycles = nullptr;  // device buffer, set by xm_debug_conv_cycles(1)

// ---- per-kernel timing with HIP events on the launch stream (bench.py roofline leg) --------
struct ProfRec {
  hipEvent_t start, stop;
  int key;  // kind * 100 + cfg * 2 + check
  double flops;
  double bytes;  // algorithmic HBM bytes of the launch: every operand once (x + f + y [+ residual])
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t prof_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
struct ProfScope {
  bool on;
  ProfRec r;
  hipStream_t st;
  ProfScope(int key, double flops, hipStream_t s, double bytes = 0) : on(g_prof_on), st(s) {
    if (!on) return;
    r.key = key;
    r.flops = flops;
    r.bytes = bytes;
    r.start = prof_event();
    r.stop = prof_event();
    (void)hipEventRecord(r.start, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.stop, st);
    g_prof.push_back(r);
  }
};


// co-resident blocks per CU of the register-staged configurations (VGPR-limited: 222 / 232 / 186 / 140 / 116 / 134 / 122 /
// 128 with eight waves)
static const int kCfgOcc[kNumBaseCfg] = {2, 2, 2, 3, 4, 3, 4, 2};

// Hybrid schedule (ConvGemmArgs::hyS): a launch of more than one round of the chip whose LAST round is partly filled
// computes the full rounds tile by tile and splits the remaining tiles along the reduction so that they fill the last
// round with short blocks -- 3.06 rounds cost 3.06 + 1/S instead of 4.  Returns the split count of the remainder (1: plain).
static int hybrid_plan(const ConvGemmArgs &a, int ci, int *full_out) {
  const bool off = !path_on(kPathHybrid);
  *full_out = 0;
  if (off || is_dma_cfg(ci) || g_force_splits > 0) return 1;
  if (a.statPart && (!a.vecStore || a.relu || a.resid)) return 1;   // conv_splitk_epilogue_stats_kernel's case only
  const int tiles = a.nbm * a.nbn, slots = 256 * kCfgOcc[ci];
  if (tiles <= slots || a.nkt < 16) return 1;
  int full = tiles / slots * slots;
  full -= full % a.nbm;                       // the remainder is a whole range of pixel tiles
  const int rem = tiles - full;
  if (rem <= 0 || rem * 5 > slots * 4) return 1;     // a last round that is > 80 % full is left alone
  const int S = std::min(std::min(slots / rem, a.nkt / 8), 16);
  if (S < 2) return 1;
  *full_out = full;
  return S;
}

static size_t gemm_slab_floats(const ConvGemmArgs &a, int ci, int *splits_out) {
  const Cfg &c = kCfgs[ci];
  int nbm = (a.M + c.bm() - 1) / c.bm(), nbn = (a.NP + c.bn() - 1) / c.bn();
  int splits = pick_splits(nbm * nbn, a.Rp / kBK);
  *splits_out = splits;
  size_t need = splits > 1 ? (size_t)splits * a.M * ((a.NP + 3) & ~3) : 0;
  if (splits == 1 && !is_dma_cfg(ci)) {
    // hybrid remainder: at most one round of tiles, up to 16 partial copies of each
    ConvGemmArgs t = a;
    t.nbm = nbm, t.nbn = nbn, t.nkt = a.Rp / kBK;
    t.statPart = nullptr;
    int full;
    const int S = hybrid_plan(t, ci, &full);
    if (S > 1) need = std::max(need, (size_t)S * a.M * ((size_t)(nbm * nbn - full) / nbm * c.bn() + 4));
  }
  return need;
}

static int launch_gemm(ConvGemmArgs &a, int mode, int ci, int splits, float *slab, hipStream_t st) {
  if (is_dma_cfg(ci) && !a.dmaOk) ci = base_cfg(ci);   // forced configuration on a geometry it cannot run
  if (ci == kCfgW8 && !path_on(kPathW8)) ci = 0;
  const Cfg &c = kCfgs[ci];
  a.nbm = (a.M + c.bm() - 1) / c.bm();
  a.nbn = (a.NP + c.bn() - 1) / c.bn();
  a.nkt = a.Rp / kBK;
  a.tilesPerSplit = (a.nkt + splits - 1) / splits;
  splits = (a.nkt + a.tilesPerSplit - 1) / a.tilesPerSplit;
  a.NPs = (a.NP + 3) & ~3;
  a.slab = splits > 1 ? slab : nullptr;
  a.dbgCycles = g_dbg_cycles;
  a.hyS = 1, a.hyFull = 0, a.hyTps = 0, a.hyP0 = 0;
  dim3 grid(a.nbm * a.nbn, splits);
  int hyS = 1;
  if (splits == 1 && slab) {
    int full;
    hyS = hybrid_plan(a, ci, &full);
    if (hyS > 1) {
      a.hyFull = full;
      a.hyTps = (a.nkt + hyS - 1) / hyS;
      hyS = (a.nkt + a.hyTps - 1) / a.hyTps;
      a.hyS = hyS;
      a.hyP0 = full / a.nbm * c.bn();
      a.NPs = (a.NP - a.hyP0 + 3) & ~3;
      a.slab = slab;
      grid = dim3(full + (a.nbm * a.nbn - full) * hyS, 1);
    }
  }
  {
    const double abytes = (double)a.xBytes + 4.0 * a.M * a.Rtrue + 4.0 * a.M * (double)a.NP * (a.resid ? 2 : 1);
    ProfScope ps(0 * 100 + ci * 2 + mode, a.algoFlops > 0 ? a.algoFlops : 2.0 * a.M * (double)a.NP * a.Rtrue, st,
                 abytes);
    if (is_dma_cfg(ci)) {
      // persistent: one round of co-resident blocks (a multiple of 8 so that every XCD gets the same share)
      static const int per_cu_tab[] = XM_DMA_PERCU;          // co-resident blocks per CU (LDS-limited)
      const int slots = std::max(8, 256 * per_cu_tab[ci - kNumBaseCfg] / std::max(1, splits));
      const int ntl = a.nbm * a.nbn;
      const int g = ntl <= slots ? ntl : slots - slots % 8;
      launch_gemm_dma(ci, a, dim3(std::max(1, g), splits), st);
    } else if (mode)
      launch_gemm_cfg<1>(ci, a, grid, st);
    else
      launch_gemm_cfg<0>(ci, a, grid, st);
  }
  XM_LAUNCH_CHECK();
  if (splits > 1 || hyS > 1) {
    const int z = splits > 1 ? splits : hyS;
    const int np = a.NP - a.hyP0;                       // pixels the slabs cover
    const bool vec = a.vecStore && (np & 3) == 0 && (a.hyP0 & 3) == 0 && (a.NPs & 3) == 0 && ((uintptr_t)a.slab & 15) == 0;
    const size_t n = (size_t)a.M * (vec ? np / 4 : np);
    if (hyS > 1 && a.statPart) {
      // the full rounds left their partial sums per pixel tile; the remainder's come from the combine kernel
      const int cols = np / 4, chunks = (cols + 255) / 256, statBase = a.hyFull / a.nbm;
      if (!vec) return fail(XM_EINVAL, "vl_nnconv: hybrid schedule with statistics needs vector stores");
      hipLaunchKernelGGL(conv_splitk_epilogue_stats_kernel, dim3(chunks, a.M), dim3(256), 0, st, a, z, cols, statBase);
      a.statNcg = statBase + chunks;
    } else if (vec)
      hipLaunchKernelGGL(conv_splitk_epilogue_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                         a, z, make_fastdiv((uint32_t)(np / 4)));
    else
      hipLaunchKernelGGL(conv_splitk_epilogue_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                         a, z, make_fastdiv((uint32_t)np));
    XM_LAUNCH_CHECK();
  }
  return XM_OK;
}

static void launch_gemm_multi_cfg(int ci, const ConvGemmMulti &m, dim3 grid, hipStream_t st) {
  dim3 block(cfg_threads(ci));
  switch (ci) {
    case 7: hipLaunchKernelGGL((conv_gemm_multi_kernel<1, 2, 4, 2>), grid, block, 0, st, m); break;
    case 0: hipLaunchKernelGGL((conv_gemm_multi_kernel<2, 2, 2, 2>), grid, block, 0, st, m); break;
    case 1: hipLaunchKernelGGL((conv_gemm_multi_kernel<2, 2, 1, 4>), grid, block, 0, st, m); break;
    case 2: hipLaunchKernelGGL((conv_gemm_multi_kernel<3, 1, 1, 4>), grid, block, 0, st, m); break;
    case 3: hipLaunchKernelGGL((conv_gemm_multi_kernel<1, 2, 2, 2>), grid, block, 0, st, m); break;
    case 4: hipLaunchKernelGGL((conv_gemm_multi_kernel<1, 1, 2, 2>), grid, block, 0, st, m); break;
    case 5: hipLaunchKernelGGL((conv_gemm_multi_kernel<1, 1, 4, 1>), grid, block, 0, st, m); break;
    default: hipLaunchKernelGGL((conv_gemm_multi_kernel<1, 1, 1, 4>), grid, block, 0, st, m); break;
  }
}

// up to 4 masked (MODE 1) implicit GEMMs without split-K in one launch, same tile configuration
static int launch_gemm_multi(const std::vector<ConvGemmArgs> &args, int ci, hipStream_t st) {
  ci = base_cfg(ci);
  if (ci == kCfgW8 && !path_on(kPathW8)) ci = 0;
  const Cfg &c = kCfgs[ci];
  ConvGemmMulti m{};
  int maxTiles = 0;
  double flops = 0, abytes = args.empty() ? 0.0 : (double)args[0].xBytes;   // the classes share one source tensor
  for (size_t i = 0; i < args.size(); ++i) {
    ConvGemmArgs a = args[i];
    a.nbm = (a.M + c.bm() - 1) / c.bm();
    a.nbn = (a.NP + c.bn() - 1) / c.bn();
    a.nkt = a.Rp / kBK;
    a.tilesPerSplit = a.nkt;
    a.NPs = (a.NP + 3) & ~3;
    a.slab = nullptr;
    maxTiles = std::max(maxTiles, a.nbm * a.nbn);
    flops += a.algoFlops > 0 ? a.algoFlops : 2.0 * a.M * (double)a.NP * a.Rtrue;
    abytes += 4.0 * a.M * a.Rtrue + 4.0 * a.M * (double)a.NP * (a.resid ? 2 : 1);
    m.c[i] = a;
  }
  {
    ProfScope ps(2 * 100 + ci * 2, flops, st, abytes);
    launch_gemm_multi_cfg(ci, m, dim3(maxTiles, 1, (unsigned)args.size()), st);
  }
  XM_LAUNCH_CHECK();
  return XM_OK;
}

// ---- halo-patch kernels (conv_halo_kernel / conv_halo_multi_kernel): <= 3 x 3 taps, unit pixel stride ---------------
// Variants: 0 = 128-row tiles (2 x 2 wave tiles), patch 512;  1 = 96-row tiles (3 x 1), patch 512;
//           2 = 96-row tiles, patch 1024 (tall columns: the student's conv2 dgrad classes have 65-row patch columns)
struct HaloVar {
  int bm, ps;
};
static const HaloVar kHaloVars[] = {{128, 512}, {96, 512}, {96, 1024}};
static const float kHaloMargin = (float)env_double("XM_HALO_MARGIN", 0.04);

// fills the patch geometry of `a` (an implicit-GEMM problem already described by its tap / gather fields) and returns
// the patch floats per channel the widest 128-pixel tile needs (0: the kernel cannot run this problem): taps in an
// nU x nV <= 3 x 3 arrangement of 9 / 6 / 4, adjacent rows / columns, unit pixel stride, channels a multiple of 8.
static int halo_setup(ConvGemmArgs &a, int nSamples) {
  const bool off = !path_on(kPathHalo);
  const int T = a.nU * a.nV;
  if (off || a.nU < 1 || a.nU > 3 || a.nV < 1 || a.nV > 3 || (T != 9 && T != 6 && T != 4)) return 0;
  if (a.gsy != 1 || a.gsx != 1 || std::abs(a.dus) != 1 || std::abs(a.dvs) != 1) return 0;
  const int KS = kHaloCB * T;
  if (a.Rtrue % KS != 0 || a.Rtrue < KS || (a.lda & 3) || ((uintptr_t)a.A & 15)) return 0;
  if (a.divMU.d != 1) return 0;
  const int rlo = a.gh0 + a.du0 + std::min(0, (a.nU - 1) * a.dus), clo = a.gw0 + a.dv0 + std::min(0, (a.nV - 1) * a.dvs);
  a.hpRmin = rlo;
  a.hpCmin = clo;
  a.hpHP = a.PI + a.nU - 1;
  a.hpWP = a.PJ + a.nV - 1;
  a.hpN = nSamples;
  for (int iv = 0; iv < a.nV; ++iv)
    for (int iu = 0; iu < a.nU; ++iu) {
      const int su = a.gh0 + a.du0 + iu * a.dus - rlo, sv = a.gw0 + a.dv0 + iv * a.dvs - clo;
      a.hpSh[iu + a.nU * iv] = (su + a.hpHP * sv) * 4;
    }
  a.hpDivHP = make_fastdiv((uint32_t)a.hpHP);
  a.hpDivWP = make_fastdiv((uint32_t)a.hpWP);
  // widest patch over all 128-pixel tiles: padded columns of the first and the last pixel of the tile + the taps
  const int BN = 128, pij = a.PI * a.PJ;
  int worst = 0;
  for (long long p0 = 0; p0 < a.NP; p0 += BN) {
    const long long p1 = std::min<long long>(p0 + BN - 1, a.NP - 1);
    const long long c0 = (p0 / pij) * a.hpWP + (p0 % pij) / a.PI, c1 = (p1 / pij) * a.hpWP + (p1 % pij) / a.PI;
    worst = std::max(worst, (int)(c1 - c0) + a.nV);
  }
  return worst * a.hpHP;
}
// Tall pixel grids (the student's conv2 dgrad classes: 63 / 62 rows) put 3-4 columns + taps + a sample gap under a
// 128-pixel tile -- more than 512 patch floats.  Enumerating the pixels with PI rounded up to a multiple of 32 makes a tile
// a whole number of columns (2 of 64): the patch shrinks below 512 floats and the 96-row / 512 variant (3 blocks per CU,
// half the patch loads) applies; the non-existent rows (1.6 %) are computed and dropped in the epilogue (piReal).
// Only for scalar-store problems whose padding waste stays under 4 %.
static int halo_setup_padded(ConvGemmArgs &a, int nSamples) {
  int need = halo_setup(a, nSamples);
  if (need <= 512 || a.vecStore || a.slab) return need;
  const int piv = (a.PI + 31) / 32 * 32;
  if (piv == a.PI || (piv - a.PI) * 25 > piv || (long long)piv * a.PJ * nSamples >= (1LL << 31)) return need;
  ConvGemmArgs b = a;
  b.piReal = a.PI;
  b.PI = piv;
  b.NP = piv * a.PJ * nSamples;
  b.divPI = make_fastdiv((uint32_t)piv);
  b.divPIJ = make_fastdiv((uint32_t)(piv * a.PJ));
  const int need2 = halo_setup(b, nSamples);
  if (need2 > 0 && need2 <= 512) {
    a = b;
    return need2;
  }
  return need;
}

static bool halo_var_ok(const ConvGemmArgs &a, int need, int v) {
  if (need <= 0 || need > kHaloVars[v].ps) return false;
  if (v == 2 && need <= 512) return false;                       // variant 1 covers it with half the patch loads
  if (kHaloVars[v].bm == 96 && (a.M % 96 != 0)) return false;    // 96-row tiles only where they waste nothing
  return true;
}

// split-K: stages of 8 T reduction steps; one round of 2 blocks per CU, >= 4 stages per split
static int halo_splits(const ConvGemmArgs &a, int v) {
  const int bm = kHaloVars[v].bm;
  const int tiles = ((a.M + bm - 1) / bm) * ((a.NP + 127) / 128), nst = a.Rtrue / (kHaloCB * a.nU * a.nV);
  if (tiles >= 384 || nst < 8) return 1;
  return std::max(1, std::min(std::min(512 / tiles, nst / 4), 16));
}
static size_t halo_slab_floats(const ConvGemmArgs &a) {
  int sp = 1;
  for (int v = 0; v < 3; ++v) sp = std::max(sp, halo_splits(a, v));
  return sp > 1 ? (size_t)sp * a.M * ((a.NP + 3) & ~3) : 0;
}

static void halo_prepare(ConvGemmArgs &a, int v) {
  const int bm = kHaloVars[v].bm;
  a.nbm = (a.M + bm - 1) / bm;
  a.nbn = (a.NP + 127) / 128;
  a.nkt = a.Rtrue / (kHaloCB * a.nU * a.nV);   // stages (the split-K bookkeeping of the kernel counts in these)
  a.tilesPerSplit = a.nkt;
  a.NPs = (a.NP + 3) & ~3;
  a.slab = nullptr;
  a.dbgCycles = nullptr;
}

// test hook: xm_debug_force_conv_halo(1 + v) asks for variant v, any other runnable one if v cannot take the problem
static int forced_halo_variant(const bool *hok) {
  if (g_force_halo >= 1 && g_force_halo <= 3 && hok[g_force_halo]) return g_force_halo;
  return hok[2] ? 2 : (hok[1] ? 1 : 3);
}

static int launch_halo(ConvGemmArgs a, int v, float *slab, hipStream_t st) {
  halo_prepare(a, v);
  int splits = slab ? halo_splits(a, v) : 1;
  a.tilesPerSplit = (a.nkt + splits - 1) / splits;
  splits = (a.nkt + a.tilesPerSplit - 1) / a.tilesPerSplit;
  a.slab = splits > 1 ? slab : nullptr;
  if (a.slab) a.statPart = nullptr;
  {
    const double abytes = (double)a.xBytes + 4.0 * a.M * a.Rtrue + 4.0 * a.M * (double)a.NP * (a.resid ? 2 : 1);
    ProfScope ps(3 * 100 + v, a.algoFlops > 0 ? a.algoFlops : 2.0 * a.M * (double)a.NP * a.Rtrue, st, abytes);
    const dim3 grid(a.nbm * a.nbn, splits), block(256);
    if (v == 0) hipLaunchKernelGGL((conv_halo_kernel<2, 2, 2, 2, 512>), grid, block, 0, st, a);
    else if (v == 1) hipLaunchKernelGGL((conv_halo_kernel<3, 1, 1, 4, 512>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv_halo_kernel<3, 1, 1, 4, 1024>), grid, block, 0, st, a);
  }
  XM_LAUNCH_CHECK();
  if (splits > 1) {
    const bool vec = a.vecStore && (a.NP & 3) == 0 && ((uintptr_t)a.slab & 15) == 0;
    const size_t n = (size_t)a.M * (vec ? a.NP / 4 : a.NP);
    if (vec)
      hipLaunchKernelGGL(conv_splitk_epilogue_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                         a, splits, make_fastdiv((uint32_t)(a.NP / 4)));
    else
      hipLaunchKernelGGL(conv_splitk_epilogue_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                         a, splits, make_fastdiv((uint32_t)a.NP));
    XM_LAUNCH_CHECK();
  }
  return XM_OK;
}

// the stride-parity classes of a strided dgrad in one launch (no split-K), all through halo variant v
static int launch_halo_multi(const std::vector<ConvGemmArgs> &args, int v, hipStream_t st) {
  ConvGemmMulti m{};
  int maxTiles = 0;
  double flops = 0, abytes = args.empty() ? 0.0 : (double)args[0].xBytes;
  for (size_t i = 0; i < args.size(); ++i) {
    ConvGemmArgs a = args[i];
    halo_prepare(a, v);
    maxTiles = std::max(maxTiles, a.nbm * a.nbn);
    flops += a.algoFlops > 0 ? a.algoFlops : 2.0 * a.M * (double)a.NP * a.Rtrue;
    abytes += 4.0 * a.M * a.Rtrue + 4.0 * a.M * (double)a.NP * (a.resid ? 2 : 1);
    m.c[i] = a;
  }
  {
    ProfScope ps(4 * 100 + v, flops, st, abytes);
    const dim3 grid(maxTiles, 1, (unsigned)args.size()), block(256);
    if (v == 0) hipLaunchKernelGGL((conv_halo_multi_kernel<2, 2, 2, 2, 512>), grid, block, 0, st, m);
    else if (v == 1) hipLaunchKernelGGL((conv_halo_multi_kernel<3, 1, 1, 4, 512>), grid, block, 0, st, m);
    else hipLaunchKernelGGL((conv_halo_multi_kernel<3, 1, 1, 4, 1024>), grid, block, 0, st, m);
  }
  XM_LAUNCH_CHECK();
  return XM_OK;
}

static int choose_cfg(long long M, long long NP, int nkt) {
  return g_force_cfg >= 0 ? g_force_cfg : pick_cfg(M, NP, nkt);
}

// ---- measured tile selection ("find mode") ---------------------------------------------------
// The first time a (direction, geometry) is seen, every tile configuration is timed on the
// caller's stream (HIP events, 2 launches each, best of) and the winner is cached for the life of
// the process.  Happens during warm-up; results are identical for every configuration.
// XM_AUTOTUNE=0 falls back to the analytic model above.
struct TuneKey {
  int kind, M, NP, Rp, mode, a, b, c, d;
  bool operator<(const TuneKey &o) const { return memcmp(this, &o, sizeof(TuneKey)) < 0; }
};
static std::map<TuneKey, int> g_tuned;

// ---- persistent tuning table ------------------------------------------------------------------
// The measured choices are kept in a text file next to the library (tune_gfx950.txt, or $XM_TUNE_FILE) and
// loaded before the first lookup: tile choices -- and therefore the summation order of every convolution --
// are then the same in every process, and shapes already in the table pay no timed launches on the caller's
// stream.  Shapes that are not in the table are still measured once per process (and written back by
// xm_tune_save).  The header carries XM_TUNE_REV, bumped whenever the kernels or the configuration list change.
constexpr int XM_TUNE_REV = 7;   // 7: configuration 7 = 128 x 128 by eight waves (the LDS-DMA ones moved to 8 ... 11); 6: fused-statistics launches are their own entries (mode + 4)
static bool g_tune_loaded = false;
static int g_tune_new = 0;  // entries measured in this process (not yet saved)

static std::string tune_default_path() {
  if (const char *e = getenv("XM_TUNE_FILE")) return e;
  Dl_info info;
  if (dladdr((const void *)&tune_default_path, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t k = p.find_last_of('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/tune_gfx950.txt";
  }
  return "tune_gfx950.txt";
}
static int tune_load_file(const char *path) {
  FILE *fp = fopen(path, "r");
  if (!fp) return 0;
  int ver = 0, rev = 0, ncfg = 0, n = 0;
  if (fscanf(fp, "xmodal-tune %d rev=%d cfgs=%d\n", &ver, &rev, &ncfg) == 3 && ver == 1 && rev == XM_TUNE_REV &&
      ncfg == kNumCfg) {
    TuneKey k;
    int cfg;
    while (fscanf(fp, "%d %d %d %d %d %d %d %d %d %d\n", &k.kind, &k.M, &k.NP, &k.Rp, &k.mode, &k.a, &k.b, &k.c, &k.d,
                  &cfg) == 10)
      if (cfg >= 0 && cfg < kNumCfg && !g_tuned.count(k)) {
        g_tuned[k] = cfg;
        ++n;
      }
  }
  fclose(fp);
  return n;
}
static void tune_load_once() {
  if (g_tune_loaded) return;
  g_tune_loaded = true;
  const char *e = getenv("XM_TUNE_FILE");
  if (e && !e[0]) return;  // XM_TUNE_FILE="" : no persistent table
  int n = tune_load_file(tune_default_path().c_str());
  if (getenv("XM_TUNE_VERBOSE")) fprintf(stderr, "[xm tune] %d entries from %s\n", n, tune_default_path().c_str());
}
static int autotune_enabled() {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("XM_AUTOTUNE");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on;
}
template <class F>
static int tune_cfg(const TuneKey &key, int fallback, hipStream_t st, F &&launch, int ncfg = kNumBaseCfg,
                    unsigned skip = 0 /* bit ci: configuration not eligible for this launch */) {
  if (g_force_cfg >= 0) return g_force_cfg < ncfg ? g_force_cfg : base_cfg(g_force_cfg);
  if (!autotune_enabled()) return fallback;
  tune_load_once();
  auto it = g_tuned.find(key);
  if (it != g_tuned.end()) return it->second;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return fallback;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fallback;
  static const bool verbose = getenv("XM_TUNE_VERBOSE") != nullptr;
  static const int reps = getenv("XM_TUNE_REPS") ? std::max(1, atoi(getenv("XM_TUNE_REPS"))) : 2;
  float best = 1e30f;
  int bi = fallback;
  // The candidates are timed on an otherwise idle device: whatever the other streams of the step had queued is drained first
  // (the host thread is in here, so nothing new arrives).  Timed next to a neighbour's kernels the choice depended on what
  // happened to be running -- the shipped table differed from one generation to the next.
  (void)hipDeviceSynchronize();
  for (int ci = 0; ci < ncfg; ++ci) {
    if (skip >> ci & 1u) continue;
    float tmin = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
      (void)hipEventRecord(e0, st);
      if (launch(ci) != XM_OK) {
        tmin = 1e30f;
        break;
      }
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) break;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) tmin = std::min(tmin, ms);
    }
    if (verbose) fprintf(stderr, " cfg%d %.3f", ci, tmin);
    if (tmin < best) {
      best = tmin;
      bi = ci;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  g_tuned[key] = bi;
  ++g_tune_new;
  if (verbose)
    fprintf(stderr, "  -> [xm tune] kind %d M %d NP %d Rp %d (%d %d %d %d %d): cfg%d %.3f ms\n", key.kind, key.M,
            key.NP, key.Rp, key.mode, key.a, key.b, key.c, key.d, bi, best);
  return bi;
}

// Incumbent (0) against challengers (1 .. n - 1): all are timed ALTERNATELY, four rounds, best-of each (the first round
// is a warm-up), and the best challenger has to win by `margin` -- a choice within the noise of two timed launches
// flipped between processes, and an alternative that is only as fast in isolation loses when its blocks share the chip
// (larger LDS footprint).  `ok[i]` = false skips a challenger.
template <class F>
static int tune_challengers(const TuneKey &key, hipStream_t st, F &&launch, int n, const bool *ok, float margin) {
  if (!autotune_enabled()) return 0;
  tune_load_once();
  auto it = g_tuned.find(key);
  if (it != g_tuned.end()) return (it->second < n && (it->second == 0 || ok[it->second])) ? it->second : 0;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 0;
  float tmin[8];
  for (int i = 0; i < 8; ++i) tmin[i] = 1e30f;
  (void)hipDeviceSynchronize();   // see tune_cfg: the candidates are timed on an otherwise idle device
  for (int rep = 0; rep < 4; ++rep)
    for (int ci = 0; ci < n && ci < 8; ++ci) {
      if (ci > 0 && !ok[ci]) continue;
      (void)hipEventRecord(e0, st);
      if (launch(ci) != XM_OK) continue;
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) continue;
      float ms = 0.f;
      if (rep > 0 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) tmin[ci] = std::min(tmin[ci], ms);
    }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  int pick = 0;
  float best = (1.f - margin) * tmin[0];
  for (int ci = 1; ci < n && ci < 8; ++ci)
    if (tmin[ci] < best) best = tmin[ci], pick = ci;
  g_tuned[key] = pick;
  ++g_tune_new;
  if (getenv("XM_TUNE_VERBOSE"))
    fprintf(stderr, "[xm tune] kind %d M %d NP %d Rp %d: incumbent %.3f ms, challengers %.3f %.3f %.3f -> %d\n", key.kind,
            key.M, key.NP, key.Rp, tmin[0], tmin[1], tmin[2], tmin[3], pick);
  return pick;
}

template <int AV>
static void launch_wgrad_av(int ci, const WgradArgs &a, dim3 grid, hipStream_t st) {
  dim3 block(256);
  switch (ci) {
    case 0: hipLaunchKernelGGL((conv_wgrad_kernel<2, 2, 2, 2, AV>), grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL((conv_wgrad_kernel<2, 2, 1, 4, AV>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((conv_wgrad_kernel<3, 1, 1, 4, AV>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((conv_wgrad_kernel<1, 2, 2, 2, AV>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 2, 2, AV>), grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 4, 1, AV>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 1, 4, AV>), grid, block, 0, st, a); break;
  }
}

// dY staging width: 4 / 2 consecutive pixels per load when groups cannot straddle two samples
// (Ho*Wo % AV == 0) and the rows are AV*4-byte aligned
static int wgrad_av(const WgradArgs &a) {
  const int hw = a.Ho * a.Wo;
  const uintptr_t p = (uintptr_t)a.dY;
  if (hw % 4 == 0 && (p & 15) == 0) return 4;
  if (hw % 2 == 0 && (p & 7) == 0) return 2;
  return 1;
}
static void launch_wgrad_cfg(int ci, const WgradArgs &a, dim3 grid, hipStream_t st) {
  switch (wgrad_av(a)) {
    case 4: launch_wgrad_av<4>(ci, a, grid, st); break;
    case 2: launch_wgrad_av<2>(ci, a, grid, st); break;
    default: launch_wgrad_av<1>(ci, a, grid, st); break;
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
  // gathers go through 32-bit-offset buffer descriptors: every tensor must stay below 4 GiB
  if ((long long)H * W * C * N >= (1LL << 30) || (long long)Ho * Wo * K * N >= (1LL << 30) ||
      (long long)FH * FW * FC * K >= (1LL << 30))
    return fail(XM_ETOOBIG, "vl_nnconv: tensor with >= 2^30 elements (4 GiB)");
  g = Geo{H, W, C, N, FH, FW, FC, K, G, K / G, Ho, Wo, FH * FW * FC, sy, sx, pt, pb, pl, pr, dy, dx};
  return XM_OK;
}

// forward tap table for the implicit GEMM: r = u + FH*(v + FW*c) -> {byte offset in X, (u,v) index}
static const int2 *fwd_taps2(const Geo &g, int Rp, bool all_valid = false) {
  int count = Rp + 3 * kBK;  // kernels fetch the table three stages ahead
  std::vector<int2> t(count);
  for (int r = 0; r < count; ++r) {
    if (r < g.R) {
      int u = r % g.FH, v = (r / g.FH) % g.FW, c = r / (g.FH * g.FW);
      // all_valid (no spatial padding, > 63 taps): every real tap shares mask bit 0 = tap (0,0)
      t[r] = make_int2(4 * (u * g.dy + g.H * (v * g.dx) + g.H * g.W * c), all_valid ? 0 : u + g.FH * v);
    } else {
      t[r] = make_int2(0, 63);
    }
  }
  return (const int2 *)cached_device_table(t.data(), t.size() * sizeof(int2));
}

// wgrad tap table: {byte offset in X, u*dy, v*dx}
static const int4 *fwd_taps(const Geo &g, int count) {
  std::vector<int4> t(count);
  for (int r = 0; r < count; ++r) {
    if (r < g.R) {
      int u = r % g.FH, v = (r / g.FH) % g.FW, c = r / (g.FH * g.FW);
      t[r] = make_int4(4 * (u * g.dy + g.H * (v * g.dx) + g.H * g.W * c), u * g.dy, v * g.dx, 0);
    } else {
      t[r] = make_int4(0, -(1 << 28), 0, 0);
    }
  }
  return (const int4 *)cached_device_table(t.data(), t.size() * sizeof(int4));
}

// ---- skinny fully-connected layers ------------------------------------------------------------
// vl_nnconv with a 1 x 1 filter over few output values: the SE gates (1 x 1 x C x N), the teacher's classifier
// (2048 -> 8), the student's fc8.  y[m, p] = act(b[m] + sum_k F[k + K m] X[p, k]).  An MFMA tile would be > 90 %
// padding and a single block would walk K alone (31 us for 2048 -> 8 at 32 samples).  Here one BLOCK owns one output
// row m and a chunk of <= 32 pixels: its 1-4 waves split K (VEC: 4 consecutive k per lane, 16-byte loads of the
// filter row and of every pixel's channel run -- needs H*W == 1; otherwise one k per lane), 32 accumulators per
// lane, a transposing butterfly (63 shuffles instead of 32 x 6) leaves the sum of pixel j in lanes 2j, 2j + 1,
// and the waves' partials are added in wave order through LDS.  ACT: 0 none, 1 relu, 2 sigmoid.
template <int ACT, bool VEC>
__global__ void __launch_bounds__(256)
fc_skinny_kernel(const float *__restrict__ x, const float *__restrict__ f, const float *__restrict__ b,
                 float *__restrict__ y, int K, int M, int HW, int NP, int chunks, FastDiv divChunks, FastDiv divHW,
                 const float *__restrict__ scale, const float *__restrict__ shift) {
  __shared__ float red[4][32];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int m = (int)xm_div(blockIdx.x, divChunks);
  const int p0 = ((int)blockIdx.x - m * chunks) * 32;
  int xb[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int p = min(p0 + j, NP - 1);
    const int n = (int)xm_div((uint32_t)p, divHW);
    xb[j] = (p - n * HW) + HW * K * n;
  }
  float acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  const float *fr = f + (size_t)K * m;
  if (VEC) {
    for (int k = (wv * 64 + lane) * 4; k < K; k += nw * 256) {
      const float4 w = *reinterpret_cast<const float4 *>(fr + k);
      // all loads of a half first, then the arithmetic: left alone, the scheduler pairs every load with its use
      // (two loads in flight, 32 serial round trips -- 20 us per call)
#pragma unroll
      for (int jh = 0; jh < 32; jh += 16) {
        f32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const f32x4 *>(x + xb[jh + j] + k);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(v[j]));  // pins the uses behind all 16 loads
#pragma unroll
        for (int j = 0; j < 16; ++j)
          acc[jh + j] = fmaf(w.w, v[j].w, fmaf(w.z, v[j].z, fmaf(w.y, v[j].y, fmaf(w.x, v[j].x, acc[jh + j]))));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    for (int k = wv * 64 + lane; k < K; k += nw * 64) {
      const float w = fr[k];
      const float *xk = x + (size_t)HW * k;
      float v[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = xk[xb[j]];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 32; ++j) asm volatile("" : "+v"(v[j]));
#pragma unroll
      for (int j = 0; j < 32; ++j) acc[j] = fmaf(w, v[j], acc[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // stage (o, h): lanes with bit o set keep the upper h values, the others the lower h, and add the partner's
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int o = 32 >> s, h = 16 >> s;
    const bool up = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      // (selecting which ELEMENT to send would halve the shuffles, but the compiler turns `up ? acc[i + h] : acc[i]`
      // into a runtime-indexed register array: 5000 compare / select instructions)
      const float lo = acc[i] + __shfl_xor(acc[i], o, 64);
      const float hi = acc[i + h] + __shfl_xor(acc[i + h], o, 64);
      acc[i] = up ? hi : lo;
    }
  }
  float v = acc[0] + __shfl_xor(acc[0], 1, 64);
  if ((lane & 1) == 0) red[wv][lane >> 1] = v;
  __syncthreads();
  const int p = p0 + (int)threadIdx.x;
  if (threadIdx.x < 32 && p < NP) {
    v = red[0][threadIdx.x];
    for (int w = 1; w < nw; ++w) v += red[w][threadIdx.x];
    if (b) v += b[m];
    if (scale) v = v * scale[m] + shift[m];     // folded test-mode bnorm: (conv + b) .* scale + shift
    if (ACT == 1) v = fmaxf(v, 0.f);
    if (ACT == 2) v = 1.f / (1.f + expf(-v));
    const int n = (int)xm_div((uint32_t)p, divHW);
    y[(p - n * HW) + (size_t)HW * (m + (size_t)M * n)] = v;
  }
}
// The same for FOUR output rows and a chunk of 16 pixels per block (H*W == 1, 16-byte loads): a block of the one-row
// kernel reads its 32 pixels' whole channel runs, i.e. the input is read M times from the L2s (512 MB for the student's
// fc7 at 32 clips: 22 us); four rows per block share every load.  64 accumulators per lane (row r, pixel j -> 16 r + j),
// the transposing butterfly over all six lane bits leaves value i in lane i, the waves' partials meet in LDS.
template <int ACT>
__global__ void __launch_bounds__(256)
fc_skinny4_kernel(const float *__restrict__ x, const float *__restrict__ f, const float *__restrict__ b,
                  float *__restrict__ y, int K, int M, int NP, int chunks, FastDiv divChunks,
                  const float *__restrict__ scale, const float *__restrict__ shift) {
  __shared__ float red[4][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int mg = (int)xm_div(blockIdx.x, divChunks);
  const int p0 = ((int)blockIdx.x - mg * chunks) * 16, m0 = 4 * mg;
  int xb[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) xb[j] = K * min(p0 + j, NP - 1);
  float acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = 0.f;
  const float *fr = f + (size_t)K * m0;
  for (int k = (wv * 64 + lane) * 4; k < K; k += nw * 256) {
    f32x4 w[4], v[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4 *>(fr + (size_t)K * r + k);
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const f32x4 *>(x + xb[j] + k);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(v[j]));   // pins the uses behind all the loads (see fc_skinny_kernel)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 16; ++j)
        acc[16 * r + j] = fmaf(w[r].w, v[j].w, fmaf(w[r].z, v[j].z, fmaf(w[r].y, v[j].y, fmaf(w[r].x, v[j].x, acc[16 * r + j]))));
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int o = 32 >> s, h = 32 >> s;
    const bool up = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const float lo = acc[i] + __shfl_xor(acc[i], o, 64);
      const float hi = acc[i + h] + __shfl_xor(acc[i + h], o, 64);
      acc[i] = up ? hi : lo;
    }
  }
  red[wv][lane] = acc[0];
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = red[0][threadIdx.x];
    for (int w = 1; w < nw; ++w) v += red[w][threadIdx.x];
    const int m = m0 + (int)(threadIdx.x >> 4), p = p0 + (int)(threadIdx.x & 15);
    if (b) v += b[m];
    if (scale) v = v * scale[m] + shift[m];
    if (ACT == 1) v = fmaxf(v, 0.f);
    if (ACT == 2) v = 1.f / (1.f + expf(-v));
    if (p < NP) y[m + (size_t)M * p] = v;
  }
}
static bool fc_skinny_ok(const Geo &g, bool dgrad = false) {
  const bool off = !path_on(kPathSkinny);
  if (off || g.FH != 1 || g.FW != 1 || g.sy != 1 || g.sx != 1 || g.dy != 1 || g.dx != 1 || g.G != 1) return false;
  if (g.pt | g.pb | g.pl | g.pr) return false;
  const long long NP = (long long)g.Ho * g.Wo * g.N;
  // A wide FC layer over many samples is a GEMM again (the student's fc7, 4096 -> 1024: forward 19.7 / 22.5 / 41.5 / 60.9 us
  // here against 38 / 40 / 42 / 46.6 us on the MFMA path at 32 / 64 / 128 / 256 samples; as dX = W' dY 28.6 / 39.6 / 67.7 / 111 us
  // against 30 / 32.5 / 40.6 / 59 us): every 16-sample chunk re-reads the 16 MB of weights.  The SE layers (<= 0.26 M weights)
  // stay here at every batch.
  if ((long long)g.C * g.K >= (1ll << 20) && NP > (dgrad ? 32 : 128)) return false;
  const bool rows4 = g.H * g.W == 1 && (g.C & 3) == 0 && (g.K & 3) == 0;   // fc_skinny4_kernel's geometry
  if (!(g.K <= 16 || NP <= (rows4 ? 512 : 128))) return false;
  return (long long)g.K * ((NP + 31) / 32) <= 65535 && (long long)g.H * g.W * g.C * g.N < (1ll << 31);
}
static int fc_skinny_forward(const float *x, const float *f, const float *b, float *y, const Geo &g, int act,
                             hipStream_t st, const float *scale = nullptr, const float *shift = nullptr) {
  const int HW = g.H * g.W, NP = HW * g.N, chunks = (NP + 31) / 32;
  const bool vec = HW == 1 && (g.C & 3) == 0 && (((uintptr_t)x | (uintptr_t)f) & 15) == 0;
  const int units = vec ? g.C / 4 : g.C;   // k positions handed out per lane step
  const int nw = std::max(1, std::min(4, (units + 63) / 64));
  const bool no4 = !path_on(kPathSkinny4);
  if (vec && (g.K & 3) == 0 && g.K >= 16 && !no4) {
    const int ch = (NP + 15) / 16;
    dim3 grid4((unsigned)(g.K / 4 * ch)), block4(64 * nw);
    FastDiv dc4 = make_fastdiv((uint32_t)ch);
#define XM_FC4_LAUNCH(A) \
  hipLaunchKernelGGL((fc_skinny4_kernel<A>), grid4, block4, 0, st, x, f, b, y, g.C, g.K, NP, ch, dc4, scale, shift)
    if (act == 2) XM_FC4_LAUNCH(2);
    else if (act == 1) XM_FC4_LAUNCH(1);
    else XM_FC4_LAUNCH(0);
#undef XM_FC4_LAUNCH
    XM_LAUNCH_CHECK();
    return XM_OK;
  }
  dim3 grid(g.K * chunks), block(64 * nw);
  FastDiv dc = make_fastdiv((uint32_t)chunks), dh = make_fastdiv((uint32_t)HW);
#define XM_FC_LAUNCH(A, V) \
  hipLaunchKernelGGL((fc_skinny_kernel<A, V>), grid, block, 0, st, x, f, b, y, g.C, g.K, HW, NP, chunks, dc, dh, scale, shift)
  if (vec) {
    if (act == 2) XM_FC_LAUNCH(2, true);
    else if (act == 1) XM_FC_LAUNCH(1, true);
    else XM_FC_LAUNCH(0, true);
  } else {
    if (act == 2) XM_FC_LAUNCH(2, false);
    else if (act == 1) XM_FC_LAUNCH(1, false);
    else XM_FC_LAUNCH(0, false);
  }
#undef XM_FC_LAUNCH
  XM_LAUNCH_CHECK();
  return XM_OK;
}

// moments_out != NULL: also the batch moments [mean, sqrt(var + eps)] of Y (the statistics half of the train-mode
// vl_nnbnorm that follows the convolution), from per-wave partial sums of the GEMM epilogue when the launch allows it
// (16-byte-store epilogue, no split-K, no filter groups), else by a pass over Y.
// ---- single-channel stem (conv_stem_kernel) ----------------------------------------------------------------------
// Forward convolutions over ONE input channel with <= 8 x 7 taps, stride 1 / 2 along H, <= 96 filters, source columns of
// <= 512 rows (the student's conv1 over 512-bin spectrograms).  `f` is the caller's filter bank, unpadded.
static bool stem_ok(const ConvGemmArgs &a, const Geo &g, const float *x, const float *f) {
  const bool off = !path_on(kPathStem);
  if (off || g_force_cfg >= 0 || g_force_splits > 0) return false;
  if (g.C != 1 || g.G != 1 || g.FC != 1 || g.dy != 1 || g.dx != 1) return false;
  if (g.FH > 8 || g.FW > kStemNV || g.FH * g.FW < 16 || g.Kg > 96) return false;
  if (g.sy != 1 && g.sy != 2) return false;
  if (!a.vecStore || a.scale || a.resid || a.gate || a.relu) return false;
  if (g.H % 4 != 0 || g.H > kStemHP - 8 || ((uintptr_t)x & 15) != 0) return false;
  if (g.pt > 4 || 4 * ((g.sy * (g.Ho - 1) - g.pt + 4 + 7) >> 2) + 3 >= kStemHP) return false;   // last 16-byte row unit a tile loads
  if (g.Ho < 128) return false;                                                  // a tile spans <= 2 output columns
  if (g_force_stem != 1 && (long long)g.Ho * g.Wo * g.N < 128 * 512) return false;   // >= one round of the chip
  (void)f;
  return true;
}

static int stem_grid(int NP) { return std::min(256 * XM_STEM_OCC, ((NP + 127) / 128 + 7) / 8 * 8); }

static int launch_stem(ConvGemmArgs a, const float *f, int R, hipStream_t st) {
  a.A = f;
  a.lda = R;
  a.nbm = 1;
  a.nbn = (a.NP + 127) / 128;
  a.slab = nullptr;
  a.hyS = 1, a.hyFull = 0, a.hyTps = 0, a.hyP0 = 0;
  a.piReal = 0;
  a.dbgCycles = g_dbg_cycles;
  const double abytes = (double)a.xBytes + 4.0 * a.M * a.Rtrue + 4.0 * a.M * (double)a.NP;
  ProfScope ps(5 * 100, 2.0 * a.M * (double)a.NP * a.Rtrue, st, abytes);
  const int grid = stem_grid(a.NP);
  if (a.gsy == 2) hipLaunchKernelGGL(conv_stem_kernel<2>, dim3(grid), dim3(256), 0, st, a, a.nbn);
  else hipLaunchKernelGGL(conv_stem_kernel<1>, dim3(grid), dim3(256), 0, st, a, a.nbn);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

// ---- three-channel stem (conv_stem3_kernel): 7 x 7 / stride 2 over RGB images, 64 filters -- the teachers' conv1 --------------
static int g_force_stem3 = -1;   // test hook (xm_debug_force_conv_stem3)
static bool stem3_ok(const ConvGemmArgs &a, const Geo &g, const float *x, bool stats) {
  if (!path_on(kPathStem3) || g_force_cfg >= 0 || g_force_splits > 0 || stats) return false;
  if (g.C != 3 || g.G != 1 || g.FC != 3 || g.FH != 7 || g.FW != 7 || g.sy != 2 || g.sx != 2 || g.dy != 1 || g.dx != 1) return false;
  if (g.Kg != 64 || !a.vecStore || a.resid || a.gate) return false;
  if ((g.Ho * g.Wo) % 128 != 0 || g.Ho < 64 || (g.H & 1) || g.pt > 4 || g.pl > 6) return false;
  if (2 * (g.Ho - 1) + 8 + (4 - g.pt) + 1 > kStem3CS) return false;                 // patch rows of a column
  if (((uintptr_t)x & 7) != 0 || (size_t)g.H * g.W * g.C * g.N * 4 >= (1ull << 31)) return false;
  return g_force_stem3 == 1 || (long long)g.Ho * g.Wo * g.N >= 128 * 512;           // >= one round of the chip
}
static int launch_stem3(ConvGemmArgs a, const float *f, int R, hipStream_t st) {
  a.A = f;
  a.lda = R;
  a.nbm = 1;
  a.nbn = a.NP / 128;
  a.slab = nullptr;
  const double abytes = (double)a.xBytes + 4.0 * a.M * a.Rtrue + 4.0 * a.M * (double)a.NP;
  // 256-pixel block tiles (four accumulator chains per wave) where they divide a sample and span <= 4 output columns
  static const bool wide_on = env_int("XM_STEM3_WIDE", 1) != 0;
  const int pij = (int)a.divPIJ.d, pi = (int)a.divPI.d;
  const bool wide = wide_on && pij % 256 == 0 && pi >= 86;
  if (wide) a.nbn = a.NP / 256;
  ProfScope ps(12 * 100 + (wide ? 1 : 0), 2.0 * a.M * (double)a.NP * a.Rtrue, st, abytes);
  const int grid = std::min(512, (a.nbn + 7) / 8 * 8);
  if (wide) hipLaunchKernelGGL(conv_stem3_kernel<2>, dim3(grid), dim3(256), 0, st, a, a.nbn);
  else hipLaunchKernelGGL(conv_stem3_kernel<1>, dim3(grid), dim3(256), 0, st, a, a.nbn);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

static int conv_forward(const float *x, const float *f, const float *b, float *y, const Geo &g,
                        const float *scale, const float *shift, const float *resid, int relu,
                        hipStream_t st, float *moments_out = nullptr, float eps = 0.f, const float *gate = nullptr) {
  // The (u,v) validity mask has 63 bits.  Without spatial padding every tap is inside the image, so
  // larger filters (the 1 x 401 STFT bank of batch.runSpec) simply do not use it.
  const bool padded = (g.pt | g.pb | g.pl | g.pr) != 0;
  const bool bigTaps = g.FH * g.FW > 63;
  if (bigTaps && padded)
    return fail(XM_ENOTSUP, "vl_nnconv: padded filters with more than 63 spatial taps (%dx%d) are not built",
                g.FH, g.FW);
  const int Rp = (g.R + kBK - 1) / kBK * kBK;
  const bool need_pad = (g.R % kBK) != 0 || ((uintptr_t)f & 15);
  const int mode = ((g.pt | g.pb | g.pl | g.pr) != 0 || Rp != g.R) ? 1 : 0;
  ConvGemmArgs proto{};
  proto.M = g.Kg;
  proto.NP = g.Ho * g.Wo * g.N;
  proto.Rp = Rp;
  // scratch must fit the split-K slab of whichever tile configuration ends up being used
  size_t slabf = 0;
  for (int c = 0; c < kNumCfg; ++c) {
    int sp;
    slabf = std::max(slabf, gemm_slab_floats(proto, c, &sp));
  }
  if (g.FH <= 3 && g.FW <= 3 && g.FH * g.FW >= 4 && g.R % (kHaloCB * g.FH * g.FW) == 0) {   // halo-patch kernel splits
    ConvGemmArgs ph = proto;
    ph.Rtrue = g.R;
    ph.nU = g.FH;
    ph.nV = g.FW;
    slabf = std::max(slabf, halo_slab_floats(ph));
  }
  // LDS-DMA eligibility: a plain GEMM in memory (1x1, unit stride, no padding), pixel quads inside one sample,
  // 16-byte aligned operands
  const bool dma_ok = !need_pad && mode == 0 && g.FH == 1 && g.FW == 1 && g.sy == 1 && g.sx == 1 && g.dy == 1 &&
                      g.dx == 1 && (g.H * g.W) % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)f & 15) == 0 &&
                      g.R % kBK == 0 && path_on(kPathDma);
  const bool no_fstats = !path_on(kPathFusedStats);
  const bool want_stats = moments_out != nullptr && g.G == 1 && !no_fstats;
  // partial sums: one {sum, sum sq} pair per row and per pixel tile (>= 32 pixels), + 64 fp64 slabs for the reduction
  // (the persistent stem kernel leaves one row per block: stem_grid(NP) can exceed NP / 32 on small problems)
  const size_t statf = want_stats ? (size_t)2 * g.K * std::max((proto.NP + 31) / 32, stem_grid(proto.NP)) : 0;
  const size_t stat2 = want_stats ? (size_t)2 * g.K * 64 : 0;
  WsCarver ws;
  int rc = ws.init(WsCarver::need(need_pad ? (size_t)g.K * Rp : 0, 4) + WsCarver::need(slabf, 4) +
                       WsCarver::need(statf, 4) + WsCarver::need(stat2, 8), st);
  if (rc) return rc;
  const int2 *taps = fwd_taps2(g, Rp, bigTaps);
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
  float *slab = slabf ? ws.take<float>(slabf) : nullptr;
  float *statp = statf ? ws.take<float>(statf) : nullptr;
  double *statp2 = stat2 ? ws.take<double>(stat2) : nullptr;
  bool stats_done = false;
  const size_t xTotal = (size_t)g.H * g.W * g.C * g.N;
  for (int grp = 0; grp < g.G; ++grp) {
    ConvGemmArgs a{};
    a.A = A + (size_t)grp * g.Kg * lda;
    size_t xoff = (size_t)grp * g.FC * g.H * g.W;
    a.X = x + xoff;
    a.xBytes = (unsigned)((xTotal - xoff) * 4);
    a.Y = y + (size_t)grp * g.Kg * g.Ho * g.Wo;
    a.taps = taps;
    a.bias = b ? b + grp * g.Kg : nullptr;
    a.scale = scale ? scale + grp * g.Kg : nullptr;
    a.shift = shift ? shift + grp * g.Kg : nullptr;
    a.resid = resid ? resid + (size_t)grp * g.Kg * g.Ho * g.Wo : nullptr;
    a.gate = gate ? gate + (size_t)grp * g.Kg : nullptr;
    a.gateStride = g.K;
    a.relu = relu;
    a.lda = lda;
    a.M = g.Kg;
    a.Rp = Rp;
    a.Rtrue = g.R;
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
    a.nU = bigTaps ? 1 : g.FH;   // mask construction only looks at tap (0,0) then (always valid)
    a.nV = bigTaps ? 1 : g.FW;
    a.du0 = 0;
    a.dus = g.dy;
    a.dv0 = 0;
    a.dvs = g.dx;
    a.osy = 1;
    a.osx = 1;
    a.oh0 = 0;
    a.ow0 = 0;
    a.OH = g.Ho;
    a.oChanStride = g.Ho * g.Wo;
    a.oSampleStride = g.Ho * g.Wo * g.K;
    a.divMU = make_fastdiv(1);
    a.oUStride = 0;
    // 4 consecutive output pixels are contiguous (same sample) and every tile starts on a multiple
    // of 32 pixels; 16-byte alignment of (y, residual) rows needs Ho*Wo % 4 == 0
    a.vecStore = ((g.Ho * g.Wo) % 4 == 0 && (((uintptr_t)a.Y | (uintptr_t)a.resid) & 15) == 0) ? 1 : 0;
    a.dmaOk = (dma_ok && a.vecStore) ? 1 : 0;   // the LDS-DMA kernel only carries the 16-byte-store epilogue
    const bool stats_ok = want_stats && a.vecStore && !relu && !resid;
    if (stats_ok) a.dmaOk = 0;                  // the partial sums ride in the register-staged kernel's epilogue
    a.aBytes = (unsigned)((size_t)g.Kg * lda * 4);
    a.tapStride = (unsigned)((size_t)g.H * g.W * 4);
    int stat_ncg = 0;
    auto run = [&](int ci) {
      int sp;
      ci = a.dmaOk ? ci : base_cfg(ci);
      gemm_slab_floats(a, ci, &sp);
      ConvGemmArgs aa = a;
      stat_ncg = 0;
      if (stats_ok && sp == 1 && (g_force_splits <= 1)) {
        const Cfg &c = kCfgs[ci];
        stat_ncg = (a.NP + c.bn() - 1) / c.bn();
        aa.statPart = statp;
        aa.statNcg = stat_ncg;
      }
      const int rc_ = launch_gemm(aa, mode, ci, sp, slab, st);
      if (aa.statPart) stat_ncg = aa.statNcg;     // (hybrid schedule: full pixel tiles + the combine kernel's chunks)
      return rc_;
    };
    // the launch with fused batch statistics is its own entry: it excludes the LDS-DMA configurations and carries a
    // longer epilogue, so whichever variant reached a shape first must not fix the choice for the other
    const int tmode = mode + (stats_ok ? 4 : 0);
    TuneKey key{0, a.M, a.NP, Rp, tmode, g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
    int ci = tune_cfg(key, pick_cfg(a.M, a.NP, Rp / kBK), st, run, a.dmaOk ? kNumCfg : kNumBaseCfg, w8_skip(a.M, a.NP));
    if (stem_ok(a, g, x, f)) {
      // the single-channel stem kernel against the best implicit-GEMM configuration (measured once per shape)
      auto run3 = [&](int h) {
        if (!h) return run(ci);
        ConvGemmArgs aa = a;
        stat_ncg = 0;
        if (stats_ok) {
          stat_ncg = stem_grid(a.NP);             // one partial per (persistent) block
          aa.statPart = statp;
          aa.statNcg = stat_ncg;
        }
        return launch_stem(aa, f, g.R, st);
      };
      bool sok[2] = {true, true};
      TuneKey skey{7, a.M, a.NP, Rp, tmode, g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
      const int pick = g_force_stem >= 0 ? g_force_stem : tune_challengers(skey, st, run3, 2, sok, kHaloMargin);
      rc = run3(pick);
      if (rc) return rc;
    } else if (stem3_ok(a, g, x, stats_ok)) {
      // the three-channel stem kernel against the best implicit-GEMM configuration (measured once per shape)
      auto run5 = [&](int h) { return h ? launch_stem3(a, f, g.R, st) : run(ci); };
      bool sok[2] = {true, true};
      TuneKey skey{12, a.M, a.NP, Rp, tmode, g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
      const int pick = g_force_stem3 >= 0 ? g_force_stem3 : tune_challengers(skey, st, run5, 2, sok, kHaloMargin);
      rc = run5(pick);
      if (rc) return rc;
    } else {
    // <= 3 x 3 taps / unit stride: the halo-patch kernel variants against the best implicit-GEMM configuration
    // (measured once per shape, alternating launches, the challenger must win by a margin)
    ConvGemmArgs ah = a;
    const int hneed = (g_force_cfg < 0 && g_force_splits == 0) ? halo_setup(ah, g.N) : 0;
    bool hok[4] = {true, halo_var_ok(ah, hneed, 0), halo_var_ok(ah, hneed, 1), halo_var_ok(ah, hneed, 2)};
    if (hok[1] || hok[2] || hok[3]) {
      auto run2 = [&](int h) {
        if (!h) return run(ci);
        ConvGemmArgs aa = ah;
        stat_ncg = 0;
        if (stats_ok && halo_splits(aa, h - 1) == 1) {
          stat_ncg = (a.NP + 127) / 128;
          aa.statPart = statp;
          aa.statNcg = stat_ncg;
        }
        return launch_halo(aa, h - 1, slab, st);
      };
      TuneKey hkey{4, a.M, a.NP, Rp, tmode, g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
      int pick;
      if (g_force_halo == 0) pick = 0;
      else if (g_force_halo > 0) pick = forced_halo_variant(hok);
      else pick = tune_challengers(hkey, st, run2, 4, hok, kHaloMargin);
      rc = run2(pick);
    } else {
      rc = run(ci);
    }
    if (rc) return rc;
    }
    if (stat_ncg > 0) {
      const int S = std::max(1, std::min(64, stat_ncg / 768));   // one launch up to ~1500 partial rows per channel
      hipLaunchKernelGGL(conv_stats_reduce_kernel, dim3((g.K + 7) / 8, S), dim3(256), 0, st, statp, moments_out,
                         statp2, g.K, stat_ncg, S, (double)a.NP, eps);
      XM_LAUNCH_CHECK();
      if (S > 1) {
        hipLaunchKernelGGL(conv_stats_finalize2_kernel, dim3((g.K + 255) / 256), dim3(256), 0, st, statp2, moments_out,
                           g.K, S, (double)a.NP, eps);
        XM_LAUNCH_CHECK();
      }
      stats_done = true;
    }
  }
  if (moments_out && !stats_done) return bn_batch_moments(y, g.Ho, g.Wo, g.K, g.N, eps, moments_out, st);
  return XM_OK;
}

// ---- prepared dgrad operands ------------------------------------------------------------------
// The dgrad GEMM reads the filter bank transposed / split by stride parity (prep_dgrad_filter_kernel).  In a
// training step that is ~20 us per layer ON THE CRITICAL PATH of the backward pass.  xm_nnconv_prepare_backward
// lets the host run the transposition early -- typically while the forward pass of the same step runs, on a
// side stream -- into a persistent buffer; the backward call then finds it and skips the kernel.  An entry is
// valid while the parameter version it was built at is current (xm_sgd_update / xm_average_update /
// xm_params_changed bump the version) and the geometry matches.
struct PrepKey {
  const float *f;
  int FH, FW, FC, K, sy, sx, dy, dx, pt, pl, H, W, fold;
  bool operator<(const PrepKey &o) const { return memcmp(this, &o, sizeof(PrepKey)) < 0; }
};
struct PrepEntry {
  float *buf = nullptr;
  size_t bytes = 0;
  unsigned long long version = ~0ull;
  hipEvent_t ev = nullptr;
  hipStream_t st = nullptr;
};
static std::map<PrepKey, PrepEntry> g_prep;
unsigned long long g_param_version = 1;   // bumped by every parameter update (misc.hip)
static PrepKey prep_key(const float *f, const Geo &g, bool fold) {
  PrepKey k;
  memset(&k, 0, sizeof k);
  k.f = f, k.FH = g.FH, k.FW = g.FW, k.FC = g.FC, k.K = g.K, k.sy = g.sy, k.sx = g.sx, k.dy = g.dy, k.dx = g.dx;
  k.pt = g.pt, k.pl = g.pl, k.H = g.H, k.W = g.W, k.fold = fold ? 1 : 0;
  return k;
}

// dX: one implicit GEMM per stride-parity class (a, b) of the input pixels.
// accum != NULL: dX = dgrad + accum (the derivative another branch of a fork already produced), added in
// the GEMM epilogue instead of by a separate pass
// prepare_only: run just the filter transpositions into the persistent cache (xm_nnconv_prepare_backward)
// ---- dgrad of 5 x 5 / stride 2 layers with both row parities per wave (conv_dgrad_s2_kernel, round 5) ---------------------
static int g_force_dgrad_s2 = -1;   // test hook (xm_debug_force_dgrad_s2): 0 = never, 1 / -1 = wherever it can run
static bool dgrad_s2_ok(const Geo &g, const float *dzdy, const float *dxo, const float *accum) {
  if (!path_on(kPathDgradS2) || g_force_dgrad_s2 == 0 || g_force_cfg >= 0 || g_force_splits > 0 || accum) return false;
  if (g.G != 1 || g.FH != 5 || g.FW != 5 || g.sy != 2 || g.sx != 2 || g.dy != 1 || g.dx != 1) return false;
  if (g.pt != 1 || g.pl < 0 || g.pl > 4 || (g.H & 1) || (g.Ho & 1) || (g.K & 7)) return false;
  if ((((uintptr_t)dzdy | (uintptr_t)dxo) & 7) != 0) return false;
  if ((size_t)g.Ho * g.Wo * g.K * g.N * 4 >= (1ull << 31)) return false;   // dY byte offsets are formed in 32-bit arithmetic
  if (g_force_dgrad_s2 == 1) return true;
  // Its blocks are large (two output columns x 64 row pairs x all filters: ~0.4 ms of a CU slot) and there are 37 of them per
  // 126 x 73 sample, so small batches pay the partly filled last round -- 32 spectrograms: 1.54 rounds of 768 slots cost 2
  // (90 against 95 TFLOP/s for the merged launch), 64: 3.08 rounds cost 4 and the 40 KB blocks are poor neighbours of the
  // filter derivative on the side stream (student step - 4 %); 256: 12.3 rounds, 124 against 115 TFLOP/s and - 0.5 ms per
  // step.  A function of the shape only: launches of at least XM_DGRAD_S2_MIN_BLOCKS blocks (default 6 rounds of the chip).
  static const long long min_blocks = env_int("XM_DGRAD_S2_MIN_BLOCKS", 6 * 768);
  const long long per_sample = (long long)((g.W + 3) / 4 + (g.W + 2) / 4) * (((g.H + 1) / 2 + 63) / 64) * ((g.C + kDgS2Rows - 1) / kDgS2Rows);
  return per_sample * g.N >= min_blocks;
}
static int launch_dgrad_s2(const float *dzdy, const float *f, float *dxo, const Geo &g, hipStream_t st) {
  DgradS2Args a{};
  a.dY = dzdy;
  a.dX = dxo;
  a.dyBytes = (unsigned)((size_t)g.Ho * g.Wo * g.K * g.N * 4);
  a.C = g.C, a.K = g.K, a.H = g.H, a.W = g.W, a.Ho = g.Ho, a.Wo = g.Wo;
  a.pl = g.pl;
  a.nbm = (g.C + kDgS2Rows - 1) / kDgS2Rows;
  a.nkg = g.K / 8;
  a.nmt = ((g.H + 1) / 2 + 63) / 64;
  int blocks = 0;
  for (int px = 0; px < 2; ++px) {
    a.xfirst[px] = ((px - g.pl) % 2 + 2) % 2;               // columns x with (x + pl) % 2 == px
    const int ncols = a.xfirst[px] < g.W ? (g.W - a.xfirst[px] + 1) / 2 : 0;
    a.ncg[px] = (ncols + 1) / 2;
    a.nv[px] = px == 0 ? 3 : 2;                              // v = px, px + 2, (px + 4)
    blocks += a.ncg[px] * a.nmt * a.nbm;
  }
  const size_t fdFloats = (size_t)a.nbm * 2 * 3 * a.nkg * kDgS2Tile;
  WsCarver ws;
  int rc = ws.init(WsCarver::need(fdFloats, 4), st);
  if (rc) return rc;
  float *Fd = ws.take<float>(fdFloats);
  a.Fd = Fd;
  hipLaunchKernelGGL(prep_dgrad_s2_kernel, dim3((unsigned)std::min<size_t>((fdFloats + 255) / 256, 2048)), dim3(256), 0, st, f,
                     Fd, g.C, g.K, a.nbm, a.nkg, g.pl);
  XM_LAUNCH_CHECK();
  {
    ProfScope ps(11 * 100, 2.0 * g.K * (double)g.Ho * g.Wo * g.N * g.R, st,
                 (double)a.dyBytes + 4.0 * g.K * g.R + 4.0 * g.C * (double)g.H * g.W * g.N);
    hipLaunchKernelGGL(conv_dgrad_s2_kernel<1>, dim3((unsigned)(blocks * g.N)), dim3(256), 0, st, a);
  }
  XM_LAUNCH_CHECK();
  return XM_OK;
}

static int conv_dgrad(const float *f, const float *dzdy, float *dxo, const Geo &g, hipStream_t st,
                      const float *accum = nullptr, bool prepare_only = false) {
  struct Cls {
    int a, b, u0, ustep, nU, v0, vstep, nV, Rc, Rp, i0, hi0, PI, j0, wi0, PJ;
    size_t aoff;
  };
  if (dgrad_s2_ok(g, dzdy, dxo, accum)) {
    // the student's conv2: both row parities per wave, whole-line stores (conv_dgrad_s2_kernel); its filter operand is laid
    // out by a 3 MB pass inside the call, so there is nothing to prepare ahead
    if (prepare_only) return XM_OK;
    return launch_dgrad_s2(dzdy, f, dxo, g, st);
  }
  // H-collapsing convolution (FC layer sliding along W only, e.g. the student's fc6: 9x1 filter on a
  // 9 x Wi map): every input row hi is touched by exactly one filter row u = hi, so folding u into
  // the GEMM rows (M = FH*FC) avoids multiplying FH-1 masked-out taps per pixel.
  const bool foldH = g.Ho == 1 && g.FH > 1 && g.FH == g.H && g.pt == 0 && g.pb == 0 && g.dy == 1 && g.sy == 1;
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
      c.Rc = (foldH ? c.nV : c.nU * c.nV) * g.Kg;
      c.Rp = (c.Rc + kBK - 1) / kBK * kBK;
      c.aoff = abytes;
      abytes += WsCarver::need((size_t)g.FC * (foldH ? g.FH : 1) * c.Rp * g.G, 4);
      cls.push_back(c);
    }
  // pixels whose class has no tap (e.g. 1x1 stride 2) receive no gradient
  if (!prepare_only && (!covers_all || cls.empty())) {
    const size_t bytes = sizeof(float) * (size_t)g.H * g.W * g.C * g.N;
    if (accum)
      XM_HIP(hipMemcpyAsync(dxo, accum, bytes, hipMemcpyDeviceToDevice, st));
    else
      XM_HIP(hipMemsetAsync(dxo, 0, bytes, st));
  }
  if (cls.empty()) return XM_OK;
  // scratch for the split-K slab of the largest class under any tile configuration
  size_t slab_max = 0;
  for (size_t i = 0; i < cls.size(); ++i) {
    const Cls &c = cls[i];
    ConvGemmArgs proto{};
    proto.M = foldH ? g.FC * g.FH : g.FC;
    proto.NP = (foldH ? 1 : c.PI) * c.PJ * g.N;
    proto.Rp = c.Rp;
    for (int ci = 0; ci < kNumBaseCfg; ++ci) {
      int sp;
      slab_max = std::max(slab_max, gemm_slab_floats(proto, ci, &sp));
    }
    if (!foldH && c.nU <= 3 && c.nV <= 3 && c.nU * c.nV >= 4 && c.Rc % (kHaloCB * c.nU * c.nV) == 0) {
      proto.Rtrue = c.Rc;
      proto.nU = c.nU;
      proto.nV = c.nV;
      slab_max = std::max(slab_max, halo_slab_floats(proto));
    }
  }
  // transposed filters: from the persistent cache when a current entry exists (or is being built now), else scratch
  const PrepKey pkey = prep_key(f, g, foldH);
  PrepEntry *pe = nullptr;
  bool have_prepared = false;
  if (prepare_only) {
    pe = &g_prep[pkey];
    if (pe->bytes < abytes) {
      if (pe->buf) {
        XM_HIP(hipDeviceSynchronize());
        (void)hipFree(pe->buf);
      }
      pe->buf = nullptr;
      XM_HIP(hipMalloc((void **)&pe->buf, abytes));
      pe->bytes = abytes;
    }
    if (!pe->ev) XM_HIP(hipEventCreateWithFlags(&pe->ev, hipEventDisableTiming));
  } else {
    auto it = g_prep.find(pkey);
    if (it != g_prep.end() && it->second.version == g_param_version && it->second.bytes >= abytes) {
      pe = &it->second;
      have_prepared = true;
      if (pe->st != st) XM_HIP(hipStreamWaitEvent(st, pe->ev, 0));
    }
  }
  WsCarver ws;
  int rc = ws.init((pe ? 0 : abytes) + WsCarver::need(slab_max, 4), st);
  if (rc) return rc;
  char *abase = pe ? (char *)pe->buf : ws.base;
  float *slab = slab_max ? (float *)(ws.base + (pe ? 0 : abytes)) : nullptr;
  const size_t dyTotal = (size_t)g.Ho * g.Wo * g.K * g.N;
  double pair_total = 0;
  for (const Cls &c : cls) pair_total += (double)(foldH ? 1 : c.PI) * c.PJ * g.N * c.Rc;
  // merged launch: 2-4 classes, no filter groups, and enough tiles per class that none would be split
  bool merge = g.G == 1 && !foldH && cls.size() >= 2 && cls.size() <= 4 && g_force_splits == 0 &&
               path_on(kPathDgradMerge);
  {
    // the merged launch never splits K: it needs enough tiles in total to fill the chip by itself
    long long tiles = 0;
    for (const Cls &c : cls) tiles += (long long)((g.FC + 127) / 128) * (((long long)c.PI * c.PJ * g.N + 255) / 256);
    if (tiles < 512) merge = false;
  }
  std::vector<ConvGemmArgs> merged;
  for (size_t ic = 0; ic < cls.size(); ++ic) {
    const Cls &c = cls[ic];
    if (c.nU * c.nV > 63)
      return fail(XM_ENOTSUP, "vl_nnconv: more than 63 spatial taps per stride class is not built");
    // tap table in dY space: r' = iu + nU*(iv + nV*k); u' = (u*dy - a)/sy; ho = i' - u'
    const int up0 = ((c.u0 * g.dy) - c.a) / g.sy, ups = c.ustep * g.dy / g.sy;
    const int vp0 = ((c.v0 * g.dx) - c.b) / g.sx, vps = c.vstep * g.dx / g.sx;
    std::vector<int2> t(c.Rp + 3 * kBK);
    for (int r = 0; r < c.Rp + 3 * kBK; ++r) {
      if (r < c.Rc && foldH) {
        int iv = r % c.nV, k = r / c.nV;
        t[r] = make_int2(4 * (-g.Ho * (vp0 + iv * vps) + g.Ho * g.Wo * k), iv);
      } else if (r < c.Rc) {
        int iu = r % c.nU, iv = (r / c.nU) % c.nV, k = r / (c.nU * c.nV);
        int up = up0 + iu * ups, vp = vp0 + iv * vps;
        t[r] = make_int2(4 * (-up - g.Ho * vp + g.Ho * g.Wo * k), iu + c.nU * iv);
      } else {
        t[r] = make_int2(0, 63);
      }
    }
    const int2 *taps = (const int2 *)cached_device_table(t.data(), t.size() * sizeof(int2));
    if (!taps) return fail(XM_ENOMEM, "vl_nnconv: tap table allocation failed");
    float *At = (float *)(abase + c.aoff);
    for (int grp = 0; grp < g.G; ++grp) {
      float *Ag = At + (size_t)grp * g.FC * (foldH ? g.FH : 1) * c.Rp;
      // pure transpose (see transpose_filter_kernel): 1 x 1 filters, or an FH x 1 filter folded into the GEMM rows
      const int Rrows = g.FC * (foldH ? g.FH : 1);
      const bool pure_t = g.FW == 1 && ((foldH && c.nU == g.FH && c.nV == 1) || (g.FH == 1 && !foldH)) &&
                          c.Rp == g.Kg && (Rrows & 3) == 0 && (g.Kg & 3) == 0 && g.G == 1 &&
                          (((uintptr_t)f | (uintptr_t)Ag) & 15) == 0 && path_on(kPathFastTranspose);
      if (!have_prepared && pure_t) {
        hipLaunchKernelGGL(transpose_filter_kernel, dim3((Rrows + 63) / 64, (g.Kg + 63) / 64), dim3(256), 0, st, f, Ag,
                           g.Kg, Rrows, c.Rp);
        XM_LAUNCH_CHECK();
      } else if (!have_prepared) {
        const int T = g.FH * g.FW;
        const int TS = T <= 14 ? 32 : (T <= 56 ? 16 : 8);
        size_t lds = sizeof(float) * (size_t)TS * (TS * T + 1);
        const int nT = c.nU * c.nV, used = (foldH ? c.nV : nT) * g.Kg;
        PrepDiv pd;
        pd.tsT = make_fastdiv((uint32_t)(TS * T)), pd.T = make_fastdiv((uint32_t)T);
        pd.tsNT = make_fastdiv((uint32_t)std::max(1, TS * nT)), pd.nT = make_fastdiv((uint32_t)std::max(1, nT));
        pd.nU = make_fastdiv((uint32_t)std::max(1, c.nU));
        pd.padc = make_fastdiv((uint32_t)std::max(1, c.Rp - used));
        pd.rows = make_fastdiv((uint32_t)(foldH ? std::max(1, c.nU) : 1));
        hipLaunchKernelGGL(prep_dgrad_filter_kernel, dim3((g.FC + TS - 1) / TS, (g.Kg + TS - 1) / TS),
                           dim3(256), lds, st, f + (size_t)g.R * g.Kg * grp, Ag, g.FH, g.FW, g.FC, g.Kg,
                           c.u0, c.ustep, c.nU, c.v0, c.vstep, c.nV, c.Rp, TS, foldH ? 1 : 0, pd);
        XM_LAUNCH_CHECK();
      }
      if (prepare_only) continue;
      // FC-shaped layer over few pixels (the student's fc7, the SE gates of a trainable teacher): dX = W^T dY is the
      // skinny forward product with the transposed filter bank as its filter
      if (pure_t && !foldH && !accum && g.FH == 1 && g.sy == 1 && g.sx == 1 && (g.pt | g.pb | g.pl | g.pr) == 0 &&
          g.H * g.W == 1 && (g.FC & 3) == 0 && g.N <= 512 && g_force_cfg < 0 && g_force_splits == 0) {
        Geo gs = g;
        gs.C = gs.FC = g.Kg;
        gs.K = gs.Kg = g.FC;
        gs.R = g.Kg;
        if (fc_skinny_ok(gs, true)) {
          int rc = fc_skinny_forward(dzdy, Ag, nullptr, dxo, gs, 0, st);
          if (rc) return rc;
          continue;
        }
      }
      ConvGemmArgs a{};
      a.A = Ag;
      a.lda = c.Rp;
      size_t xoff = (size_t)grp * g.Kg * g.Ho * g.Wo;
      a.X = dzdy + xoff;
      a.xBytes = (unsigned)((dyTotal - xoff) * 4);
      a.Y = dxo + (size_t)grp * g.FC * g.H * g.W;
      a.resid = accum ? accum + (size_t)grp * g.FC * g.H * g.W : nullptr;
      a.taps = taps;
      a.M = foldH ? g.FC * g.FH : g.FC;
      a.Rp = c.Rp;
      a.Rtrue = c.Rc;
      a.PI = foldH ? 1 : c.PI;
      a.PJ = c.PJ;
      a.NP = a.PI * c.PJ * g.N;
      a.divPIJ = make_fastdiv((uint32_t)(a.PI * c.PJ));
      a.divPI = make_fastdiv((uint32_t)a.PI);
      a.gsy = 1;
      a.gsx = 1;
      a.gh0 = c.i0;
      a.gw0 = c.j0;
      a.LimH = g.Ho;
      a.LimW = g.Wo;
      a.xSampleStride = g.Ho * g.Wo * g.K;
      a.nU = foldH ? 1 : c.nU;
      a.nV = c.nV;
      a.du0 = foldH ? 0 : -up0;
      a.dus = -ups;
      a.dv0 = -vp0;
      a.dvs = -vps;
      a.osy = g.sy;
      a.osx = g.sx;
      a.oh0 = c.hi0;
      a.ow0 = c.wi0;
      a.OH = g.H;
      a.oChanStride = g.H * g.W;
      a.oSampleStride = g.H * g.W * g.C;
      a.divMU = make_fastdiv(foldH ? (uint32_t)g.FH : 1u);
      a.oUStride = foldH ? 1 : 0;
      a.vecStore = (!foldH && g.sy == 1 && g.sx == 1 && (c.PI * c.PJ) % 4 == 0 && c.PI == g.H &&
                    c.PJ == g.W && (((uintptr_t)a.Y | (uintptr_t)a.resid) & 15) == 0) ? 1 : 0;
      if (foldH) {
        a.gh0 = 0;  // the single dY row; destination row comes from the GEMM row (m % FH)
        a.oh0 = 0;
        a.osy = 0;
      }
      // algorithmic work of dgrad == forward MACs (2*Ho*Wo*N*K*R), apportioned over the classes by
      // their share of (pixel, tap) pairs; masked-out pairs are not work
      a.algoFlops = 2.0 * g.Ho * g.Wo * (double)g.N * g.Kg * g.R * ((double)a.NP * c.Rc) / pair_total;
      if (merge) {
        merged.push_back(a);
        continue;
      }
      auto run = [&](int ci) {
        int sp;
        gemm_slab_floats(a, ci, &sp);
        ConvGemmArgs aa = a;
        return launch_gemm(aa, 1, ci, sp, slab, st);
      };
      TuneKey key{1, a.M, a.NP, c.Rp, c.nU * 64 + c.nV, g.sy * 16 + g.sx, a.PI, c.PJ, g.Ho};
      int ci = tune_cfg(key, pick_cfg(a.M, a.NP, c.Rp / kBK), st, run, kNumBaseCfg, w8_skip(a.M, a.NP));
      ConvGemmArgs ah = a;
      const int hneed = (g_force_cfg < 0 && g_force_splits == 0 && !foldH) ? halo_setup_padded(ah, g.N) : 0;
      bool hok[4] = {true, halo_var_ok(ah, hneed, 0), halo_var_ok(ah, hneed, 1), halo_var_ok(ah, hneed, 2)};
      if (hok[1] || hok[2] || hok[3]) {
        auto run2 = [&](int h) { return h ? launch_halo(ah, h - 1, slab, st) : run(ci); };
        TuneKey hkey{5, a.M, a.NP, c.Rp, c.nU * 64 + c.nV, g.sy * 16 + g.sx, a.PI, c.PJ, g.Ho};
        int pick;
        if (g_force_halo == 0) pick = 0;
        else if (g_force_halo > 0) pick = forced_halo_variant(hok);
        else pick = tune_challengers(hkey, st, run2, 4, hok, kHaloMargin);
        rc = run2(pick);
      } else {
        rc = run(ci);
      }
      if (rc) return rc;
    }
  }
  if (prepare_only) {
    pe->version = g_param_version;
    pe->st = st;
    XM_HIP(hipEventRecord(pe->ev, st));
    return XM_OK;
  }
  if (merge) {
    // all stride-parity classes in one launch (each alone is a fraction more than one round of the chip)
    auto run = [&](int ci) { return launch_gemm_multi(merged, ci, st); };
    long long npSum = 0;
    int rpMax = 0;
    for (const ConvGemmArgs &a : merged) {
      npSum += a.NP;
      rpMax = std::max(rpMax, a.Rp);
    }
    TuneKey key{3, merged[0].M, (int)std::min<long long>(npSum, 1 << 30), rpMax, (int)merged.size(),
                g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
    int ci = tune_cfg(key, pick_cfg(merged[0].M, npSum, rpMax / kBK), st, run, kNumBaseCfg, w8_skip(merged[0].M, npSum));
    // every class a halo-patch problem (unit-stride gathers of <= 3 x 3 taps in dY space)?  Then the same variant for all
    std::vector<ConvGemmArgs> mh = merged;
    bool hok[4] = {true, g_force_cfg < 0, g_force_cfg < 0, g_force_cfg < 0};
    for (ConvGemmArgs &a : mh) {
      const int need = hok[1] || hok[2] || hok[3] ? halo_setup_padded(a, g.N) : 0;
      for (int v = 0; v < 3; ++v) hok[v + 1] = hok[v + 1] && halo_var_ok(a, need, v);
    }
    if (hok[3] && !hok[2])      // the tall-patch variant may serve classes that would also fit the short one
      ;
    else if (!hok[3]) {         // classes of mixed patch height: let the tall variant take them all if each fits 1024
      bool all = g_force_cfg < 0;
      for (ConvGemmArgs &a : mh) {
        ConvGemmArgs t = a;
        const int need = all ? halo_setup_padded(t, g.N) : 0;
        all = all && need > 0 && need <= 1024 && t.M % 96 == 0;
      }
      hok[3] = all && !hok[2];
    }
    if (hok[1] || hok[2] || hok[3]) {
      auto run2 = [&](int h) { return h ? launch_halo_multi(mh, h - 1, st) : run(ci); };
      TuneKey hkey{6, merged[0].M, (int)std::min<long long>(npSum, 1 << 30), rpMax, (int)merged.size(),
                   g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
      int pick;
      if (g_force_halo == 0) pick = 0;
      else if (g_force_halo > 0) pick = forced_halo_variant(hok);
      else pick = tune_challengers(hkey, st, run2, 4, hok, kHaloMargin);
      rc = run2(pick);
    } else {
      rc = run(ci);
    }
    if (rc) return rc;
  }
  return XM_OK;
}

// one wgrad launch (+ split reduction) with tile configuration ci; `part` has room for 1024 slabs
static int wgrad_run(const float *x, const float *dzdy, float *dfo, const Geo &g, int ci, float *part,
                     hipStream_t st) {
  ci = wgrad_cfg(ci);
  const Cfg &c = kCfgs[ci];
  const int NP = g.Ho * g.Wo * g.N;
  const int nkt = (NP + kBK - 1) / kBK;
  const int nbm = (g.Kg + c.bm() - 1) / c.bm(), nbn = (g.R + c.bn() - 1) / c.bn();
  const int Rn = nbn * c.bn();
  // split the pixel reduction so that the grid fills one round of the chip (no tail round);
  // smaller tiles co-reside more blocks per CU
  const int tiles = nbm * nbn;
  const int slots = 256 * (c.bm() * c.bn() >= 128 * 128 ? 2 : (c.bm() * c.bn() >= 64 * 128 ? 3 : 6));
  int splits = std::max(1, std::min(nkt / 8, slots / tiles));
  splits = std::min(splits, 1024);
  int tps = (nkt + splits - 1) / splits;
  splits = (nkt + tps - 1) / tps;
  const int4 *taps = fwd_taps(g, Rn);
  if (!taps) return fail(XM_ENOMEM, "vl_nnconv: tap table allocation failed");
  const size_t slab = (size_t)g.Kg * g.R;
  for (int grp = 0; grp < g.G; ++grp) {
    WgradArgs a{};
    a.dY = dzdy + (size_t)grp * g.Kg * g.Ho * g.Wo;
    a.X = x + (size_t)grp * g.FC * g.H * g.W;
    float *dst = dfo + (size_t)grp * g.Kg * g.R;
    a.out = splits > 1 ? part : dst;
    a.taps = taps;
    {
      size_t xoff = (size_t)grp * g.FC * g.H * g.W, doff = (size_t)grp * g.Kg * g.Ho * g.Wo;
      a.xBytes = (unsigned)(((size_t)g.H * g.W * g.C * g.N - xoff) * 4);
      a.dyBytes = (unsigned)(((size_t)g.Ho * g.Wo * g.K * g.N - doff) * 4);
    }
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
    {
      const int av = wgrad_av(a);
      ProfScope ps(1 * 100 + ci * 4 + (av == 4 ? 2 : av == 2 ? 1 : 0), 2.0 * g.Kg * (double)NP * g.R, st,
                   (double)a.xBytes + (double)a.dyBytes + 4.0 * g.Kg * g.R);
      launch_wgrad_cfg(ci, a, dim3(nbm * nbn, splits), st);
    }
    XM_LAUNCH_CHECK();
    if (splits > 1) {
      launch_reduce_splits(part, dst, slab, splits, slab, st);
      XM_LAUNCH_CHECK();
    }
  }
  return XM_OK;
}

// ---- filter derivative of the single-channel stem (conv_stem_wgrad_kernel) ------------------------------------------
static bool stem_wgrad_ok(const Geo &g, const float *x, const float *dzdy, bool structural_only = false) {
  const bool off = !path_on(kPathStem) || !path_on(kPathStemWgrad);
  if (!structural_only && (off || g_force_cfg >= 0 || g_force_splits > 0)) return false;
  if (g.C != 1 || g.G != 1 || g.FC != 1 || g.dy != 1 || g.dx != 1) return false;
  if (g.FH > 8 || g.FW > kStemNV || g.FH * g.FW < 16 || g.FH * g.FW > 64 || g.Kg > 96) return false;
  if (g.sy != 1 && g.sy != 2) return false;
  if (g.H % 4 != 0 || g.H > kStemHP - 8 || (((uintptr_t)x | (uintptr_t)dzdy) & 15) != 0) return false;
  if ((g.Ho * g.Wo) % 4 != 0) return false;                                      // dY pixel quads stay inside a sample
  if (g.pt > 4 || 4 * ((g.sy * (g.Ho - 1) - g.pt + 4 + 7) >> 2) + 3 >= kStemHP) return false;
  if (g.Ho < 128) return false;
  if (!structural_only && g_force_stem != 1 && (long long)g.Ho * g.Wo * g.N < 128 * 512) return false;
  return true;
}

// BNP (bnp != NULL): `dzdy` is the convolution's own OUTPUT, the derivative is rebuilt on the fly (StemWgradArgs)
struct StemBnp {
  const float *dP;
  const unsigned char *amax;
  const float *rowc;
  int pHo, pWo;
  float *dbias;      // NULL, or dzdb of the convolution (column R of the partials multiplies ones)
};
static int launch_stem_wgrad(const float *x, const float *dzdy, float *dfo, const Geo &g, float *part, int grid, hipStream_t st,
                             const StemBnp *bnp = nullptr) {
  StemWgradArgs a{};
  a.dY = dzdy;
  a.X = x;
  a.part = part;
  a.M = g.Kg;
  a.R = g.R;
  a.nU = g.FH;
  a.nV = g.FW;
  a.PI = g.Ho;
  a.PJ = g.Wo;
  a.NP = g.Ho * g.Wo * g.N;
  a.divPIJ = make_fastdiv((uint32_t)(g.Ho * g.Wo));
  a.divPI = make_fastdiv((uint32_t)g.Ho);
  a.gsx = g.sx;
  a.gh0 = -g.pt;
  a.gw0 = -g.pl;
  a.LimH = g.H;
  a.LimW = g.W;
  a.xSampleStride = g.H * g.W;
  a.dySampleStride = g.Ho * g.Wo * g.K;
  a.dyChanStride = g.Ho * g.Wo;
  a.onesCol = -1;
  if (bnp) {
    a.dP = bnp->dP;
    a.amax = bnp->amax;
    a.rowc = bnp->rowc;
    a.pHo = bnp->pHo;
    a.pWo = bnp->pWo;
    const size_t pooled = (size_t)bnp->pHo * bnp->pWo * g.K * g.N;
    a.dpBytes = (unsigned)(pooled * 4);
    a.amBytes = (unsigned)pooled;
    a.dyBytes = (unsigned)((size_t)g.Ho * g.Wo * g.K * g.N * 4);
    if (bnp->dbias) a.onesCol = g.R;
  }
  static bool attr_done = false;
  if (!attr_done) {
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemWgSmem));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemWgSmem));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_bnp_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemWgSmemBnp));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_bnp_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemWgSmemBnp));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_bnp_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemWgSmemBnp));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_bnp_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemWgSmemBnp));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_bnp_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemWgSmemBnp));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_bnp_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemWgSmemBnp));
    attr_done = true;
  }
  const int ntiles = (a.NP + 127) / 128;
  {
    // BNP reads the convolution's output once (as the plain kernel reads dY) + the pooled derivative and its table
    const double abytes = 4.0 * g.H * g.W * g.N + 4.0 * a.M * (double)a.NP + 4.0 * a.M * a.R +
                          (bnp ? 5.0 * bnp->pHo * bnp->pWo * (double)g.K * g.N : 0.0);
    ProfScope ps(6 * 100 + (bnp ? 1 : 0), 2.0 * a.M * (double)a.NP * a.R, st, abytes);
    if (bnp) {
      const int ones = a.onesCol < 0 ? 0 : (a.onesCol < 32 ? 1 : 2);
#define XM_BNP_LAUNCH(SY_, ON_) hipLaunchKernelGGL((conv_stem_wgrad_bnp_kernel<SY_, ON_>), dim3(grid), dim3(256), kStemWgSmemBnp, st, a, ntiles)
      if (g.sy == 2) {
        if (ones == 0) XM_BNP_LAUNCH(2, 0); else if (ones == 1) XM_BNP_LAUNCH(2, 1); else XM_BNP_LAUNCH(2, 2);
      } else {
        if (ones == 0) XM_BNP_LAUNCH(1, 0); else if (ones == 1) XM_BNP_LAUNCH(1, 1); else XM_BNP_LAUNCH(1, 2);
      }
#undef XM_BNP_LAUNCH
    } else {
      if (g.sy == 2) hipLaunchKernelGGL(conv_stem_wgrad_kernel<2>, dim3(grid), dim3(256), kStemWgSmem, st, a, ntiles);
      else hipLaunchKernelGGL(conv_stem_wgrad_kernel<1>, dim3(grid), dim3(256), kStemWgSmem, st, a, ntiles);
    }
  }
  XM_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv_stem_wgrad_reduce_kernel, dim3(a.M), dim3(1024), 0, st, part, dfo, grid, a.M, a.R,
                     bnp ? bnp->dbias : nullptr);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

// ---- filter derivative of 3 x 3 / stride 1 / pad 1 layers from an input patch (conv_wgrad_patch_kernel) -------------------
// Hosts that declared XM_EXEC_SINGLE_STREAM only (xm_set_exec_hint; the reference's own call sequence, cnn_train_dag ->
// vl_nnconv one after the other -- the MEX binding sets it; bench.py --serial): alone the kernel is 11 ... 22 % faster than the generic one (profiles/r04/wgrad_patch_bench.txt), a step on one
// stream 1.7 %; launched on a side stream next to the dgrad of the same layer it fills the chip with three 48 KB blocks per CU
// for its whole life and the pair takes LONGER than with the generic kernel (student step at 64: - 1.5 %).
// The choice depends on the shape, the tuning table and that explicit hint -- never on what the process called before.
static bool wgrad_patch_ok(const Geo &g, const float *x, const float *dzdy, hipStream_t st) {
  if (!path_on(kPathWgradPatch) || g_force_cfg >= 0 || g_force_splits > 0) return false;
  // a candidate for hosts that declared one stream, and -- whatever the host does -- for launches of at least
  // XM_WGRAD_PATCH_MIN_STAGES output columns (default 4096: the batch-256 layers, which fill the chip for ~2 ms by
  // themselves; like the eight-wave rule "from 1024 tiles" a function of the shape only)
  static const long long min_stages = env_int("XM_WGRAD_PATCH_MIN_STAGES", 4096);
  if (g_force_wgrad_patch < 0 && !(g_exec_hint & XM_EXEC_SINGLE_STREAM) && (long long)g.N * g.W < min_stages) return false;
  if (g.G != 1 || g.FH != 3 || g.FW != 3 || g.sy != 1 || g.sx != 1 || g.dy != 1 || g.dx != 1) return false;
  if (g.pt != 1 || g.pb != 1 || g.pl != 1 || g.pr != 1) return false;
  if (g.H != 30 || g.Ho != g.H || g.Wo != g.W) return false;          // instantiated row counts (HH)
  if ((((uintptr_t)x | (uintptr_t)dzdy) & 7) != 0) return false;
  if ((size_t)g.H * g.W * g.C * g.N * 4 >= (1ull << 31) || (size_t)g.Ho * g.Wo * g.K * g.N * 4 >= (1ull << 31)) return false;
  return (long long)g.N * g.W >= 64;                                    // enough stages to split
}
static int launch_wgrad_patch(const float *x, const float *dzdy, float *dfo, const Geo &g, float *part, int max_splits,
                              hipStream_t st) {
  WgradPatchArgs a{};
  a.dY = dzdy;
  a.X = x;
  a.xBytes = (unsigned)((size_t)g.H * g.W * g.C * g.N * 4);
  a.dyBytes = (unsigned)((size_t)g.Ho * g.Wo * g.K * g.N * 4);
  a.M = g.Kg;
  a.R = g.R;
  a.ldo = g.R;
  a.C = g.C;
  a.W = g.W;
  a.K = g.K;
  a.nStages = g.N * g.W;
  a.nbm = (g.Kg + 127) / 128;
  a.nbn = (g.R + 127) / 128;
  const int tiles = a.nbm * a.nbn;
  static const int slots = std::max(64, (int)env_int("XM_WGRAD_PATCH_SLOTS", 768));   // one round of 3 blocks per CU
  int splits = std::max(1, std::min(std::min(a.nStages / 8, slots / std::max(1, tiles)), max_splits));
  a.stagesPerSplit = (a.nStages + splits - 1) / splits;
  splits = (a.nStages + a.stagesPerSplit - 1) / a.stagesPerSplit;
  const size_t slab = (size_t)g.Kg * g.R;
  a.splitStride = slab;
  a.out = splits > 1 ? part : dfo;
  {
    ProfScope ps(9 * 100, 2.0 * g.Kg * (double)g.Ho * g.Wo * g.N * g.R, st, (double)a.xBytes + (double)a.dyBytes + 4.0 * slab);
    hipLaunchKernelGGL(conv_wgrad_patch_kernel<30>, dim3(tiles, splits), dim3(256), 0, st, a);
  }
  XM_LAUNCH_CHECK();
  if (splits > 1) {
    launch_reduce_splits(part, dfo, slab, splits, slab, st);
    XM_LAUNCH_CHECK();
  }
  return XM_OK;
}

// ---- filter derivative of 5 x 5 / stride 2 layers from an input patch (conv_wgrad_patch_s2_kernel, round 5) ------------------
// The student's conv2 (44 % of its arithmetic).  Unlike the 3 x 3 patch kernel above this one is a candidate for every caller
// (measured once per shape against the best generic configuration): 53 KB of LDS per block, three blocks per CU.
static int g_force_wgrad_patch_s2 = -1; // test hook (xm_debug_force_wgrad_patch_s2)
static bool wgrad_patch_s2_ok(const Geo &g, const float *x, const float *dzdy) {
  if (!path_on(kPathWgradPatchS2) || g_force_cfg >= 0 || g_force_splits > 0) return false;
  if (g.G != 1 || g.FH != 5 || g.FW != 5 || g.sy != 2 || g.sx != 2 || g.dy != 1 || g.dx != 1) return false;
  if (g.pt < 1 || g.pt > 2 || g.pl < 0 || g.pl > 4) return false;
  if ((g.H & 1) || (g.Ho & 1)) return false;                            // 8-byte loads of row pairs
  if ((((uintptr_t)x | (uintptr_t)dzdy) & 7) != 0) return false;
  // byte offsets are formed in 32-bit arithmetic: both tensors below 2 GiB (904 MB / 585 MB at 256 spectrograms)
  if ((size_t)g.H * g.W * g.C * g.N * 4 >= (1ull << 31) || (size_t)g.Ho * g.Wo * g.K * g.N * 4 >= (1ull << 31)) return false;
  return (long long)g.N * g.Wo * ((g.Ho + 31) / 32) >= 64;              // enough stages to split
}
static int launch_wgrad_patch_s2(const float *x, const float *dzdy, float *dfo, const Geo &g, float *part, int max_splits,
                                 hipStream_t st) {
  WgradPatchS2Args a{};
  a.dY = dzdy;
  a.X = x;
  a.xBytes = (unsigned)((size_t)g.H * g.W * g.C * g.N * 4);
  a.dyBytes = (unsigned)((size_t)g.Ho * g.Wo * g.K * g.N * 4);
  a.M = g.Kg;
  a.R = g.R;
  a.ldo = g.R;
  a.C = g.C;
  a.H = g.H, a.W = g.W, a.Ho = g.Ho, a.Wo = g.Wo, a.K = g.K;
  a.pt = g.pt, a.pl = g.pl;
  a.nSeg = (g.Ho + 31) / 32;
  a.nStages = g.N * g.Wo * a.nSeg;
  a.nbm = (g.Kg + 127) / 128;
  a.nbn = (g.R + 127) / 128;
  const int tiles = a.nbm * a.nbn;
  static const int slots = std::max(64, (int)env_int("XM_WGRAD_PATCH_SLOTS", 768));   // one round of 3 blocks per CU
  int splits = std::max(1, std::min(std::min(a.nStages / 8, slots / std::max(1, tiles)), max_splits));
  a.stagesPerSplit = (a.nStages + splits - 1) / splits;
  splits = (a.nStages + a.stagesPerSplit - 1) / a.stagesPerSplit;
  a.splits = splits;
  const size_t slab = (size_t)g.Kg * g.R;
  a.splitStride = slab;
  a.out = splits > 1 ? part : dfo;
  {
    ProfScope ps(10 * 100, 2.0 * g.Kg * (double)g.Ho * g.Wo * g.N * g.R, st, (double)a.xBytes + (double)a.dyBytes + 4.0 * slab);
    hipLaunchKernelGGL((conv_wgrad_patch_s2_kernel<5, 2>), dim3(tiles * splits), dim3(256), 0, st, a);
  }
  XM_LAUNCH_CHECK();
  if (splits > 1) {
    launch_reduce_splits(part, dfo, slab, splits, slab, st);
    XM_LAUNCH_CHECK();
  }
  return XM_OK;
}

static int conv_wgrad(const float *x, const float *dzdy, float *dfo, const Geo &g, hipStream_t st) {
  // analytic fallback: minimal padded work (split-K supplies the parallelism)
  int fb = 0;
  {
    double best = 1e300;
    for (int i = 0; i < kNumWgradCfg; ++i) {
      const Cfg &cc = kCfgs[i];
      double eff = cfg_eff(cc);
      double cost = (double)((g.Kg + cc.bm() - 1) / cc.bm() * cc.bm()) *
                    ((g.R + cc.bn() - 1) / cc.bn() * cc.bn()) * eff;
      if (cost < best) {
        best = cost;
        fb = i;
      }
    }
  }
  const int NP = g.Ho * g.Wo * g.N;
  const int nkt = (NP + kBK - 1) / kBK;
  const size_t slab = (size_t)g.Kg * g.R;
  const int max_splits = std::max(1, std::min(1024, nkt / 8));
  const bool stem = stem_wgrad_ok(g, x, dzdy);
  const int stem_grid_ = stem ? stem_grid(NP) : 0;
  WsCarver ws;
  int rc = ws.init(WsCarver::need(slab * max_splits, 4) + WsCarver::need((size_t)stem_grid_ * 96 * 64, 4), st);
  if (rc) return rc;
  float *part = ws.take<float>(slab * max_splits);
  float *spart = stem ? ws.take<float>((size_t)stem_grid_ * 96 * 64) : nullptr;
  auto run = [&](int ci) { return wgrad_run(x, dzdy, dfo, g, ci, part, st); };
  TuneKey key{2, g.Kg, NP, g.R, g.G, g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
  int ci = tune_cfg(key, fb, st, run, kNumWgradCfg);
  if (stem) {
    // the single-channel stem's own wgrad kernel against the best generic configuration (measured once per shape)
    auto run2 = [&](int h) { return h ? launch_stem_wgrad(x, dzdy, dfo, g, spart, stem_grid_, st) : run(ci); };
    bool sok[2] = {true, true};
    TuneKey skey{8, g.Kg, NP, g.R, g.G, g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
    const int pick = g_force_stem >= 0 ? g_force_stem : tune_challengers(skey, st, run2, 2, sok, kHaloMargin);
    return run2(pick);
  }
  if (wgrad_patch_ok(g, x, dzdy, st)) {
    // the patch kernel against the best generic configuration (measured once per shape; it removes work -- gathers and
    // address arithmetic -- so it does not have to clear the margin the equal-work challengers need: DESIGN.md 2.1f)
    auto run3 = [&](int h) { return h ? launch_wgrad_patch(x, dzdy, dfo, g, part, max_splits, st) : run(ci); };
    bool pok[2] = {true, true};
    TuneKey pkey{9, g.Kg, NP, g.R, g.G, g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
    const int pick = g_force_wgrad_patch >= 0 ? g_force_wgrad_patch : tune_challengers(pkey, st, run3, 2, pok, 0.01f);
    return run3(pick);
  }
  if (wgrad_patch_s2_ok(g, x, dzdy)) {
    auto run4 = [&](int h) { return h ? launch_wgrad_patch_s2(x, dzdy, dfo, g, part, max_splits, st) : run(ci); };
    bool pok[2] = {true, true};
    TuneKey pkey{10, g.Kg, NP, g.R, g.G, g.sy * 16 + g.sx, g.FH * 64 + g.FW, g.H, g.W};
    const int pick = g_force_wgrad_patch_s2 >= 0 ? g_force_wgrad_patch_s2 : tune_challengers(pkey, st, run4, 2, pok, 0.01f);
    return run4(pick);
  }
  return run(ci);
}

// ---- conv1 -> bnorm -> relu -> pool through the Gram matrix of the input patches (stem_pool_kernels.h) -------------------
static bool stem_pool_ok(const Geo &g, const float *x) {
  if (!path_on(kPathStem)) return false;
  if (g.C != 1 || g.G != 1 || g.FC != 1 || g.dy != 1 || g.dx != 1) return false;
  if (g.FH > 8 || g.FW > kStemNV || g.FH * g.FW < 16 || g.FH * g.FW > 63 || g.Kg > 96) return false;
  if ((g.sy != 1 && g.sy != 2) || g.sx + g.FW > kSpNC) return false;      // two output columns sit on <= 9 source columns
  if (g.H % 4 != 0 || g.H > kStemHP - 8 || ((uintptr_t)x & 15) != 0) return false;
  if (g.pt > 4 || 4 * ((g.sy * (g.Ho - 1) - g.pt + 4 + 7) >> 2) + 3 >= kStemHP) return false;
  return g.Ho >= 32;
}

static int stem_pool_units(const Geo &g) { return g.N * ((g.Wo + 1) / 2) * ((g.Ho + 255) / 256); }
static void stem_pool_args(StemPoolArgs &a, const float *x, const Geo &g) {
  a.X = x;
  a.M = g.Kg;
  a.R = g.R;
  a.nU = g.FH;
  a.nV = g.FW;
  a.PI = g.Ho;
  a.PJ = g.Wo;
  const int G = (g.Ho + 255) / 256;
  a.divJG = make_fastdiv((uint32_t)(((g.Wo + 1) / 2) * G));
  a.divG = make_fastdiv((uint32_t)G);
  a.gsx = g.sx;
  a.gh0 = -g.pt;
  a.gw0 = -g.pl;
  a.LimH = g.H;
  a.LimW = g.W;
  a.xSampleStride = g.H * g.W;
}

static int stem_pool_grid(int nunits, int occ) { return std::min(256 * occ, (nunits + 7) / 8 * 8); }
static size_t stem_gram_need(const Geo &g) {
  return WsCarver::need((size_t)stem_pool_grid(stem_pool_units(g), 3) * 64 * 64, 4);
}

// G (fp64 [64][64]) of the patches of x; scratch from the caller's carver
static int launch_stem_gram(WsCarver &ws, const float *x, const Geo &g, double *gram, hipStream_t st) {
  StemPoolArgs a{};
  stem_pool_args(a, x, g);
  const int nunits = stem_pool_units(g), grid = stem_pool_grid(nunits, 3);
  a.part = ws.take<float>((size_t)grid * 64 * 64);
  static bool attr_done = false;
  if (!attr_done) {
    XM_HIP(hipFuncSetAttribute((const void *)stem_gram_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kSpGramSmem));
    XM_HIP(hipFuncSetAttribute((const void *)stem_gram_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kSpGramSmem));
    attr_done = true;
  }
  {
    // three of the four 32 x 32 tiles of the padded 64 x 64 product are computed
    ProfScope ps(13 * 100, 2.0 * (g.R + 1) * (g.R + 1) * (double)g.Ho * g.Wo * g.N, st, 4.0 * g.H * g.W * g.N);
    if (g.sy == 2) hipLaunchKernelGGL(stem_gram_kernel<2>, dim3(grid), dim3(256), kSpGramSmem, st, a, nunits);
    else hipLaunchKernelGGL(stem_gram_kernel<1>, dim3(grid), dim3(256), kSpGramSmem, st, a, nunits);
  }
  XM_LAUNCH_CHECK();
  hipLaunchKernelGGL(stem_gram_reduce_kernel, dim3(64), dim3(1024), 0, st, a.part, gram, grid);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

}  // namespace xm

using namespace xm;

extern "C" {

// ONE entry for the test / tools switches (round-5 review: the library exported eleven xm_debug_* functions that set
// process-global state).  Returns the previous value, INT_MIN for an unknown key.  Keys and values: include/xmodal_prof.h.
int xm_debug_set(const char *key, int value) {
  auto tri = [](int &g, int v) { int old = g; g = v < 0 ? -1 : (v ? 1 : 0); return old; };
  const std::string k = key ? key : "";
  if (k == "conv_cfg") { int old = g_force_cfg; g_force_cfg = (value >= 0 && value < kNumCfg) ? value : -1; return old; }
  if (k == "conv_splits") { int old = g_force_splits; g_force_splits = value > 0 ? value : 0; return old; }
  if (k == "conv_halo") { int old = g_force_halo; g_force_halo = value < 0 ? -1 : std::min(value, 3); return old; }
  if (k == "conv_stem") return tri(g_force_stem, value);
  if (k == "conv_stem3") return tri(g_force_stem3, value);
  if (k == "wgrad_patch") return tri(g_force_wgrad_patch, value);
  if (k == "wgrad_patch_s2") return tri(g_force_wgrad_patch_s2, value);
  if (k == "dgrad_s2") return tri(g_force_dgrad_s2, value);
  if (k == "comm_single") return comm_force_single(value);
  return INT_MIN;
}
int xm_debug_get(const char *key) {
  const std::string k = key ? key : "";
  if (k == "num_conv_cfgs") return kNumCfg;
  if (k == "conv_cfg") return g_force_cfg;
  if (k == "conv_splits") return g_force_splits;
  if (k == "conv_halo") return g_force_halo;
  if (k == "conv_stem") return g_force_stem;
  if (k == "conv_stem3") return g_force_stem3;
  if (k == "wgrad_patch") return g_force_wgrad_patch;
  if (k == "wgrad_patch_s2") return g_force_wgrad_patch_s2;
  if (k == "dgrad_s2") return g_force_dgrad_s2;
  return INT_MIN;
}

int xm_set_exec_hint(unsigned flags) {
  if (flags & ~(unsigned)XM_EXEC_SINGLE_STREAM) return fail(XM_EINVAL, "xm_set_exec_hint: unknown flag bits 0x%x", flags);
  g_exec_hint = flags;
  return XM_OK;
}
unsigned xm_get_exec_hint(void) { return g_exec_hint; }


// ---- persistent tuning table (include/xmodal.h) ----------------------------------------------
int xm_tune_load(const char *path) {
  g_tune_loaded = true;
  return tune_load_file(path && path[0] ? path : tune_default_path().c_str());
}
int xm_tune_save(const char *path) {
  tune_load_once();   // merge with what is on disk
  std::string p = path && path[0] ? std::string(path) : tune_default_path();
  std::string tmp = p + ".tmp";
  FILE *fp = fopen(tmp.c_str(), "w");
  if (!fp) return fail(XM_EINVAL, "xm_tune_save: cannot write %s", tmp.c_str());
  fprintf(fp, "xmodal-tune 1 rev=%d cfgs=%d\n", XM_TUNE_REV, kNumCfg);
  for (auto &kv : g_tuned) {
    const TuneKey &k = kv.first;
    fprintf(fp, "%d %d %d %d %d %d %d %d %d %d\n", k.kind, k.M, k.NP, k.Rp, k.mode, k.a, k.b, k.c, k.d, kv.second);
  }
  fclose(fp);
  if (rename(tmp.c_str(), p.c_str()) != 0) return fail(XM_EINVAL, "xm_tune_save: rename to %s failed", p.c_str());
  g_tune_new = 0;
  return XM_OK;
}
int xm_tune_entries(int *total, int *unsaved) {
  tune_load_once();
  if (total) *total = (int)g_tuned.size();
  if (unsaved) *unsaved = g_tune_new;
  return XM_OK;
}
// on = 1: every block (< 4096) of every later conv_gemm launch records {first clock, last clock, HW_ID, XCC_ID};
// out (4 * nblocks words) receives the records of the most recent launch (caller synchronises first).
// Debugging / tools only.
#ifdef XM_DEBUG_CYCLES   // (tools build only: XM_DEBUG_CYCLES=1 python -m mcncrossmodalemotions_amd.build)
int xm_debug_conv_cycles(int on, unsigned long long *out, int nblocks) {
  const size_t bytes = 4096 * 4 * sizeof(unsigned long long);
  if (on && !g_dbg_cycles) {
    if (hipMalloc((void **)&g_dbg_cycles, bytes) != hipSuccess) return XM_ENOMEM;
    (void)hipMemset(g_dbg_cycles, 0, bytes);
  }
  if (out && g_dbg_cycles)
    (void)hipMemcpy(out, g_dbg_cycles, (size_t)std::min(nblocks, 4096) * 32, hipMemcpyDeviceToHost);
  if (!on && g_dbg_cycles) {
    (void)hipFree(g_dbg_cycles);
    g_dbg_cycles = nullptr;
  }
  return XM_OK;
}
#endif

// ---- include/xmodal_prof.h ------------------------------------------------------------------
int xm_prof_enable(int on) {
  if (on) {
    for (auto &r : g_prof) {
      g_event_pool.push_back(r.start);
      g_event_pool.push_back(r.stop);
    }
    g_prof.clear();
  }
  g_prof_on = on != 0;
  return XM_OK;
}

// Aggregates the recorded launches by kernel instantiation.  Caller must have synchronised the
// stream(s).  Returns the number of distinct kernels; fills up to `cap` entries.
int xm_prof_collect(int cap, int *keys, double *total_ms, double *total_flops, long long *launches) {
  std::vector<int> ks;
  std::vector<double> ms, fl;
  std::vector<long long> cnt;
  for (auto &r : g_prof) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) continue;
    size_t i = 0;
    for (; i < ks.size(); ++i)
      if (ks[i] == r.key) break;
    if (i == ks.size()) {
      ks.push_back(r.key);
      ms.push_back(0);
      fl.push_back(0);
      cnt.push_back(0);
    }
    ms[i] += t;
    fl[i] += r.flops;
    cnt[i] += 1;
  }
  for (size_t i = 0; i < ks.size() && (int)i < cap; ++i) {
    keys[i] = ks[i];
    total_ms[i] = ms[i];
    total_flops[i] = fl[i];
    launches[i] = cnt[i];
  }
  return (int)ks.size();
}

// algorithmic bytes (every operand once) summed per kernel instantiation, same order / keys as xm_prof_collect
int xm_prof_collect_bytes(int cap, int *keys, double *total_bytes) {
  std::vector<int> ks;
  std::vector<double> by;
  for (auto &r : g_prof) {
    size_t i = 0;
    for (; i < ks.size(); ++i)
      if (ks[i] == r.key) break;
    if (i == ks.size()) {
      ks.push_back(r.key);
      by.push_back(0);
    }
    by[i] += r.bytes;
  }
  for (size_t i = 0; i < ks.size() && (int)i < cap; ++i) {
    keys[i] = ks[i];
    total_bytes[i] = by[i];
  }
  return (int)ks.size();
}

// human-readable kernel name of a profiler key, matching the rocprofv3 kernel-trace name
int xm_prof_kernel_name(int key, char *buf, int len) {
  int kind = key / 100;
  if (kind == 3 || kind == 4) {
    const int v = key % 100;
    snprintf(buf, len, "%s<%s, %d>", kind == 3 ? "conv_halo_kernel" : "conv_halo_multi_kernel",
             v == 0 ? "2, 2, 2, 2" : "3, 1, 1, 4", v == 2 ? 1024 : 512);
    return XM_OK;
  }
  if (kind == 9) {
    snprintf(buf, len, "conv_wgrad_patch_kernel<30>");
    return XM_OK;
  }
  if (kind == 10) {
    snprintf(buf, len, "conv_wgrad_patch_s2_kernel<5, 2>");
    return XM_OK;
  }
  if (kind == 12) {
    snprintf(buf, len, key % 100 ? "conv_stem3_kernel<2>" : "conv_stem3_kernel<1>");
    return XM_OK;
  }
  if (kind == 11) {
    snprintf(buf, len, "conv_dgrad_s2_kernel<1>");
    return XM_OK;
  }
  if (kind == 13 || kind == 14 || kind == 15) {
    snprintf(buf, len, kind == 13 ? "stem_gram_kernel<2>" : kind == 14 ? (key % 100 ? "conv_stem_wgrad_pool_kernel<2, true>" : "conv_stem_wgrad_pool_kernel<2, false>") : "conv_stem_bnpool_fwd_kernel");
    return XM_OK;
  }
  if (kind == 5 || kind == 6) {
    snprintf(buf, len, kind == 5 ? "conv_stem_kernel<2>" : (key % 100 ? "conv_stem_wgrad_bnp_kernel<2, 2>" : "conv_stem_wgrad_kernel<2>"));
    return XM_OK;
  }
  int ci = (kind == 0 || kind == 2) ? (key % 100) / 2 : (key % 100) / 4;   // keys: ProfScope call sites (ci * 2 [+ mode] / ci * 4 + vec)
  if (ci < 0 || ci >= kNumCfg) return XM_EINVAL;
  const Cfg &c = kCfgs[ci];
  if (kind == 0 && is_dma_cfg(ci))
    snprintf(buf, len, "conv_gemm_dma_kernel<%d, %d, %d, %d, %d>", c.tm, c.tn, c.wgm, c.wgn,
             ci == kNumBaseCfg ? XM_DMA_NST7 : ci == kNumBaseCfg + 1 ? 3 : ci == kNumBaseCfg + 2 ? XM_DMA_NST9 : XM_DMA_NST10);
  else if (kind == 0)
    snprintf(buf, len, "conv_gemm_kernel<%d, %d, %d, %d, %d>", c.tm, c.tn, c.wgm, c.wgn, key % 2);
  else if (kind == 2)
    snprintf(buf, len, "conv_gemm_multi_kernel<%d, %d, %d, %d>", c.tm, c.tn, c.wgm, c.wgn);
  else
    snprintf(buf, len, "conv_wgrad_kernel<%d, %d, %d, %d, %d>", c.tm, c.tn, c.wgm, c.wgn, 1 << (key % 4));
  return XM_OK;
}

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
  if ((flags & XM_FUSE_RELU) && (flags & XM_FUSE_SIGMOID))
    return fail(XM_EINVAL, "vl_nnconv(fused): relu and sigmoid are exclusive");
  hipStream_t st = (hipStream_t)stream;
  if (!residual && fc_skinny_ok(g))
    return fc_skinny_forward(x, f, b, y, g, (flags & XM_FUSE_SIGMOID) ? 2 : ((flags & XM_FUSE_RELU) ? 1 : 0), st, scale,
                             shift);
  rc = conv_forward(x, f, b, y, g, scale, shift, residual, (flags & XM_FUSE_RELU) ? 1 : 0, st);
  if (rc || !(flags & XM_FUSE_SIGMOID)) return rc;
  return xm_nnsigmoid(y, (size_t)g.Ho * g.Wo * g.K * g.N, nullptr, y, stream);  // in place (elementwise)
}

int xm_nnconv_forward_moments(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                              int FC, int K, const float *b, float *y, int sy, int sx, int pt, int pb,
                              int pl, int pr, int dy, int dx, float epsilon, float *moments_out, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, C, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx);
  if (rc) return rc;
  if (!x || !f || !y || !moments_out) return fail(XM_EINVAL, "vl_nnconv(+moments): NULL tensor");
  hipStream_t st = (hipStream_t)stream;
  if (fc_skinny_ok(g)) {
    rc = fc_skinny_forward(x, f, b, y, g, 0, st);
    return rc ? rc : bn_batch_moments(y, g.Ho, g.Wo, g.K, g.N, epsilon, moments_out, st);
  }
  return conv_forward(x, f, b, y, g, nullptr, nullptr, nullptr, 0, st, moments_out, epsilon);
}

int xm_nnconv_forward_gated(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                            int FC, int K, const float *b, float *y, int sy, int sx, int pt, int pb,
                            int pl, int pr, int dy, int dx, const float *scale, const float *shift,
                            const float *gate, const float *residual, int flags, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, C, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx);
  if (rc) return rc;
  if (!x || !f || !y || !gate) return fail(XM_EINVAL, "vl_nnconv(gated): NULL tensor");
  if ((scale == nullptr) != (shift == nullptr))
    return fail(XM_EINVAL, "vl_nnconv(gated): scale and shift must be given together");
  if (flags & XM_FUSE_SIGMOID) return fail(XM_ENOTSUP, "vl_nnconv(gated): sigmoid epilogue is not built");
  return conv_forward(x, f, b, y, g, scale, shift, residual, (flags & XM_FUSE_RELU) ? 1 : 0, (hipStream_t)stream,
                      nullptr, 0.f, gate);
}

int xm_nnconv_forward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                      int FC, int K, const float *b, float *y, int sy, int sx, int pt, int pb,
                      int pl, int pr, int dy, int dx, void *stream) {
  return xm_nnconv_forward_fused(x, H, W, C, N, f, FH, FW, FC, K, b, y, sy, sx, pt, pb, pl, pr, dy,
                                 dx, nullptr, nullptr, nullptr, 0, stream);
}

int xm_nnconv_prepare_backward(int H, int W, int C, int N, const float *f, int FH, int FW, int FC, int K,
                               int sy, int sx, int pt, int pb, int pl, int pr, int dy, int dx, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, C, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx);
  if (rc) return rc;
  if (!f) return fail(XM_EINVAL, "vl_nnconv: F is NULL");
  return conv_dgrad(f, nullptr, nullptr, g, (hipStream_t)stream, nullptr, true);
}

int xm_params_changed(void) {
  ++g_param_version;
  return XM_OK;
}

// dzdf (+ dzdb) of a first-layer convolution whose output feeds vl_nnbnorm -> vl_nnrelu -> vl_nnpool('max') and nothing
// else (the student's conv1 -> bn1 -> relu1 -> pool1, emoVoxZoo.m:50-62 / Appendix B.1), together with the bnorm's dg / db:
// the bnorm's DZDX -- the widest tensor of the backward pass, 462 MB at 32 spectrograms -- is rebuilt inside the
// filter-derivative kernel from the pooled derivative and the routing table instead of being written by
// xm_nnbnorm_relu_pool_backward and read back by xm_nnconv_backward.  XM_ENOTSUP when the geometry is not the one the
// kernel is written for (single-channel stem, 3 x 3 / stride-2 unpadded pooling): the caller then makes the two calls.
int xm_nnconv_backward_filter_bnrelupool(const float *x, int H, int W, int C, int N, int FH, int FW, int FC, int K, int sy,
                                         int sx, int pt, int pb, int pl, int pr, int dy, int dx, const float *y,
                                         const float *bn_g, const float *bn_b, const float *moments, int train, int ph,
                                         int pw, int psy, int psx, int ppt, int ppb, int ppl, int ppr,
                                         const unsigned char *argmax, const float *y_pool, const float *dzdy_pool,
                                         float *df_out, float *dbias_out, float *dg_out, float *db_out, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, C, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx);
  if (rc) return rc;
  if (!x || !y || !bn_g || !bn_b || !moments || !argmax || !dzdy_pool || !df_out)
    return fail(XM_EINVAL, "vl_nnconv(filter derivative through bnorm+relu+pool): NULL tensor");
  hipStream_t st = (hipStream_t)stream;
  const int pHo = out_size(g.Ho, ppt, ppb, ph, 1, psy), pWo = out_size(g.Wo, ppl, ppr, pw, 1, psx);
  const bool ok = stem_wgrad_ok(g, x, y, true) && g.G == 1 && g.K == g.Kg && y_pool != nullptr && ph == 3 && pw == 3 &&
                  psy == 2 && psx == 2 && ppt == 0 && ppb == 0 && ppl == 0 && ppr == 0 && (g.Ho & 1) == 0 && pHo >= 1 &&
                  pWo >= 1 && (!dbias_out || g.R < 64) && !too_big((long long)pHo * pWo, g.K, g.N, 4) &&
                  g_force_stem != 0;
  if (!ok)
    return fail(XM_ENOTSUP, "vl_nnconv(filter derivative through bnorm+relu+pool): geometry not covered by the fused kernel");
  const int grid = stem_grid(g.Ho * g.Wo * g.N);
  WsCarver ws;
  rc = ws.init(WsCarver::need((size_t)6 * g.K, 4) + WsCarver::need((size_t)grid * 96 * 64, 4) + bnpool_sums_need(g.K, g.N), st);
  if (rc) return rc;
  float *rowc = ws.take<float>((size_t)6 * g.K);
  float *spart = ws.take<float>((size_t)grid * 96 * 64);
  rc = bnpool_backward_sums(ws, y, g.Ho, g.Wo, g.K, g.N, bn_g, bn_b, moments, train, ph, pw, psy, psx, ppt, ppb, ppl, ppr,
                            argmax, y_pool, dzdy_pool, dg_out, db_out, rowc, st);
  if (rc) return rc;
  StemBnp bnp{dzdy_pool, argmax, rowc, pHo, pWo, dbias_out};
  return launch_stem_wgrad(x, y, df_out, g, spart, grid, st, &bnp);
}

// G = P~' P~ of the im2col patches (+ a column of ones) of a single-channel first-layer convolution: fp64 [64][64],
// row / column t = u + FH v (t < FH FW), t = FH FW: the ones column; entries beyond are zero.
int xm_stem_gram(const float *x, int H, int W, int N, int FH, int FW, int sy, int sx, int pt, int pb, int pl, int pr,
                 double *gram, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, 1, N, FH, FW, 1, 1, sy, sx, pt, pb, pl, pr, 1, 1);
  if (rc) return rc;
  if (!x || !gram) return fail(XM_EINVAL, "xm_stem_gram: NULL tensor");
  if (!stem_pool_ok(g, x)) return fail(XM_ENOTSUP, "xm_stem_gram: geometry not covered");
  hipStream_t st = (hipStream_t)stream;
  WsCarver ws;
  rc = ws.init(stem_gram_need(g), st);
  if (rc) return rc;
  return launch_stem_gram(ws, x, g, gram, st);
}

// Batch moments [mean, sqrt(var + eps)] of Y = vl_nnconv(X, F, B) for a single-channel first layer, from G alone
int xm_stem_gram_moments(const double *gram, const float *f, const float *b, int FH, int FW, int K, float epsilon,
                         float *moments_out, void *stream) {
  if (!gram || !f || !moments_out) return fail(XM_EINVAL, "xm_stem_gram_moments: NULL tensor");
  if (FH * FW < 1 || FH * FW > 63 || K < 1) return fail(XM_EINVAL, "xm_stem_gram_moments: bad shape");
  hipLaunchKernelGGL(stem_gram_moments_kernel, dim3(K), dim3(64), 0, (hipStream_t)stream, gram, f, b, K, FH * FW,
                     (double)epsilon, moments_out);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

// Y_POOL, ARGMAX, MOMENTS of vl_nnpool(vl_nnrelu(vl_nnbnorm(vl_nnconv(X, F, B), G, BB)), [3 3], 'stride', 2) for a
// single-channel first layer in one kernel (conv_stem_bnpool_fwd_kernel): the convolution's output is never written.
// Train mode (moments_in NULL): G = xm_stem_gram(X) is left in `gram` (fp64 [64][64], caller-owned: the backward call
// takes it) and the batch moments come from it.  The table marks windows whose maximum did not pass the ReLU with 255.
int xm_nnconv_bnorm_relu_pool_forward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW, int FC,
                                      int K, const float *bias, int sy, int sx, int pt, int pb, int pl, int pr, int dy, int dx,
                                      const float *bn_g, const float *bn_b, float epsilon, const float *moments_in, int ph,
                                      int pw, int psy, int psx, int ppt, int ppb, int ppl, int ppr, double *gram,
                                      float *y_pool, unsigned char *argmax, float *moments_out, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, C, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx);
  if (rc) return rc;
  if (!x || !f || !bn_g || !bn_b || !y_pool || !argmax || (!moments_in && (!gram || !moments_out)))
    return fail(XM_EINVAL, "vl_nnconv + bnorm + relu + pool (fused forward): NULL tensor");
  hipStream_t st = (hipStream_t)stream;
  const int pHo = out_size(g.Ho, ppt, ppb, ph, 1, psy), pWo = out_size(g.Wo, ppl, ppr, pw, 1, psx);
  const long long pooled = (long long)pHo * pWo * g.K * g.N;
  const bool ok = stem_pool_ok(g, x) && g.K == g.Kg && g.sy == 2 && g.sx == 2 && g.FH <= 7 && g.FW == kStemNV && g.K % 8 == 0 &&
                  ph == 3 && pw == 3 && psy == 2 && psx == 2 && ppt == 0 && ppb == 0 && ppl == 0 && ppr == 0 && pHo >= 1 &&
                  pWo >= 1 && pooled < (1LL << 30) && g_force_stem != 0 && (((uintptr_t)y_pool) & 3) == 0;
  if (!ok)
    return fail(XM_ENOTSUP, "vl_nnconv + bnorm + relu + pool (fused forward): geometry not covered by the fused kernel");
  const float *moments = moments_in;
  if (!moments_in) {
    WsCarver ws;
    rc = ws.init(stem_gram_need(g), st);
    if (rc) return rc;
    rc = launch_stem_gram(ws, x, g, gram, st);
    if (rc) return rc;
    hipLaunchKernelGGL(stem_gram_moments_kernel, dim3(g.K), dim3(64), 0, st, gram, f, bias, g.K, g.R, (double)epsilon,
                       moments_out);
    XM_LAUNCH_CHECK();
    moments = moments_out;
  }
  StemFwdArgs a{};
  a.X = x, a.F = f, a.bias = bias, a.bn_g = bn_g, a.bn_b = bn_b, a.moments = moments;
  a.Y = y_pool, a.amax = argmax;
  a.M = g.K, a.R = g.R, a.nU = g.FH, a.nV = g.FW;
  a.PI = g.Ho, a.PJ = g.Wo, a.pHo = pHo, a.pWo = pWo;
  a.NS = (pHo + 62) / 63;
  const int slots = 256 * XM_SF_OCC * 4;
  // segments of window columns per (sample, strip, row tile): the count that minimises (rounds of the resident waves) x
  // (output columns of a unit, one of them computed twice per segment).  A partly filled last round costs a whole one:
  // 11 segments at 32 spectrograms = 2112 units for 2048 waves ran 285 us, 10 segments 206 us; at 256 spectrograms one
  // segment leaves a quarter of the waves idle and two make 1.5 rounds, four make exactly three.
  {
    const long long base = (long long)g.N * a.NS * 3;
    long long best = -1;
    a.SG = 1;
    for (int sg = 1; sg <= std::min(16, pWo); ++sg) {
      const long long rounds = (base * sg + slots - 1) / slots;
      const long long cost = rounds * (2 * ((pWo + sg - 1) / sg) + 1);
      if (best < 0 || cost < best) best = cost, a.SG = sg;
    }
  }
  a.nunits = g.N * a.NS * a.SG * 3;
  a.div3 = make_fastdiv(3u), a.divSG = make_fastdiv((uint32_t)a.SG), a.divNS = make_fastdiv((uint32_t)a.NS);
  a.gh0 = -g.pt, a.gw0 = -g.pl, a.LimH = g.H, a.LimW = g.W, a.xSampleStride = g.H * g.W;
  a.yBytes = (unsigned)(pooled * 4), a.amBytes = (unsigned)pooled;
  static bool attr_done = false;
  if (!attr_done) {
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_bnpool_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSfSmem));
    attr_done = true;
  }
  const int grid = std::min(256 * XM_SF_OCC, (a.nunits + 3) / 4);
  {
    ProfScope ps(15 * 100, 2.0 * g.K * (double)g.Ho * g.Wo * g.N * g.R, st, 4.0 * g.H * g.W * g.N + 5.0 * (double)pooled);
    hipLaunchKernelGGL(conv_stem_bnpool_fwd_kernel, dim3(grid), dim3(256), kSfSmem, st, a);
  }
  XM_LAUNCH_CHECK();
  return XM_OK;
}

// xm_nnconv_backward_filter_bnrelupool without a pass over the convolution's output (stem_pool_kernels.h): the
// filter bank F (and bias B) take Y's place; `gram` = xm_stem_gram of X (NULL: computed here).
int xm_nnconv_backward_filter_bnrelupool_gram(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                                              int FC, int K, const float *bias, int sy, int sx, int pt, int pb, int pl,
                                              int pr, int dy, int dx, const float *bn_g, const float *moments, int train,
                                              int ph, int pw, int psy, int psx, int ppt, int ppb, int ppl, int ppr,
                                              const unsigned char *argmax, const float *y_pool, const float *dzdy_pool,
                                              const double *gram, float *df_out, float *dbias_out, float *dg_out,
                                              float *db_out, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, C, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx);
  if (rc) return rc;
  if (!x || !f || !bn_g || !moments || !argmax || !dzdy_pool || !df_out)
    return fail(XM_EINVAL, "vl_nnconv(filter derivative through bnorm+relu+pool, gram): NULL tensor");
  hipStream_t st = (hipStream_t)stream;
  const int pHo = out_size(g.Ho, ppt, ppb, ph, 1, psy), pWo = out_size(g.Wo, ppl, ppr, pw, 1, psx);
  const long long pooled = (long long)pHo * pWo * g.K * g.N;
  const bool ok = stem_pool_ok(g, x) && g.K == g.Kg && ph == 3 && pw == 3 && psy == 2 && psx == 2 && ppt == 0 && ppb == 0 &&
                  ppl == 0 && ppr == 0 && pHo >= 4 && pWo >= 1 && pooled < (1LL << 30) && g_force_stem != 0 &&
                  ((((uintptr_t)dzdy_pool | (uintptr_t)y_pool) & 3) == 0);   // (y_pool NULL: the table marks closed windows itself)
  if (!ok)
    return fail(XM_ENOTSUP, "vl_nnconv(filter derivative through bnorm+relu+pool, gram): geometry not covered by the fused kernel");
  StemPoolArgs a{};
  stem_pool_args(a, x, g);
  // wave units (sample, column pair, 64-row chunk pair) x 3 row tiles of filters; grid: whole XCD rows of blocks whose
  // waves split evenly over the row tiles (a multiple of 24 blocks), two blocks per CU
  const int ncp = (g.Ho + 63) / 64, npair = (g.Wo + 1) / 2;
  const int nunits = g.N * npair * ncp;
  a.divJG = make_fastdiv((uint32_t)npair);
  a.divG = make_fastdiv((uint32_t)ncp);
  const int grid = std::min(504, ((nunits * 3 + 3) / 4 + 23) / 24 * 24);      // two blocks per CU, 63 per XCD
  const int nwc = grid * 4 / 3;
  WsCarver ws;
  rc = ws.init(WsCarver::need((size_t)grid * 4 * 32 * 64, 4) + WsCarver::need(64 * 64, 8) + stem_gram_need(g), st);
  if (rc) return rc;
  a.part = ws.take<float>((size_t)grid * 4 * 32 * 64);
  double *gr = ws.take<double>(64 * 64);
  if (train && !gram) {
    rc = launch_stem_gram(ws, x, g, gr, st);
    if (rc) return rc;
    gram = gr;
  }
  a.dP = dzdy_pool;
  a.yP = y_pool ? y_pool : dzdy_pool;
  a.amax = argmax;
  a.pHo = pHo;
  a.pWo = pWo;
  a.dpBytes = (unsigned)(pooled * 4);
  a.amBytes = (unsigned)pooled;
  static bool attr_done = false;
  if (!attr_done) {
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_pool_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSp3Smem));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_pool_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSp3Smem));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_pool_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kSp3Smem));
    XM_HIP(hipFuncSetAttribute((const void *)conv_stem_wgrad_pool_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kSp3Smem));
    attr_done = true;
  }
  {
    const double abytes = 4.0 * g.H * g.W * g.N + (y_pool ? 9.0 : 5.0) * (double)pooled + 4.0 * a.M * a.R;
    ProfScope ps(14 * 100 + (y_pool ? 1 : 0), 2.0 * a.M * (double)g.Ho * g.Wo * g.N * a.R, st, abytes);
#define XM_SPW_LAUNCH(SY_, G_) hipLaunchKernelGGL((conv_stem_wgrad_pool_kernel<SY_, G_>), dim3(grid), dim3(256), kSp3Smem, st, a, nunits)
    if (g.sy == 2) {
      if (y_pool) XM_SPW_LAUNCH(2, true); else XM_SPW_LAUNCH(2, false);
    } else {
      if (y_pool) XM_SPW_LAUNCH(1, true); else XM_SPW_LAUNCH(1, false);
    }
#undef XM_SPW_LAUNCH
  }
  XM_LAUNCH_CHECK();
  hipLaunchKernelGGL(stem_pool_finalize_kernel, dim3(g.K), dim3(1024), 0, st, a.part, nwc, gram, f, bias, bn_g, moments, g.K,
                     g.R, train ? 1 : 0, df_out, dbias_out, dg_out, db_out);
  XM_LAUNCH_CHECK();
  return XM_OK;
}

int xm_nnconv_backward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                       int FC, int K, const float *dzdy, float *dx_out, float *df_out,
                       float *db_out, int sy, int sx, int pt, int pb, int pl, int pr, int dy,
                       int dx, void *stream) {
  return xm_nnconv_backward_accum(x, H, W, C, N, f, FH, FW, FC, K, dzdy, dx_out, df_out, db_out, sy, sx, pt,
                                  pb, pl, pr, dy, dx, nullptr, stream);
}

int xm_nnconv_backward_accum(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                             int FC, int K, const float *dzdy, float *dx_out, float *df_out,
                             float *db_out, int sy, int sx, int pt, int pb, int pl, int pr, int dy,
                             int dx, const float *dx_accum, void *stream) {
  Geo g;
  int rc = make_geo(g, H, W, C, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx);
  if (rc) return rc;
  if (!dzdy) return fail(XM_EINVAL, "vl_nnconv: DZDY is NULL in backward mode");
  hipStream_t st = (hipStream_t)stream;
  if (db_out) {
    int S = std::max(1, std::min(N, 2048 / K));
    if ((long long)g.Ho * g.Wo < 1024) S = 1;
    WsCarver ws;
    rc = ws.init(WsCarver::need((size_t)K * S, 8), st);
    if (rc) return rc;
    double *part = ws.take<double>((size_t)K * S);
    const int HWo = g.Ho * g.Wo;
    hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(K, S), dim3(256), 0, st, dzdy, part, HWo, K, N, S,
                       make_fastdiv((uint32_t)((HWo & 3) == 0 ? HWo >> 2 : HWo)));
    XM_LAUNCH_CHECK();
    hipLaunchKernelGGL(bias_grad_finalize_kernel, dim3((K + 3) / 4), dim3(256), 0, st, part, db_out,
                       K, S);
    XM_LAUNCH_CHECK();
  }
  if (df_out) {
    if (!x) return fail(XM_EINVAL, "vl_nnconv: X is NULL but DZDF requested");
    rc = conv_wgrad(x, dzdy, df_out, g, st);
    if (rc) return rc;
  }
  if (dx_out) {
    if (!f) return fail(XM_EINVAL, "vl_nnconv: F is NULL but DZDX requested");
    if (dx_accum == dx_out) return fail(XM_EINVAL, "vl_nnconv: dx_accum must not alias dx_out");
    rc = conv_dgrad(f, dzdy, dx_out, g, st, dx_accum);
    if (rc) return rc;
  }
  return XM_OK;
}
}
