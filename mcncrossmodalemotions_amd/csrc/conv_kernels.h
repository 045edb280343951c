// Device kernels for vl_nnconv on gfx950: implicit-GEMM convolution on the fp32 MFMA pipe.
//
// Why fp32-input MFMA: the parity bar is 1e-4 against an fp32 CPU path, gfx950 has no TF32/xf32
// and v_mfma_f32_32x32x2_f32 is an exact fp32 fmaf chain at the full 157 TFLOP/s vector rate.
//
// GEMM view (all three directions share it):   D[m][p] = sum_r A[m][r] * G(p, r)
//   forward : m = output channel, p = output pixel (ho,wo,n), r = filter tap (u,v,c)
//             A = the filter bank itself (FH x FW x FC x K column-major == [K][R] row-major)
//   dgrad   : m = input channel,  p = input pixel of one stride-parity class, r = (u',v',k)
//             A = transposed / parity-split filters (prep_dgrad_filter)
//   G(p, r) = X[base(p) + off(r)] if the tap lands inside the source image, else 0 -- an
//             im2col that only ever exists as LDS tiles (never in HBM).
// MATLAB layout is H-fastest, so pixels are the contiguous axis of both the gather source and
// the destination: pixels go on the MFMA "column" (lane & 31) axis, which makes every global
// access of a wave a run of consecutive addresses along H.
#pragma once
#include "xm_common.h"

namespace xm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvGemmArgs {
  const float *A;     // [M][lda], lda % 4 == 0, 16-byte aligned, zero padded to Rp columns
  const float *X;     // gather source
  float *Y;           // destination
  const int4 *taps;   // Rp entries {off, du, dv, 0}; padding entries have du = -2^28
  const float *bias, *scale, *shift, *resid;
  int relu;
  int lda, M, Rp;
  int PI, PJ, NP;     // pixel grid (i fastest) and total pixel count PI*PJ*N
  FastDiv divPIJ, divPI;
  int gsy, gsx, gh0, gw0;  // gather origin of pixel (i,j): (i*gsy + gh0, j*gsx + gw0)
  int LimH, LimW, xSampleStride;
  int osy, osx, oh0, ow0, OH;  // destination position of pixel (i,j): (oh0 + i*osy, ow0 + j*osx)
  int oChanStride, oSampleStride;
  int nbm, nbn;       // tile counts
};

// XCD-aware, bijective block remap: consecutive logical tiles (which share the same pixel tile)
// land on the same XCD / L2.  Hardware places block b on XCD b % 8 (speed only, never
// correctness -- MI355X_MICROARCH.md "Workgroup dispatch").
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int nx = 8;
  int q = nblk / nx, r = nblk % nx;
  int xcd = bid % nx, idx = bid / nx;
  int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

constexpr int kBK = 16;  // reduction depth per LDS stage
constexpr int kNG = 4;   // float4 groups per stage (kBK / 4)

// LDS image of both operands: [stage][group g = k/4][row][k%4] so that one ds_read_b128 hands a
// lane four consecutive k of its row.  MFMA 32x32x2 takes k from lane>>5, so lanes 0-31 read
// group 2s and lanes 32-63 group 2s+1; the e-th MFMA of a chunk then sums k = 8s+e and 8s+4+e
// -- A and B use the same assignment, so the dot product is complete and exact.
template <int TM, int TN, int WGM, int WGN, bool CHECK>
__global__ void __launch_bounds__(256)
conv_gemm_kernel(const ConvGemmArgs a) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  static_assert(WGM * WGN == 4, "4 waves per block");
  static_assert(BN == 32 || BN == 64 || BN == 128 || BN == 256, "BN must divide 256");
  constexpr int PLA = BM * 4 + 4, PLB = BN * 4 + 4;  // plane strides (floats), 16-B multiples
  constexpr int NUA = (kNG * BM + 255) / 256;        // float4 units per thread, A
  constexpr int NUB = (kNG * BN + 255) / 256;        // float4 units per thread, B
  constexpr bool UNIFORM = BN >= 64;                 // tap group is wave-uniform
  __shared__ __attribute__((aligned(16))) float smem[2 * kNG * (PLA + PLB)];
  float *sA = smem;
  float *sB = smem + 2 * kNG * PLA;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave % WGM, wn = wave / WGM;
  const int tile = xcd_remap(blockIdx.x, a.nbm * a.nbn);
  const int bm = tile % a.nbm, bn = tile / a.nbm;

  // ---- per-thread gather geometry (fixed for the whole reduction) ----
  const int pl = t % BN, gB0 = t / BN;
  int ph, pw, xbase;
  {
    int p = bn * BN + pl;
    bool pvalid = p < a.NP;
    uint32_t pc = pvalid ? p : a.NP - 1;
    uint32_t n = xm_div(pc, a.divPIJ);
    uint32_t q = pc - n * a.divPIJ.d;
    uint32_t j = xm_div(q, a.divPI);
    uint32_t i = q - j * a.divPI.d;
    ph = (int)i * a.gsy + a.gh0;
    pw = (int)j * a.gsx + a.gw0;
    xbase = ph + a.LimH * pw + (int)n * a.xSampleStride;
    if (CHECK && !pvalid) ph = -(1 << 28);
  }
  const float *__restrict__ X = a.X;
  const float *__restrict__ A = a.A;
  const int4 *__restrict__ taps = a.taps;

  float4 ra[NUA], rb[NUB];

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NUA; ++i) {
      int u = t + 256 * i;
      int g = u % kNG, m = u / kNG;
      int gm = bm * BM + m;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((kNG * BM % 256 == 0 || u < kNG * BM) && gm < a.M)
        v = *reinterpret_cast<const float4 *>(A + (size_t)gm * a.lda + kt * kBK + g * 4);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NUB; ++i) {
      int g = gB0 + i * (256 / BN);
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (kNG * BN % 256 == 0 || g < kNG) {
        int r0 = kt * kBK + g * 4;
        if (UNIFORM) r0 = __builtin_amdgcn_readfirstlane(r0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int4 tp = taps[r0 + e];
          if (CHECK) {
            bool ok = (unsigned)(ph + tp.y) < (unsigned)a.LimH &&
                      (unsigned)(pw + tp.z) < (unsigned)a.LimW;
            if (ok) v[e] = X[xbase + tp.x];
          } else {
            v[e] = X[xbase + tp.x];
          }
        }
      }
      rb[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NUA; ++i) {
      int u = t + 256 * i;
      if (kNG * BM % 256 == 0 || u < kNG * BM) {
        int g = u % kNG, m = u / kNG;
        *reinterpret_cast<float4 *>(sA + (buf * kNG + g) * PLA + m * 4) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NUB; ++i) {
      int g = gB0 + i * (256 / BN);
      if (kNG * BN % 256 == 0 || g < kNG)
        *reinterpret_cast<float4 *>(sB + (buf * kNG + g) * PLB + pl * 4) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = a.Rp / kBK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int half = lane >> 5, l31 = lane & 31;
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int g = 2 * s + half;
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const float4 *>(sA + (cur * kNG + g) * PLA +
                                                  ((wm * TM + i) * 32 + l31) * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const float4 *>(sB + (cur * kNG + g) * PLB +
                                                  ((wn * TN + j) * 32 + l31) * 4);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nkt) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias -> per-channel affine -> residual -> relu -> store ----
  // C/D map of 32x32 MFMA: col = lane & 31 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int p = bn * BN + (wn * TN + j) * 32 + l31;
    if (p >= a.NP) continue;
    uint32_t n = xm_div((uint32_t)p, a.divPIJ);
    uint32_t q = (uint32_t)p - n * a.divPIJ.d;
    uint32_t jj = xm_div(q, a.divPI);
    uint32_t ii = q - jj * a.divPI.d;
    int obase = (a.oh0 + (int)ii * a.osy) + a.OH * (a.ow0 + (int)jj * a.osx) +
                (int)n * a.oSampleStride;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = bm * BM + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < a.M) {
          float v = acc[i][j][r];
          if (a.bias) v += a.bias[m];
          if (a.scale) v = v * a.scale[m] + a.shift[m];
          int off = obase + m * a.oChanStride;
          if (a.resid) v += a.resid[off];
          if (a.relu) v = fmaxf(v, 0.f);
          a.Y[off] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad:  dF[k][r] = sum_p dY[k][p] * G(p, r)      (p = flat output pixel (ho, wo, n))
// MFMA rows = output channel k, MFMA cols = tap r (contiguous in dF), reduction = pixels.
// Split over the pixel range (grid.y); partials go to a workspace slab per split and are summed
// by reduce_splits_kernel in a fixed order (deterministic, no atomics).
struct WgradArgs {
  const float *dY;   // [Ho*Wo][K][N]
  const float *X;    // gather source (forward input)
  float *out;        // [splits][M][ldo]
  const int4 *taps;  // Rn entries (forward tap table), padded entries du = -2^28
  int M, R, Rn, ldo;  // Rn = taps rounded up to BN
  int Ho, Wo, NP;     // output pixel grid, NP = Ho*Wo*N
  FastDiv divHW, divHo;
  int sy, sx, pt, pl_;  // forward stride / pad (gather origin = ho*sy - pt)
  int H, W, xSampleStride;
  int dyChanStride, dySampleStride;  // Ho*Wo, Ho*Wo*Ktotal
  int nbm, nbn, tilesPerSplit, nkt;
  size_t splitStride;
};

template <int TM, int TN, int WGM, int WGN>
__global__ void __launch_bounds__(256)
conv_wgrad_kernel(const WgradArgs a) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  static_assert(WGM * WGN == 4, "4 waves per block");
  static_assert(BN == 32 || BN == 64 || BN == 128 || BN == 256, "BN must divide 256");
  constexpr int PLA = BM * 4 + 4, PLB = BN * 4 + 4;
  constexpr int NUA = (kNG * BM + 255) / 256;
  constexpr int NUB = (kNG * BN + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * kNG * (PLA + PLB)];
  float *sA = smem;
  float *sB = smem + 2 * kNG * PLA;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave % WGM, wn = wave / WGM;
  const int tile = xcd_remap(blockIdx.x, a.nbm * a.nbn);
  const int bm = tile % a.nbm, bn = tile / a.nbm;
  const int split = blockIdx.y;
  const int kt0 = split * a.tilesPerSplit;
  const int kt1 = min(a.nkt, kt0 + a.tilesPerSplit);

  // the thread's tap (fixed for the whole reduction): B units are (tap = pl, pixel group g)
  const int pl = t % BN, gB0 = t / BN;
  const int4 tp = a.taps[bn * BN + pl];
  const float *__restrict__ X = a.X;
  const float *__restrict__ dY = a.dY;

  float4 ra[NUA], rb[NUB];

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NUA; ++i) {
      int u = t + 256 * i;
      int g = u % kNG, m = u / kNG;
      int gm = bm * BM + m;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if ((kNG * BM % 256 == 0 || u < kNG * BM) && gm < a.M) {
        uint32_t p = (uint32_t)(kt * kBK + g * 4);
        uint32_t n = xm_div(p, a.divHW);
        uint32_t q = p - n * a.divHW.d;
        int base = gm * a.dyChanStride;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if ((int)p + e < a.NP) v[e] = dY[base + (int)q + (int)n * a.dySampleStride];
          if (++q == a.divHW.d) {
            q = 0;
            ++n;
          }
        }
      }
      ra[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int i = 0; i < NUB; ++i) {
      int g = gB0 + i * (256 / BN);
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (kNG * BN % 256 == 0 || g < kNG) {
        uint32_t p = (uint32_t)(kt * kBK + g * 4);
        uint32_t n = xm_div(p, a.divHW);
        uint32_t q = p - n * a.divHW.d;
        uint32_t wo = xm_div(q, a.divHo);
        uint32_t ho = q - wo * a.divHo.d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int hi = (int)ho * a.sy - a.pt + tp.y;
          int wi = (int)wo * a.sx - a.pl_ + tp.z;
          bool ok = (int)p + e < a.NP && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
          if (ok) v[e] = X[((int)ho * a.sy - a.pt) + a.H * ((int)wo * a.sx - a.pl_) +
                           (int)n * a.xSampleStride + tp.x];
          if (++ho == a.divHo.d) {
            ho = 0;
            if (++wo == (uint32_t)a.Wo) {
              wo = 0;
              ++n;
            }
          }
        }
      }
      rb[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NUA; ++i) {
      int u = t + 256 * i;
      if (kNG * BM % 256 == 0 || u < kNG * BM) {
        int g = u % kNG, m = u / kNG;
        *reinterpret_cast<float4 *>(sA + (buf * kNG + g) * PLA + m * 4) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NUB; ++i) {
      int g = gB0 + i * (256 / BN);
      if (kNG * BN % 256 == 0 || g < kNG)
        *reinterpret_cast<float4 *>(sB + (buf * kNG + g) * PLB + pl * 4) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;
  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < kt1) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int g = 2 * s + half;
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const float4 *>(sA + (cur * kNG + g) * PLA +
                                                  ((wm * TM + i) * 32 + l31) * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const float4 *>(sB + (cur * kNG + g) * PLB +
                                                  ((wn * TN + j) * 32 + l31) * 4);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < kt1) store_tile(cur ^ 1);
    __syncthreads();
  }

  float *out = a.out + (size_t)split * a.splitStride;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int r = bn * BN + (wn * TN + j) * 32 + l31;
    if (r >= a.R) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        int m = bm * BM + (wm * TM + i) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        if (m < a.M) out[(size_t)m * a.ldo + r] = acc[i][j][rr];
      }
  }
}

}  // namespace xm
