// Round 6: the student's first layers  conv1 -> bn1 -> relu1 -> pool1  (emoVoxCeleb/emoVoxZoo.m:50-62)
// WITHOUT a pass over conv1's output on the way back.
//
// conv1 has ONE input channel: its output is x[m][p] = sum_t f~[m][t] P~[p][t] with P~ the im2col patches of the
// spectrogram (R = FH * FW taps) extended by a column of ones (t = R, f~[m][R] = bias[m]).  Everything the backward of
// vl_nnpool('max') o vl_nnrelu o vl_nnbnorm o vl_nnconv needs from x is then a contraction with the (R+1) x (R+1) GRAM
// matrix G = P~' P~ of the patches -- a property of the INPUT alone (157 MB at 256 spectrograms, against 3.7 GB for x):
//     dz[m][p]  = routed, ReLU-masked pooled derivative (non-zero only at window maxima)
//     A[m][t]   = sum_p dz[m][p] P~[p][t]                                    (MFMA, this file: conv_stem_wgrad_pool_kernel)
//     S1[m]     = sum_p dz            = A[m][R]                               (= db of the bnorm)
//     S2[m]     = sum_p dz (x - mu)   = sum_t f~[m][t] A[m][t] - mu S1        (dg = S2 / sigma)
//     Cx[m][t]  = sum_p (x - mu) P~   = (f~ G)[m][t] - mu G[R][t]
//     dF~[m][t] = g/sigma (A[m][t] - S1/P G[R][t] - S2/(sigma^2 P) Cx[m][t])  (t < R: dzdf, t = R: dzdb of the convolution)
// which is vl_nnbnorm's train-mode derivative dx = g/sigma (dz - mean(dz) - xhat mean(dz xhat)) pushed through
// dF = sum_p dx P~ term by term (test mode: dF~ = g/sigma A).  The sums S1, S2 come out of A as well, so the separate
// sums pass over the pooled tensors (bnpool_bwd_partial_pooled_kernel) disappears, and x is not read at all.
// Numerics (tools-free check on the host, 2 spectrograms, fp32 partial sums per ~9 k pixels then fp64): 1.5e-7 of the
// largest filter-derivative entry from the fp64 composition; the Gram route to the batch moments: sigma to 2e-8.
//
// dz is built in LDS by SCATTER from the pooled side: a pooled element (channel, ph, pw) with routing code dh + 3 dw lands
// on conv pixel (2 ph + dh, 2 pw + dw) -- each pooled element is touched once per candidate column instead of every conv
// pixel testing its <= 2 x 2 covering windows (conv_stem_wgrad_bnp_kernel: 48 pooled vector-memory instructions per wave
// and 32 pixels, each fetching 12 + 4 bytes per lane; here 16-byte loads of whole pooled columns).  ds_add_f32 from one
// wave executes in program order and lanes of one instruction never share a target (their pooled rows are >= 4 apart),
// so the sum of the <= 4 contributions to a pixel has a fixed order.
#pragma once
#include "conv_kernels.h"

namespace xm {

struct StemPoolArgs {
  const float *X;                    // the convolution's input [H][W][1][N]
  float *part;                       // gram: [grid][64][64], wgrad: [grid][96][64]
  int M, R, nU, nV;                  // filters, taps (nU * nV), filter rows / columns
  int PI, PJ;                        // output pixel grid
  FastDiv divJG, divG;               // (column pairs) * (256-row groups), 256-row groups: block unit -> (sample, pair, group)
  int gsx, gh0, gw0, LimH, LimW;     // forward gather geometry (as ConvGemmArgs)
  int xSampleStride;
  const float *dP, *yP;              // pooled DZDY, pooled forward output [pHo][pWo][M][N]
  const unsigned char *amax;         // routing table (first maximum, code = dh + 3 dw)
  unsigned dpBytes, amBytes;
  int pHo, pWo;
  int dbg;                           // experiment switches (XM_SP_DBG; 0 in the product)
};

// ---- work decomposition ------------------------------------------------------------------------------------------------------
// A BLOCK unit is (sample, pair of output columns 2 jp, 2 jp + 1, group of 256 output rows); wave w of the block owns the
// 64 rows [64 cp, 64 cp + 64), cp = 4 group + w, of both columns: four 32-pixel MFMA tiles (column jj, 32-row chunk h).
// Why pairs of columns and 64 rows: the pooled tensors are [window row][window column][filter][sample]; a pooled column
// (126 floats) serves THREE output columns and a 128-byte line of it 64 output rows.  The first version of this kernel
// (a wave = one column x 32 rows, 17 window rows per filter and slot) pulled 39 GB through the L2s for 2.2 GB of pooled
// operands at 256 spectrograms and ran at the L2 -> L1 rate (2.7 ms); here a wave loads 32 consecutive window rows
// (= one line) of window columns jp - 1 and jp ONCE per row tile of filters and keeps them in registers for its four tiles.
constexpr int kSpHW = 168;                    // row pitch of a source column in the patch: 35 units of 16 bytes + 28 (168 = 40 mod 64:
                                              // the taps u + 168 v of a pixel fall into different LDS banks, as kStemHW)
constexpr int kSpNC = 9;                      // source columns under two output columns (stride 2: 7 + 2)
constexpr int kSpPatch = kSpNC * kSpHW + 4;   // floats of a wave's source patch + a dummy unit
constexpr int kSpTP = 36;                     // row pitch (floats) of a dz region: 32 pixels + 4
constexpr int kSpWave = 2 * kSpPatch + 32 * kSpTP;   // two patches (the next unit's is written while the current one is read) + the dz region
constexpr int kSpCst = 64;                    // 32 ones + 32 zeros: what the B lanes of the ones column / the padding columns read
constexpr int kSpSmem = (4 * kSpWave + kSpCst) * 4;      // bytes per block of conv_stem_wgrad_pool_kernel (67 KB: two blocks per CU)
constexpr int kSpGramSmem = (4 * kSpPatch + kSpCst > 64 * 64 ? 4 * kSpPatch + kSpCst : 64 * 64) * 4;   // bytes per block of stem_gram_kernel

struct SpUnit {
  int n, jp, cp;         // sample, column pair, 64-row chunk pair of this wave
  bool live;             // the unit exists and the wave's rows exist
  bool col1, chunk1;     // the second column / the second 32-row chunk exist
};

#define XM_SP_COMMON(a)                                                                                              \
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;                        \
  const int lc = lane >> 3, lk = lane & 7;                                                                           \
  const int per = (nunits + 7) >> 3, tbase = (blockIdx.x & 7) * per, tstep = gridDim.x >> 3;                         \
  const int tend = min(per, nunits - tbase);                                                                         \
  const int wv = __builtin_amdgcn_readfirstlane(wave);                                                               \
  auto wave_unit = [&](int unit) {                                                                                   \
    SpUnit c;                                                                                                        \
    const uint32_t u = (uint32_t)unit;                                                                               \
    c.n = (int)xm_div(u, a.divJG);                                                                                   \
    const uint32_t rem = u - (uint32_t)c.n * a.divJG.d;                                                              \
    c.jp = (int)xm_div(rem, a.divG);                                                                                 \
    c.cp = 4 * ((int)rem - c.jp * (int)a.divG.d) + wv;                                                               \
    c.live = 64 * c.cp < a.PI;                                                                                       \
    c.col1 = 2 * c.jp + 1 < a.PJ;                                                                                    \
    c.chunk1 = 64 * c.cp + 32 < a.PI;                                                                                \
    return c;                                                                                                        \
  };                                                                                                                 \
  f32x4 ld[6];                                                                                                       \
  int ldst[6];                                                                                                       \
  bool ldz[6];                                                                                                       \
  auto issue_patch = [&](const SpUnit &c) {                                                                          \
    const int iF = 64 * c.cp, iL = min(iF + 63, a.PI - 1);                                                           \
    const int lo4 = (SY * iF + a.gh0 + 4) >> 2;                                                                      \
    const int n0 = c.live ? ((SY * iL + a.gh0 + 4 + 7) >> 2) - lo4 + 1 : 0;                                          \
    const int ncol = a.gsx + a.nV;                                                                                   \
    _Pragma("unroll") for (int it = 0; it < 6; ++it) {                                                               \
      const int col = it < 5 ? lc : 8, q = it < 5 ? lk + 8 * it : lane;                                              \
      const bool wr = col < ncol && q < n0 && (it < 5 || lane < 40);                                                 \
      const int cc = a.gsx * 2 * c.jp + a.gw0 + col, r = 4 * (lo4 + q) - 4;                                          \
      const bool in = wr && cc >= 0 && cc < a.LimW && r >= 0 && r < a.LimH;                                          \
      ldst[it] = wr ? col * kSpHW + 4 * q : kSpNC * kSpHW;                                                           \
      ldz[it] = !in;                                                                                                 \
      ld[it] = *reinterpret_cast<const f32x4 *>(in ? a.X + (size_t)c.n * a.xSampleStride + (size_t)cc * a.LimH + r : a.X); \
    }                                                                                                                \
  };                                                                                                                 \
  auto write_patch = [&](float *sW) {                                                                                \
    _Pragma("unroll") for (int it = 0; it < 6; ++it)                                                                 \
      *reinterpret_cast<f32x4 *>(sW + ldst[it]) = ldz[it] ? f32x4{0.f, 0.f, 0.f, 0.f} : ld[it];                      \
  };                                                                                                                 \
  /* B operand base of (column jj, chunk h): pixel 16 half + s of the chunk sits SY s floats further on */            \
  auto patch_base = [&](const SpUnit &c, int jj, int h) {                                                            \
    const int iF = 64 * c.cp;                                                                                        \
    return (iF + 32 * h + 16 * half) * SY + a.gh0 + 4 - 4 * ((SY * iF + a.gh0 + 4) >> 2) + a.gsx * jj * kSpHW;      \
  };                                                                                                                 \
  /* this lane's two taps (column tiles jt = 0, 1): offset inside the patch.  Lanes of the ones column (n == R) and of the   */ \
  /* padding columns read a run of ones / zeros instead (cstoff: float offset inside the constants region): the B read of a  */ \
  /* step is base + immediate for every lane, no select per read                                                              */ \
  int tapoff[2], cstoff[2];                                                                                          \
  bool isTap[2];                                                                                                     \
  _Pragma("unroll") for (int jt = 0; jt < 2; ++jt) {                                                                 \
    const int n = l31 + 32 * jt;                                                                                     \
    const int v = n < a.R ? n / a.nU : 0, u = n < a.R ? n - v * a.nU : 0;                                            \
    tapoff[jt] = u + kSpHW * v;                                                                                      \
    isTap[jt] = n < a.R;                                                                                             \
    cstoff[jt] = n == a.R ? 0 : 32;                                                                                  \
  }

// ---- G = P~' P~ ------------------------------------------------------------------------------------------------------------
// MFMA rows = columns = taps (+ ones) padded to 64, reduction = output pixels.  The A fragment of a 32 x 32 x 2 MFMA
// (lane l: A[l % 32][l / 32]) and its B fragment (lane l: B[l / 32][l % 32]) are the SAME register when A = B': one
// ds_read_b32 per (pixel pair, column tile) feeds three MFMAs (tiles (0,0), (1,0), (1,1); (0,1) is the mirror image).
// Every wave walks its own units with a wave-private patch; no barrier in the loop.
template <int SY>
__global__ void __launch_bounds__(256, 3)
stem_gram_kernel(const StemPoolArgs a, const int nunits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 4 patches, then (after the loop) the [64][64] partial
  XM_SP_COMMON(a)
  for (int i = t; i < kSpGramSmem / 16; i += 256) reinterpret_cast<f32x4 *>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float *const sW = smem + wave * kSpPatch;
  float *const cst = smem + 4 * kSpPatch;
  __syncthreads();
  if (t < 32) cst[t] = 1.f;
  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int q = blockIdx.x >> 3;
  __syncthreads();
  SpUnit cur = wave_unit(tbase + min(q, max(tend, 1) - 1));
  if (q < tend) {
    issue_patch(cur);
    write_patch(sW);
  }
  for (; q < tend; q += tstep) {
    const int unit = tbase + q;
    const bool more = q + tstep < tend;
    const SpUnit nxt = wave_unit(more ? unit + tstep : unit);   // (the last unit is staged once more: no branch around the loads)
    issue_patch(nxt);
    __builtin_amdgcn_sched_barrier(0);
    if (cur.live) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if (jj == 0 || cur.col1) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (h == 0 || cur.chunk1) {
              const int nval = a.PI - 64 * cur.cp - 32 * h - 16 * half;      // this half's pixels s < nval exist
              const int pbo = patch_base(cur, jj, h);
              const float *pl[2];
#pragma unroll
              for (int jt = 0; jt < 2; ++jt) pl[jt] = isTap[jt] ? sW + pbo + tapoff[jt] : cst + cstoff[jt];
              if (64 * cur.cp + 32 * h + 32 <= a.PI) {                       // (wave-uniform) all 32 pixels exist
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                  const float b0 = pl[0][SY * s], b1 = pl[1][SY * s];
                  acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, b0, acc[0], 0, 0, 0);
                  acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, b0, acc[1], 0, 0, 0);
                  acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, b1, acc[2], 0, 0, 0);
                }
              } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                  const float b0 = s < nval ? pl[0][SY * s] : 0.f, b1 = s < nval ? pl[1][SY * s] : 0.f;
                  acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, b0, acc[0], 0, 0, 0);
                  acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, b0, acc[1], 0, 0, 0);
                  acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, b1, acc[2], 0, 0, 0);
                }
              }
            }
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    write_patch(sW);
    cur = nxt;
  }
  // the four waves add their accumulators in LDS in wave order; the block leaves one partial [64][64] (tile (0, 1) = 0)
  float *const sD = smem;
  __syncthreads();
  for (int i = t; i < 64 * 64 / 4; i += 256) reinterpret_cast<f32x4 *>(sD)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int wq = 0; wq < 4; ++wq) {
    __syncthreads();
    if (wave == wq) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float *d = sD + (32 * (i > 0) + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 * (i > 1) + l31;
          *d = *d + acc[i][r];
        }
    }
  }
  __syncthreads();
  float *out = a.part + (size_t)blockIdx.x * (64 * 64);
  for (int i = t; i < 64 * 64 / 4; i += 256) reinterpret_cast<f32x4 *>(out)[i] = reinterpret_cast<const f32x4 *>(sD)[i];
}

// G[i][j] = sum over the blocks' partials in a fixed order, fp64, mirrored into the tile the kernel does not compute
__global__ void __launch_bounds__(1024)
stem_gram_reduce_kernel(const float *__restrict__ part, double *__restrict__ gram, int nblk) {
  __shared__ double red[16][64];
  const int j = threadIdx.x & 63, g = threadIdx.x >> 6, i = blockIdx.x;
  const int si = (i < 32 && j >= 32) ? j : i, sj = (i < 32 && j >= 32) ? i : j;
  double v = 0.0;
  for (int b = g; b < nblk; b += 16) v += (double)part[(size_t)b * (64 * 64) + si * 64 + sj];
  red[g][j] = v;
  __syncthreads();
  if (g == 0) {
    v = red[0][j];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += red[k][j];
    gram[i * 64 + j] = v;
  }
}

// ---- A = dz P~ with dz scattered from the pooled derivative ---------------------------------------------------------------
// Per row tile rt of 32 filters a wave loads, for window columns p = 0: jp - 1 and p = 1: jp, the 32 window rows
// [32 cp, 32 cp + 32) of its 64 output rows as 16-byte quads (lane -> filter lane / 4 (+ 16), quad lane % 4 of chunk h) --
// pooled derivative, pooled forward output (ReLU gate) and routing codes -- plus the single window row 32 cp - 1 in front
// (lanes 0-31: p = 0, lanes 32-63: p = 1).  A quad that would reach past its pooled column is shifted back (no load
// leaves the tensor) and its first `shift` elements are ignored.  The derivatives are gated once (y_pool > 0) and stay in
// registers for the four (column, chunk) steps of the row tile; each step zero-fills the wave's dz region [32 filters][32
// pixels], scatters the elements whose code names this column (dw) with ds_add_f32 -- window row r of the chunk lands on
// pixel 2 r + dh; row 15 with dh = 2 on pixel 0 of the NEXT chunk -- and multiplies: 32 MFMAs per step.
// Output column 2 jp takes dw = 0 of window column jp and dw = 2 of jp - 1; column 2 jp + 1 takes dw = 1 of jp.
// GATE: the ReLU gate is taken from y_pool (> 0); false: the table marks closed windows itself (code 255, as
// conv_stem_bnpool_fwd_kernel writes it) and y_pool is not read.
template <int SY, bool GATE>
__global__ void __launch_bounds__(256, 2)
conv_stem_wgrad_pool_kernel(const StemPoolArgs a, const int nunits) {
  constexpr int TM = 3, TP = kSpTP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  XM_SP_COMMON(a)
  for (int i = t; i < 4 * kSpWave / 4; i += 256) reinterpret_cast<f32x4 *>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float *const sW0 = smem + wave * kSpWave;           // this wave's two source patches
  float *const reg = sW0 + 2 * kSpPatch;              // this wave's dz region [32][TP]
  float *const cst = smem + 4 * kSpWave;
  for (int i = t; i < kSpCst; i += 256) cst[i] = i < 32 ? 1.f : 0.f;
  const __amdgpu_buffer_rsrc_t dprsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dP, 0, a.dpBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yprsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.yP, 0, a.dpBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t amrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.amax, 0, a.amBytes, 0x00020000);
  const int pHW = a.pHo * a.pWo;
  const int cl = lane >> 2, qd = lane & 3;

  // ---- per-unit lane geometry of the candidates ---------------------------------------------------------------------------
  struct Cand {
    unsigned voff[2];     // per chunk h: element offset of the lane's quad inside a (sample, 16-filter group, window column 0) block
    unsigned lim[2];      // per chunk: 4 x 8 bit, element e routes into the chunk iff (code - 3 dw) < lim[e]  (0: ignore the element)
    int rowb[2];          // per chunk: LDS float offset of pixel row (2 e + dh) = 0 of the lane's quad inside a 16-filter group
    bool x15;             // this lane's element 3 of chunk 0 is window row 15: with dh = 2 it lands on pixel 0 of chunk 1
    unsigned exoff;       // the window row 32 cp - 1: element offset inside a (sample, row tile) block, or out of range
    bool ok[2];           // window column p exists (wave-uniform)
    int sbase;            // element offset of (sample, filter 0, window column 0, window row 0)
    int jp;
  };
  auto cand_of = [&](const SpUnit &c, bool on) {
    Cand k;
    k.jp = c.jp;
    k.ok[1] = on && c.live && c.jp < a.pWo;
    k.ok[0] = on && c.live && c.jp >= 1 && c.jp - 1 < a.pWo;
    k.x15 = false;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ph0 = 32 * c.cp + 16 * h + 4 * qd;        // nominal first window row of the lane's quad
      const int phs = min(ph0, a.pHo - 4);                // shifted back so that the quad stays inside its pooled column
      const int shift = ph0 - phs;                        // elements e < shift belong to the quad before
      k.voff[h] = (unsigned)(cl * pHW + phs);
      unsigned lim = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rel = phs + e - 32 * c.cp - 16 * h;     // window row relative to the chunk: pixel row = 2 rel + dh
        const bool valid = e >= shift && shift < 4;
        lim |= (valid ? (rel == 15 ? 2u : 3u) : 0u) << (8 * e);
        if (h == 0 && e == 3 && valid && rel == 15) k.x15 = true;
      }
      k.lim[h] = lim;
      k.rowb[h] = cl * TP + 2 * (phs - 32 * c.cp - 16 * h);
    }
    const bool exok = c.cp > 0 && (half ? k.ok[1] : k.ok[0]);
    k.exoff = exok ? (unsigned)((l31 * a.pWo + c.jp - 1 + half) * a.pHo + 32 * c.cp - 1) : 0x3FFFFFFFu;
    k.sbase = c.n * a.M * pHW;
    return k;
  };
  // pooled operands of row tile rt: [window column p][group of 16 filters][chunk]
  f32x4 cdp[2][2][2], cyp[2][2][2];
  unsigned ccd[2][2][2];
  float edp, eyp = 0.f;
  unsigned ecd;
  auto issue_cand = [&](const Cand &k, int rt) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (k.ok[p] && !(a.dbg & 1)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          typedef unsigned u4 __attribute__((ext_vector_type(4)));
          const int so = k.sbase + (32 * rt + 16 * it) * pHW + (k.jp - 1 + p) * a.pHo;     // scalar part (elements)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            cdp[p][it][h] = __builtin_bit_cast(f32x4, (u4)__builtin_amdgcn_raw_buffer_load_b128(dprsrc, (int)(k.voff[h] * 4u), so * 4, 0));
            if (GATE) cyp[p][it][h] = __builtin_bit_cast(f32x4, (u4)__builtin_amdgcn_raw_buffer_load_b128(yprsrc, (int)(k.voff[h] * 4u), so * 4, 0));
            ccd[p][it][h] = __builtin_amdgcn_raw_buffer_load_b32(amrsrc, (int)k.voff[h], so, 0);
          }
        }
      }
    }
    const int so = k.sbase + 32 * rt * pHW;
    if (a.dbg & 1) return;
    edp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dprsrc, (int)(k.exoff * 4u), so * 4, 0));
    if (GATE) eyp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yprsrc, (int)(k.exoff * 4u), so * 4, 0));
    ecd = __builtin_amdgcn_raw_buffer_load_b8(amrsrc, (int)k.exoff, so, 0);
  };
  // the ReLU gate, once per row tile: the derivative of a window whose maximum did not pass the ReLU is zero
  auto gate_cand = [&](const Cand &k, int rt) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
      if (k.ok[p]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const bool mok = 32 * rt + 16 * it + cl < a.M;       // (filters that do not exist: nothing is routed)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) cdp[p][it][h][e] = (mok && (!GATE || cyp[p][it][h][e] > 0.f)) ? cdp[p][it][h][e] : 0.f;
        }
      }
    edp = ((!GATE || eyp > 0.f) && 32 * rt + l31 < a.M) ? edp : 0.f;
  };
  // dz of (row tile rt, column jj, chunk h) into the wave's region.  Order of the <= 4 additions to a pixel: the window
  // row in front (row 32 cp - 1 / row 15 of chunk 0) first, then window column jp - 1, then jp.
  auto scatter = [&](const Cand &k, int rt, int jj, int h) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) *reinterpret_cast<f32x4 *>(reg + (lc + 8 * kk) * TP + 4 * lk) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (h == 0) {
      // lanes 0-31: window column jp - 1 (dw = 2: column 2 jp only), lanes 32-63: window column jp (dw = jj)
      const unsigned want = half ? 3u * (unsigned)jj + 2u : 8u;
      const bool act = half || jj == 0;
      if (act && ecd == want && edp != 0.f) __hip_atomic_fetch_add(reg + l31 * TP, edp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if ((p == 1 || jj == 0) && k.ok[p]) {
        const unsigned d3 = p == 1 ? 3u * (unsigned)jj : 6u;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          if (h == 1) {
            // window row 15 of chunk 0 with dh = 2: pixel 0 of this chunk
            const unsigned code = ccd[p][it][0] >> 24;
            if (k.x15 && code - d3 == 2u && cdp[p][it][0][3] != 0.f)
              __hip_atomic_fetch_add(reg + (16 * it + cl) * TP, cdp[p][it][0][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
          float *const rb = reg + 16 * it * TP + k.rowb[h];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned code = (ccd[p][it][h] >> (8 * e)) & 0xFFu, d = code - d3;
            const bool hit = cdp[p][it][h][e] != 0.f && d < ((k.lim[h] >> (8 * e)) & 0xFFu);
            if (hit) __hip_atomic_fetch_add(rb + 2 * e + (int)d, cdp[p][it][h][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
        }
      }
    }
  };

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jt][r] = 0.f;

  int q = blockIdx.x >> 3;
  __syncthreads();                       // zero fill complete
  SpUnit cur = wave_unit(tbase + min(q, max(tend, 1) - 1));
  Cand kc = cand_of(cur, q < tend);
  int pbuf = 0;
  if (q < tend) {
    issue_patch(cur);
    issue_cand(kc, 0);
    write_patch(sW0);
  }
  for (; q < tend; q += tstep) {
    const int unit = tbase + q;
    const bool more = q + tstep < tend;
    const SpUnit nxt = wave_unit(more ? unit + tstep : unit);
    const float *const sW = sW0 + pbuf * kSpPatch;
    const float *tr = reg + l31 * TP + 16 * half;      // A operand: row l31, pixels 16 half + 4 c .. + 3
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      gate_cand(kc, i);                                 // (waits for this row tile's operands)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool run = cur.live && (jj == 0 || cur.col1) && (h == 0 || cur.chunk1);
          if (run && !(a.dbg & 2)) scatter(kc, i, jj, h);
          __builtin_amdgcn_sched_barrier(0);
          // the next unit's patch: requested under the first step's MFMAs (few registers are live here), parked in the
          // other patch buffer behind them
          if (i == 0 && jj == 0 && h == 0) issue_patch(nxt);
          if (jj == 1 && h == 1) {
            // the row tile's operands are dead: request the next row tile's (the next unit's after the last one) under
            // the last step's MFMAs
            if (i < TM - 1) issue_cand(kc, i + 1);
            else {
              kc = cand_of(nxt, more);
              issue_cand(kc, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (run && !(a.dbg & 4)) {
            const int pbo = patch_base(cur, jj, h);
            const float *pl[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) pl[jt] = isTap[jt] ? sW + pbo + tapoff[jt] : cst + cstoff[jt];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x4 af = *reinterpret_cast<const f32x4 *>(tr + 4 * c);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float b0 = pl[0][SY * (4 * c + e)], b1 = pl[1][SY * (4 * c + e)];
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], b1, acc[i][1], 0, 0, 0);
              }
            }
          }
          if (i == 0 && jj == 0 && h == 0) {
            __builtin_amdgcn_sched_barrier(0);
            write_patch(sW0 + (pbuf ^ 1) * kSpPatch);
          }
        }
      }
    }
    cur = nxt;
    pbuf ^= 1;
  }
  // the four waves add their accumulators in LDS in wave order; the block leaves one partial [96][64]
  float *const sD = smem;
  for (int wq = 0; wq < 4; ++wq) {
    __syncthreads();
    if (wave == wq) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float *d = sD + (32 * i + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 * jt + l31;
            *d = wq ? *d + acc[i][jt][r] : acc[i][jt][r];
          }
    }
  }
  __syncthreads();
  float *out = a.part + (size_t)blockIdx.x * (96 * 64);
  for (int i = t; i < 96 * 64 / 4; i += 256) reinterpret_cast<f32x4 *>(out)[i] = reinterpret_cast<const f32x4 *>(sD)[i];
}

// ---- dF~, dg, db from A and G (fp64; one block per filter) -----------------------------------------------------------------
__global__ void __launch_bounds__(256)
stem_pool_finalize_kernel(const float *__restrict__ part, int nblk, const double *__restrict__ gram,
                          const float *__restrict__ f, const float *__restrict__ bias, const float *__restrict__ bn_g,
                          const float *__restrict__ moments, int M, int R, int train, float *__restrict__ df,
                          float *__restrict__ dbias, float *__restrict__ dg, float *__restrict__ db) {
  __shared__ double red[4][64];
  __shared__ double sA[64], sF[64], sS[2];
  const int tt = threadIdx.x & 63, grp = threadIdx.x >> 6, m = blockIdx.x;
  double v = 0.0;
  for (int b = grp; b < nblk; b += 4) v += (double)part[(size_t)b * (96 * 64) + m * 64 + tt];
  red[grp][tt] = v;
  __syncthreads();
  if (grp == 0) {
    sA[tt] = (red[0][tt] + red[1][tt]) + (red[2][tt] + red[3][tt]);
    sF[tt] = tt < R ? (double)f[(size_t)m * R + tt] : (tt == R && bias ? (double)bias[m] : 0.0);
  }
  __syncthreads();
  const double mu = (double)moments[m], sig = (double)moments[M + m], gs = (double)bn_g[m] / sig;
  if (threadIdx.x == 0) {
    const double s1 = sA[R];
    double s2 = 0.0;
    for (int k = 0; k <= R; ++k) s2 += sF[k] * sA[k];
    s2 -= mu * s1;
    sS[0] = s1, sS[1] = s2;
    if (dg) dg[m] = (float)(s2 / sig);
    if (db) db[m] = (float)s1;
  }
  __syncthreads();
  if (grp == 0 && tt <= R) {
    double o = sA[tt];
    if (train) {
      const double P = gram[R * 64 + R];
      double cx = 0.0;
      for (int k = 0; k <= R; ++k) cx += sF[k] * gram[k * 64 + tt];
      cx -= mu * gram[R * 64 + tt];
      o = o - sS[0] / P * gram[R * 64 + tt] - sS[1] / (sig * sig * P) * cx;
    }
    o *= gs;
    if (tt < R) df[(size_t)m * R + tt] = (float)o;
    else if (dbias) dbias[m] = (float)o;
  }
}

// batch moments of the convolution's output from G (vl_nnbnorm's [mean, sqrt(var + eps)]): mu = f~ G[:, R] / P,
// E[x^2] = f~ G f~' / P, fp64
__global__ void __launch_bounds__(64)
stem_gram_moments_kernel(const double *__restrict__ gram, const float *__restrict__ f, const float *__restrict__ bias, int M,
                         int R, double eps, float *__restrict__ moments) {
  __shared__ double sF[64];
  const int tt = threadIdx.x, m = blockIdx.x;
  sF[tt] = tt < R ? (double)f[(size_t)m * R + tt] : (tt == R && bias ? (double)bias[m] : 0.0);
  __syncthreads();
  double row = 0.0;                     // (f~ G)[tt]
  if (tt <= R)
    for (int k = 0; k <= R; ++k) row += sF[k] * gram[k * 64 + tt];
  const double P = gram[R * 64 + R];
  double e2 = tt <= R ? row * sF[tt] : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o, 64);
  const double mu = __shfl(row, R, 64) / P;
  if (tt == 0) {
    const double var = fmax(e2 / P - mu * mu, 0.0);
    moments[m] = (float)mu;
    moments[M + m] = (float)sqrt(var + eps);
  }
}

}  // namespace xm

namespace xm {

// ---- forward: vl_nnpool('max', 3 x 3 / 2) o vl_nnrelu o vl_nnbnorm o vl_nnconv in ONE kernel --------------------------------
// The convolution's output (3.7 GB at 256 spectrograms) is neither written nor read.  The batch moments of the train-mode
// bnorm come from the Gram matrix of the input patches (stem_gram_moments_kernel) BEFORE this kernel starts, so the
// normalisation is folded into the operand: A[m][k] = g/sigma F[m][k], and the constant g/sigma (bias - mu) + b rides on
// the one spare reduction index (k = 7: filter column 0, row 7, whose B operand is forced to 1) -- the MFMA accumulator IS
// bnorm's output.
// A wave owns (sample, strip of 63 window rows, row tile of 32 filters, segment of window columns) and walks the output
// columns of its segment.  Per output column: four 32-pixel MFMA tiles a0, a1 (even rows 126 s + 2 q, q = 0 .. 63) and
// b0, b1 (odd rows): window row q of the strip is max(a[q], b[q], a[q + 1]) -- three registers of the SAME lane except
// a[q + 1], one DPP wave shift (lane 31 takes a1's lane 0) -- so 126 window rows are exactly two strips of 63 (one even
// row computed twice).  Horizontally the window is closed over three consecutive columns with a running (best, code)
// per lane: first maximum in MatConvNet's scan order (column-major: strict > keeps the earlier element).
// Source patch: a wave-private ring of 7 source columns, each as FOUR PHASE PLANES (plane rho holds source rows
// 4 k + rho): lane q reads source row 4 q + const, i.e. consecutive floats of one plane -- no bank conflicts; two new
// source columns per output column (3 x 16-byte loads per lane) written behind the column's B reads.  The column loop is
// unrolled by the ring period so that every LDS address is base + immediate.  No barrier in the loop.
// Routing table: code = dh + 3 dw of the first maximum, 255 for a window whose maximum did not pass the ReLU (its
// derivative is zero whatever it is routed to): conv_stem_wgrad_pool_kernel then needs no pass over y_pool.
struct StemFwdArgs {
  const float *X, *F, *bias, *bn_g, *bn_b, *moments;
  float *Y;
  unsigned char *amax;
  int M, R, nU, nV;
  int PI, PJ, pHo, pWo;
  int NS, SG, nunits;                // strips of 63 window rows, segments of window columns, wave units = N * NS * SG * 3
  FastDiv div3, divSG, divNS;
  int gh0, gw0, LimH, LimW, xSampleStride;
  unsigned yBytes, amBytes;
};
constexpr int kSfPlane = 68;                      // floats per phase plane: 64 + 4
constexpr int kSfSlot = 4 * kSfPlane;             // a source column of the ring
constexpr int kSfRing = 7 * kSfSlot;              // floats per wave
constexpr int kSfA = kStemNV * 3 * 2 * 32 * 4;    // the folded filter bank in MFMA operand order (conv_stem_kernel's sA)
constexpr int kSfSmem = (kSfA + 4 * kSfRing) * 4; // bytes per block (51.9 KB: three blocks per CU)

#ifndef XM_SF_OCC
#define XM_SF_OCC 2
#endif
__global__ void __launch_bounds__(256, XM_SF_OCC)
conv_stem_bnpool_fwd_kernel(const StemFwdArgs a) {
  constexpr int TM = 3, NV = kStemNV;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const sA = smem;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  float *const ring = smem + kSfA + wave * kSfRing;
  for (int i = t; i < 4 * kSfRing / 4; i += 256) reinterpret_cast<f32x4 *>(smem + kSfA)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // folded filter bank: lane (l31, half) of row tile i reads A[32 i + l31][k = 8 v + e + 4 half], e = 0 .. 3
  for (int idx = t; idx < kSfA; idx += 256) {
    const int e = idx & 3, l = (idx >> 2) & 31, h = (idx >> 7) & 1, vi = idx >> 8, i = vi % TM, v = vi / TM;
    const int m = 32 * i + l, u = e + 4 * h;
    float w = 0.f;
    if (m < a.M) {
      const float sc = a.bn_g[m] / a.moments[a.M + m];
      if (u < a.nU && v < a.nV) w = sc * a.F[(size_t)m * a.R + u + a.nU * v];
      else if (u == 7 && v == 0) w = sc * ((a.bias ? a.bias[m] : 0.f) - a.moments[m]) + a.bn_b[m];
    }
    sA[idx] = w;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.Y, 0, a.yBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t amrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.amax, 0, a.amBytes, 0x00020000);
  const int pHW = a.pHo * a.pWo;
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  for (int unit = blockIdx.x * 4 + wv; unit < a.nunits; unit += gridDim.x * 4) {
    // ---- unit -> (sample, strip, segment, row tile) --------------------------------------------------------------------
    uint32_t uu = (uint32_t)unit;
    const uint32_t q3 = xm_div(uu, a.div3);
    const int rt = (int)(uu - q3 * 3u);
    const uint32_t qs = xm_div(q3, a.divSG);
    const int seg = (int)(q3 - qs * a.divSG.d);
    const int n = (int)xm_div(qs, a.divNS), strip = (int)(qs - (uint32_t)n * a.divNS.d);
    const int pw0 = seg * a.pWo / a.SG, pw1 = (seg + 1) * a.pWo / a.SG;
    const int ncol = 2 * (pw1 - pw0) + 1;                  // output columns 2 pw0 .. 2 pw1
    // source rows: tile a, lane q, tap u: 2 (126 strip + 2 q) + gh0 + u = 4 q + b0 + u; units of 4 rows from U0
    const int b0 = 252 * strip + a.gh0;
    const int U0 = b0 >> 2, delta = b0 - 4 * U0;           // (arithmetic shift: floor for the negative first unit)
    // B read of (tile type tb, e): off = delta + e + 4 half + 2 tb -> plane off & 3, index q + (off >> 2)
    unsigned ladd[2][4];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int off = delta + e + 4 * half + 2 * tb;
        ladd[tb][e] = (unsigned)(((off & 3) * kSfPlane + (off >> 2) + l31) * 4);
      }
    const f32x4 *const pa = reinterpret_cast<const f32x4 *>(sA) + rt * 64 + half * 32 + l31;   // + v * TM * 64
    // staging of two source columns: lane -> (column idx / 68, unit idx % 68), idx = lane + 64 it
    auto load_cols = [&](f32x4 (&nl)[3], int scA, int count) {
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int idx = lane + 64 * it, cs = idx >= kSfPlane ? 1 : 0, k = idx - kSfPlane * cs;
        const int sc = scA + cs, r = 4 * (U0 + k);
        const bool in = cs < count && idx < 2 * kSfPlane && sc >= 0 && sc < a.LimW && r >= 0 && r < a.LimH;
        nl[it] = *reinterpret_cast<const f32x4 *>(in ? a.X + (size_t)n * a.xSampleStride + (size_t)sc * a.LimH + r : a.X);
        if (!in) nl[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto write_cols = [&](const f32x4 (&nl)[3], int slotA, int slotB) {
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const int idx = lane + 64 * it, cs = idx >= kSfPlane ? 1 : 0, k = idx - kSfPlane * cs;
        if (idx < 2 * kSfPlane && (cs ? slotB : slotA) >= 0) {
          float *d = ring + (cs ? slotB : slotA) * kSfSlot + k;
          d[0] = nl[it].x, d[kSfPlane] = nl[it].y, d[2 * kSfPlane] = nl[it].z, d[3 * kSfPlane] = nl[it].w;
        }
      }
    };
    // prologue: the seven source columns under the first output column into slots 0 .. 6
    const int sc0 = 2 * (2 * pw0) + a.gw0;
    {
      f32x4 nl[3];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        load_cols(nl, sc0 + 2 * pr, pr < 3 ? 2 : 1);
        write_cols(nl, 2 * pr, pr < 3 ? 2 * pr + 1 : -1);     // (the fourth call loads one column)
      }
    }
    // running window state: best value and packed codes of the open window, per (window tile w, register i)
    float best[2][16];
    unsigned codes[8];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int i = 0; i < 16; ++i) best[w][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) codes[i] = 0u;
    // stores: lane part of the offset (window row q of window tile w; 4 half filters up), invalid windows out of range
    unsigned vy[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int wq = 32 * w + l31, ph = 63 * strip + wq;
      vy[w] = (wq < 63 && ph < a.pHo) ? (unsigned)(4 * half * pHW + ph) : 0x3FFFFFFFu;
    }
    const int sbase = (n * a.M + 32 * rt) * pHW;

    auto column = [&](auto rtag, int c) {
      constexpr int r = decltype(rtag)::value;               // position in the ring period: slots (2 r + v) % 7
      f32x4 nl[3];
      load_cols(nl, sc0 + 2 * c + 7, c + 1 < ncol ? 2 : 0);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc[2][2];
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[w][tb][i] = 0.f;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const f32x4 af = pa[v * TM * 64];
          const int slot = (2 * r + v) % 7;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int imm = (slot * kSfSlot + 32 * w) * 4;
            float bA = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ring) + ladd[0][e] + imm);
            float bB = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ring) + ladd[1][e] + imm);
            if (v == 0 && e == 3) {                          // k = 7: the constant term multiplies ones
              bA = half ? 1.f : bA;
              bB = half ? 1.f : bB;
            }
            acc[w][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bA, acc[w][0], 0, 0, 0);
            acc[w][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bB, acc[w][1], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      write_cols(nl, (2 * r) % 7, (2 * r + 1) % 7);          // (LDS operations of a wave execute in order: behind the B reads)
      // ---- pooling ------------------------------------------------------------------------------------------------------
      const bool odd = c & 1;
      const int pw = pw0 + (c >> 1) - 1;                     // the window column an even output column closes
#pragma unroll
      for (int w = 0; w < 2; ++w) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float av = fmaxf(acc[w][0][i], 0.f), bv = fmaxf(acc[w][1][i], 0.f);
          float an = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, av), 0x130, 0xf, 0xf, false));
          if (w == 0) {
            // a[q + 1] of lanes 31 / 63: lanes 0 / 32 of the second even tile
            const float a1 = fmaxf(acc[1][0][i], 0.f);
            const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a1), 0));
            const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a1), 32));
            asm volatile("v_writelane_b32 %0, %1, 31\n\tv_writelane_b32 %0, %2, 63" : "+v"(an) : "s"(s0), "s"(s1));
          }
          const float m1 = fmaxf(av, bv);
          unsigned dh = bv > av ? 1u : 0u;
          const float V = fmaxf(m1, an);
          dh = an > m1 ? 2u : dh;
          const int ci = (w * 16 + i) >> 2, sh = 8 * ((w * 16 + i) & 3);
          const unsigned cur = (codes[ci] >> sh) & 0xFFu;
          if (odd) {
            const bool take = V > best[w][i];
            best[w][i] = take ? V : best[w][i];
            codes[ci] = take ? (codes[ci] & ~(0xFFu << sh)) | ((dh + 3u) << sh) : codes[ci];
          } else {
            if (c > 0) {
              const bool take = V > best[w][i];
              const float yv = take ? V : best[w][i];
              const unsigned cv = yv > 0.f ? (take ? dh + 6u : cur) : 255u;
              if (32 * rt + 8 * (i >> 2) < a.M) {            // (wave-uniform: M % 8 == 0)
                const int so = sbase + (8 * (i >> 2) + (i & 3)) * pHW + pw * a.pHo;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), yrsrc, (int)(vy[w] * 4u), so * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)cv, amrsrc, (int)vy[w], so, 0);
              }
            }
            best[w][i] = V;
            codes[ci] = (codes[ci] & ~(0xFFu << sh)) | (dh << sh);
          }
        }
      }
    };
    for (int c0 = 0; c0 < ncol; c0 += 7) {
      if (c0 + 0 < ncol) column(std::integral_constant<int, 0>{}, c0 + 0);
      if (c0 + 1 < ncol) column(std::integral_constant<int, 1>{}, c0 + 1);
      if (c0 + 2 < ncol) column(std::integral_constant<int, 2>{}, c0 + 2);
      if (c0 + 3 < ncol) column(std::integral_constant<int, 3>{}, c0 + 3);
      if (c0 + 4 < ncol) column(std::integral_constant<int, 4>{}, c0 + 4);
      if (c0 + 5 < ncol) column(std::integral_constant<int, 5>{}, c0 + 5);
      if (c0 + 6 < ncol) column(std::integral_constant<int, 6>{}, c0 + 6);
    }
  }
}

}  // namespace xm
