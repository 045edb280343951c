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
// dz is never a tensor: conv_stem_wgrad_pool_kernel composes it on chip from the pooled side -- a pooled element
// (channel, ph, pw) with routing code dh + 3 dw lands on conv pixel (2 ph + dh, 2 pw + dw); each pooled element is loaded
// once per candidate column as part of a 16-byte quad of its pooled column (conv_stem_wgrad_bnp_kernel, the round-5 kernel:
// 48 pooled vector-memory instructions per wave and 32 pixels, each fetching 12 + 4 bytes per lane) and the <= 4
// contributions to a pixel are added in registers in a fixed order.  The forward half of the layer
// (conv_stem_bnpool_fwd_kernel, second part of this file) writes the pooled output and the table without writing x either.
#pragma once
#include "conv_kernels.h"

namespace xm {

struct StemPoolArgs {
  const float *X;                    // the convolution's input [H][W][1][N]
  float *part;                       // gram: [grid][64][64], wgrad: [logical wave][32][64]
  int M, R, nU, nV;                  // filters, taps (nU * nV), filter rows / columns
  int PI, PJ;                        // output pixel grid
  FastDiv divJG, divG;               // stem_gram_kernel: (column pairs) * (256-row groups), 256-row groups: block unit -> (sample, pair, group)
                                     // conv_stem_wgrad_pool_kernel: column pairs per sample, 64-row chunk pairs per column: wave unit -> (sample, pair, chunk pair)
  int gsx, gh0, gw0, LimH, LimW;     // forward gather geometry (as ConvGemmArgs)
  int xSampleStride;
  const float *dP, *yP;              // pooled DZDY, pooled forward output [pHo][pWo][M][N]
  const unsigned char *amax;         // routing table (first maximum, code = dh + 3 dw)
  unsigned dpBytes, amBytes;
  int pHo, pWo;
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
constexpr int kSpCst = 64;                    // 32 ones + 32 zeros: what the B lanes of the ones column / the padding columns read
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
  // (eight loads in flight per thread: with one, the 48 partials of a thread were 48 serial memory round trips -- 26 us)
  double v = 0.0;
  int b = g;
  for (; b + 16 * 7 < nblk; b += 16 * 8) {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = part[(size_t)(b + 16 * k) * (64 * 64) + si * 64 + sj];
#pragma unroll
    for (int k = 0; k < 8; ++k) v += (double)x[k];
  }
  for (; b < nblk; b += 16) v += (double)part[(size_t)b * (64 * 64) + si * 64 + sj];
  red[g][j] = v;
  __syncthreads();
  if (g == 0) {
    v = red[0][j];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += red[k][j];
    gram[i * 64 + j] = v;
  }
}

// ---- A = dz P~ with dz composed from the pooled derivative -----------------------------------------------------------------
// A WAVE owns one row tile of 32 filters (rt = logical wave index % 3, fixed for the kernel: its 32 accumulator registers
// hold its share of A for the whole launch) and walks (sample, column pair jp, 64-row chunk pair cp) units.  Per unit it
// loads, for window columns p = 0: jp - 1 and p = 1: jp, the 32 window rows [32 cp, 32 cp + 32) of its 64 output rows as
// 16-byte quads (lane -> filter lane / 4 (+ 16), quad lane % 4 of chunk h) -- pooled derivative, routing codes and (GATE)
// the pooled forward output -- plus the single window row 32 cp - 1 in front (lanes with lane % 4 == 0: p = 0, == 1: p = 1).
// A quad that would reach past its pooled column reads the head of the next one (masked); in the last column of the tensor
// it is shifted back (no load leaves the tensor) and re-aligned after the load.  The derivatives are gated once and stay in
// registers for the four (column, chunk) steps of the unit; each step composes the wave's dz region [32 filters][32 pixels]
// in registers (compose_slice below), stores it with two 16-byte LDS stores per filter and multiplies: 32 MFMAs per step.
// Output column 2 jp takes dw = 0 of window column jp and dw = 2 of jp - 1; column 2 jp + 1 takes dw = 1 of jp.  The next
// unit's operands and source patch are requested a whole unit ahead; the patch is parked in LDS behind the unit's last B
// reads (LDS operations of a wave execute in order, no barrier in the loop).
// (Earlier versions, profiles/r06/stem_chain_dissection.txt: 96 accumulators per wave with the row tiles as an inner loop;
// three waves per SIMD; scatter with ds_add_f32 into a zero-filled region -- each slower)
// GATE: the ReLU gate is taken from y_pool (> 0); false: the table marks closed windows itself (code 255, as
// conv_stem_bnpool_fwd_kernel writes it) and y_pool is not read.
// Logical wave index: XCD x (blocks b % 8 == x) holds a contiguous range, so that the waves that work on neighbouring
// units at the same time -- and share pooled lines and source columns -- share an L2.
constexpr int kSp3Wave = kSpPatch + 32 * kSpTP;          // a wave's source patch + its dz region
constexpr int kSp3Smem = (4 * kSp3Wave + kSpCst) * 4;    // 43 KB per block (two blocks per CU: 229 registers per lane)
template <int SY, bool GATE>
__global__ void __launch_bounds__(256, 2)
conv_stem_wgrad_pool_kernel(const StemPoolArgs a, const int nunits) {
  constexpr int TP = kSpTP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  const int lc = lane >> 3, lk = lane & 7;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int nb8 = gridDim.x >> 3;                                            // blocks per XCD (grid % 24 == 0)
  const int wl = ((blockIdx.x & 7) * nb8 + (blockIdx.x >> 3)) * 4 + wv;      // logical wave
  const int rt = wl % 3, wc = wl / 3, nwc = (gridDim.x * 4) / 3;
  for (int i = t; i < 4 * kSp3Wave / 4; i += 256) reinterpret_cast<f32x4 *>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float *const sW = smem + wave * kSp3Wave;           // this wave's source patch
  float *const reg = sW + kSpPatch;                   // this wave's dz region [32][TP]
  float *const cst = smem + 4 * kSp3Wave;
  for (int i = t; i < kSpCst; i += 256) cst[i] = i < 32 ? 1.f : 0.f;
  const __amdgpu_buffer_rsrc_t dprsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dP, 0, a.dpBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yprsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.yP, 0, a.dpBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t amrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.amax, 0, a.amBytes, 0x00020000);
  const int pHW = a.pHo * a.pWo;
  const int cl = lane >> 2, qd = lane & 3;
  const bool mok[2] = {32 * rt + cl < a.M, 32 * rt + 16 + cl < a.M};      // filters that do not exist: nothing is routed

  auto unit_of = [&](int u) {
    SpUnit c;
    const uint32_t uu = (uint32_t)min(u, nunits - 1);
    const uint32_t q = xm_div(uu, a.divG);                // divG here: chunk pairs per column
    c.cp = (int)(uu - q * a.divG.d);
    c.n = (int)xm_div(q, a.divJG);                        // divJG here: column pairs per sample
    c.jp = (int)(q - (uint32_t)c.n * a.divJG.d);
    c.live = u < nunits;
    c.col1 = 2 * c.jp + 1 < a.PJ;
    c.chunk1 = 64 * c.cp + 32 < a.PI;
    return c;
  };
  f32x4 ld[6];
  int ldst[6];
  bool ldz[6];
  auto issue_patch = [&](const SpUnit &c) {
    const int iF = 64 * c.cp, iL = min(iF + 63, a.PI - 1);
    const int lo4 = (SY * iF + a.gh0 + 4) >> 2;
    const int n0 = c.live ? ((SY * iL + a.gh0 + 4 + 7) >> 2) - lo4 + 1 : 0;
    const int ncol = a.gsx + a.nV;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int col = it < 5 ? lc : 8, q = it < 5 ? lk + 8 * it : lane;
      const bool wr = col < ncol && q < n0 && (it < 5 || lane < 40);
      const int cc = a.gsx * 2 * c.jp + a.gw0 + col, r = 4 * (lo4 + q) - 4;
      const bool in = wr && cc >= 0 && cc < a.LimW && r >= 0 && r < a.LimH;
      ldst[it] = wr ? col * kSpHW + 4 * q : kSpNC * kSpHW;
      ldz[it] = !in;
      ld[it] = *reinterpret_cast<const f32x4 *>(in ? a.X + (size_t)c.n * a.xSampleStride + (size_t)cc * a.LimH + r : a.X);
    }
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int it = 0; it < 6; ++it) *reinterpret_cast<f32x4 *>(sW + ldst[it]) = ldz[it] ? f32x4{0.f, 0.f, 0.f, 0.f} : ld[it];
  };
  int tapoff[2], cstoff[2];
  bool isTap[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int n = l31 + 32 * jt;
    const int v = n < a.R ? n / a.nU : 0, u = n < a.R ? n - v * a.nU : 0;
    tapoff[jt] = u + kSpHW * v;
    isTap[jt] = n < a.R;
    cstoff[jt] = n == a.R ? 0 : 32;
  }

  // ---- per-unit lane geometry of the candidates ---------------------------------------------------------------------------
  struct Cand {
    unsigned voff[2];     // per chunk h: element offset of the lane's quad inside a (sample, 16-filter group, window column 0) block
    unsigned vmask;       // bit 4 h + e: element e of the chunk-h quad is a window row of this column (< pHo)
    int shift[2];         // != 0 only for the LAST pooled column of the tensor: the quad was loaded `shift` rows early
    bool anyshift;        // (wave-uniform) some lane of this unit has a shifted quad
    unsigned exoff;       // the window row 32 cp - 1 (lanes with qd < 2: window column p = qd): element offset inside a
                          // (sample, 16-filter group) block, or out of range
    bool ok[2];           // window column p exists (wave-uniform)
    int sbase;            // element offset of (sample, filter 32 rt, window column 0, window row 0)
    int jp;
  };
  auto cand_of = [&](const SpUnit &c) {
    Cand k;
    k.jp = c.jp;
    k.ok[1] = c.live && c.jp < a.pWo;
    k.ok[0] = c.live && c.jp >= 1 && c.jp - 1 < a.pWo;
    // a quad that reaches past its pooled column reads the head of the next one (ignored: vmask) -- except in the very last
    // column of the tensor, where it would leave the buffer: there it is loaded early and re-aligned after the load
    const bool lastplane = c.n * a.M + 32 * rt + 32 >= (a.dpBytes >> 2) / (unsigned)pHW;      // (wave-uniform)
    k.anyshift = lastplane && c.jp >= a.pWo - 1 && 32 * c.cp + 32 > a.pHo;
    k.vmask = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ph0 = 32 * c.cp + 16 * h + 4 * qd;        // first window row of the lane's quad
      const int phs = k.anyshift ? min(ph0, a.pHo - 4) : ph0;
      k.shift[h] = ph0 - phs;
      k.voff[h] = (unsigned)(cl * pHW + phs);
#pragma unroll
      for (int e = 0; e < 4; ++e) k.vmask |= (ph0 + e < a.pHo ? 1u : 0u) << (4 * h + e);
    }
    const bool exok = c.cp > 0 && qd < 2 && (qd ? k.ok[1] : k.ok[0]);
    k.exoff = exok ? (unsigned)((cl * a.pWo + c.jp - 1 + qd) * a.pHo + 32 * c.cp - 1) : 0x3FFFFFFFu;
    k.sbase = (c.n * a.M + 32 * rt) * pHW;
    return k;
  };
  // pooled operands of a unit: [window column p][group of 16 filters][chunk].  n*: as loaded, for the NEXT unit (requested a
  // whole unit ahead); c*: gated, the unit being multiplied.  e*: the window row in front of the unit's first one
  f32x4 cdp[2][2][2], ndp[2][2][2], nyp[2][2][2];
  unsigned ccd[2][2][2], ncd[2][2][2];
  float edp[2], nedp[2] = {0.f, 0.f}, neyp[2] = {0.f, 0.f};
  unsigned ecd[2], necd[2] = {0u, 0u};
  auto issue_cand = [&](const Cand &k) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (k.ok[p]) {
          typedef unsigned u4 __attribute__((ext_vector_type(4)));
          const int so = k.sbase + 16 * it * pHW + (k.jp - 1 + p) * a.pHo;     // scalar part (elements)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            ndp[p][it][h] = __builtin_bit_cast(f32x4, (u4)__builtin_amdgcn_raw_buffer_load_b128(dprsrc, (int)(k.voff[h] * 4u), so * 4, 0));
            if (GATE) nyp[p][it][h] = __builtin_bit_cast(f32x4, (u4)__builtin_amdgcn_raw_buffer_load_b128(yprsrc, (int)(k.voff[h] * 4u), so * 4, 0));
            ncd[p][it][h] = __builtin_amdgcn_raw_buffer_load_b32(amrsrc, (int)k.voff[h], so, 0);
          }
        }
      }
      const int so = k.sbase + 16 * it * pHW;
      nedp[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dprsrc, (int)(k.exoff * 4u), so * 4, 0));
      if (GATE) neyp[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yprsrc, (int)(k.exoff * 4u), so * 4, 0));
      necd[it] = __builtin_amdgcn_raw_buffer_load_b8(amrsrc, (int)k.exoff, so, 0);
    }
  };
  // gate, once per unit: the derivative of a window whose maximum did not pass the ReLU, of a filter or a window row that
  // does not exist, is zero
  auto gate_cand = [&](const Cand &k) {
    if (k.anyshift) {
      // (one unit per launch) quads loaded `shift` rows early: element j of the quad is loaded element j + shift
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int st = 0; st < 3; ++st)
              if (k.shift[h] > st) {
                ndp[p][it][h] = f32x4{ndp[p][it][h].y, ndp[p][it][h].z, ndp[p][it][h].w, 0.f};
                if (GATE) nyp[p][it][h] = f32x4{nyp[p][it][h].y, nyp[p][it][h].z, nyp[p][it][h].w, 0.f};
                ncd[p][it][h] >>= 8;
              }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool v = k.ok[p] && mok[it] && ((k.vmask >> (4 * h + e)) & 1u) && (!GATE || nyp[p][it][h][e] > 0.f);
            cdp[p][it][h][e] = v ? ndp[p][it][h][e] : 0.f;
          }
          ccd[p][it][h] = ncd[p][it][h];
        }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      edp[it] = (mok[it] && (!GATE || neyp[it] > 0.f)) ? nedp[it] : 0.f;      // (a row that does not exist was loaded out of range: 0)
      ecd[it] = necd[it];
    }
  };
  // dz of (column jj, chunk h) into the wave's region, composed in REGISTERS: lane (cl, qd) owns window rows 4 qd .. 4 qd + 3
  // of the chunk for filters cl and 16 + cl, i.e. pixel rows 8 qd .. 8 qd + 8 -- element e with code dh + 3 dw adds to row
  // 2 e + dh when dw names this column (three compare / select / add per element, no LDS atomics, no zero fill, no
  // lane predicate) -- the ninth row belongs to the next lane (DPP inside the quad of lanes) or, from lane qd = 3, to pixel
  // row 0 of the next chunk (kept in c8 from step h = 0 to step h = 1; the unit's first chunk takes the window row in front
  // of the unit, e*).  Two 16-byte LDS stores per filter.
  // (First version: ds_add_f32 per routed element behind a lane predicate, ~60 per unit and wave: 0.4 ms of the launch.)
  float c8[2] = {0.f, 0.f};
  float r9[9];
  // slice idx = 0 .. 15 of the composition of step (jj, h): filter group it = idx / 8, window column p = (idx / 4) % 2,
  // element e = idx % 4; the slice behind a group's last element finishes the group (carry rows, two 16-byte stores)
  auto compose_slice = [&](int jj, int h, int idx, float *dst) {
    const int it = idx >> 3, p = (idx >> 2) & 1, e = idx & 3;
    if ((idx & 7) == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) r9[i] = 0.f;
    }
    if (p == 1 || jj == 0) {
      const unsigned d3 = p == 1 ? 3u * (unsigned)jj : 6u;
      const unsigned d = ((ccd[p][it][h] >> (8 * e)) & 0xFFu) - d3;
      const float v = cdp[p][it][h][e];
      r9[2 * e] += d == 0u ? v : 0.f;
      r9[2 * e + 1] += d == 1u ? v : 0.f;
      r9[2 * e + 2] += d == 2u ? v : 0.f;
    }
    if ((idx & 7) == 7) {
      // pixel row 0: the ninth row of the lane before (quad of lanes), of lane qd = 3 in the chunk before, or the window
      // row in front of the unit
      const float fromprev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r9[8]), 0x90 /* quad_perm [0,0,1,2] */, 0xf, 0xf, false));
      float first;
      if (h == 0) {
        // lanes qd = 0 / 1 hold the row in front for window column jp - 1 (dw = 2: column 2 jp only) / jp (dw = jj)
        const unsigned want = qd ? 3u * (unsigned)jj + 2u : 8u;
        const float xv = ((qd || jj == 0) && ecd[it] == want) ? edp[it] : 0.f;
        first = xv + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xv), 0xF5 /* quad_perm [1,1,3,3] */, 0xf, 0xf, false));
      } else {
        first = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c8[it]), 0xFF /* quad_perm [3,3,3,3] */, 0xf, 0xf, false));
      }
      r9[0] += qd == 0 ? first : fromprev;
      if (h == 0) c8[it] = r9[8];
      float *const d = dst + (16 * it + cl) * TP + 8 * qd;
      *reinterpret_cast<f32x4 *>(d) = f32x4{r9[0], r9[1], r9[2], r9[3]};
      *reinterpret_cast<f32x4 *>(d + 4) = f32x4{r9[4], r9[5], r9[6], r9[7]};
    }
  };
  auto compose = [&](int jj, int h, float *dst) {
#pragma unroll
    for (int idx = 0; idx < 16; ++idx) compose_slice(jj, h, idx, dst);
  };

  f32x16 acc[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;

  __syncthreads();                       // zero fill and constants complete
  int u = wc;
  SpUnit cur = unit_of(u), nxt = unit_of(u + nwc);
  Cand kn = cand_of(nxt);                               // geometry of the unit whose operands are in flight (n*)
  if (u < nunits) {
    const Cand k0 = cand_of(cur);
    issue_patch(cur);
    issue_cand(k0);
    write_patch();
    gate_cand(k0);
    __builtin_amdgcn_sched_barrier(0);
    issue_cand(kn);                                     // the next unit's operands and patch: a whole unit ahead
    if (!GATE) issue_patch(nxt);                        // (GATE: y_pool's quads leave no registers for a patch in flight)
  }
  const float *tr = reg + l31 * TP + 16 * half;        // A operand: row l31, pixels 16 half + 4 c .. + 3
  for (; u < nunits; u += nwc) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int jj = st >> 1, h = st & 1;
      const bool run = (jj == 0 || cur.col1) && (h == 0 || cur.chunk1);
      if (run) compose(jj, h, reg);
      __builtin_amdgcn_sched_barrier(0);
      if (run) {
        const int iF = 64 * cur.cp;
        const int pbo = (iF + 32 * h + 16 * half) * SY + a.gh0 + 4 - 4 * ((SY * iF + a.gh0 + 4) >> 2) + a.gsx * jj * kSpHW;
        const float *pl[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) pl[jt] = isTap[jt] ? sW + pbo + tapoff[jt] : cst + cstoff[jt];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 af = *reinterpret_cast<const f32x4 *>(tr + 4 * c);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float b0 = pl[0][SY * (4 * c + e)], b1 = pl[1][SY * (4 * c + e)];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], b1, acc[1], 0, 0, 0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // this unit's operands are dead: gate the next unit's (requested a unit ago), request the ones after them, park the
    // next unit's patch behind this unit's last B reads (LDS operations of a wave execute in order)
    if (GATE) issue_patch(nxt);
    gate_cand(kn);
    write_patch();
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
    nxt = unit_of(u + 2 * nwc);
    kn = cand_of(nxt);
    issue_cand(kn);
    if (!GATE) issue_patch(nxt);
  }
  // the wave's partial [32][64]
  float *out = a.part + (size_t)wl * (32 * 64);
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 * jt + l31] = acc[jt][r];
}

// ---- dF~, dg, db from A and G (fp64; one block per filter) -----------------------------------------------------------------
__global__ void __launch_bounds__(1024)
stem_pool_finalize_kernel(const float *__restrict__ part, int nwc, const double *__restrict__ gram,
                          const float *__restrict__ f, const float *__restrict__ bias, const float *__restrict__ bn_g,
                          const float *__restrict__ moments, int M, int R, int train, float *__restrict__ df,
                          float *__restrict__ dbias, float *__restrict__ dg, float *__restrict__ db) {
  // part: [logical wave 3 wc + rt][32][64] (conv_stem_wgrad_pool_kernel); filter m sits in row m % 32 of the waves rt = m / 32.
  // Sixteen groups of 64 taps, eight loads in flight per thread, fixed summation order.
  __shared__ double red[16][64];
  __shared__ double sA[64], sF[64], sS[2];
  const int tt = threadIdx.x & 63, grp = threadIdx.x >> 6, m = blockIdx.x;
  const int rt = m >> 5, row = m & 31;
  double v = 0.0;
  int b = grp;
  for (; b + 16 * 7 < nwc; b += 16 * 8) {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = part[(size_t)(3 * (b + 16 * k) + rt) * (32 * 64) + row * 64 + tt];
#pragma unroll
    for (int k = 0; k < 8; ++k) v += (double)x[k];
  }
  for (; b < nwc; b += 16) v += (double)part[(size_t)(3 * b + rt) * (32 * 64) + row * 64 + tt];
  red[grp][tt] = v;
  __syncthreads();
  if (grp == 0) {
    double sum = red[0][tt];
#pragma unroll
    for (int k = 1; k < 16; ++k) sum += red[k][tt];
    sA[tt] = sum;
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
constexpr int kSfSmem = (kSfA + 4 * kSfRing) * 4; // bytes per block (51.9 KB; XM_SF_OCC blocks per CU: 227 registers per lane)

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
    // staging of two source columns, 68 units of 16 bytes each: loads 0 / 2 take units 0 .. 63 of column A / B (lane =
    // unit), loads 1 / 3 units 64 .. 67 (lanes 0 .. 3) -- every instruction has ONE destination slot, so the LDS address is
    // lane base + immediate
    auto load_cols = [&](f32x4 (&nl)[4], int scA, int count) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int cs = it >> 1, k = lane + 64 * (it & 1);
        const int sc = scA + cs, r = 4 * (U0 + k);
        const bool in = cs < count && k < kSfPlane && sc >= 0 && sc < a.LimW && r >= 0 && r < a.LimH;
        nl[it] = *reinterpret_cast<const f32x4 *>(in ? a.X + (size_t)n * a.xSampleStride + (size_t)sc * a.LimH + r : a.X);
        if (!in) nl[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto write_cols = [&](const f32x4 (&nl)[4], int slotA, int slotB) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int slot = (it >> 1) ? slotB : slotA;
        if (slot >= 0 && ((it & 1) == 0 || lane < kSfPlane - 64)) {
          float *d = ring + slot * kSfSlot + 64 * (it & 1) + lane;
          d[0] = nl[it].x, d[kSfPlane] = nl[it].y, d[2 * kSfPlane] = nl[it].z, d[3 * kSfPlane] = nl[it].w;
        }
      }
    };
    // prologue: the seven source columns under the first output column into slots 0 .. 6
    const int sc0 = 2 * (2 * pw0) + a.gw0;
    {
      f32x4 nl[4];
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

    const int ngrp = min(4, (a.M - 32 * rt) >> 3);           // groups of 8 filters of this row tile that exist (M % 8 == 0)
    auto column = [&](auto rtag, auto fulltag, int c) {
      constexpr int r = decltype(rtag)::value;               // position in the ring period: slots (2 r + v) % 7
      constexpr bool FULL = decltype(fulltag)::value;        // all 32 filters of the row tile exist
      f32x4 nl[4];
      load_cols(nl, sc0 + 2 * c + 7, c + 1 < ncol ? 2 : 0);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 acc[2][2];
#pragma unroll
      for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[w][tb][i] = 0.f;
      // (the wave in its MFMA phase goes first: two waves of a SIMD that start together then settle half a period apart,
      // one multiplying while the other pools and stores -- without it both phases of both waves simply added up)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        // (an opaque zero per tile pair: without it the seven filter fragments of the row tile are hoisted out of the unit's
        // column loop -- 28 registers held for the whole unit in a kernel that sits on its register limit)
        int zero = 0;
        asm volatile("" : "+v"(zero));
        const f32x4 *const paw = pa + zero;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const f32x4 af = paw[v * TM * 64];
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
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      write_cols(nl, (2 * r) % 7, (2 * r + 1) % 7);          // (LDS operations of a wave execute in order: behind the B reads)
      // ---- pooling ------------------------------------------------------------------------------------------------------
      // on the bnorm's RAW output: max and the ReLU commute, and the first maximum of the rectified window is the first
      // maximum of the raw one whenever it is positive -- a window whose maximum is <= 0 is closed (value 0, code 255)
      const int pwoff = (pw0 + (c >> 1) - 1) * a.pHo;        // the window column an even output column closes
      // MODE 0: odd column (update), 1: first column of the segment (open a window), 2: even column (close one, open the next)
      auto pool = [&](auto modetag) {
        constexpr int MODE = decltype(modetag)::value;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float av = acc[w][0][i], bv = acc[w][1][i];
            float an = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, av), 0x130, 0xf, 0xf, false));
            if (w == 0) {
              // a[q + 1] of lanes 31 / 63: lanes 0 / 32 of the second even tile
              // (everything in asm: with the readlane builtin on a vector element the compiler read element 0 for every i)
              asm volatile("v_readlane_b32 s6, %1, 0\n\tv_readlane_b32 s7, %1, 32\n\ts_nop 3\n\t"
                           "v_writelane_b32 %0, s6, 31\n\tv_writelane_b32 %0, s7, 63"
                           : "+v"(an) : "v"(acc[1][0][i]) : "s6", "s7");
            }
            const float V = fmaxf(fmaxf(av, bv), an);
            unsigned dh = V == bv ? 1u : 2u;
            dh = V == av ? 0u : dh;
            const int ci = (w * 16 + i) >> 2, sh = 8 * ((w * 16 + i) & 3);
            if (MODE == 0) {
              const bool take = V > best[w][i];
              best[w][i] = fmaxf(V, best[w][i]);
              codes[ci] = take ? (codes[ci] & ~(0xFFu << sh)) | ((dh + 3u) << sh) : codes[ci];
            } else {
              if (MODE == 2 && (FULL || (i >> 2) < ngrp)) {
                const bool take = V > best[w][i];
                const float yv = fmaxf(fmaxf(V, best[w][i]), 0.f);
                const unsigned cur = (codes[ci] >> sh) & 0xFFu;
                const unsigned cv = yv > 0.f ? (take ? dh + 6u : cur) : 255u;
                const int so = sbase + (8 * (i >> 2) + (i & 3)) * pHW + pwoff;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), yrsrc, (int)(vy[w] * 4u), so * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)cv, amrsrc, (int)vy[w], so, 0);
              }
              best[w][i] = V;
              codes[ci] = (codes[ci] & ~(0xFFu << sh)) | (dh << sh);
            }
          }
        }
      };
      if (c & 1) pool(std::integral_constant<int, 0>{});
      else if (c == 0) pool(std::integral_constant<int, 1>{});
      else pool(std::integral_constant<int, 2>{});
    };
    const bool full = 32 * rt + 32 <= a.M;
#define XM_SF_COL(R_)                                                                          \
  if (c0 + R_ < ncol) {                                                                        \
    if (full) column(std::integral_constant<int, R_>{}, std::true_type{}, c0 + R_);            \
    else column(std::integral_constant<int, R_>{}, std::false_type{}, c0 + R_);                \
  }
    for (int c0 = 0; c0 < ncol; c0 += 7) {
      XM_SF_COL(0) XM_SF_COL(1) XM_SF_COL(2) XM_SF_COL(3) XM_SF_COL(4) XM_SF_COL(5) XM_SF_COL(6)
    }
#undef XM_SF_COL
  }
}

}  // namespace xm
