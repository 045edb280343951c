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
//   wgrad   : m = output channel, p = filter tap, reduction over output pixels
//   G(p, r) = X[base(p) + off(r)] if the tap lands inside the source image, else 0 -- an
//             im2col that only ever exists as LDS tiles (never in HBM).
// MATLAB layout is H-fastest, so pixels are the contiguous axis of both the gather source and
// the destination: pixels go on the MFMA "column" (lane & 31) axis, which makes every global
// access of a wave a run of consecutive addresses along H.
//
// CDNA4 specifics used here
//   * gathers are raw buffer loads: an out-of-image tap gets byte offset 0xFFFFFFFF, the buffer
//     bounds check returns 0.0f -- no select, no branch, no mask in the staging path;
//   * validity of a tap depends only on its (u,v) for a given pixel: each thread computes a
//     64-bit "invalid (u,v)" mask once, a tap then costs bfe + or + add;
//   * the K loop is unrolled x2 so that every LDS address is base + immediate;
//   * sched_group_barrier interleaves the next stage's address arithmetic and buffer loads
//     between the current stage's MFMAs (the MFMA pipe is busy 64 cycles per instruction, the
//     wave can issue ~15 other instructions meanwhile), so one wave per SIMD already overlaps;
//   * split-K (grid.y) for layers with few output tiles; partial slabs are combined in a fixed
//     order by conv_splitk_epilogue_kernel (deterministic, no atomics).
#pragma once
#include <type_traits>

#include "xm_common.h"

namespace xm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvGemmArgs {
  const float *A;     // [M][lda], lda % 4 == 0, 16-byte aligned, zero padded to Rp columns
  const float *X;     // gather source
  float *Y;           // destination
  float *slab;        // splits > 1: [splits][M][NPs] raw partial sums
  const int2 *taps;   // Rp + 3*kBK entries {byte offset, (u,v) index}; padding entries use index 63
  const float *bias, *scale, *shift, *resid;
  // != NULL: per-(row, sample) multiplier applied after scale / shift and before the residual: gate[row + gateStride *
  // sample] -- the SE excite a .* x folded into the 1 x 1 projection that produces x (xm_nnconv_forward_gated)
  const float *gate;
  int gateStride;
  int relu;
  unsigned xBytes;    // size of the gather source in bytes (buffer bounds check)
  unsigned aBytes;    // size of A in bytes (LDS-DMA variant: A is read through a buffer descriptor too)
  unsigned tapStride; // LDS-DMA variant: byte distance between consecutive taps (1x1: H*W*4)
  int dmaOk;          // 1: this launch may use conv_gemm_dma_kernel
  int lda, M, Rp, Rtrue;
  int PI, PJ, NP, NPs;  // pixel grid (i fastest), total pixel count PI*PJ*N, slab row pitch
  FastDiv divPIJ, divPI;
  int gsy, gsx, gh0, gw0;  // gather origin of pixel (i,j): (i*gsy + gh0, j*gsx + gw0)
  int LimH, LimW, xSampleStride;
  int nU, nV, du0, dus, dv0, dvs;  // tap (iu,iv) sits at (du0 + iu*dus, dv0 + iv*dvs) from the origin
  int osy, osx, oh0, ow0, OH;      // destination of pixel (i,j): (oh0 + i*osy, ow0 + j*osx)
  int oChanStride, oSampleStride;
  // GEMM row m -> destination offset (m / oMU) * oChanStride + (m % oMU) * oUStride.  oMU == 1 is the
  // plain channel map; oMU == FH folds the filter rows of an H-collapsing conv (Ho == 1) into M.
  FastDiv divMU;
  int oUStride;
  int vecStore;       // 1: 4 consecutive pixels are contiguous & 16-B friendly -> dwordx4 epilogue
  int nbm, nbn;       // tile counts
  int tilesPerSplit, nkt;
  // Hybrid schedule of conv_gemm_kernel (hyS > 1): the first hyFull tiles (whole rounds of the chip) are computed by one
  // block each; each of the remaining tiles -- the partly filled last round -- is split hyS ways along the reduction
  // (hyTps stages per split), its partial sums go to the slab ([split][M][NPs], pixel index relative to hyP0) and
  // conv_splitk_epilogue_kernel combines them over the pixel range [hyP0, NP).  grid.x = hyFull + (tiles - hyFull) * hyS.
  int hyFull, hyS, hyTps, hyP0;
  // != NULL (vecStore, no split-K, not the LDS-DMA kernel): every BLOCK also leaves {sum, sum of squares} of the
  // values it stores, per row: statPart[pixel tile bn][row][2] (rows contiguous: one coalesced run per block).  The batch
  // moments of the train-mode bnorm that follows the convolution then cost no second pass over Y (conv_forward).
  float *statPart;
  int statNcg;        // number of pixel tiles (= nbn)
  // conv_halo_kernel (3 x 3 taps, unit stride): geometry of the zero-padded input patch a pixel tile works from.
  // Patch row 0 / padded column 0 of a sample are source row hpRmin / source column hpCmin; a column has hpHP rows, a
  // sample hpWP padded columns; tap t = iu + 3 iv sits hpSh[t] BYTES behind a pixel's own patch position.
  int hpHP, hpWP, hpRmin, hpCmin, hpN;
  // != 0: the pixel grid is enumerated with PI rounded UP (PI, divPI, divPIJ, NP describe the padded grid) so that a
  // 128-pixel tile is a whole number of columns; rows >= piReal of every column do not exist and are never stored
  int piReal;
  int hpSh[9];
  FastDiv hpDivHP, hpDivWP;
  double algoFlops;   // profiler only: algorithmic FLOPs of this launch (0 = 2*M*NP*Rtrue)
  unsigned long long *dbgCycles;  // debug (xm_debug_conv_cycles): per-block {first clock, last clock, HW_ID, XCC_ID}
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

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
// AV consecutive floats with one instruction (whole group in or out of range)
template <int AV>
__device__ __forceinline__ auto buf_load_v(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  if constexpr (AV == 4) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f4, v);
  } else {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f2, v);
  }
}

// 4x4 transpose across the 4 lanes of a quad (DPP quad_perm, full-rate VALU): on entry lane i holds
// v[k] = element (row k, column i); on exit v[k] = element (row i, column k).
__device__ __forceinline__ float dpp_quad(float x, const int ctrl_is_xor1) {
  int xi = __builtin_bit_cast(int, x);
  int r = ctrl_is_xor1 ? __builtin_amdgcn_mov_dpp(xi, 0xB1, 0xF, 0xF, true)   // quad_perm [1,0,3,2]
                       : __builtin_amdgcn_mov_dpp(xi, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ void quad_transpose4(float (&v)[4], int iq) {
  const bool b0 = iq & 1, b1 = iq & 2;
  float s0 = b0 ? v[0] : v[1], s1 = b0 ? v[2] : v[3];
  s0 = dpp_quad(s0, 1);
  s1 = dpp_quad(s1, 1);
  if (b0) { v[0] = s0; v[2] = s1; } else { v[1] = s0; v[3] = s1; }
  float t0 = b1 ? v[0] : v[2], t1 = b1 ? v[1] : v[3];
  t0 = dpp_quad(t0, 0);
  t1 = dpp_quad(t1, 0);
  if (b1) { v[0] = t0; v[1] = t1; } else { v[2] = t0; v[3] = t1; }
}

// sum over the 8 lanes l31 = iq + 4 q (q = 0..7) of each 32-lane half: two DPP rotations inside the 16-lane rows, one
// swizzle across the two rows.  Every lane ends up with the sum of its residue class iq = l31 & 3.
__device__ __forceinline__ float quad_class_sum8(float x) {
  int xi = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x124, 0xF, 0xF, false));   // row_ror:4
  xi = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x128, 0xF, 0xF, false));   // row_ror:8
  xi = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(xi, 0x401F));                      // lane ^ 16
  return x;
}

// sched_group_barrier masks
#define XM_SGB_VALU 0x002
#define XM_SGB_MFMA 0x008
#define XM_SGB_VMEM_RD 0x020
#define XM_SGB_DS_RD 0x100
#define XM_SGB_DS_WR 0x200

// LDS image of both operands: [stage][group g = k/4][row][k%4] so that one ds_read_b128 hands a
// lane four consecutive k of its row.  MFMA 32x32x2 takes k from lane>>5, so lanes 0-31 read
// group 2s and lanes 32-63 group 2s+1; the e-th MFMA of a chunk then sums k = 8s+e and 8s+4+e
// -- A and B use the same assignment, so the dot product is complete and exact.
#define XM_MFMA_E(E, AF, BF)                                                   \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                               \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                           \
      if (TM * TN == 1 && ((E) & 1))                                           \
        accx = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[i][E], BF[j][E], accx, 0, 0, 0); \
      else                                                                     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[i][E], BF[j][E], acc[i][j], 0, 0, 0); \
    }
// Consecutive MFMAs always target DIFFERENT accumulators (k outer, tile inner): an instruction
// issued between two MFMAs on the same accumulator costs ~43 extra cycles on gfx950
// (MI355X_MICROARCH.md), and the interleaved staging code sits exactly there.  A 1x1 wave tile
// splits its chain over two accumulators (odd / even k), added once at the end.
#define XM_MFMA_CHUNK(AF, BF)                                                  \
  XM_MFMA_E(0, AF, BF)                                                         \
  XM_MFMA_E(1, AF, BF)                                                         \
  XM_MFMA_E(2, AF, BF)                                                         \
  XM_MFMA_E(3, AF, BF)
// fragments of 8-k chunk S (0 / 1) of LDS buffer CUR
#define XM_READ_FRAGS(CUR, S, AF, BF)                                          \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                               \
    AF[i] = *reinterpret_cast<const f32x4 *>(sAr + ((CUR) * kNG + 2 * (S)) * PLA + i * 128); \
  _Pragma("unroll") for (int j = 0; j < TN; ++j)                               \
    BF[j] = *reinterpret_cast<const f32x4 *>(sBr + ((CUR) * kNG + 2 * (S)) * PLB + j * 128);

// interleave recipe for one 8-k chunk: the fragment reads of the NEXT chunk first (they have the
// whole chunk of MFMAs to land), then each MFMA followed by a few VALU and one buffer load
#define XM_INTERLEAVE(NVALU)                                                   \
  __builtin_amdgcn_sched_group_barrier(XM_SGB_DS_RD, TM + TN, 0);              \
  _Pragma("unroll") for (int q_ = 0; q_ < 4 * TM * TN; ++q_) {                 \
    __builtin_amdgcn_sched_group_barrier(XM_SGB_MFMA, 1, 0);                   \
    __builtin_amdgcn_sched_group_barrier(XM_SGB_VALU, NVALU, 0);              \
    __builtin_amdgcn_sched_group_barrier(XM_SGB_VMEM_RD, 1, 0);                \
  }

// Epilogue shared by the implicit-GEMM kernels: split-K slab store, or y = act((acc + bias) * scale + shift +
// residual) with scalar-cache row constants and 16-byte stores through an in-quad transpose.
// Epilogue stores.  ASMST (the persistent LDS-DMA kernel): inline-asm stores the compiler does not track.  That
// kernel keeps its epilogue INSIDE the k loop; with ordinary stores the compiler's wait-count pass sees "VMEM
// possibly outstanding" on the loop back edge and puts `s_waitcnt vmcnt(0)` at the loop header -- in front of
// every k stage -- which also drains the hand-counted LDS-DMA prefetch ring (the asm loads share vmcnt): the ring
// degenerated to one stage of look-ahead and the epilogue loads each waited for the stores issued before them.
// gfx9 stores read their data registers in issue order, so nothing has to wait for an untracked store.
template <bool ASMST>
__device__ __forceinline__ void xm_st16(float *p, f32x4 v) {
  if constexpr (ASMST)
    // trailing s_nop 1: the compiler pads nothing after an asm statement, and a VALU write to the data registers in
    // the two wait states behind a > 8-byte store would overtake the store's read of them
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else
    *reinterpret_cast<f32x4 *>(p) = v;
}
template <bool ASMST>
__device__ __forceinline__ void xm_st4(float *p, float v) {
  if constexpr (ASMST)
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else
    *p = v;
}

template <int TM, int TN, int WGM, int WGN, bool ASMST = false>
__device__ __forceinline__ void conv_gemm_epilogue(const ConvGemmArgs &a, f32x16 (&acc)[TM][TN], int bm, int bn,
                                                   int split, int wm, int wn, int half, int l31,
                                                   float *sred = nullptr /* LDS, >= 2 * WGN * BM floats (statPart) */,
                                                   bool toSlab = true /* a.slab != NULL: this block writes partials */) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  // C/D map of 32x32 MFMA: col = lane & 31 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  if (a.slab && toSlab) {
    // split-K: raw partial sums, [split][m][p] with p contiguous (hybrid schedule: p relative to hyP0)
    float *out = a.slab + (size_t)split * a.M * a.NPs - a.hyP0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int p = bn * BN + (wn * TN + j) * 32 + l31;
      if (p >= a.NP) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = bm * BM + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (m < a.M) xm_st4<ASMST>(out + (size_t)m * a.NPs + p, acc[i][j][r]);
        }
    }
    return;
  }
  // ---- epilogue: y = act((acc + bias) * scale + shift + residual) ----
  // Per-row constants: per-lane loads of the rows the lane stores, all in flight together (both paths below).
  const int wbase = __builtin_amdgcn_readfirstlane(bm * BM + wm * TM * 32);
  int obase[TN], gbase[TN];   // gbase: gateStride * sample of the lane's pixel (a.gate)
  bool pok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int p = bn * BN + (wn * TN + j) * 32 + l31;
    pok[j] = p < a.NP;
    uint32_t pc = pok[j] ? p : a.NP - 1;
    uint32_t n = xm_div(pc, a.divPIJ);
    gbase[j] = (int)n * a.gateStride;
    uint32_t q = pc - n * a.divPIJ.d;
    uint32_t jj = xm_div(q, a.divPI);
    uint32_t ii = q - jj * a.divPI.d;
    obase[j] = (a.oh0 + (int)ii * a.osy) + a.OH * (a.ow0 + (int)jj * a.osx) + (int)n * a.oSampleStride;
    if (a.piReal) pok[j] = pok[j] && (int)ii < a.piReal;
  }
  if (a.vecStore) {
    // Wide path (dword stores are issue-bound at ~2.5 TB/s on this chip): each group of 4
    // accumulator registers holds 4 consecutive channel rows of one pixel; a 4x4 transpose inside
    // the lane quad turns that into 4 consecutive pixels of one row -> one 16-byte store per lane.
    const int iq = l31 & 3;
    // Per-row constants of the rows THIS lane stores after the transpose (one row per (i, g4)), fetched with
    // per-lane loads that are all in flight together.  (They used to come through the scalar cache group by group
    // -- 8-16 s_loads and an s_waitcnt per group: a chain of eight scalar round trips per tile, 17-20 % of the
    // store-heavy layers.)  Same fma operands as before the transpose, so the results are bit-identical.
    float rmul[TM][4], radd[TM][4];
    {
      // one uniform branch per KIND of constant, all its loads back to back (a branch per row group makes the
      // compiler wait for each group before it issues the next)
      int rowc[TM][4];
      float bi_[TM][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          rowc[i][g4] = min(wbase + i * 32 + 8 * g4 + 4 * half + iq, a.M - 1);
          rmul[i][g4] = 1.f;
          radd[i][g4] = 0.f;
          bi_[i][g4] = 0.f;
        }
      if (a.scale) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            rmul[i][g4] = a.scale[rowc[i][g4]];
            radd[i][g4] = a.shift[rowc[i][g4]];
          }
      }
      if (a.bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) bi_[i][g4] = a.bias[rowc[i][g4]];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) radd[i][g4] += bi_[i][g4] * rmul[i][g4];
      }
    }
    float st1[TM][4], st2[TM][4];   // a.statPart: this lane's share of {sum, sum of squares} per row it stores
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) st1[i][g4] = 0.f, st2[i][g4] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int offs[4][TN];
      bool ok[4][TN];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int row = wbase + i * 32 + 8 * g4 + 4 * half + iq;  // the row this lane stores after the transpose
        const int rowc = min(row, a.M - 1);
        uint32_t mc = xm_div((uint32_t)rowc, a.divMU);
        const int moff = (int)mc * a.oChanStride + (rowc - (int)mc * (int)a.divMU.d) * a.oUStride;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          offs[g4][j] = obase[j] - iq + moff;  // pixel quad base (4 consecutive pixels contiguous)
          ok[g4][j] = row < a.M && pok[j];
        }
      }
      // ASMST: the residual quads of the whole row tile first -- a load issued behind untracked stores makes the
      // compiler's vmcnt(0) in front of its use wait for those stores as well (in-order counter).
      // explicit 16-byte accesses: written element by element the compiler keeps four dword loads / stores
      // (it cannot prove the alignment), i.e. 4x the VMEM instructions and quarter-filled cache lines
      float gv[4][TN];   // a.gate: multiplier of (row this lane stores, sample of its pixel quad)
      if (a.gate) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int rowc_ = min(wbase + i * 32 + 8 * g4 + 4 * half + iq, a.M - 1);
#pragma unroll
          for (int j = 0; j < TN; ++j) gv[g4][j] = a.gate[rowc_ + gbase[j]];
        }
      }
      f32x4 rv[ASMST ? 4 : 1][ASMST ? TN : 1];
      if (ASMST && a.resid) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            rv[g4 % (ASMST ? 4 : 1)][j % (ASMST ? TN : 1)] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok[g4][j]) rv[g4 % (ASMST ? 4 : 1)][j % (ASMST ? TN : 1)] = *reinterpret_cast<const f32x4 *>(a.resid + offs[g4][j]);
          }
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * g4 + k];
          quad_transpose4(v, iq);
          if (ok[g4][j]) {
            f32x4 o = {v[0] * rmul[i][g4] + radd[i][g4], v[1] * rmul[i][g4] + radd[i][g4],
                       v[2] * rmul[i][g4] + radd[i][g4], v[3] * rmul[i][g4] + radd[i][g4]};
            if (a.gate) o *= gv[g4][j];
            if (a.resid) {
              if constexpr (ASMST)
                o += rv[g4 % (ASMST ? 4 : 1)][j % (ASMST ? TN : 1)];
              else
                o += *reinterpret_cast<const f32x4 *>(a.resid + offs[g4][j]);
            }
            if (!ASMST && a.statPart) {
              st1[i][g4] += (o.x + o.y) + (o.z + o.w);
              st2[i][g4] += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
            }
            if (a.relu) {
              o.x = fmaxf(o.x, 0.f);
              o.y = fmaxf(o.y, 0.f);
              o.z = fmaxf(o.z, 0.f);
              o.w = fmaxf(o.w, 0.f);
            }
            xm_st16<ASMST>(a.Y + offs[g4][j], o);
          }
        }
      }
    }
    if (!ASMST && a.statPart) {
      // lanes iq + 4 q of a half hold pixel quads of the SAME row: sum them (DPP), park the wave's 32 * TM row sums in
      // LDS, add the WGN waves that share the rows in wave order, and store the block's rows as ONE contiguous run.
      // (Per-wave partials stored straight from the lanes -- 8-byte writes, every lane a different row -- made the
      // student's first layer 58 us slower: 3.6 M quarter-filled cache lines.)
      __syncthreads();                       // every wave is past its last fragment read: the operand tiles are dead
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float u = quad_class_sum8(st1[i][g4]), v = quad_class_sum8(st2[i][g4]);
          const int rl = (wm * TM + i) * 32 + 8 * g4 + 4 * half + iq;       // row inside the block tile
          if ((l31 >> 2) == 0) *reinterpret_cast<float2 *>(sred + 2 * (wn * BM + rl)) = make_float2(u, v);
        }
      __syncthreads();
      const int t = threadIdx.x;
      if (t < BM && bm * BM + t < a.M) {
        float u = 0.f, v = 0.f;
#pragma unroll
        for (int w = 0; w < WGN; ++w) {
          const float2 q = *reinterpret_cast<const float2 *>(sred + 2 * (w * BM + t));
          u += q.x;
          v += q.y;
        }
        *reinterpret_cast<float2 *>(a.statPart + ((size_t)bn * a.M + bm * BM + t) * 2) = make_float2(u, v);
      }
    }
    return;
  }
  if constexpr (ASMST) return;  // the LDS-DMA kernel is only dispatched with vecStore (conv_forward: dmaOk)
  // scalar-store path (pixel quads straddle samples: H*W % 4 != 0).  Row constants: per-lane loads, all 16 rows of a
  // row tile in flight together (see the wide path).
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float rmul[16], radd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mc_ = min(wbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.M - 1);
      float m_ = 1.f, t_ = 0.f;
      if (a.scale) {
        m_ = a.scale[mc_];
        t_ = a.shift[mc_];
      }
      if (a.bias) t_ += a.bias[mc_] * m_;  // (acc + b) * s + t == acc * s + (b * s + t)
      rmul[r] = m_;
      radd[r] = t_;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      uint32_t mc = xm_div((uint32_t)min(m, a.M - 1), a.divMU);
      const int moff = (int)mc * a.oChanStride + (min(m, a.M - 1) - (int)mc * (int)a.divMU.d) * a.oUStride;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (m < a.M && pok[j]) {
          int off = obase[j] + moff;
          float v = acc[i][j][r] * rmul[r] + radd[r];
          if (a.gate) v *= a.gate[min(m, a.M - 1) + gbase[j]];
          if (a.resid) v += a.resid[off];
          if (a.relu) v = fmaxf(v, 0.f);
          xm_st4<ASMST>(a.Y + off, v);
        }
      }
    }
  }
}

// MODE 0: every tap is inside the image (pad == 0, Rp == R);  MODE 1: (u,v) validity mask.
template <int TM, int TN, int WGM, int WGN, int MODE>
__device__ __forceinline__ void conv_gemm_body(const ConvGemmArgs &a) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  constexpr int NT = 64 * WGM * WGN;                 // 4 waves per block; 8 in the low-register configuration (kCfgs[7])
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves per block");
  static_assert(BN == 32 || BN == 64 || BN == 128 || BN == 256, "BN must divide the block");
  constexpr int PLA = BM * 4 + 4, PLB = BN * 4 + 4;  // plane strides (floats), 16-B multiples
  constexpr int NUA = (kNG * BM + NT - 1) / NT;        // float4 units per thread, A
  constexpr int NUB = (kNG * BN + NT - 1) / NT;        // float4 units per thread, B
  constexpr bool UNIFORM = BN >= 64;                 // tap group is wave-uniform
  __shared__ __attribute__((aligned(16))) float smem[2 * kNG * (PLA + PLB)];
  float *sA = smem;
  float *sB = smem + 2 * kNG * PLA;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave % WGM, wn = wave / WGM;
  int tile, split, kt0, kt1;
  bool toSlab = true;
  if (a.hyS > 1) {
    // hybrid: whole tiles for the full rounds, the remainder tiles split along the reduction (see ConvGemmArgs)
    const int b = (int)blockIdx.x;
    if (b < a.hyFull) {
      tile = xcd_remap(b, a.hyFull);
      split = 0, kt0 = 0, kt1 = a.nkt;
      toSlab = false;
    } else {
      const int r = b - a.hyFull;
      tile = a.hyFull + r / a.hyS;
      split = r - (r / a.hyS) * a.hyS;
      kt0 = split * a.hyTps;
      kt1 = min(a.nkt, kt0 + a.hyTps);
    }
  } else {
    tile = xcd_remap(blockIdx.x, a.nbm * a.nbn);
    split = blockIdx.y;
    kt0 = split * a.tilesPerSplit;
    kt1 = min(a.nkt, kt0 + a.tilesPerSplit);
  }
  const int bm = tile % a.nbm, bn = tile / a.nbm;

  // ---- per-thread gather geometry (fixed for the whole reduction) ----
  const int pl = t % BN, gB0 = t / BN;
  unsigned xbase4;            // byte offset of the pixel's gather origin (may wrap below zero)
  unsigned inv_lo = 0, inv_hi = 0x80000000u;  // bit (u,v) set <=> that tap is outside the image
  {
    int p = bn * BN + pl;
    uint32_t pc = p < a.NP ? p : a.NP - 1;  // clamp: columns >= NP are never stored
    uint32_t n = xm_div(pc, a.divPIJ);
    uint32_t q = pc - n * a.divPIJ.d;
    uint32_t j = xm_div(q, a.divPI);
    uint32_t i = q - j * a.divPI.d;
    int ph = (int)i * a.gsy + a.gh0;
    int pw = (int)j * a.gsx + a.gw0;
    xbase4 = (unsigned)(ph + a.LimH * pw + (int)n * a.xSampleStride) * 4u;
    if (MODE == 1) {
      // validity is separable: tap (iu, iv) is outside the image iff its row OR its column is.
      // nU + nV checks, then the per-column copies of the row mask are OR-ed into a 64-bit mask.
      unsigned rowinv = 0;
      for (int iu = 0; iu < a.nU; ++iu)
        rowinv |= ((unsigned)(ph + a.du0 + iu * a.dus) < (unsigned)a.LimH ? 0u : 1u) << iu;
      const unsigned rowall = a.nU >= 32 ? 0xFFFFFFFFu : ((1u << a.nU) - 1u);
      unsigned long long inv = 0;
      for (int iv = 0; iv < a.nV; ++iv) {
        bool okw = (unsigned)(pw + a.dv0 + iv * a.dvs) < (unsigned)a.LimW;
        inv |= (unsigned long long)(okw ? rowinv : rowall) << (iv * a.nU);
      }
      inv_lo |= (unsigned)inv;
      inv_hi |= (unsigned)(inv >> 32);
    }
  }
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, a.xBytes, 0x00020000);
  const int2 *__restrict__ taps = a.taps;

  // A operand: per-unit row pointer (rows >= M clamped: they only feed rows never stored)
  const float *aptr[NUA];
#pragma unroll
  for (int i = 0; i < NUA; ++i) {
    int u = t + NT * i;
    if (kNG * BM % NT != 0 && u >= kNG * BM) u = kNG * BM - 1;
    int g = u % kNG, m = u / kNG;
    int gm = min(bm * BM + m, a.M - 1);
    aptr[i] = a.A + (size_t)gm * a.lda + g * 4;
  }
  // LDS addresses: everything below is base + compile-time immediate
  float *sAw[NUA], *sBw[NUB];
#pragma unroll
  for (int i = 0; i < NUA; ++i) {
    int u = t + NT * i;
    int g = u % kNG, m = u / kNG;
    sAw[i] = sA + g * PLA + m * 4;
  }
#pragma unroll
  for (int i = 0; i < NUB; ++i) {
    int g = gB0 + i * (NT / BN);
    sBw[i] = sB + g * PLB + pl * 4;
  }
  const int half = lane >> 5, l31 = lane & 31;
  const float *sAr = sA + half * PLA + (wm * TM * 32 + l31) * 4;
  const float *sBr = sB + half * PLB + (wn * TN * 32 + l31) * 4;

  // Two register sets: loads are issued TWO stages ahead of their use (prefetch distance 2), so a
  // stage's global-load latency is covered by two compute phases and twice as many loads are in
  // flight -- what the short-K / small-tile layers need (their compute phase is ~512 cycles).
  f32x4 ra0[NUA], rb0[NUB], ra1[NUA], rb1[NUB];
  int tpo[NUB * 4], tpi[NUB * 4];  // tap entries of the next tile to be loaded

#define XM_FETCH_TAPS(KT)                                                      \
  _Pragma("unroll") for (int i = 0; i < NUB; ++i) {                            \
    int g_ = gB0 + i * (NT / BN);                                              \
    if (kNG * BN % NT != 0) g_ = min(g_, kNG - 1);                             \
    int r0_ = (KT) * kBK + g_ * 4;                                             \
    if (UNIFORM) r0_ = __builtin_amdgcn_readfirstlane(r0_);                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                            \
      int2 t_ = taps[r0_ + e];                                                 \
      tpo[4 * i + e] = t_.x;                                                   \
      tpi[4 * i + e] = t_.y;                                                   \
    }                                                                          \
  }

#define XM_LOAD_TILE(KT, RA, RB)                                               \
  _Pragma("unroll") for (int i = 0; i < NUA; ++i)                              \
    RA[i] = *reinterpret_cast<const f32x4 *>(aptr[i] + (KT) * kBK);            \
  _Pragma("unroll") for (int i = 0; i < NUB; ++i) {                            \
    unsigned off_[4];                                                          \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                            \
      off_[e] = xbase4 + (unsigned)tpo[4 * i + e];                             \
      if (MODE == 1) {                                                         \
        int idx_ = tpi[4 * i + e];                                             \
        unsigned w_ = idx_ >= 32 ? inv_hi : inv_lo;                            \
        off_[e] |= (unsigned)(((int)(w_ << (31 - (idx_ & 31)))) >> 31);        \
      }                                                                        \
    }                                                                          \
    RB[i].x = buf_load(xrsrc, off_[0]);                                        \
    RB[i].y = buf_load(xrsrc, off_[1]);                                        \
    RB[i].z = buf_load(xrsrc, off_[2]);                                        \
    RB[i].w = buf_load(xrsrc, off_[3]);                                        \
  }

#define XM_STORE_TILE(BUF, RA, RB)                                             \
  _Pragma("unroll") for (int i = 0; i < NUA; ++i)                              \
    if (kNG * BM % NT == 0 || t + NT * i < kNG * BM)                           \
      *reinterpret_cast<f32x4 *>(sAw[i] + (BUF) * kNG * PLA) = RA[i];          \
  _Pragma("unroll") for (int i = 0; i < NUB; ++i)                              \
    if (kNG * BN % NT == 0 || gB0 + i * (NT / BN) < kNG)                       \
      *reinterpret_cast<f32x4 *>(sBw[i] + (BUF) * kNG * PLB) = RB[i];

  // One pipeline stage on LDS buffer CUR.  The MFMA operands are double-buffered in REGISTERS: the
  // fragments of a chunk are read from LDS while the previous chunk's MFMAs run, so no MFMA ever
  // waits on an LDS round trip.  To make that work across stages with two LDS buffers, the
  // store-to-LDS + barrier sit in the MIDDLE of the stage:
  //   issue stage KT+2's global loads into (LA, LB)
  //   read chunk 1 of CUR -> (af1, bf1)        | MFMAs of chunk 0 from (af0, bf0)
  //   park stage KT+1 (in flight in (SA, SB) since the previous stage) in LDS[CUR^1]; barrier
  //   read chunk 0 of CUR^1 -> (af0, bf0)      | MFMAs of chunk 1 from (af1, bf1)
  // LDS[CUR^1] was last read (its chunk 1) at the start of the previous stage, i.e. before the
  // previous mid-stage barrier, so overwriting it here is safe.
#define XM_FETCH_EARLY(KT) XM_FETCH_TAPS(KT)
#define XM_FETCH_LATE(KT)
#define XM_STAGE_LD(KT, CUR, LA, LB, SA, SB)                                   \
  XM_LOAD_TILE((KT) + 2, LA, LB)                                               \
  XM_FETCH_EARLY((KT) + 3)                                                     \
  XM_READ_FRAGS(CUR, 1, af1, bf1)                                              \
  XM_MFMA_CHUNK(af0, bf0)                                                      \
  XM_INTERLEAVE(MODE == 1 ? 6 : 4)                                             \
  __builtin_amdgcn_sched_barrier(0);                                           \
  XM_FETCH_LATE((KT) + 3)                                                      \
  XM_STORE_TILE((CUR) ^ 1, SA, SB)                                             \
  __syncthreads();                                                             \
  XM_READ_FRAGS((CUR) ^ 1, 0, af0, bf0)                                        \
  XM_MFMA_CHUNK(af1, bf1)                                                      \
  XM_INTERLEAVE(MODE == 1 ? 6 : 4)                                             \
  __builtin_amdgcn_sched_barrier(0);
  // same without issuing new loads (the last stages of the reduction)
#define XM_STAGE_NL(CUR, SA, SB)                                               \
  XM_READ_FRAGS(CUR, 1, af1, bf1)                                              \
  XM_MFMA_CHUNK(af0, bf0)                                                      \
  __builtin_amdgcn_sched_barrier(0);                                           \
  XM_STORE_TILE((CUR) ^ 1, SA, SB)                                             \
  __syncthreads();                                                             \
  XM_READ_FRAGS((CUR) ^ 1, 0, af0, bf0)                                        \
  XM_MFMA_CHUNK(af1, bf1)                                                      \
  __builtin_amdgcn_sched_barrier(0);
#define XM_STAGE_LAST(CUR)                                                     \
  XM_READ_FRAGS(CUR, 1, af1, bf1)                                              \
  XM_MFMA_CHUNK(af0, bf0)                                                      \
  XM_MFMA_CHUNK(af1, bf1)

  f32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
  f32x16 acc[TM][TN], accx;
#pragma unroll
  for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // the tap table carries 3*kBK padding entries, so fetching ahead is always legal
  if (kt0 < kt1) {
    XM_FETCH_TAPS(kt0)
    XM_LOAD_TILE(kt0, ra0, rb0)
    XM_FETCH_TAPS(kt0 + 1)
    if (kt0 + 1 < kt1) {
      XM_LOAD_TILE(kt0 + 1, ra1, rb1)
      XM_FETCH_TAPS(kt0 + 2)
    }
    XM_STORE_TILE(0, ra0, rb0)
    __syncthreads();
    XM_READ_FRAGS(0, 0, af0, bf0)
    int kt = kt0;
    for (; kt + 3 < kt1; kt += 2) {
      XM_STAGE_LD(kt, 0, ra0, rb0, ra1, rb1)
      XM_STAGE_LD(kt + 1, 1, ra1, rb1, ra0, rb0)
    }
    const int rem = kt1 - kt;  // 1, 2 or 3 stages left; LDS[0] holds stage kt, set 1 stage kt+1
    if (rem == 3) {
      XM_STAGE_LD(kt, 0, ra0, rb0, ra1, rb1)
      XM_STAGE_NL(1, ra0, rb0)
      XM_STAGE_LAST(0)
    } else if (rem == 2) {
      XM_STAGE_NL(0, ra1, rb1)
      XM_STAGE_LAST(1)
    } else {
      XM_STAGE_LAST(0)
    }
  }
#undef XM_FETCH_TAPS
#undef XM_FETCH_EARLY
#undef XM_FETCH_LATE
#undef XM_LOAD_TILE
#undef XM_STORE_TILE
#undef XM_STAGE_LD
#undef XM_STAGE_NL
#undef XM_STAGE_LAST

  if (TM * TN == 1) acc[0][0] += accx;
  conv_gemm_epilogue<TM, TN, WGM, WGN>(a, acc, bm, bn, split, wm, wn, half, l31, smem, toSlab);
}

template <int TM, int TN, int WGM, int WGN, int MODE>
__global__ void __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2)   // (threads, waves per SIMD): two blocks per CU
conv_gemm_kernel(const ConvGemmArgs a) {
#ifdef XM_DEBUG_CYCLES
  // per-block shader-clock trace (tools/conv_bench.py --cycles).  Compile-time only: even as a never-taken
  // branch it cost the whole step 4 % (the start-clock value stays live across the main loop).
  const unsigned long long c0 = __builtin_readcyclecounter();
#endif
  conv_gemm_body<TM, TN, WGM, WGN, MODE>(a);
#ifdef XM_DEBUG_CYCLES
  if (a.dbgCycles && threadIdx.x == 0) {
    const unsigned b = blockIdx.x + gridDim.x * blockIdx.y;
    if (b < 4096) {
      a.dbgCycles[4 * b + 0] = c0;
      a.dbgCycles[4 * b + 1] = __builtin_readcyclecounter();
      a.dbgCycles[4 * b + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
      a.dbgCycles[4 * b + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------
// LDS-DMA variant for the layers that are plain GEMMs in memory: 1x1 convolutions with unit stride, no padding and
// H*W % 4 == 0 (27 of the 36 1x1 layers of the ResNet-50 teachers; their dgrad as well).
//   D[m][p] = sum_c A[m][c] * X[q(p) + HW * (c + C n(p))]
// Both operands go global -> LDS with `buffer_load_dwordx4 ... lds` (CDNA4: 16 bytes per lane, no VGPR staging, no
// ds_write, no per-element address arithmetic):
//   A  [stage][g = k/4][row][k%4]   lane = row, 16 B = 4 consecutive k          (as conv_gemm_kernel)
//   B  [stage][k][pixel]            lane = (k row, pixel quad), 16 B = 4 consecutive pixels of one channel
// The MFMA B fragment is then four ds_read_b32 (one per k) instead of one ds_read_b128; everything else --
// fragment / accumulator mapping, epilogue, split-K slabs -- is conv_gemm_kernel's.
// Pipeline: NST LDS slots; per stage ONE barrier:
//   s_waitcnt vmcnt(own loads of later stages) -> s_barrier -> issue the loads of stage kt+NST-1 into the slot the
//   barrier has just freed -> MFMAs of stage kt.
// The loads are inline asm: for the builtin form the compiler drains vmcnt(0) in front of every barrier (it cannot
// tell the LDS slots apart), which would serialise the pipeline.
struct DmaGeo {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  // raw buffer descriptor (what make_buffer_rsrc builds): base address, stride 0, size in bytes, DATA_FORMAT = 32 bit
  static __device__ __forceinline__ i32x4 rsrc(const void *base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)(uintptr_t)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
  }
  static __device__ __forceinline__ void load16(unsigned lds_byte, unsigned voff, const i32x4 &rsrc, unsigned soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 2\n\t"   // m0 write -> LDS-DMA use, and 5 wait states in all between a VALU-written SGPR operand and its read
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff)
        : "memory");
  }
};

template <int N>
__device__ __forceinline__ void dma_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PERSISTENT: the grid is one round of co-resident blocks (host: a.nbm * a.nbn tiles over gridDim.x blocks); a
// block walks its tiles with the LDS ring running straight through the tile boundaries -- the loads of the next
// tile are in flight while the current one is multiplied and stored, so a short reduction (K = 64 ... 512: 4 ... 32
// stages) no longer pays a memory round trip per tile.  Tiles of one pixel column (same X tile, different filter
// rows) go to neighbouring blocks of ONE XCD at the same time, so X is read from HBM once.
template <int TM, int TN, int WGM, int WGN, int NST>
__global__ void __launch_bounds__(256, (32 * TM * WGM) * (32 * TN * WGN) >= 128 * 128 ? 2 : 3)
conv_gemm_dma_kernel(const ConvGemmArgs a) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  static_assert(WGM * WGN == 4, "4 waves per block");
  static_assert(BM == 64 || BM == 128, "A rows are loaded 64 at a time");
  static_assert(BN == 64 || BN == 128 || BN == 256, "B rows are loaded 256 pixels per instruction");
  constexpr int PLA = BM * 4 + 4;            // floats per k-group plane of A
  constexpr int SA = kNG * PLA;              // floats per stage, A
  constexpr int SB = kBK * BN;               // floats per stage, B (row pitch BN: DMA destinations are linear)
  constexpr int NIA = kNG * (BM / 64);       // A load instructions per stage per block
  constexpr int RPI = 256 / BN;              // k rows per B load instruction
  constexpr int NIB = kBK / RPI;             // B load instructions per stage per block
  constexpr int NWA = (NIA + 3) / 4, NWB = (NIB + 3) / 4;  // per wave
  static_assert(NIA % 4 == 0 && NIB % 4 == 0, "every wave issues the same number of loads (vmcnt bookkeeping)");
  __shared__ __attribute__((aligned(16))) float smem[NST * (SA + SB)];
  float *sA = smem;
  float *sB = smem + NST * SA;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave % WGM, wn = wave / WGM;
  const int split = blockIdx.y;
  const int kt0 = split * a.tilesPerSplit;
  const int kt1 = min(a.nkt, kt0 + a.tilesPerSplit);
  const int nst = kt1 - kt0;                 // stages per tile (this split)
  const int half = lane >> 5, l31 = lane & 31;

  // ---- this block's tiles: XCD x = blockIdx % 8 owns a contiguous segment of the logical tile list (bm fastest);
  // its blocks take the segment round-robin ----
  const int ntl = a.nbm * a.nbn;
  const int G = (int)gridDim.x;
  int seg0, segn, bstep, bidx;
  if (G >= 8 && G % 8 == 0) {
    const int xcd = (int)blockIdx.x % 8, q = ntl / 8, r = ntl % 8;
    seg0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    segn = q + (xcd < r ? 1 : 0);
    bstep = G / 8;
    bidx = (int)blockIdx.x / 8;
  } else {
    seg0 = 0, segn = ntl, bstep = G, bidx = (int)blockIdx.x;
  }
  const int mycnt = bidx < segn ? (segn - bidx + bstep - 1) / bstep : 0;
  const int total = mycnt * nst;             // stages this block runs through

  const DmaGeo::i32x4 arsrc = DmaGeo::rsrc(a.A, a.aBytes);
  const DmaGeo::i32x4 xrsrc = DmaGeo::rsrc(a.X, a.xBytes);
  const unsigned sAbase = (unsigned)(uintptr_t)sA, sBbase = (unsigned)(uintptr_t)sB;   // LDS byte addresses

  // ---- issue side: per-lane global offsets of the tile whose stages are being requested ----
  unsigned avoff[NWA], bvoff[NWB];
  unsigned alds[NWA], blds[NWB];             // wave-uniform LDS byte offsets inside a stage
#pragma unroll
  for (int i = 0; i < NWA; ++i) {
    const int id = wave + 4 * i;
    alds[i] = (unsigned)(((id % kNG) * PLA + (id / kNG) * 64 * 4) * 4);
  }
#pragma unroll
  for (int i = 0; i < NWB; ++i) blds[i] = (unsigned)((wave + 4 * i) * RPI * BN * 4);
  auto set_issue_tile = [&](int tile) {
    const int bm = tile % a.nbm, bn = tile / a.nbm;
#pragma unroll
    for (int i = 0; i < NWA; ++i) {
      const int id = wave + 4 * i;
      const int row = min(bm * BM + (id / kNG) * 64 + lane, a.M - 1);   // rows >= M feed rows that are never stored
      avoff[i] = ((unsigned)row * (unsigned)a.lda + 4u * (id % kNG)) * 4u;
    }
    const int qd = lane % (BN / 4), rr = lane / (BN / 4);
    int p = bn * BN + 4 * qd;
    p = p < a.NP ? p : a.NP - 4;                                        // NP % 4 == 0; clamped quads are never stored
    const uint32_t n = xm_div((uint32_t)p, a.divPIJ);
    const uint32_t q = (uint32_t)p - n * a.divPIJ.d;
    const unsigned pix = (q + n * (unsigned)a.xSampleStride) * 4u;
#pragma unroll
    for (int i = 0; i < NWB; ++i) bvoff[i] = pix + (unsigned)((wave + 4 * i) * RPI + rr) * (unsigned)a.tapStride;
  };
  int ij = 0, ikt = kt0, islot = 0;          // next stage to request: tile sequence index, k stage, LDS slot
  auto issue_next = [&]() {
    const unsigned asoff = (unsigned)ikt * (kBK * 4u);                       // 16 k = 64 bytes along a filter row
    const unsigned bsoff = (unsigned)ikt * (unsigned)kBK * (unsigned)a.tapStride;
#pragma unroll
    for (int i = 0; i < NWA; ++i)
      DmaGeo::load16(__builtin_amdgcn_readfirstlane(sAbase + (unsigned)islot * (SA * 4u) + alds[i]), avoff[i], arsrc, asoff);
#pragma unroll
    for (int i = 0; i < NWB; ++i)
      DmaGeo::load16(__builtin_amdgcn_readfirstlane(sBbase + (unsigned)islot * (SB * 4u) + blds[i]), bvoff[i], xrsrc, bsoff);
    islot = islot + 1 == NST ? 0 : islot + 1;
    if (++ikt == kt1) {
      ikt = kt0;
      if (++ij < mycnt) set_issue_tile(seg0 + bidx + ij * bstep);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float *sAr = sA + half * PLA + (wm * TM * 32 + l31) * 4;
  const float *sBr = sB + (4 * half) * BN + wn * TN * 32 + l31;

  // fragments of 8-k chunk C (0 / 1) of LDS slot SLOT: A one ds_read_b128 per row tile, B four ds_read_b32 (k rows)
#define XM_DREAD(SLOT, C, AF, BF)                                              \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                               \
    AF[i] = *reinterpret_cast<const f32x4 *>(sAr + (SLOT) * SA + (2 * (C)) * PLA + i * 128); \
  _Pragma("unroll") for (int j = 0; j < TN; ++j)                               \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                              \
      BF[j][e] = sBr[(SLOT) * SB + (8 * (C) + e) * BN + j * 32];
#define XM_DMFMA(AF, BF)                                                       \
  _Pragma("unroll") for (int e = 0; e < 4; ++e)                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                             \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[i][e], BF[j][e], acc[i][j], 0, 0, 0);
  // the fragment reads of the NEXT chunk are spread between the MFMAs of the current one
#define XM_DSCHED                                                              \
  _Pragma("unroll") for (int q_ = 0; q_ < 4 * TM * TN; ++q_) {                 \
    __builtin_amdgcn_sched_group_barrier(XM_SGB_MFMA, 1, 0);                   \
    __builtin_amdgcn_sched_group_barrier(XM_SGB_DS_RD, 1, 0);                  \
  }                                                                            \
  __builtin_amdgcn_sched_barrier(0);

  f32x4 af0[TM], af1[TM];
  float bf0[TN][4], bf1[TN][4];
  if (total > 0) {
    set_issue_tile(seg0 + bidx);
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_)
      if (s_ < total) issue_next();
    // stage 0 landed?  (up to NST-2 later stages may stay in flight)
    {
      const int later0 = min(total - 1, NST - 2);
      if (NST >= 6 && later0 >= 4) dma_wait<4 * (NWA + NWB)>();
      else if (NST >= 5 && later0 >= 3) dma_wait<3 * (NWA + NWB)>();
      else if (NST >= 4 && later0 >= 2) dma_wait<2 * (NWA + NWB)>();
      else if (NST >= 3 && later0 >= 1) dma_wait<NWA + NWB>();
      else dma_wait<0>();
    }
    __syncthreads();
    int slot = 0, ckt = 0, ctile = seg0 + bidx;   // compute side: stage inside the tile, logical tile
    XM_DREAD(slot, 0, af0, bf0)
    for (int g = 0; g < total; ++g) {
      // first half: MFMAs of chunk 0 | fragment reads of chunk 1
      XM_DREAD(slot, 1, af1, bf1)
      XM_DMFMA(af0, bf0)
      XM_DSCHED
      const int nslot = slot + 1 == NST ? 0 : slot + 1;
      if (g + 1 < total) {
        // stage g+1 must have landed; stages g+2 .. g+NST-2 (already requested) may stay in flight
        const int later = min(total - 2 - g, NST - 3);
        if (NST >= 6 && later >= 3) dma_wait<3 * (NWA + NWB)>();
        else if (NST >= 5 && later >= 2) dma_wait<2 * (NWA + NWB)>();
        else if (NST >= 4 && later >= 1) dma_wait<NWA + NWB>();
        else dma_wait<0>();
        __syncthreads();   // stage g+1 visible to all waves; every wave is done with stage g-1 (its slot is free)
        if (g + NST - 1 < total) issue_next();
        // second half: MFMAs of chunk 1 | fragment reads of chunk 0 of the next stage
        XM_DREAD(nslot, 0, af0, bf0)
      }
      XM_DMFMA(af1, bf1)
      XM_DSCHED
      slot = nslot;
      if (++ckt == nst) {
        // tile complete: store it (the next tile's loads are already in flight) and start over
        conv_gemm_epilogue<TM, TN, WGM, WGN, true>(a, acc, ctile % a.nbm, ctile / a.nbm, split, wm, wn, half, l31);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        ckt = 0;
        ctile += bstep;
      }
    }
  }
#undef XM_DREAD
#undef XM_DMFMA
#undef XM_DSCHED
}

// ------------------------------------------------------------------------------------------
// Halo-patch variant for unit-stride gathers with T = nU x nV <= 3 x 3 taps: the 3 x 3 convolutions (student conv3-5,
// the sixteen 3 x 3 layers of the ResNet-50 teachers) and their dgrads, and the stride-parity classes of a strided
// dgrad (the student's 5 x 5 / 2 conv2: classes of 3x3, 3x2, 2x3 and 2x2 taps) -- half of the FLOPs of a step.
// conv_gemm_kernel fetches every input element once PER TAP (T gathers through the texture path, each with its own
// address arithmetic and padding mask) and meets at a barrier every 16 reduction steps.  Here a stage is 8 input
// channels: the zero-padded input PATCH under the block's 128 output pixels (all taps of all its pixels: PS floats per
// channel) goes global -> registers -> LDS ONCE, padding rows / columns and sample gaps as out-of-range buffer loads
// (zeros); the taps then read their B operand straight from the patch -- `ds_read_b32` at the lane's own patch position
// + a per-tap shift + a compile-time channel offset, no VALU, no masks -- for 8 T reduction steps (16 T MFMAs per wave
// tile) between barriers.  The filter operand is reordered while it is parked in LDS so that MFMA e of tap t multiplies
// channels e (lanes 0-31) / e + 4 (lanes 32-63):  k = 8 t + c.
//   LDS  A [2 T k-groups][BM rows][4]  +  patch [8 channels][PS]   (T = 9, BM = 128, PS = 512: 37 + 16 KB; one stage --
//   the next stage's operands wait in registers while this one is multiplied, two blocks per CU overlap the store phases)
// Measured on the way (x_fill3x3, TFLOP/s): multiply from LDS only 151; + store phase / barriers 142; + the loads in one
// burst behind the barrier 119 (a wave issues in order: 25 wave-wide loads, most touching 64 cache lines, hold its
// issue slot); loads spread over the taps of the stage 129-135.  The filter quads of a wave-wide load sit in 64
// different cache lines (two threads per row, compile-time reorder): 5 % -- a pre-ordered filter copy would remove it.
// Epilogue, accumulator map, split-K slabs: conv_gemm_kernel's.
constexpr int kHaloCB = 8;

template <int T, int TM, int TN, int WGM, int WGN, int PS>
__device__ __forceinline__ void conv_halo_body(const ConvGemmArgs &a, float *smem) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  static_assert(WGM * WGN == 4 && BN == 128 && 2 * BM <= 256, "four waves, 128 pixels, two staging threads per row");
  static_assert(PS == 512 || PS == 1024, "patch positions map to threads as 4 (t % (PS / 4))");
  constexpr int KS = kHaloCB * T;              // reduction steps per stage
  constexpr int NGRP = 2 * T;                  // k-groups of four
  constexpr int PLA = BM * 4 + 4;              // floats per k-group plane
  constexpr int UC = PS / 4;                   // staging threads per channel
  constexpr int CPP = 256 / UC;                // channels per staging pass (2 / 1)
  constexpr int NI = kHaloCB / CPP;            // patch units (float4) per thread and stage (4 / 8)
  constexpr int NLT = T < 7 ? T : 7;           // the stage's global loads are spread over this many taps
  float *sA = smem;
  float *sP = smem + NGRP * PLA;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave % WGM, wn = wave / WGM;
  const int tile = xcd_remap(blockIdx.x, a.nbm * a.nbn);
  const int bm = tile % a.nbm, bn = tile / a.nbm;
  const int half = lane >> 5, l31 = lane & 31;
  // stages = channels / 8; split-K (grid.y) hands every split a contiguous range of them
  const int split = blockIdx.y;
  const int st0 = split * a.tilesPerSplit;
  const int nst = min(a.nkt, st0 + a.tilesPerSplit) - st0;

  // ---- A staging: thread (row = t / 2, h = t % 2) moves 4 T consecutive reduction indices of its row per stage ----
  const bool arowOk = t < 2 * BM;
  const float *arow = a.A + (size_t)min(bm * BM + min(t >> 1, BM - 1), a.M - 1) * a.lda + 4 * T * (t & 1) + (size_t)st0 * KS;
  float *sAw = sA + (t & 1) * PLA + min(t >> 1, BM - 1) * 4;

  // ---- patch staging: thread owns patch positions 4 (t % UC) .. + 3 of channels CPP i + t / UC ----
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, a.xBytes, 0x00020000);
  unsigned poff[4], pinv[4];
  int jjp0;
  {
    const uint32_t p0 = (uint32_t)min(bn * BN, a.NP - 1);
    const uint32_t n0 = xm_div(p0, a.divPIJ);
    const uint32_t q0 = p0 - n0 * a.divPIJ.d;
    jjp0 = (int)(n0 * (uint32_t)a.hpWP + xm_div(q0, a.divPI));
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t pos = 4u * (uint32_t)(t % UC) + d;
      const uint32_t col = xm_div(pos, a.hpDivHP), r = pos - col * (uint32_t)a.hpHP;
      const uint32_t jjp = (uint32_t)jjp0 + col;
      const uint32_t n = xm_div(jjp, a.hpDivWP), jl = jjp - n * (uint32_t)a.hpWP;
      const int sr = (int)r + a.hpRmin, sc = (int)jl + a.hpCmin;
      const bool ok = (unsigned)sr < (unsigned)a.LimH && (unsigned)sc < (unsigned)a.LimW && (int)n < a.hpN;
      poff[d] = (unsigned)(sr + a.LimH * sc + (int)n * a.xSampleStride) * 4u;
      pinv[d] = ok ? 0u : 0xFFFFFFFFu;
    }
  }
  const unsigned chBytes = (unsigned)(a.LimH * a.LimW) * 4u;
  const int chalf = __builtin_amdgcn_readfirstlane(t / UC);      // wave-uniform (UC is a multiple of 64)
  float *sPw = sP + chalf * PS + 4 * (t % UC);

  // ---- per-lane patch position of the wave's pixels ----
  const float *pb[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const uint32_t p = (uint32_t)min(bn * BN + (wn * TN + j) * 32 + l31, a.NP - 1);
    const uint32_t n = xm_div(p, a.divPIJ);
    const uint32_t q = p - n * a.divPIJ.d;
    const uint32_t jj = xm_div(q, a.divPI), ii = q - jj * a.divPI.d;
    pb[j] = sP + half * 4 * PS + (int)ii + a.hpHP * ((int)(n * (uint32_t)a.hpWP + jj) - jjp0);
  }
  const float *sAr = sA + half * PLA + (wm * TM * 32 + l31) * 4;

  f32x4 ra[T], rb[NI];
  // part P of NLT of the loads of stage S: filter quads P, P + NLT, ... and patch dwords likewise.  (The loads of
  // stage s + 1 are spread over the taps of stage s, each behind a few MFMAs -- see the header.)
#define XM_HLOAD_PART(S, P)                                                    \
  if ((P) < NLT) {                                                             \
    _Pragma("unroll") for (int i_ = (P); i_ < T; i_ += NLT)                    \
      ra[i_] = *reinterpret_cast<const f32x4 *>(arow + (S) * KS + 4 * i_);     \
    _Pragma("unroll") for (int u_ = (P); u_ < 4 * NI; u_ += NLT) {             \
      const unsigned cb_ = (unsigned)((st0 + (S)) * kHaloCB + CPP * (u_ >> 2) + chalf) * chBytes; \
      rb[u_ >> 2][u_ & 3] = buf_load(xrsrc, (poff[u_ & 3] + cb_) | pinv[u_ & 3]); \
    }                                                                          \
  }
  // element e of unit i is reduction index 4 T h + 4 i + e of the stage: channel 4 h + (4 i + e) / T, tap (4 i + e) % T,
  // i.e. k = 8 tap + channel -> k-group 2 tap + h, slot (4 i + e) / T  (h is folded into sAw)
#define XM_HSTORE                                                              \
  if (arowOk) {                                                                \
    _Pragma("unroll") for (int i = 0; i < T; ++i)                              \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                            \
        sAw[2 * ((4 * i + e) % T) * PLA + (4 * i + e) / T] = ra[i][e];         \
  }                                                                            \
  _Pragma("unroll") for (int i = 0; i < NI; ++i)                               \
    *reinterpret_cast<f32x4 *>(sPw + CPP * i * PS) = rb[i];
#define XM_HREAD(TAP, AF, BF)                                                  \
  {                                                                            \
    const int sh_ = a.hpSh[TAP] >> 2;                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                             \
      AF[i] = *reinterpret_cast<const f32x4 *>(sAr + 2 * (TAP) * PLA + i * 128); \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                             \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                            \
        BF[j][e] = pb[j][sh_ + e * PS];                                        \
  }
#define XM_HMFMA(AF, BF)                                                       \
  _Pragma("unroll") for (int e = 0; e < 4; ++e)                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                             \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[i][e], BF[j][e], acc[i][j], 0, 0, 0);
#define XM_HSCHED_LD                                                           \
  _Pragma("unroll") for (int q_ = 0; q_ < 4 * TM * TN; ++q_) {                 \
    __builtin_amdgcn_sched_group_barrier(XM_SGB_MFMA, 1, 0);                   \
    __builtin_amdgcn_sched_group_barrier(XM_SGB_DS_RD, 1, 0);                  \
    if (q_ % 2 == 1) {                                                         \
      __builtin_amdgcn_sched_group_barrier(XM_SGB_VALU, 3, 0);                 \
      __builtin_amdgcn_sched_group_barrier(XM_SGB_VMEM_RD, 1, 0);              \
    }                                                                          \
  }                                                                            \
  __builtin_amdgcn_sched_barrier(0);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 af[2][TM];
  float bf[2][TN][4];

#pragma unroll
  for (int P = 0; P < NLT; ++P) {
    XM_HLOAD_PART(0, P)
  }
  for (int s = 0; s < nst; ++s) {
    if (s > 0) __syncthreads();     // every wave has read the last fragments of stage s - 1
    XM_HSTORE
    __syncthreads();
    XM_HREAD(0, af[0], bf[0])
    __builtin_amdgcn_sched_barrier(0);
    const int sn = min(s + 1, nst - 1);   // (the last stage re-requests itself: no branch inside the scheduling regions)
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      if (tap + 1 < T) {
        XM_HREAD(tap + 1, af[(tap + 1) & 1], bf[(tap + 1) & 1])
      }
      XM_HLOAD_PART(sn, tap)
      XM_HMFMA(af[tap & 1], bf[tap & 1])
      XM_HSCHED_LD
    }
  }
#undef XM_HLOAD_PART
#undef XM_HSTORE
#undef XM_HREAD
#undef XM_HMFMA
#undef XM_HSCHED_LD
  __syncthreads();                  // the epilogue's statistics path reuses the operand tiles
  conv_gemm_epilogue<TM, TN, WGM, WGN>(a, acc, bm, bn, split, wm, wn, half, l31, smem);
}

template <int TM, int WGM, int PS>
constexpr int halo_smem_floats() {
  return 2 * 9 * (32 * TM * WGM * 4 + 4) + kHaloCB * PS;   // sized for T = 9
}

// one problem per launch; T = a.nU * a.nV (9 / 6 / 4)
template <int TM, int TN, int WGM, int WGN, int PS>
__global__ void __launch_bounds__(256, 2)
conv_halo_kernel(const ConvGemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[halo_smem_floats<TM, WGM, PS>()];
  const int T = a.nU * a.nV;
  if (T == 9) conv_halo_body<9, TM, TN, WGM, WGN, PS>(a, smem);
  else if (T == 6) conv_halo_body<6, TM, TN, WGM, WGN, PS>(a, smem);
  else conv_halo_body<4, TM, TN, WGM, WGN, PS>(a, smem);
}

// Several independent implicit GEMMs in ONE launch (blockIdx.z picks the problem): the stride-parity
// classes of a strided dgrad.  Each class alone is ~1.1 rounds of the chip (a 14 % full second round);
// together they pack into whole rounds, and the larger 96-row tile becomes the best choice.
struct ConvGemmMulti {
  ConvGemmArgs c[4];
};
template <int TM, int TN, int WGM, int WGN>
__global__ void __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2)
conv_gemm_multi_kernel(const ConvGemmMulti m) {
  const ConvGemmArgs &a = m.c[blockIdx.z];
  if ((int)blockIdx.x >= a.nbm * a.nbn) return;
  conv_gemm_body<TM, TN, WGM, WGN, 1>(a);
}

// the stride-parity classes of a strided dgrad through the halo-patch body, one launch (blockIdx.z = class)
template <int TM, int TN, int WGM, int WGN, int PS>
__global__ void __launch_bounds__(256, 2)
conv_halo_multi_kernel(const ConvGemmMulti m) {
  __shared__ __attribute__((aligned(16))) float smem[halo_smem_floats<TM, WGM, PS>()];
  const ConvGemmArgs &a = m.c[blockIdx.z];
  if ((int)blockIdx.x >= a.nbm * a.nbn) return;
  const int T = a.nU * a.nV;
  if (T == 9) conv_halo_body<9, TM, TN, WGM, WGN, PS>(a, smem);
  else if (T == 6) conv_halo_body<6, TM, TN, WGM, WGN, PS>(a, smem);
  else conv_halo_body<4, TM, TN, WGM, WGN, PS>(a, smem);
}

// ---- three-channel stem (round 5): 7 x 7 / stride 2 over 224 x 224 x 3 faces, the teachers' conv1 ---------------------------
// K = 147: the implicit-GEMM kernel pads it to 160, gathers 147 taps per pixel (dword loads with padding masks) and runs at
// 72 TFLOP/s (0.84 ms at 256 faces).  Same idea as the stride-2 patch kernels above, for the forward direction: a tile = 128
// consecutive output pixels of one sample (1 - 3 output columns); the input patch under it -- 3 channels x 11 source columns x
// 232 rows (source rows -4 ... 227), every element loaded ONCE with 8-byte loads, padding as out-of-range loads -- sits in LDS
// as [channel][column][row]; tap (u, v, c) of pixel (i, j) is at lane base + (c * 11 + v) * 232 + u with lane base =
// 2 (j - j0) * 232 + 2 i + 1: ds_read_b32 with immediate offsets, no masks, no VALU.  The filter bank lives in LDS for the whole
// (persistent) kernel with the filter rows padded to 8 (k' = u + 8 (v + 7 c), u = 7: zero weights -- 168 reduction steps for 147
// taps): an MFMA step multiplies k' = 2 t (lanes 0-31) and 2 t + 1 (lanes 32-63), i.e. rows u and u + 1 of one filter column,
// so the two half-waves' patch offsets differ by exactly 1 and everything stays an immediate.  Image [k' / 4][row][4] stored as
// {4g, 4g + 2, 4g + 1, 4g + 3}: each half-wave reads ITS two weights of a group with one ds_read_b64.
// One patch buffer (30.6 KB) + the filter image (43 KB): two blocks per CU; the next tile's patch waits in registers (15 8-byte
// asm loads, issued before the MFMAs) and is parked between two barriers; the epilogue (bias / folded bnorm / relu; row
// constants loaded once, they are the same for every tile) leaves through asm stores the waits do not count:
// `s_waitcnt vmcnt(8)` in front of the patch store = the loads are in, this tile's 8 stores may still be in flight.
constexpr int kStem3CS = 232, kStem3G = 42;

// TN = 32-pixel tiles per wave: 1 (128-pixel block tiles, 11 source columns) or 2 (256 pixels, 13 columns: FOUR independent
// accumulator chains per wave instead of two)
template <int TN>
__global__ void __launch_bounds__(256, 2)
conv_stem3_kernel(const ConvGemmArgs a, const int ntiles) {
  constexpr int NC = 3, FW = 7, S = 2, NSC = TN == 1 ? 11 : 13, CS = kStem3CS, G = kStem3G, PR = CS / 2, BP = 128 * TN;
  constexpr int PATCH = NC * NSC * CS;                     // 7656 floats
  constexpr int NLD = (NC * NSC * PR + 255) / 256;         // 15 / 18 row-pair loads per thread and tile
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float sA[G * 64 * 4];
  __shared__ __attribute__((aligned(16))) float sP[PATCH + 8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  // the filter bank in MFMA operand order (zero rows / taps where the layer has none)
  for (int idx = t; idx < G * 64 * 4; idx += 256) {
    const int q = idx & 3, row = (idx >> 2) & 63, g = idx >> 8;
    const int kp = 4 * g + ((q & 1) << 1) + (q >> 1), u = kp & 7, vc = kp >> 3, v = vc % FW, c = vc / FW;
    sA[idx] = (row < a.M && u < a.nU && v < a.nV && c < NC) ? a.A[(size_t)row * a.lda + u + a.nU * (v + a.nV * c)] : 0.f;
  }
  for (int i = t; i < (PATCH + 8) / 4; i += 256) reinterpret_cast<f32x4 *>(sP)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int PI = (int)a.divPI.d, PIJ = (int)a.divPIJ.d;   // output rows per column, output pixels per sample
  const int tps = PIJ / BP;                                // tiles per sample (host: PIJ % BP == 0)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, a.xBytes, 0x00020000);
  // XCD-contiguous tile ranges (as conv_stem_kernel): XCD x walks tiles [x * per, (x + 1) * per)
  const int per = (ntiles + 7) >> 3, tbase = (blockIdx.x & 7) * per, tstep = gridDim.x >> 3;
  const int tend = min(per, ntiles - tbase);

  // ---- patch staging map (fixed for the whole kernel): thread -> (channel, patch column, row pair) ----
  unsigned voP[NLD];
  int ldP[NLD], scP[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int idx = t + 256 * j;
    const int c = idx / (NSC * PR), rem = idx - c * (NSC * PR), sc = rem / PR, pr = rem - sc * PR;
    const int row = 2 * pr - 4;                            // source row of the pair's first element
    const bool ok = idx < NC * NSC * PR && (unsigned)row < (unsigned)a.LimH;   // (LimH even: a pair is in or out as a whole)
    voP[j] = ok ? (unsigned)(((c * a.LimW + sc) * a.LimH + row) * 4) : 0xFFFFFFFFu;   // + sample and first source column below
    scP[j] = ok ? sc : -(1 << 20);
    ldP[j] = idx < NC * NSC * PR ? (c * NSC + sc) * CS + 2 * pr : PATCH;
  }
  // ---- epilogue constants of the rows this lane stores after the in-quad transpose (the same for every tile) ----
  const int iq = l31 & 3;
  float rmul[2][4], radd[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int rowc = min(32 * i + 8 * g4 + 4 * half + iq, a.M - 1);
      const float m_ = a.scale ? a.scale[rowc] : 1.f;
      float t_ = a.scale ? a.shift[rowc] : 0.f;
      if (a.bias) t_ += a.bias[rowc] * m_;
      rmul[i][g4] = m_;
      radd[i][g4] = t_;
    }

  f32x2 ld[NLD];
  auto issue_loads = [&](int tile) {
    const int n = tile / tps, q0 = (tile - n * tps) * BP, j0 = q0 / PI;
    const int c0 = S * j0 + a.gw0;                         // source column of patch column 0 (gw0 = - left padding)
    const unsigned base = (unsigned)((n * a.xSampleStride + c0 * a.LimH) * 4);
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const unsigned off = (unsigned)(c0 + scP[j]) < (unsigned)a.LimW ? voP[j] + base : 0xFFFFFFFFu;
      asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(ld[j]) : "v"(off), "s"(xrsrc) : "memory");
    }
  };
  auto park_patch = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) *reinterpret_cast<f32x2 *>(sP + ldP[j]) = ld[j];
  };
  // (an operand must not be listed twice: the tail of the list is its own statement for the wider tile)
#define XM_S3_WAIT(N)                                                                                         \
  {                                                                                                           \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                    \
                 : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(ld[4]), "+v"(ld[5]), "+v"(ld[6]), "+v"(ld[7]), \
                   "+v"(ld[8]), "+v"(ld[9]), "+v"(ld[10]), "+v"(ld[11]), "+v"(ld[12]), "+v"(ld[13]), "+v"(ld[14]) \
                 :                                                                                            \
                 : "memory");                                                                                 \
    if (NLD > 15)                                                                                             \
      asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(ld[NLD - 3]), "+v"(ld[NLD - 2]), "+v"(ld[NLD - 1]) : : "memory"); \
  }
  static_assert(NLD == 15 || NLD == 18, "XM_S3_WAIT lists the load registers");

  int q = blockIdx.x >> 3;
  __syncthreads();                                         // filter image and zero fill
  if (q < tend) {
    issue_loads(tbase + q);
    XM_S3_WAIT(0)
    park_patch();
  }
  __syncthreads();
  const float *pA = sA + (l31 * 4 + 2 * half);             // + 32 rows per row tile, + 256 floats per group
  for (; q < tend; q += tstep) {
    const int tile = tbase + q;
    const bool more = q + tstep < tend;
    if (more) issue_loads(tile + tstep);
    // this lane's pixel
    const int n = tile / tps, q0 = (tile - n * tps) * BP, j0 = q0 / PI;
    int qq[TN];
    const float *pB[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      qq[j] = q0 + 32 * (wave * TN + j) + l31;
      const int jj = qq[j] / PI, ii = qq[j] - jj * PI;
      pB[j] = sP + (S * (jj - j0)) * CS + S * ii + (4 + a.gh0) + half;   // gh0 = - top padding; patch row 0 = source row -4
    }
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      // group g: k' = 4 g ... 4 g + 3 = filter rows u0 ... u0 + 3 of filter column vc = g / 2, u0 = 4 (g & 1)
      const int vc = g >> 1, u0 = 4 * (g & 1), v = vc % FW, c = vc / FW;
      f32x2 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x2 *>(pA + g * 256 + i * 128);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float bf[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = pB[j][(c * NSC + v) * CS + u0 + 2 * e];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    // ---- epilogue: y = act(acc * scale + (bias * scale + shift)), 16-byte stores through the in-quad transpose ----
#pragma unroll
    for (int j = 0; j < TN; ++j) {
    float *yb = a.Y + (size_t)n * a.oSampleStride + (qq[j] - iq);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int row = 32 * i + 8 * g4 + 4 * half + iq;
        float v4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v4[k] = acc[i][j][4 * g4 + k];
        quad_transpose4(v4, iq);
        f32x4 o = {v4[0] * rmul[i][g4] + radd[i][g4], v4[1] * rmul[i][g4] + radd[i][g4],
                   v4[2] * rmul[i][g4] + radd[i][g4], v4[3] * rmul[i][g4] + radd[i][g4]};
        if (a.relu) {
          o.x = fmaxf(o.x, 0.f);
          o.y = fmaxf(o.y, 0.f);
          o.z = fmaxf(o.z, 0.f);
          o.w = fmaxf(o.w, 0.f);
        }
        xm_st16<true>(yb + (size_t)row * a.oChanStride, o);   // (host: exactly 64 filters, so every lane issues its 8 TN stores)
      }
    }
    __syncthreads();                                       // every wave has read its taps: the patch may be replaced
    if (more) {
      if (TN == 1) {
        XM_S3_WAIT(8);                                     // the next patch is in; this tile's 8 TN stores may still be in flight
      } else {
        XM_S3_WAIT(16);
      }
      park_patch();
    }
    __syncthreads();
  }
#undef XM_S3_WAIT
}


// ---- single-channel stem (the student's conv1: 7 x 7 taps, stride 2, 1 -> 96 channels over 512 x W spectrograms) ----
// K = 49 and a 462 MB output at 32 spectrograms: the layer is bounded by its stores (0.09 ms at the write rate this
// store pattern reaches; tools/store_mfma_probe.hip: 0.107 ms with the MFMAs next to them), the generic kernel needs
// 0.30 ms -- every block pays a gather prologue (49 taps x 128 pixels of dword loads), 4 K-stages of 16, its stores, and
// then leaves.  This kernel is PERSISTENT, keeps everything a tile needs on chip and has NO barrier in its tile loop:
//   * the filter bank lives in LDS for the whole kernel in MFMA operand order (k = 8 v + u: filter column v, filter
//     row u, row 7 = 0; one 16-byte read per (column, row tile)), so a tile issues no global loads for A at all;
//   * every WAVE owns 32 consecutive output pixels of the 128-pixel tile (a piece of one output column, or the end of
//     one and the start of the next) and stages the source rows under them itself: 7 source columns x <= 24 16-byte
//     row units per column group into its own LDS patch (3 loads per lane, issued BEFORE the MFMAs of the current tile
//     and written behind them: LDS operations of a wave execute in order, so the reads of the current patch are done).
//     A lane's tap (u, v) is then at base + v * 104 + u, all immediates;
//   * waves therefore drift apart freely: one wave's epilogue and stores run under another wave's MFMAs (with a barrier
//     per tile the four waves of a block -- and, measured, the two blocks of a CU -- stay in phase: 165 us);
//   * the stores of a tile are plain stores; nothing waits on them until the patch loads issued AFTER them are needed,
//     a whole tile later.  No load is issued under a branch and no epilogue constant is loaded inside the loop (a
//     register that may be pending at the loop's back edge costs an s_waitcnt vmcnt(0) = a wait for all stores).
// Tiles are the same 96 x 128 tiles in the same enumeration as conv_gemm_kernel<3, 1, 1, 4, 1> and the epilogue is the
// same arithmetic (bias, 16-byte stores); the summation order per output is k = 8 v + u ascending.  Statistics for the
// following train-mode bnorm: per-lane partial sums over all tiles of the block, ONE reduction per block at the end.
#ifndef XM_STEM_OCC
#define XM_STEM_OCC 2
#endif
constexpr int kStemHP = 520;   // source columns of <= 512 rows (+ 4 rows of padding either side)
constexpr int kStemNV = 7;     // filter columns; 8 filter rows (the 8th has zero weights) per column
constexpr int kStemHW = 104;   // row pitch of a wave's patch per (column group, source column): 24 units of 16 bytes + 8 (104 = 40
                               // mod 64: the 49 taps u + 104 v of a pixel fall into 49 different LDS banks -- conv_stem_wgrad_kernel reads one tap per lane)
constexpr int kStemTP = 36;    // row pitch (floats) of the epilogue's transpose tile: 32 pixels + 4

template <int SY>
__global__ void __launch_bounds__(256, XM_STEM_OCC)
conv_stem_kernel(const ConvGemmArgs a, const int ntiles) {
  constexpr int TM = 3, NV = kStemNV, HW = kStemHW, GRP = NV * HW, WPATCH = 2 * GRP + 4;   // floats (+ a dummy unit)
  __shared__ __attribute__((aligned(16))) float sP[4 * WPATCH];             // [wave][column group][v][row] + dummy
  __shared__ __attribute__((aligned(16))) float sA[NV * TM * 2 * 32 * 4];   // [v][row tile][half][l31][e]
  __shared__ __attribute__((aligned(16))) float sT[4 * 32 * kStemTP];        // [wave][channel row][pixel]: epilogue transpose;
  float *const sred = sT;                                                   // after the loop: 2 * 4 * 96 floats of partial sums
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  for (int i = t; i < 4 * WPATCH / 4; i += 256) reinterpret_cast<f32x4 *>(sP)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the filter bank, in the order the MFMA A operand wants it: lane (l31, half) of row tile i reads
  // A[32 i + l31][k = 8 v + e + 4 half], e = 0 .. 3, with one 16-byte LDS read per (v, i)
  for (int idx = t; idx < NV * TM * 2 * 32 * 4; idx += 256) {
    const int e = idx & 3, l = (idx >> 2) & 31, h = (idx >> 7) & 1, vi = idx >> 8, i = vi % TM, v = vi / TM;
    const int m = 32 * i + l, u = e + 4 * h;
    sA[idx] = (m < a.M && u < a.nU && v < a.nV) ? a.A[(size_t)m * a.lda + u + a.nU * v] : 0.f;
  }

  // XCD-contiguous tile ranges: XCD x (blocks b % 8 == x) walks tiles [x * per, (x + 1) * per).  With the plain order
  // (block b: tiles b, b + grid, ...) every output row is written in 512-byte pieces that alternate between the eight
  // L2s: tools/store_mfma_probe.hip measures 142 us for this tile shape and output layout ([sample][row][pixel]) against
  // 117 us with contiguous ranges (MFMAs alone: 114) -- the stores only hide under the MFMAs when each L2 writes runs
  // of consecutive tiles.  (Neighbouring tiles also share 5 of their 7 source columns through the XCD's L2.)
  const int per = (ntiles + 7) >> 3, tbase = (blockIdx.x & 7) * per, tstep = gridDim.x >> 3;
  const int tend = min(per, ntiles - tbase);
  const int PIJ = (int)a.divPIJ.d, PI = (int)a.divPI.d;
  float *const sW = sP + wave * WPATCH;   // this wave's patch

  // Patch staging of a wave: lane -> (source column lc = lane / 8, row units lk, lk + 8, lk + 16 of the wave's unit
  // list: the units of column group 0 first, then those of group 1).  Every lane ALWAYS issues its three loads and
  // ALWAYS writes three units (a unit outside the image or beyond the list reads the first 16 bytes of X and is written
  // as zeros / to the dummy unit): no branch around a load (see above).
  const int lc = lane >> 3, lk = lane & 7;
  f32x4 ld[3];
  int ldst[3];      // LDS float index of each unit inside the wave's patch
  bool ldz[3];      // write zeros
  // column groups of the wave's 32 pixels in tile `tile`: (sample, column, row) of the first pixel and of the last one
  // (wave-uniform: scalar registers, scalar divisions)
  struct Cols {
    int nF, jF, iF, nL, jL, iL;
  };
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto wave_cols = [&](int tile) {
    Cols c;
    const uint32_t pF = min((uint32_t)tile * 128u + 32u * wv, (uint32_t)a.NP - 1u), pL = min(pF + 31u, (uint32_t)a.NP - 1u);
    c.nF = (int)xm_div(pF, a.divPIJ);
    uint32_t q = pF - (uint32_t)c.nF * PIJ;
    c.jF = (int)xm_div(q, a.divPI);
    c.iF = (int)q - c.jF * PI;
    c.nL = (int)xm_div(pL, a.divPIJ);
    q = pL - (uint32_t)c.nL * PIJ;
    c.jL = (int)xm_div(q, a.divPI);
    c.iL = (int)q - c.jL * PI;
    return c;
  };
  auto issue_loads = [&](const Cols &c) {
    const bool two = c.nF != c.nL || c.jF != c.jL;
    const int hi0 = two ? PI - 1 : c.iL;
    const int lo4[2] = {(SY * c.iF + a.gh0 + 4) >> 2, (a.gh0 + 4) >> 2};
    const int n0 = ((SY * hi0 + a.gh0 + 4 + 7) >> 2) - lo4[0] + 1;
    const int n1 = two ? ((SY * c.iL + a.gh0 + 4 + 7) >> 2) - lo4[1] + 1 : 0;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int q = lk + 8 * it;
      const int g = q >= n0 ? 1 : 0, u = q - (g ? n0 : 0);
      const bool wr = lc < NV && q < n0 + n1;
      const int n = g ? c.nL : c.nF, j = g ? c.jL : c.jF;
      const int cc = a.gsx * j + a.gw0 + lc, r = 4 * (lo4[g] + u) - 4;
      const bool in = wr && lc < a.nV && cc >= 0 && cc < a.LimW && r >= 0 && r < a.LimH;
      ldst[it] = wr ? (g * NV + lc) * HW + 4 * u : 2 * GRP;
      ldz[it] = !in;
      // an asm load: the compiler does not count it, so the wait in front of its use is written by hand (XM_STEM_PATCH_WAIT)
      const float *src = in ? a.X + (size_t)n * a.xSampleStride + (size_t)cc * a.LimH + r : a.X;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[it]) : "v"(src) : "memory");
    }
  };
  // s_waitcnt vmcnt(N) for the three patch loads: N = number of vector memory operations issued behind them
#define XM_STEM_PATCH_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]) : : "memory")
  auto write_patch = [&]() {
#pragma unroll
    for (int it = 0; it < 3; ++it)
      *reinterpret_cast<f32x4 *>(sW + ldst[it]) = ldz[it] ? f32x4{0.f, 0.f, 0.f, 0.f} : ld[it];
  };

  int q = blockIdx.x >> 3;
  __syncthreads();                       // zero fill and filter bank are complete
  const f32x4 *pa = reinterpret_cast<const f32x4 *>(sA) + half * 32 + l31;
  float rbias[TM][4];   // bias of the rows this lane stores after the epilogue's transpose
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      rbias[i][g4] = a.bias ? a.bias[min(32 * i + 8 * g4 + 4 * half + (l31 & 3), a.M - 1)] : 0.f;
  // a.statPart: this lane's share of {sum, sum of squares} of the rows it stores, over ALL tiles of the block (one
  // reduction per block at the end, statPart[block][row]: a per-tile reduction costs a barrier, 24 DPP chains and an
  // LDS round trip per tile -- 45 us of the 210 the kernel took with it)
  float sacc1[TM][4], sacc2[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) sacc1[i][g4] = 0.f, sacc2[i][g4] = 0.f;
#ifdef XM_DEBUG_CYCLES
  unsigned long long dc[4] = {0, 0, 0, 0}, c0, c1;
#define XM_STEM_T(k) c1 = __builtin_readcyclecounter(), dc[k] += c1 - c0, c0 = c1
  c0 = __builtin_readcyclecounter();
#else
#define XM_STEM_T(k)
#endif
  Cols cur = wave_cols(tbase + min(q, max(tend, 1) - 1));
  if (q < tend) {
    issue_loads(cur);
    XM_STEM_PATCH_WAIT(0);
    write_patch();
  }
  for (; q < tend; q += tstep) {
    const int tile = tbase + q;
    const Cols nxt = wave_cols(q + tstep < tend ? tile + tstep : tile);   // (the last tile is staged once more: no branch)
    // this lane's pixel: 32 consecutive pixels span at most two output columns (PI >= 128) -- no division
    int ii = cur.iF + l31, jj = cur.jF, n = cur.nF, grp = 0;
    if (ii >= PI) ii -= PI, jj = cur.jL, n = cur.nL, grp = 1;
    const uint32_t qq = (uint32_t)(ii + PI * jj);                              // pixel offset inside its sample
    const int w0 = grp ? ((a.gh0 + 4) >> 2) : ((SY * cur.iF + a.gh0 + 4) >> 2);   // first row unit of the group's window
    const float *pb = sW + grp * GRP + SY * ii + a.gh0 + 4 - 4 * w0 + 4 * half;
    cur = nxt;
    f32x16 acc[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    XM_STEM_T(0);   // loads issued, lane geometry
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      f32x4 af[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = pa[(v * TM + i) * 64];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float bv = pb[v * HW + e];
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bv, acc[i][0], 0, 0, 0);
      }
    }
    // The next tile's patch loads are issued BEHIND the MFMAs and IN FRONT of this tile's stores: loads and stores share
    // one in-order counter, so waiting for a load also waits for every store issued before it.  In this order the wait
    // in front of write_patch() below is vmcnt(12) (written by hand: XM_STEM_PATCH_WAIT) -- it leaves this tile's 12 stores in flight and only needs the stores
    // of the PREVIOUS tile, which have had a whole iteration to drain (with the loads in front of the MFMAs the wait
    // needed the previous tile's stores after one MFMA phase).  The epilogue covers the load latency.
    __builtin_amdgcn_sched_barrier(0);
    issue_loads(nxt);
    __builtin_amdgcn_sched_barrier(0);
    XM_STEM_T(1);   // MFMA phase + loads issued
    // Epilogue (dense forward output, 16-byte stores, bias [+ statistics]): the arithmetic of conv_gemm_epilogue's
    // vector path.  The 4 x 4 transpose that turns "4 channel rows of one pixel" into "4 pixels of one channel row" goes
    // through a wave-private LDS tile (16 dword writes + 4 16-byte reads per row tile) instead of the DPP quad
    // transpose (16 VALU operations per 4 registers: with K = 49 the epilogue was as long as the MFMA phase).
    {
      const int iq = l31 & 3;
      const bool ok = (uint32_t)tile * 128u + 32u * wave + l31 < (uint32_t)a.NP;   // quads never straddle NP (NP % 4 == 0)
      // pixel quad base: dense output, p - n * PIJ is the offset inside the sample
      float *yq = a.Y + (size_t)n * a.oSampleStride + (qq - iq);
      float *const tw = sT + wave * (32 * kStemTP) + 4 * half * kStemTP + l31;              // + row(r) * TP
      const float *const tr = sT + wave * (32 * kStemTP) + (4 * half + iq) * kStemTP + (l31 & ~3);   // + 8 g4 * TP
      // a wave whose 32 pixels and 96 rows all exist (every wave but those of the last tile) runs without any lane
      // predicate: no exec-mask branch around the LDS reads and the stores, so that they are issued back to back
      auto rows = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f32x4 v[4];
#pragma unroll
          for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2)) * kStemTP] = acc[i][0][r];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) v[g4] = *reinterpret_cast<const f32x4 *>(tr + 8 * g4 * kStemTP);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int row = 32 * i + 8 * g4 + 4 * half + iq;
            const f32x4 o = {v[g4].x * 1.f + rbias[i][g4], v[g4].y * 1.f + rbias[i][g4], v[g4].z * 1.f + rbias[i][g4],
                             v[g4].w * 1.f + rbias[i][g4]};
            if (FULL || ok) {
              sacc1[i][g4] += (o.x + o.y) + (o.z + o.w);
              sacc2[i][g4] += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
              if (FULL || row < a.M) *reinterpret_cast<f32x4 *>(yq + (size_t)row * a.oChanStride) = o;
            }
          }
        }
      };
      const bool full = (uint32_t)tile * 128u + 32u * wave + 32u <= (uint32_t)a.NP && a.M == 32 * TM;   // wave-uniform
      if (full) {
        rows(std::true_type{});
        XM_STEM_PATCH_WAIT(12);   // exactly 12 stores, no branch: they stay in flight
      } else {
        rows(std::false_type{});
        XM_STEM_PATCH_WAIT(0);
      }
    }
    XM_STEM_T(2);   // epilogue + wait for the loads
    __builtin_amdgcn_sched_barrier(0);
    write_patch();                       // (LDS operations of a wave execute in order: behind its reads of the old patch)
    XM_STEM_T(3);   // wait for the loads, patch write
  }
#ifdef XM_DEBUG_CYCLES
  if (a.dbgCycles && t == 0 && blockIdx.x < 4096)
    for (int k = 0; k < 4; ++k) a.dbgCycles[blockIdx.x * 4 + k] = dc[k];
#endif
  if (a.statPart) {
    // 8 lanes of a row class (DPP), the 4 waves of the block through LDS in wave order, one contiguous run of rows per
    // block (blocks without tiles store zeros: conv_stats_reduce_kernel adds all gridDim.x partials)
    const int iq = l31 & 3;
    __syncthreads();                     // every wave is done with its transpose tile (sred lives there)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float u = quad_class_sum8(sacc1[i][g4]), w = quad_class_sum8(sacc2[i][g4]);
        const int rl = i * 32 + 8 * g4 + 4 * half + iq;
        if ((l31 >> 2) == 0) *reinterpret_cast<float2 *>(sred + 2 * (wave * 96 + rl)) = make_float2(u, w);
      }
    __syncthreads();
    if (t < 96 && t < a.M) {
      float u = 0.f, w = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        const float2 z = *reinterpret_cast<const float2 *>(sred + 2 * (wv * 96 + t));
        u += z.x;
        w += z.y;
      }
      *reinterpret_cast<float2 *>(a.statPart + ((size_t)blockIdx.x * a.M + t) * 2) = make_float2(u, w);
    }
  }
}

// ---- filter derivative of the single-channel stem ------------------------------------------------------------------
// dF[m][u, v] = sum over output pixels p of dY[m][p] * X[origin(p) + (u, v)]: 96 x 49 outputs, a reduction over 1.2 M
// pixels (462 MB of dY at 32 spectrograms).  The generic wgrad kernel covers the 96 x 49 output with 128 x 64 of MFMA
// tiles (1.74 x the work) and gathers the im2col operand tap by tap: 0.22 ms.  Here (MFMA rows = filters, columns =
// taps padded to 64, reduction = pixels) every WAVE walks its own 32-pixel segments like conv_stem_kernel:
//   * its source patch (the same wave-private LDS patch, same staging) gives the B operand: lane = tap, a pixel's tap
//     is patch[base(pixel) + u + 104 v];
//   * its dY tile (96 rows x 32 pixels) is loaded with 12 16-byte loads per lane (8 rows x 128 bytes per instruction),
//     parked in a wave-private LDS tile and read back as the A operand with one 16-byte read per (row tile, 4 steps):
//     MFMA step s multiplies pixels s (lanes 0-31) and 16 + s (lanes 32-63) of the segment;
//   * 96 accumulator registers per wave hold its share of dF for the whole kernel; at the end the four waves add theirs
//     in LDS in wave order, the block writes one partial [96][64] and conv_stem_wgrad_reduce_kernel adds the blocks in
//     block order (deterministic, no atomics).
struct StemWgradArgs {
  const float *dY, *X;
  float *part;                       // [grid][96][64]
  int M, R, nU, nV;                  // filters, taps (nU * nV), filter rows / columns
  int PI, PJ, NP;                    // output pixel grid
  FastDiv divPIJ, divPI;
  int gsx, gh0, gw0, LimH, LimW;     // as ConvGemmArgs (forward gather geometry)
  int xSampleStride, dySampleStride, dyChanStride;
  // conv_stem_wgrad_bnp_kernel only: dY is the convolution's OUTPUT x -- the input of the vl_nnbnorm -> vl_nnrelu ->
  // vl_nnpool('max', 3 x 3 / stride 2, no padding) chain behind it -- and the bnorm's DZDX is rebuilt on the fly from the
  // pooled derivative + the routing table, never written to HBM
  const float *dP;                   // pooled DZDY [pHo][pWo][M][N]
  const unsigned char *amax;         // routing table of the pooling (first maximum, code = dh + 3 dw)
  const float *rowc;                 // [M][6]: g/sigma, mu, b, k1 hi, k1 lo, k2   (bnpool_rowconst_kernel)
  unsigned dpBytes, amBytes, dyBytes;
  int pHo, pWo;                      // pooled grid
  int onesCol;                       // >= 0: that B column multiplies ones: partial[.][m][onesCol] = sum of dY[m] (dzdb)
};
constexpr int kStemWgTP = 36;        // row pitch (floats) of a wave's dY tile: 32 pixels + 4
constexpr int kStemWgWave = 2 * kStemNV * kStemHW + 4 + 96 * kStemWgTP;   // floats of LDS per wave: patch + dummy unit + dY tile
constexpr int kStemWgSmem = (4 * kStemWgWave + 4) * 4;                    // bytes per block (+ a zero unit)
constexpr int kStemWgSmemBnp = kStemWgSmem + 96 * 6 * 4;   // + the bnorm's per-channel constants (two blocks per CU still
                                                           // fit: 2 x 80976 B <= 160 KB at any allocation granule up to 1280 B)

template <int SY>
__global__ void __launch_bounds__(256, 2)
conv_stem_wgrad_kernel(const StemWgradArgs a, const int ntiles) {
  constexpr int TM = 3, NV = kStemNV, HW = kStemHW, GRP = NV * HW, WPATCH = 2 * GRP + 4, TP = kStemWgTP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  for (int i = t; i < 4 * kStemWgWave / 4 + 1; i += 256) reinterpret_cast<f32x4 *>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float *const sW = smem + wave * kStemWgWave;        // this wave's source patch
  float *const tW = sW + WPATCH;                      // this wave's dY tile [96][TP]
  const int per = (ntiles + 7) >> 3, tbase = (blockIdx.x & 7) * per, tstep = gridDim.x >> 3;
  const int tend = min(per, ntiles - tbase);
  const int PIJ = (int)a.divPIJ.d, PI = (int)a.divPI.d;

  // ---- staging: source patch exactly as conv_stem_kernel, dY tile: lane -> (row lane / 8 + 8 k, pixel quad lane % 8)
  const int lc = lane >> 3, lk = lane & 7;
  f32x4 ld[3], dl[12];
  int ldst[3];
  bool ldz[3], dpin = false;
  struct Cols {
    int nF, jF, iF, nL, jL, iL;
  };
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto wave_cols = [&](int tile) {
    Cols c;
    const uint32_t pF = min((uint32_t)tile * 128u + 32u * wv, (uint32_t)a.NP - 1u), pL = min(pF + 31u, (uint32_t)a.NP - 1u);
    c.nF = (int)xm_div(pF, a.divPIJ);
    uint32_t q = pF - (uint32_t)c.nF * PIJ;
    c.jF = (int)xm_div(q, a.divPI);
    c.iF = (int)q - c.jF * PI;
    c.nL = (int)xm_div(pL, a.divPIJ);
    q = pL - (uint32_t)c.nL * PIJ;
    c.jL = (int)xm_div(q, a.divPI);
    c.iL = (int)q - c.jL * PI;
    return c;
  };
  auto issue_loads = [&](const Cols &c, int tile) {
    const bool two = c.nF != c.nL || c.jF != c.jL;
    const int hi0 = two ? PI - 1 : c.iL;
    const int lo4[2] = {(SY * c.iF + a.gh0 + 4) >> 2, (a.gh0 + 4) >> 2};
    const int n0 = ((SY * hi0 + a.gh0 + 4 + 7) >> 2) - lo4[0] + 1;
    const int n1 = two ? ((SY * c.iL + a.gh0 + 4 + 7) >> 2) - lo4[1] + 1 : 0;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int q = lk + 8 * it;
      const int g = q >= n0 ? 1 : 0, u = q - (g ? n0 : 0);
      const bool wr = lc < NV && q < n0 + n1;
      const int n = g ? c.nL : c.nF, j = g ? c.jL : c.jF;
      const int cc = a.gsx * j + a.gw0 + lc, r = 4 * (lo4[g] + u) - 4;
      const bool in = wr && lc < a.nV && cc >= 0 && cc < a.LimW && r >= 0 && r < a.LimH;
      ldst[it] = wr ? (g * NV + lc) * HW + 4 * u : 2 * GRP;
      ldz[it] = !in;
      ld[it] = *reinterpret_cast<const f32x4 *>(in ? a.X + (size_t)n * a.xSampleStride + (size_t)cc * a.LimH + r : a.X);
    }
    // dY: pixel quad of this lane (quads never straddle samples: PIJ % 4 == 0), rows lc + 8 k
    const uint32_t pq = (uint32_t)tile * 128u + 32u * wv + 4u * lk;
    int qq = c.iF + PI * c.jF + 4 * lk, n = c.nF;
    if (qq >= PIJ) qq -= PIJ, ++n;
    const float *row0 = a.dY + (size_t)n * a.dySampleStride + qq;
    dpin = pq < (uint32_t)a.NP;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int m = lc + 8 * k;
      // (rows / pixels that do not exist read dY[0] and are zeroed when the tile is WRITTEN: a select here would put
      // the wait for the load in front of the MFMAs)
      dl[k] = *reinterpret_cast<const f32x4 *>(dpin && m < a.M ? row0 + (size_t)m * a.dyChanStride : a.dY);
    }
  };
  auto write_tiles = [&]() {
#pragma unroll
    for (int it = 0; it < 3; ++it)
      *reinterpret_cast<f32x4 *>(sW + ldst[it]) = ldz[it] ? f32x4{0.f, 0.f, 0.f, 0.f} : ld[it];
#pragma unroll
    for (int k = 0; k < 12; ++k)
      *reinterpret_cast<f32x4 *>(tW + (lc + 8 * k) * TP + 4 * lk) = dpin && lc + 8 * k < a.M ? dl[k] : f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // this lane's two taps (column tiles jt = 0, 1): offset inside a patch column group, or invalid
  int tapoff[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int n = l31 + 32 * jt;
    const int v = n < a.R ? n / a.nU : 0, u = n < a.R ? n - v * a.nU : 0;
    tapoff[jt] = u + HW * v;
  }
  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jt][r] = 0.f;

  int q = blockIdx.x >> 3;
  __syncthreads();                       // zero fill complete
  Cols cur = wave_cols(tbase + min(q, max(tend, 1) - 1));
  if (q < tend) {
    issue_loads(cur, tbase + q);
    write_tiles();
  }
  for (; q < tend; q += tstep) {
    const int tile = tbase + q;
    const bool more = q + tstep < tend;
    const Cols nxt = wave_cols(more ? tile + tstep : tile);
    issue_loads(nxt, more ? tile + tstep : tile);    // (the last tile is staged once more: no branch around the loads)
    __builtin_amdgcn_sched_barrier(0);               // the loads stay in front of the MFMAs
    // B operand: pixel 16 half + s of the segment sits in column group 0 until the column ends (w pixels), then in group 1
    const int w = PI - cur.iF - 16 * half;           // steps s < w are in group 0
    const int b0 = (cur.iF + 16 * half) * SY + a.gh0 + 4 - 4 * ((SY * cur.iF + a.gh0 + 4) >> 2);
    const int b1 = GRP + (cur.iF + 16 * half - PI) * SY + a.gh0 + 4 - 4 * ((a.gh0 + 4) >> 2);
    const float *tr = tW + l31 * TP + 16 * half;     // A operand: row 32 i + l31, pixels 16 half + 4 c .. + 3
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 af[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(tr + 32 * i * TP + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = 4 * c + e;
        const float *px = sW + (s < w ? b0 : b1) + SY * s;
        float bv[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) bv[jt] = px[tapoff[jt]];   // (columns >= R accumulate finite garbage that is never stored)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bv[jt], acc[i][jt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);               // ... and the uses of the loads behind them
    write_tiles();                                   // (LDS operations of a wave execute in order)
    cur = nxt;
  }
  // the four waves add their accumulators in LDS in wave order; the block leaves one partial [96][64]
  float *const sD = smem;                            // 96 x 64 floats (the tiles are dead)
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

// dF[r + R m] = sum over the blocks' partials in a fixed order; grid = M blocks of 64 taps x 16 block groups
__global__ void __launch_bounds__(1024)
conv_stem_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ df, int nblk, int M, int R,
                              float *__restrict__ db /* NULL, or dzdb: column R of the partials (StemWgradArgs::onesCol) */) {
  __shared__ float red[16][64];
  const int r = threadIdx.x & 63, g = threadIdx.x >> 6, m = blockIdx.x;
  float v = 0.f;
  for (int b = g; b < nblk; b += 16) v += part[(size_t)b * (96 * 64) + m * 64 + r];
  red[g][r] = v;
  __syncthreads();
  if (g == 0 && (r < R || (db && r == R))) {
    v = red[0][r];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += red[k][r];
    if (r < R) df[r + (size_t)R * m] = v;
    else db[m] = v;
  }
}

// ---- the same filter derivative THROUGH vl_nnpool('max') o vl_nnrelu o vl_nnbnorm ------------------------------------
// xm_nnconv_backward_filter_bnrelupool: dY above is the derivative the bnorm hands to its input -- 462 MB at 32
// spectrograms, written by bnpool_bwd_apply_patch_kernel and read once, here.  This kernel takes the bnorm's INPUT x
// (= the convolution's own output, same layout) in its place and rebuilds the derivative per element:
//     dz = [g/sigma (x - mu) + b > 0] * sum over the <= 2 x 2 covering windows w [argmax(w) == this element] dP(w)
//     dx = g/sigma dz - k1 - k2 (x - mu)                         (k1, k2: per-channel constants, bnpool_rowconst_kernel)
// A lane's pixel quad = two stride cells of the 3 x 3 / stride-2 pooling (quad origins and PI are even); cell (kh, column
// j): the even row 2 kh is covered by window rows kh - 1 (as their row 2) and kh (row 0), the odd row by kh (row 1); the
// column by window columns (j >> 1) - 1 (j even only; column 2 of the window) and j >> 1 (column j & 1).  Per cell and
// window column ONE 8-byte load fetches the derivatives of window rows (kh - 1, kh) and one 2-byte load their routing
// codes; a candidate that does not exist gets the expected code 255 (never recorded): no validity masks.  The first cell
// of a column (kh = 0) loads the pair one row later, so that no load starts in front of the tensor.  Contributions are
// added in the order of bnpool_bwd_apply_patch_kernel (window column outer, window row inner).  The transform runs in
// fp32 with k1 = g/sigma mean(dz) split in two floats (a rounded k1 would shift every element of a channel the same
// way; sum(dx) must stay at rounding-noise level, xm_common.h).
// Schedule: the MFMA loop runs ROW TILE by row tile (3 phases of 32 MFMAs); the 32 rows of the dY tile a phase has read
// are dead afterwards (LDS operations of a wave execute in order), so production runs exactly two phases ahead of
// consumption -- phase p of tile T writes row tile (p + 2) % 3 of tile T + (p > 0) -- with the x quads requested one
// phase before that and the pooled operands inside the phase, two row groups at a time: 32 + 24 registers in flight
// instead of a whole tile's (the tile-ahead variant of conv_stem_wgrad_kernel needed 48 + 36 and spilled).
// ONES: 0 = no bias derivative; 1 / 2 = the ones column sits in column tile 0 / 1 (only that tile's B operand carries the select)
template <int SY, int ONES>
__global__ void __launch_bounds__(256, 2)
conv_stem_wgrad_bnp_kernel(const StemWgradArgs a, const int ntiles) {
  constexpr int TM = 3, NV = kStemNV, HW = kStemHW, GRP = NV * HW, WPATCH = 2 * GRP + 4, TP = kStemWgTP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  for (int i = t; i < 4 * kStemWgWave / 4 + 1; i += 256) reinterpret_cast<f32x4 *>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float *const sW = smem + wave * kStemWgWave;        // this wave's source patch
  float *const tW = sW + WPATCH;                      // this wave's dY tile [96][TP]
  float *const sRC = smem + 4 * kStemWgWave + 4;      // the bnorm's per-channel constants [96][6]
  for (int i = t; i < 96 * 6; i += 256) sRC[i] = i < 6 * a.M ? a.rowc[i] : 0.f;
  const int per = (ntiles + 7) >> 3, tbase = (blockIdx.x & 7) * per, tstep = gridDim.x >> 3;
  const int tend = min(per, ntiles - tbase);
  const int PIJ = (int)a.divPIJ.d, PI = (int)a.divPI.d;
  const int pHW = a.pHo * a.pWo;
  const __amdgpu_buffer_rsrc_t dprsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dP, 0, a.dpBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t amrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.amax, 0, a.amBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dY, 0, a.dyBytes, 0x00020000);
  // All streaming loads are buffer loads with a per-lane byte offset that is fixed for a tile + a SCALAR offset for the
  // channel row (rows 32 rt + 8 kk + lc: the lane's lc sits in the per-lane part): 64-bit per-row addresses kept live
  // across the tile loop were what made the first version of this kernel spill.  Rows / quads that do not exist read
  // in-range garbage or, past the end of the tensor, zeros (range check) and are zeroed when the tile is WRITTEN.

  const int lc = lane >> 3, lk = lane & 7;
  struct Cols {
    int nF, jF, iF, nL, jL, iL;
  };
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto wave_cols = [&](int tile) {
    Cols c;
    const uint32_t pF = min((uint32_t)tile * 128u + 32u * wv, (uint32_t)a.NP - 1u), pL = min(pF + 31u, (uint32_t)a.NP - 1u);
    c.nF = (int)xm_div(pF, a.divPIJ);
    uint32_t q = pF - (uint32_t)c.nF * PIJ;
    c.jF = (int)xm_div(q, a.divPI);
    c.iF = (int)q - c.jF * PI;
    c.nL = (int)xm_div(pL, a.divPIJ);
    q = pL - (uint32_t)c.nL * PIJ;
    c.jL = (int)xm_div(q, a.divPI);
    c.iL = (int)q - c.jL * PI;
    return c;
  };
  // ---- source patch: exactly conv_stem_wgrad_kernel's --------------------------------------------------------------
  f32x4 ld[3];
  int ldst[3];
  bool ldz[3];
  auto issue_patch = [&](const Cols &c) {
    const bool two = c.nF != c.nL || c.jF != c.jL;
    const int hi0 = two ? PI - 1 : c.iL;
    const int lo4[2] = {(SY * c.iF + a.gh0 + 4) >> 2, (a.gh0 + 4) >> 2};
    const int n0 = ((SY * hi0 + a.gh0 + 4 + 7) >> 2) - lo4[0] + 1;
    const int n1 = two ? ((SY * c.iL + a.gh0 + 4 + 7) >> 2) - lo4[1] + 1 : 0;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int q = lk + 8 * it;
      const int g = q >= n0 ? 1 : 0, u = q - (g ? n0 : 0);
      const bool wr = lc < NV && q < n0 + n1;
      const int n = g ? c.nL : c.nF, j = g ? c.jL : c.jF;
      const int cc = a.gsx * j + a.gw0 + lc, r = 4 * (lo4[g] + u) - 4;
      const bool in = wr && lc < a.nV && cc >= 0 && cc < a.LimW && r >= 0 && r < a.LimH;
      ldst[it] = wr ? (g * NV + lc) * HW + 4 * u : 2 * GRP;
      ldz[it] = !in;
      ld[it] = *reinterpret_cast<const f32x4 *>(in ? a.X + (size_t)n * a.xSampleStride + (size_t)cc * a.LimH + r : a.X);
    }
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int it = 0; it < 3; ++it)
      *reinterpret_cast<f32x4 *>(sW + ldst[it]) = ldz[it] ? f32x4{0.f, 0.f, 0.f, 0.f} : ld[it];
  };

  // ---- production of one row tile (rows 32 rt + lc + 8 kk, kk = 0 .. 3; pixel quad lk) of a tile ---------------------
  // Measured on the way (whole chain at 32 spectrograms, this mapping: 373 ... 385 us against 482 for the two separate
  // passes; the sums stage is ~60 us of both): pooled loads removed -110 us, decode arithmetic removed -30 us -- the
  // kernel is bound by the NUMBER of vector-memory instructions (~16 clocks of the CU's address path each: 96 pooled + 12
  // + 3 per wave and tile), not by bytes or latency: touching the pooled lines one phase early (3 more instructions per
  // phase, counted by the compiler or not) cost +80 us, a one-stride-cell-per-lane mapping (half the cache lines per
  // pooled instruction, but 24 instead of 12 x loads) +50 us, two decode variants behind a wave-uniform branch +50 us,
  // a streaming cache policy on the x loads nothing.
  struct Prod {            // per lane and tile
    unsigned xoff;         // byte offset of x at (row lc, the lane's pixel quad)
    bool in;               // the quad exists (quads never straddle NP: NP % 4 == 0)
  };
  auto prod_of = [&](const Cols &c, int tile) {
    Prod p;
    const uint32_t pq = (uint32_t)tile * 128u + 32u * wv + 4u * lk;
    int qq = c.iF + PI * c.jF + 4 * lk, n = c.nF;
    if (qq >= PIJ) qq -= PIJ, ++n;
    p.xoff = (unsigned)(n * a.dySampleStride + qq + lc * a.dyChanStride) * 4u;
    p.in = pq < (uint32_t)a.NP;
    return p;
  };
  // Pooled operands: per lane and window-column slot s (s = 1: window column jA >> 1, s = 0: the one before it) ONE
  // 12-byte load fetches the derivatives of three consecutive elements of the pooled plane and ONE 4-byte load their
  // routing codes -- window rows (khA - 1, khA, khA + 1) for the two stride cells A, B of an ordinary quad.  Consecutive
  // elements wrap into the next window column, which is exactly what a quad that straddles two image columns needs (its
  // second cell is the first cell of the next column).  Where the triple starts is the only case analysis: at window
  // row 0 for the first cell of a column, pHo - 3 when cell B is the last cell of its column (or A is and no further
  // window column exists), pHo - 2 for a straddling quad -- always such that every load that carries an EXISTING window
  // lies inside the tensor as a whole: a load that straddles either end of the buffer is dropped entirely (seen twice:
  // a 2-byte load whose second byte is past the end returned zero for both; a 12-byte load that starts 8 bytes in
  // front of the tensor returned zero for its third, in-range element -- the range check looks at the per-lane offset).  Which pixel takes
  // what from which element follows from ONE coverage formula, evaluated per tile into expected codes
  // (dh + 3 dw, or 255 = "this window does not cover this pixel / does not exist"): no per-case decode.  (Testing only
  // the six (pixel, window) pairs per slot that can match in an ordinary quad, behind a wave-uniform branch for the six
  // of eight segments of a column whose lanes are all ordinary, measured 447 against 378 us for the chain: like the
  // first version's two decode variants, a branch in this loop costs more than the arithmetic it saves.)
  // The code dword is loaded one byte early (its unused byte in front) except for triples that start a column.
  int pbase[2];          // per slot: element index of the triple's first element, channel row lc
  int pcs[2];            // per slot: 8 * (bytes the code dword starts in front of the triple)
  unsigned pexp[2][4];   // per slot and pixel: expected codes of the triple's three windows, one byte each
  bool pin = false;      // Prod::in of the tile these belong to
  bool pfull = false;    // wave-uniform: all 32 pixels and all 96 rows of the tile exist (no zeroing at the write)
  auto pooled_geometry = [&](const Cols &c, const Prod &pr, int tile) {
    const int q0 = c.iF + PI * c.jF + 4 * lk;
    pin = pr.in;
    pfull = (uint32_t)tile * 128u + 32u * wv + 32u <= (uint32_t)a.NP && a.M == 32 * TM;
    int ci[2], cj[2], n = c.nF;                     // cells A, B: first row, column (quads never straddle samples)
    {
      int qq = q0;
      if (qq >= PIJ) qq -= PIJ, ++n;
#pragma unroll
      for (int cell = 0; cell < 2; ++cell) {
        const int qc = qq + 2 * cell;
        cj[cell] = (int)xm_div((uint32_t)qc, a.divPI);
        ci[cell] = qc - PI * cj[cell];
      }
    }
    const int khA = ci[0] >> 1, khB = ci[1] >> 1;
    const bool straddle = cj[1] != cj[0];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      int wo = (cj[0] >> 1) - (sl ? 0 : 1);
      int hb;
      if (khA == 0) hb = 0;
      else if (!straddle) hb = khB >= a.pHo ? a.pHo - 3 : khA - 1;
      else if (wo < 0) hb = 0, wo += 1;      // (only cell B's window column exists: the triple starts AT it -- a load
                                             // that starts in front of the tensor is dropped as a whole, see below)
      else hb = wo + 1 < a.pWo ? a.pHo - 2 : a.pHo - 3;
      pbase[sl] = hb + a.pHo * wo + (n * a.M + lc) * pHW;
      pcs[sl] = hb == 0 ? 0 : 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pi = ci[e >> 1] + (e & 1), pj = cj[e >> 1];
        unsigned packed = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int w = hb + t;
          const int ho = w >= a.pHo ? w - a.pHo : w, wot = wo + (w >= a.pHo ? 1 : 0);
          const int dh = pi - 2 * ho, dw = pj - 2 * wot;
          const bool ok = (unsigned)dh <= 2u && (unsigned)dw <= 2u && wot >= 0 && wot < a.pWo;
          packed |= (ok ? (unsigned)(dh + 3 * dw) : 255u) << (8 * t);
        }
        pexp[sl][e] = packed | 0xFF000000u;
      }
    }
  };
  auto x_issue = [&](f32x4 (&xq)[4], const Prod &pr, int rt) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)pr.xoff, (32 * rt + 8 * kk) * a.dyChanStride * 4, 0);
      xq[kk] = __builtin_bit_cast(f32x4, v);
    }
  };
  typedef float f32x3 __attribute__((ext_vector_type(3)));
  f32x3 pd[2][2];        // [row group of the half][slot]: derivatives of the triple
  unsigned pa[2][2];     // their routing codes
  auto pooled_issue = [&](int rt, int h) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int mrow = (32 * rt + 8 * (2 * h + kk)) * pHW;     // scalar part of the channel row
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        typedef unsigned u3 __attribute__((ext_vector_type(3)));
        const u3 v = __builtin_amdgcn_raw_buffer_load_b96(dprsrc, pbase[sl] * 4, mrow * 4, 0);
        pd[kk][sl] = __builtin_bit_cast(f32x3, v);
        pa[kk][sl] = __builtin_amdgcn_raw_buffer_load_b32(amrsrc, pbase[sl] - (pcs[sl] >> 3), mrow, 0);
      }
    }
  };
  auto decode_write = [&](const f32x4 (&xq)[4], int rt, int h) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int row = 32 * rt + lc + 8 * (2 * h + kk);
      const int m = min(row, a.M - 1);
      const float2 rca = *reinterpret_cast<const float2 *>(sRC + 6 * m);        // g/sigma, mu
      const float2 rcb = *reinterpret_cast<const float2 *>(sRC + 6 * m + 2);    // b, k1 hi
      const float2 rcc = *reinterpret_cast<const float2 *>(sRC + 6 * m + 4);    // k1 lo, k2
      float dz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {     // (window column outer, window row inner: bnpool_bwd_apply_patch_kernel's order)
        const unsigned cd = pa[kk][sl] >> pcs[sl];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned x_ = cd ^ pexp[sl][e];      // byte t == 0  <=>  window t routes to this pixel
          dz[e] += (x_ & 0x000000FFu) == 0u ? pd[kk][sl].x : 0.f;
          dz[e] += (x_ & 0x0000FF00u) == 0u ? pd[kk][sl].y : 0.f;
          dz[e] += (x_ & 0x00FF0000u) == 0u ? pd[kk][sl].z : 0.f;
        }
      }
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t_ = xq[2 * h + kk][e] - rca.y;
        const float d_ = (rca.x * t_ + rcb.x > 0.f) ? dz[e] : 0.f;
        o[e] = (rca.x * d_ - rcb.y) - (rcc.y * t_ + rcc.x);
      }
      if (!pfull) o = (pin && row < a.M) ? o : f32x4{0.f, 0.f, 0.f, 0.f};      // (wave-uniform: the last tile / fewer than 96 rows)
      *reinterpret_cast<f32x4 *>(tW + row * TP + 4 * lk) = o;
    }
  };

  // this lane's two taps (column tiles jt = 0, 1): offset inside a patch column group; the column that multiplies ones
  int tapoff[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int n = l31 + 32 * jt;
    const int v = n < a.R ? n / a.nU : 0, u = n < a.R ? n - v * a.nU : 0;
    tapoff[jt] = u + HW * v;
  }
  const bool onesLane = ONES && a.onesCol == l31 + 32 * (ONES - 1);
  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jt][r] = 0.f;

  int q = blockIdx.x >> 3;
  __syncthreads();                       // zero fill and constants complete
  f32x4 xc[4], xn[4];                    // x quads of the row tile produced in this phase / requested for the next one
  Cols cur = wave_cols(tbase + min(q, max(tend, 1) - 1));
  if (q < tend) {
    // prologue: row tiles 0 and 1 of the first tile, its patch, and the x quads of its row tile 2 (produced in phase 0)
    const Prod pr = prod_of(cur, tbase + q);
    pooled_geometry(cur, pr, tbase + q);
    issue_patch(cur);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      x_issue(xc, pr, rt);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pooled_issue(rt, h);
        decode_write(xc, rt, h);
      }
    }
    x_issue(xc, pr, 2);
    write_patch();
  }
  for (; q < tend; q += tstep) {
    const int tile = tbase + q;
    const bool more = q + tstep < tend;
    const int ntile = more ? tile + tstep : tile;    // (the last tile is staged once more: no branch around the loads)
    const Cols nxt = wave_cols(ntile);
    const Prod prn = prod_of(nxt, ntile);
    // B operand: pixel 16 half + s of the segment sits in column group 0 until the column ends (w pixels), then in group 1
    const int w = PI - cur.iF - 16 * half;           // steps s < w are in group 0
    const int b0 = (cur.iF + 16 * half) * SY + a.gh0 + 4 - 4 * ((SY * cur.iF + a.gh0 + 4) >> 2);
    const int b1 = GRP + (cur.iF + 16 * half - PI) * SY + a.gh0 + 4 - 4 * ((a.gh0 + 4) >> 2);
    const float *tr = tW + l31 * TP + 16 * half;     // A operand: row 32 i + l31, pixels 16 half + 4 c .. + 3
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // phase i multiplies row tile i and produces row tile (i + 2) % 3 -- of this tile in phase 0 (geometry still this
      // tile's), of the next one afterwards
      const int prt = (i + 2) % 3;
      if (i == 1) pooled_geometry(nxt, prn, ntile);
      if (i == 2) issue_patch(nxt);
      x_issue(xn, prn, i);                            // the row tile the NEXT phase produces: (next tile, row tile i)
      pooled_issue(prt, 0);
      __builtin_amdgcn_sched_barrier(0);             // the loads stay in front of the MFMAs
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 af = *reinterpret_cast<const f32x4 *>(tr + 32 * i * TP + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int s = 4 * c + e;
          const float *px = sW + (s < w ? b0 : b1) + SY * s;
          float bv[2];
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) bv[jt] = px[tapoff[jt]];   // (columns >= R accumulate finite garbage that is never stored)
          if (ONES) bv[ONES ? ONES - 1 : 0] = onesLane ? 1.f : bv[ONES ? ONES - 1 : 0];
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bv[jt], acc[i][jt], 0, 0, 0);
        }
        if (c == 1) {
          __builtin_amdgcn_sched_barrier(0);
          decode_write(xc, prt, 0);
          pooled_issue(prt, 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      decode_write(xc, prt, 1);
      if (i == 2) write_patch();                     // (LDS operations of a wave execute in order: behind the last B reads)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) xc[kk] = xn[kk];
    }
    cur = nxt;
  }
  // the four waves add their accumulators in LDS in wave order; the block leaves one partial [96][64]
  float *const sD = smem;                            // 96 x 64 floats (the tiles are dead)
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

// combine split-K slabs in split order and apply the fused epilogue.  VEC (vecStore destinations, slab pitch and
// pixel count multiples of 4): a thread owns 4 consecutive pixels of one row -- 16-byte slab loads, one 16-byte
// store; otherwise one (m, p) per thread.  divCols = NP / 4 resp. NP.
template <bool VEC>
__global__ void __launch_bounds__(256)
conv_splitk_epilogue_kernel(const ConvGemmArgs a, int splits, FastDiv divCols) {
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
  const uint32_t cols = divCols.d;
  if (idx >= (uint32_t)a.M * cols) return;
  const int m = (int)xm_div(idx, divCols);
  const int pr = (int)(idx - (uint32_t)m * cols) * (VEC ? 4 : 1);   // relative to hyP0 (0 for a plain split-K launch)
  const int p = pr + a.hyP0;
  uint32_t n = xm_div((uint32_t)p, a.divPIJ);
  uint32_t q = (uint32_t)p - n * a.divPIJ.d;
  uint32_t jj = xm_div(q, a.divPI);
  uint32_t ii = q - jj * a.divPI.d;
  uint32_t mc = xm_div((uint32_t)m, a.divMU);
  int off = (a.oh0 + (int)ii * a.osy) + a.OH * (a.ow0 + (int)jj * a.osx) + (int)n * a.oSampleStride +
            (int)mc * a.oChanStride + (m - (int)mc * (int)a.divMU.d) * a.oUStride;
  if (VEC) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z) v += *reinterpret_cast<const f32x4 *>(a.slab + ((size_t)z * a.M + m) * a.NPs + pr);
    if (a.bias) v += a.bias[m];
    if (a.scale) v = v * a.scale[m] + a.shift[m];
    if (a.gate) v *= a.gate[m + (int)n * a.gateStride];
    if (a.resid) v += *reinterpret_cast<const f32x4 *>(a.resid + off);
    if (a.relu) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<f32x4 *>(a.Y + off) = v;
  } else {
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += a.slab[((size_t)z * a.M + m) * a.NPs + pr];
    if (a.bias) v += a.bias[m];
    if (a.scale) v = v * a.scale[m] + a.shift[m];
    if (a.gate) v *= a.gate[m + (int)n * a.gateStride];
    if (a.resid) v += a.resid[off];
    if (a.relu) v = fmaxf(v, 0.f);
    a.Y[off] = v;
  }
}

// The same combine for the remainder of a HYBRID launch that carries batch statistics (a.statPart; vector stores, no
// ReLU / residual): grid (ceil(cols / 256), M) -- a block owns 256 pixel quads of ONE row, so that it can leave the
// row's {sum, sum of squares} over its quads as one more partial entry: statPart[statBase + blockIdx.x][row].
__global__ void __launch_bounds__(256)
conv_splitk_epilogue_stats_kernel(const ConvGemmArgs a, int splits, int cols, int statBase) {
  __shared__ float red[2][4];
  const int m = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  float s1 = 0.f, s2 = 0.f;
  if (c < cols) {
    const int pr = c * 4, p = pr + a.hyP0;
    const uint32_t n = xm_div((uint32_t)p, a.divPIJ);
    const uint32_t q = (uint32_t)p - n * a.divPIJ.d;
    const uint32_t jj = xm_div(q, a.divPI);
    const uint32_t ii = q - jj * a.divPI.d;
    const uint32_t mc = xm_div((uint32_t)m, a.divMU);
    const int off = (a.oh0 + (int)ii * a.osy) + a.OH * (a.ow0 + (int)jj * a.osx) + (int)n * a.oSampleStride +
                    (int)mc * a.oChanStride + (m - (int)mc * (int)a.divMU.d) * a.oUStride;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z) v += *reinterpret_cast<const f32x4 *>(a.slab + ((size_t)z * a.M + m) * a.NPs + pr);
    // same operand order as conv_gemm_epilogue: acc * scale + (bias * scale + shift)
    float rmul = 1.f, radd = 0.f;
    if (a.scale) rmul = a.scale[m], radd = a.shift[m];
    if (a.bias) radd += a.bias[m] * rmul;
    v = v * rmul + radd;
    if (a.gate) v *= a.gate[m + (int)n * a.gateStride];
    s1 = (v.x + v.y) + (v.z + v.w);
    s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    *reinterpret_cast<f32x4 *>(a.Y + off) = v;
  }
  s1 = xm_wave_sum(s1);
  s2 = xm_wave_sum(s2);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[0][wv] = s1, red[1][wv] = s2;
  __syncthreads();
  if (threadIdx.x == 0)
    *reinterpret_cast<float2 *>(a.statPart + ((size_t)(statBase + blockIdx.x) * a.M + m) * 2) =
        make_float2((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
}

// ------------------------------------------------------------------------------------------
// wgrad:  dF[k][r] = sum_p dY[k][p] * G(p, r)      (p = flat output pixel (ho, wo, n))
// MFMA rows = output channel k, MFMA cols = tap r (contiguous in dF), reduction = pixels.
// A stage covers 16 consecutive flat pixels; both operands are contiguous in memory along the
// pixel axis, so staging lanes run along pixels (coalesced) and each thread decodes one pixel per
// stage; its channel rows / filter taps are fixed for the whole kernel and live in registers.
// Split over the pixel range (grid.y); partials go to a workspace slab per split and are summed
// by reduce_splits_kernel in a fixed order (deterministic, no atomics).
struct WgradArgs {
  const float *dY;   // [Ho*Wo][K][N]
  const float *X;    // gather source (forward input)
  float *out;        // [splits][M][ldo]
  const int4 *taps;  // Rn entries {byte offset, du, dv, 0}; padded entries du = -2^28
  unsigned xBytes, dyBytes;
  int M, R, Rn, ldo;  // Rn = taps rounded up to BN
  int Ho, Wo, NP;     // output pixel grid, NP = Ho*Wo*N
  FastDiv divHW, divHo;
  int sy, sx, pt, pl_;  // forward stride / pad (gather origin = ho*sy - pt)
  int H, W, xSampleStride;
  int dyChanStride, dySampleStride;  // Ho*Wo, Ho*Wo*Ktotal
  int nbm, nbn, tilesPerSplit, nkt;
  size_t splitStride;
};

// AV = pixels per dY staging load (1, 2 or 4): the dY rows are contiguous along the pixel axis, so when
// Ho*Wo is a multiple of AV (groups never straddle two samples) a thread loads AV consecutive pixels of
// a row with ONE dwordx2/x4 buffer load and parks them with ONE ds_write_b64/b128 -- they are exactly
// AV consecutive k of the LDS image [k/4][row][k%4].  4x fewer VMEM + LDS instructions on the A side.
template <int TM, int TN, int WGM, int WGN, int AV>
__global__ void __launch_bounds__(256, 2)
conv_wgrad_kernel(const WgradArgs a) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  static_assert(WGM * WGN == 4, "4 waves per block");
  static_assert(BM % 16 == 0 && BN % 16 == 0, "rows are staged 16 at a time");
  static_assert(AV == 1 || AV == 2 || AV == 4, "dY staging width");
  // plane pitch = rows*4 + 16 floats: the scalar ds_write_b32 staging stores of a wave (16 pixels
  // x 4 rows) then hit each bank at most twice (free), and ds_read_b128 stays 16-B aligned
  constexpr int PLA = BM * 4 + 16, PLB = BN * 4 + 16;
  constexpr int APG = 16 / AV;             // pixel groups per 16-pixel stage (A side)
  constexpr int ARP = 256 / APG;           // dY rows covered by one pass of the block
  constexpr int NEA = (BM + ARP - 1) / ARP, NEB = BN / 16;  // loads per thread per stage
  typedef float avec __attribute__((ext_vector_type(AV == 1 ? 2 : AV)));  // AV == 1 uses .x only
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

  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, a.xBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dyrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)a.dY, 0, a.dyBytes, 0x00020000);

  // Staging map: lane -> pixel.  Thread t owns pixel slot px = t % 16 of every stage and rows
  // rsub + 16*j: a wave-wide load instruction then reads 4 runs of 16 consecutive pixels
  // (64 contiguous bytes each for stride 1) instead of 64 scattered addresses.
  const int px = t & 15, rsub = t >> 4;
  const int apg = t % APG, arow = t / APG;  // A side: pixel group / first row of this thread
  unsigned arow4[NEA];  // dY byte offset of channel row j (rows >= M clamped: never stored)
#pragma unroll
  for (int j = 0; j < NEA; ++j) {
    int gm = min(bm * BM + arow + ARP * j, a.M - 1);
    arow4[j] = (unsigned)(gm * a.dyChanStride) * 4u;
  }
  int tpo[NEB], tpy[NEB], tpz[NEB];  // the thread's taps: byte offset, du, dv (fixed all kernel)
#pragma unroll
  for (int j = 0; j < NEB; ++j) {
    int4 tp = a.taps[bn * BN + rsub + 16 * j];
    tpo[j] = tp.x;
    tpy[j] = tp.y;
    tpz[j] = tp.z;
  }
  // LDS: [stage][g = px/4][row][px%4]
  float *sAw = sA + ((apg * AV) >> 2) * PLA + arow * 4 + ((apg * AV) & 3);
  float *sBw = sB + (px >> 2) * PLB + rsub * 4 + (px & 3);
  const int half = lane >> 5, l31 = lane & 31;
  const float *sAr = sA + half * PLA + (wm * TM * 32 + l31) * 4;
  const float *sBr = sB + half * PLB + (wn * TN * 32 + l31) * 4;

  // two register sets, loads issued two stages ahead (see conv_gemm_kernel): the small-tile /
  // huge-reduction layers (conv1: 1.2 M pixels into a 96 x 49 filter gradient) are latency-bound
  avec ra0[NEA], ra1[NEA];
  float rb0[NEB], rb1[NEB];

#define XM_WLOAD_TILE(KT, RA, RB)                                              \
  {                                                                            \
    uint32_t p_ = (uint32_t)((KT) * kBK + px);                                 \
    uint32_t n_ = xm_div(p_, a.divHW);                                         \
    uint32_t q_ = p_ - n_ * a.divHW.d;                                         \
    uint32_t wo_ = xm_div(q_, a.divHo);                                        \
    uint32_t ho_ = q_ - wo_ * a.divHo.d;                                       \
    const unsigned pm_ = (int)p_ < a.NP ? 0u : 0xFFFFFFFFu; /* past the last pixel -> 0 */ \
    const int hb_ = (int)ho_ * a.sy - a.pt, wb_ = (int)wo_ * a.sx - a.pl_;     \
    const unsigned xo_ = (unsigned)(hb_ + a.H * wb_ + (int)n_ * a.xSampleStride) * 4u; \
    if (AV == 1) {                                                             \
      const unsigned dyo_ = ((q_ + n_ * (unsigned)a.dySampleStride) * 4u) | pm_; \
      _Pragma("unroll") for (int j = 0; j < NEA; ++j)                          \
        RA[j].x = buf_load(dyrsrc, (arow4[j] + dyo_) | pm_);                   \
    } else {                                                                   \
      /* the thread's pixel group (AV consecutive pixels of one sample: Ho*Wo % AV == 0) */ \
      uint32_t pa_ = (uint32_t)((KT) * kBK + apg * AV);                        \
      uint32_t na_ = xm_div(pa_, a.divHW);                                     \
      uint32_t qa_ = pa_ - na_ * a.divHW.d;                                    \
      const unsigned pma_ = (int)pa_ < a.NP ? 0u : 0xFFFFFFF0u;                \
      const unsigned dya_ = (qa_ + na_ * (unsigned)a.dySampleStride) * 4u;     \
      _Pragma("unroll") for (int j = 0; j < NEA; ++j)                          \
        RA[j] = buf_load_v<AV>(dyrsrc, (arow4[j] + dya_) | pma_);              \
    }                                                                          \
    _Pragma("unroll") for (int j = 0; j < NEB; ++j) {                          \
      bool ok_ = ((unsigned)(hb_ + tpy[j]) < (unsigned)a.H) &                  \
                 ((unsigned)(wb_ + tpz[j]) < (unsigned)a.W);                   \
      unsigned o_ = (xo_ + (unsigned)tpo[j]) | pm_;                            \
      RB[j] = buf_load(xrsrc, ok_ ? o_ : 0xFFFFFFFFu);                         \
    }                                                                          \
  }

#define XM_WSTORE_TILE(BUF, RA, RB)                                            \
  _Pragma("unroll") for (int j = 0; j < NEA; ++j)                              \
    if (BM % ARP == 0 || arow + ARP * j < BM) {                                \
      if (AV == 1)                                                             \
        sAw[(BUF) * kNG * PLA + j * ARP * 4] = RA[j].x;                        \
      else                                                                     \
        *reinterpret_cast<avec *>(sAw + (BUF) * kNG * PLA + j * ARP * 4) = RA[j]; \
    }                                                                          \
  _Pragma("unroll") for (int j = 0; j < NEB; ++j)                              \
    sBw[(BUF) * kNG * PLB + j * 64] = RB[j];

  // same mid-stage store + barrier pipeline as conv_gemm_kernel (register double-buffered fragments)
#define XM_WSTAGE_LD(KT, CUR, LA, LB, SA, SB)                                  \
  XM_WLOAD_TILE((KT) + 2, LA, LB)                                              \
  XM_READ_FRAGS(CUR, 1, af1, bf1)                                              \
  XM_MFMA_CHUNK(af0, bf0)                                                      \
  XM_INTERLEAVE(8)                                                             \
  __builtin_amdgcn_sched_barrier(0);                                           \
  XM_WSTORE_TILE((CUR) ^ 1, SA, SB)                                            \
  __syncthreads();                                                             \
  XM_READ_FRAGS((CUR) ^ 1, 0, af0, bf0)                                        \
  XM_MFMA_CHUNK(af1, bf1)                                                      \
  XM_INTERLEAVE(8)                                                             \
  __builtin_amdgcn_sched_barrier(0);
#define XM_WSTAGE_NL(CUR, SA, SB)                                              \
  XM_READ_FRAGS(CUR, 1, af1, bf1)                                              \
  XM_MFMA_CHUNK(af0, bf0)                                                      \
  __builtin_amdgcn_sched_barrier(0);                                           \
  XM_WSTORE_TILE((CUR) ^ 1, SA, SB)                                            \
  __syncthreads();                                                             \
  XM_READ_FRAGS((CUR) ^ 1, 0, af0, bf0)                                        \
  XM_MFMA_CHUNK(af1, bf1)                                                      \
  __builtin_amdgcn_sched_barrier(0);
#define XM_WSTAGE_LAST(CUR)                                                    \
  XM_READ_FRAGS(CUR, 1, af1, bf1)                                              \
  XM_MFMA_CHUNK(af0, bf0)                                                      \
  XM_MFMA_CHUNK(af1, bf1)

  f32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
  f32x16 acc[TM][TN], accx;
#pragma unroll
  for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kt0 < kt1) {
    XM_WLOAD_TILE(kt0, ra0, rb0)
    if (kt0 + 1 < kt1) {
      XM_WLOAD_TILE(kt0 + 1, ra1, rb1)
    }
    XM_WSTORE_TILE(0, ra0, rb0)
    __syncthreads();
    XM_READ_FRAGS(0, 0, af0, bf0)
    int kt = kt0;
    for (; kt + 3 < kt1; kt += 2) {
      XM_WSTAGE_LD(kt, 0, ra0, rb0, ra1, rb1)
      XM_WSTAGE_LD(kt + 1, 1, ra1, rb1, ra0, rb0)
    }
    const int rem = kt1 - kt;
    if (rem == 3) {
      XM_WSTAGE_LD(kt, 0, ra0, rb0, ra1, rb1)
      XM_WSTAGE_NL(1, ra0, rb0)
      XM_WSTAGE_LAST(0)
    } else if (rem == 2) {
      XM_WSTAGE_NL(0, ra1, rb1)
      XM_WSTAGE_LAST(1)
    } else {
      XM_WSTAGE_LAST(0)
    }
  }
#undef XM_WLOAD_TILE
#undef XM_WSTORE_TILE
#undef XM_WSTAGE_LD
#undef XM_WSTAGE_NL
#undef XM_WSTAGE_LAST

  if (TM * TN == 1) acc[0][0] += accx;
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

#undef XM_READ_FRAGS
#undef XM_MFMA_CHUNK
#undef XM_MFMA_E
#undef XM_INTERLEAVE

// ---- filter derivative of a 3 x 3 / stride 1 / pad 1 layer from an input PATCH (round 4) -----------------------------------
// conv_wgrad_kernel gathers its im2col operand tap by tap: for a 3 x 3 layer every input element is fetched nine times
// (dword loads, each with its padding test and address arithmetic) and parked with nine ds_write_b32.  Here a stage is ONE
// output column of one sample (HH pixels = the reduction indices of the stage):
//   * the dY tile [128 filters][HH pixels]: 8-byte loads of contiguous rows (per-thread offsets fixed for the whole kernel,
//     the column rides in the scalar offset), LDS image [k/4][row][k%4] as in conv_wgrad_kernel (one ds_read_b128 = 4 k);
//   * the input patch under the column: 3 input columns x HH rows of the <= 16 channels the block's 128 (tap, channel)
//     columns touch, each element loaded ONCE (8-byte loads; columns outside the image are out-of-range loads = zeros) into
//     [channel][column][row] with 6 zero rows between columns -- the rows above / below the image -- so that tap (u, v) of
//     pixel k is at lane base + k: ds_read_b32 with immediate offsets, no masks, no VALU in the loop.
// HH pixels = HH / 2 MFMA 32x32x2 steps per tile pair, no padded reduction steps: full 8-k chunks pair k = 8c + e (lanes
// 0-31) with 8c + 4 + e (lanes 32-63) as everywhere else; the tail (HH % 8 = 6: k = 24 ... 29) pairs (24, 28), (25, 29),
// (26, 27).  Double-buffered LDS, one barrier per stage (60 MFMAs per wave at HH = 30), split over the stage range.
struct WgradPatchArgs {
  const float *dY, *X;
  float *out;                       // [splits][M][ldo]
  unsigned xBytes, dyBytes;
  int M, R, ldo, C;                 // filters, 9 * C columns, row pitch of out, input channels
  int W, K;                         // image columns (= output columns), filters of the whole tensor (sample stride of dY)
  int nStages, stagesPerSplit;      // stages = N * W
  int nbm, nbn;
  size_t splitStride;
};

template <int HH>
__global__ void __launch_bounds__(256, 3)
conv_wgrad_patch_kernel(const WgradPatchArgs a) {
  static_assert(HH % 2 == 0 && HH % 8 == 6 && HH <= 30, "rows per column: pairs, a 6-pixel tail, LDS budget");
  constexpr int BM = 128, BN = 128, NCK = HH / 8, G = (HH + 7) / 8 * 2;   // full 8-k chunks, float4 groups per stage
  constexpr int PLA = BM * 4 + 16;                                          // plane pitch of the dY image (floats)
  constexpr int CS = HH + 6, PC = 3 * CS + 4, NCHN = 16, PGUARD = 4;        // patch: column / channel pitch, channels, guard
  constexpr int SA = G * PLA, SP = PGUARD + NCHN * PC + 4, STG = SA + SP;   // floats per stage buffer (the last 4: spare)
  constexpr int HP = HH / 2;                                                // pixel pairs per column
  constexpr int NLA = (BM * HP + 255) / 256, NLB = (NCHN * 3 * HP + 255) / 256;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float smem[2 * STG];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  const int wm = wave & 1, wn = wave >> 1;
  const int tile = xcd_remap(blockIdx.x, a.nbm * a.nbn);
  const int bm = tile % a.nbm, bn = tile / a.nbm;
  const int s0 = blockIdx.y * a.stagesPerSplit, s1 = min(a.nStages, s0 + a.stagesPerSplit);
  const int ci0 = (bn * BN) / 9;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, a.xBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dyrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dY, 0, a.dyBytes, 0x00020000);

  for (int i = t; i < 2 * STG / 4; i += 256) reinterpret_cast<f32x4 *>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging maps (fixed for the whole kernel)
  unsigned voA[NLA], voB[NLB];
  int ldA[NLA], ldB[NLB], colB[NLB];
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int idx = t + 256 * j;
    const int row = idx / HP, pr = idx - row * HP;
    const bool ok = idx < BM * HP;
    const int gm = min(bm * BM + row, a.M - 1);
    voA[j] = ok ? (unsigned)((gm * a.W * HH + 2 * pr) * 4) : 0xFFFFFFFFu;
    ldA[j] = ok ? (pr >> 1) * PLA + row * 4 + 2 * (pr & 1) : STG - 4;      // (the spare floats at the end of a buffer)
  }
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int idx = t + 256 * j;
    const int ch = idx / (3 * HP), rem = idx - ch * (3 * HP), col = rem / HP, pr = rem - col * HP;
    const bool ok = idx < NCHN * 3 * HP && ci0 + ch < a.C;
    voB[j] = ok ? (unsigned)((((ci0 + ch) * a.W + col - 1) * HH + 2 * pr) * 4) : 0xFFFFFFFFu;   // + the stage's column below
    colB[j] = ok ? col - 1 : -(1 << 20);
    ldB[j] = idx < NCHN * 3 * HP ? SA + PGUARD + ch * PC + col * CS + 2 * pr : STG - 2;
  }
  // fragment addresses
  const float *sAr = smem + half * PLA + (wm * 64 + l31) * 4;
  int bB[2], bT[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = min(bn * BN + (wn * 2 + j) * 32 + l31, a.R - 1);
    const int ci = r / 9, tap = r - ci * 9, v = tap / 3, u = tap - v * 3;
    bB[j] = SA + PGUARD + (ci - ci0) * PC + v * CS + (u - 1) + 4 * half;   // k = 8 c + e (+ 4 for lanes 32-63)
    bT[j] = bB[j] - 3 * half;                                               // tail pair (26, 27): k = 26 + half
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x2 ra[NLA], rb[NLB];
#define XM_WP_LOAD()   /* the stage (ln, lc), then on to the next column */                          \
  {                                                                                                 \
    const int n_ = ln, c_ = lc;                                                                     \
    if (++lc == a.W) lc = 0, ++ln;                                                                  \
    const unsigned sA_ = (unsigned)(((n_ * a.K) * a.W + c_) * HH * 4);                              \
    const unsigned sB_ = (unsigned)(((n_ * a.C) * a.W + c_) * HH * 4);                              \
    _Pragma("unroll") for (int j = 0; j < NLA; ++j)                                                 \
      ra[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dyrsrc, (int)voA[j], (int)sA_, 0)); \
    _Pragma("unroll") for (int j = 0; j < NLB; ++j) {                                               \
      const bool okc_ = (unsigned)(c_ + colB[j]) < (unsigned)a.W;                                   \
      rb[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, (int)(okc_ ? voB[j] + sB_ : 0xFFFFFFFFu), 0, 0)); \
    }                                                                                               \
  }
#define XM_WP_STORE(BUF)                                                                            \
  _Pragma("unroll") for (int j = 0; j < NLA; ++j)                                                   \
    *reinterpret_cast<f32x2 *>(smem + (BUF) * STG + ldA[j]) = ra[j];                                \
  _Pragma("unroll") for (int j = 0; j < NLB; ++j)                                                   \
    *reinterpret_cast<f32x2 *>(smem + (BUF) * STG + ldB[j]) = rb[j];

  __syncthreads();   // the zero fill
  if (s0 < s1) {
    int ln = s0 / a.W, lc = s0 - ln * a.W;
    XM_WP_LOAD()
    XM_WP_STORE(0)
    __syncthreads();
    if (s0 + 1 < s1) XM_WP_LOAD()
    int cur = 0;
    for (int s = s0; s < s1; ++s) {
      // registers hold stage s + 1 (requested a whole stage ago): park it in the buffer the previous barrier freed, request
      // stage s + 2, multiply stage s -- the LDS stores and the loads sit next to the first MFMAs, nothing is exposed
      if (s + 1 < s1) {
        XM_WP_STORE(cur ^ 1)
      }
      if (s + 2 < s1) XM_WP_LOAD()
      const float *A = sAr + cur * STG;
      const float *P = smem + cur * STG;
#pragma unroll
      for (int c = 0; c <= NCK; ++c) {
        f32x4 af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4 *>(A + 2 * c * PLA + i * 128);
        if (c < NCK) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float bf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = P[bB[j] + 8 * c + e];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j], acc[i][j], 0, 0, 0);
          }
        } else {
          // tail: k = 8 c ... 8 c + 5.  (8c, 8c + 4), (8c + 1, 8c + 5) as in a full chunk, then (8c + 2, 8c + 3): lanes 32-63
          // take their dY value from the chunk's first group as well
          f32x4 as[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) as[i] = *reinterpret_cast<const f32x4 *>(A - half * PLA + 2 * c * PLA + i * 128);
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            float bf[2], av[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = e < 2 ? P[bB[j] + 8 * c + e] : P[bT[j] + 8 * c + 2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = e < 2 ? af[i][e] : (half ? as[i][3] : as[i][2]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bf[j], acc[i][j], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      cur ^= 1;
    }
  }
#undef XM_WP_LOAD
#undef XM_WP_STORE
  float *out = a.out + (size_t)blockIdx.y * a.splitStride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = bn * BN + (wn * 2 + j) * 32 + l31;
    if (r >= a.R) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int m = bm * BM + (wm * 2 + i) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        if (m < a.M) out[(size_t)m * a.ldo + r] = acc[i][j][rr];
      }
  }
}

// ---- filter derivative of a 5 x 5 / stride 2 layer from an input PATCH (round 5: the student's conv2) ----------------------
// The same idea as conv_wgrad_patch_kernel for the layer that carries 44 % of the student's arithmetic: conv_wgrad_kernel
// gathers its im2col operand tap by tap -- every input element of a 5 x 5 / stride 2 layer is fetched 25 / 4 times (dword
// loads with padding test + address arithmetic, parked one by one; PMC: 4.2 x the algorithmic bytes).  Here a stage is a
// SEGMENT of 32 output rows of one output column of one sample (k = the 32 reduction indices of the stage):
//   * the dY tile [128 filters][32 pixels]: 8-byte loads of contiguous rows, LDS image [k/4][row][k%4]; output rows that do
//     not exist are loaded as zeros; a segment of exactly 30 rows (the second half of the student's 62-row columns) runs 15
//     reduction steps with conv_wgrad_patch_kernel's tail pairing, so that layer has no padded reduction step at all;
//   * the input patch under the segment: FW input columns x 68 rows (2 * 32 + FH - 1 = 67, from an even row) of the <= 7
//     channels the block's 128 (tap, channel) columns touch, each element loaded ONCE (8-byte loads; rows / columns outside
//     the image are out-of-range loads = zeros) into [channel][column][row]; tap (u, v) of pixel k sits at lane base + 2 k:
//     ds_read_b32 with immediate offsets, no masks, no VALU in the loop.  Pitches: column 70 = 6 (mod 64), channel 350 = 30
//     (mod 64): bank = u + 6 v + 30 c grows with the column index and 32 consecutive (u, v, c) columns span < 64 of it, so an
//     MFMA tile's B read touches 32 different LDS banks (even pitches: the 8-byte stores of the staging path stay aligned).
// 64 MFMAs per wave and stage behind ONE barrier, double-buffered LDS (2 x 26.7 KB: three blocks per CU), prefetch distance 2,
// split over the stage range; partials combined by reduce_splits_kernel in a fixed order.  Blocks that work on the same stage
// range (the tiles of one split: they share the dY tile / the patch) are placed on ONE XCD, so each L2 fetches a stage once.
struct WgradPatchS2Args {
  const float *dY, *X;
  float *out;                       // [splits][M][ldo]
  unsigned xBytes, dyBytes;
  int M, R, ldo, C;                 // filters, FH * FW * C columns, row pitch of out, input channels
  int H, W, Ho, Wo, K;              // input rows / columns, output rows / columns, filters of the whole tensor
  int pt, pl;                       // top (1 or 2) / left padding
  int nSeg;                         // 32-row segments per output column
  int nStages, stagesPerSplit;      // stages = N * Wo * nSeg
  int nbm, nbn, splits;
  size_t splitStride;
};

template <int FHW, int S>
__global__ void __launch_bounds__(256, 3)
conv_wgrad_patch_s2_kernel(const WgradPatchS2Args a) {
  static_assert(FHW == 5 && S == 2, "pitches / channel count are laid out for 5 x 5 taps, stride 2");
  constexpr int BM = 128, BN = 128, HH = 32, G = HH / 4;                      // pixels per stage, float4 groups
  constexpr int PLA = BM * 4 + 16;                                             // plane pitch of the dY image (floats)
  constexpr int PR = (S * (HH - 1) + FHW + 2) / 2;                             // row pairs of the patch (34: rows -2 ... 65; pt >= 1)
  constexpr int CS = 70, PC = FHW * CS, NCHN = (BN + FHW * FHW - 2) / (FHW * FHW) + 1, PGUARD = 4;
  static_assert(CS >= 2 * PR && CS % 2 == 0 && CS % 64 == FHW + 1 && PC % 64 == 30 && NCHN == 7, "patch pitches");
  constexpr int SA = G * PLA, SP = PGUARD + NCHN * PC + 2, STG = SA + SP;     // floats per stage buffer
  static_assert(STG % 4 == 0, "16-byte aligned stage buffers");
  constexpr int HP = HH / 2;                                                   // pixel pairs per dY row segment
  constexpr int NLA = BM * HP / 256, NLB = (NCHN * FHW * PR + 255) / 256;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float smem[2 * STG];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  const int wm = wave & 1, wn = wave >> 1;
  // block -> (tile, split): groups of 8 splits, one per XCD (hardware places block b on XCD b % 8)
  const int ntile = a.nbm * a.nbn;
  int tile, split;
  {
    const int L = blockIdx.x, grp = L / (8 * ntile), rem = L - grp * 8 * ntile;
    const int nsp = min(8, a.splits - grp * 8);
    split = grp * 8 + rem % nsp;
    tile = rem / nsp;
  }
  const int bm = tile % a.nbm, bn = tile / a.nbm;
  const int s0 = split * a.stagesPerSplit, s1 = min(a.nStages, s0 + a.stagesPerSplit);
  const int ci0 = (bn * BN) / (FHW * FHW);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, a.xBytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dyrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dY, 0, a.dyBytes, 0x00020000);

  for (int i = t; i < 2 * STG / 4; i += 256) reinterpret_cast<f32x4 *>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging maps (fixed for the whole kernel)
  unsigned voA[NLA], voB[NLB];
  int ldB[NLB], colB[NLB], rowB[NLB];
  const int prA = 2 * (t % HP);                                                // first pixel of this thread's dY pairs
  const int ldA0 = ((t % HP) >> 1) * PLA + (t / HP) * 4 + 2 * (t & 1);         // + 16 rows per j
#pragma unroll
  for (int j = 0; j < NLA; ++j) {
    const int row = t / HP + (256 / HP) * j;
    const int gm = min(bm * BM + row, a.M - 1);
    voA[j] = (unsigned)((gm * a.Wo * a.Ho + prA) * 4);
  }
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int idx = t + 256 * j;
    const int ch = idx / (FHW * PR), rem = idx - ch * (FHW * PR), col = rem / PR, pr = rem - col * PR;
    const bool ok = idx < NCHN * FHW * PR && ci0 + ch < a.C;
    voB[j] = (unsigned)((((ci0 + ch) * a.W + col - a.pl) * a.H + 2 * pr - 2) * 4);   // + the stage's column / row below
    colB[j] = ok ? col - a.pl : -(1 << 20);
    rowB[j] = 2 * pr - 2;
    ldB[j] = idx < NCHN * FHW * PR ? SA + PGUARD + ch * PC + col * CS + 2 * pr : STG - 2;
  }
  // fragment addresses
  const float *sAr = smem + half * PLA + (wm * 64 + l31) * 4;
  int bB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = min(bn * BN + (wn * 2 + j) * 32 + l31, a.R - 1);
    const int ci = r / (FHW * FHW), tap = r - ci * (FHW * FHW), v = tap / FHW, u = tap - v * FHW;
    bB[j] = SA + PGUARD + (ci - ci0) * PC + v * CS + u + (2 - a.pt) + S * 4 * half;   // k = 8 c + e (+ 4 for lanes 32-63)
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x2 ra[NLA], rb[NLB];
  int ln, lc, lh, ch = 0;          // load side: sample / column / segment of the next stage to request; compute side: segment
#define XM_WS_LOAD()   /* the stage (ln, lc, lh), then on to the next segment / column / sample */              \
  {                                                                                                 \
    const int n_ = ln, c_ = lc, h_ = lh;                                                            \
    if (++lh == a.nSeg) { lh = 0; if (++lc == a.Wo) lc = 0, ++ln; }                                 \
    const unsigned sA_ = (unsigned)((((n_ * a.K) * a.Wo + c_) * a.Ho + HH * h_) * 4);               \
    const unsigned sB_ = (unsigned)((((n_ * a.C) * a.W + S * c_) * a.H + S * HH * h_) * 4);         \
    const bool okA_ = HH * h_ + prA < a.Ho;                                                         \
    _Pragma("unroll") for (int j = 0; j < NLA; ++j)                                                 \
      ra[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dyrsrc, (int)(okA_ ? voA[j] : 0xFFFFFFFFu), (int)sA_, 0)); \
    _Pragma("unroll") for (int j = 0; j < NLB; ++j) {                                               \
      const bool ok_ = (unsigned)(S * c_ + colB[j]) < (unsigned)a.W && (unsigned)(S * HH * h_ + rowB[j]) < (unsigned)a.H; \
      rb[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, (int)(ok_ ? voB[j] + sB_ : 0xFFFFFFFFu), 0, 0)); \
    }                                                                                               \
  }
#define XM_WS_STORE(BUF)                                                                            \
  _Pragma("unroll") for (int j = 0; j < NLA; ++j)                                                   \
    *reinterpret_cast<f32x2 *>(smem + (BUF) * STG + ldA0 + (256 / HP) * 4 * j) = ra[j];             \
  _Pragma("unroll") for (int j = 0; j < NLB; ++j)                                                   \
    *reinterpret_cast<f32x2 *>(smem + (BUF) * STG + ldB[j]) = rb[j];

  __syncthreads();   // the zero fill
  if (s0 < s1) {
    lh = ch = s0 % a.nSeg;
    lc = (s0 / a.nSeg) % a.Wo;
    ln = s0 / (a.nSeg * a.Wo);
    XM_WS_LOAD()
    XM_WS_STORE(0)
    __syncthreads();
    if (s0 + 1 < s1) XM_WS_LOAD()
    int cur = 0;
    for (int s = s0; s < s1; ++s) {
      // registers hold stage s + 1 (requested a whole stage ago): park it in the buffer the previous barrier freed, request
      // stage s + 2, multiply stage s
      if (s + 1 < s1) {
        XM_WS_STORE(cur ^ 1)
      }
      if (s + 2 < s1) XM_WS_LOAD()
      const float *A = sAr + cur * STG;
      const float *P = smem + cur * STG;
      // a segment of 30 rows (the second half of a 62-row column) multiplies 15 reduction steps instead of 16: its last
      // chunk pairs (24, 28), (25, 29), (26, 27) -- lanes 32-63 take the dY value of k = 27 from the chunk's first group
      const bool tail30 = a.Ho - HH * ch == HH - 2;
      if (++ch == a.nSeg) ch = 0;
#define XM_WS_FULL(c)                                                                               \
  {                                                                                                 \
    f32x4 af[2];                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4 *>(A + 2 * (c) * PLA + i * 128); \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                 \
      float bf[2];                                                                                  \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[j] = P[bB[j] + S * (8 * (c) + e)];           \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                               \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j], acc[i][j], 0, 0, 0);    \
    }                                                                                               \
  }
      XM_WS_FULL(0)
      XM_WS_FULL(1)
      XM_WS_FULL(2)
      {
        // last chunk: steps (24, 28), (25, 29) are common; the third is (26, 30) or -- tail -- (26, 27): only the ADDRESSES of
        // the operands of lanes 32-63 differ (one select per stage); the fourth step (27, 31) exists in a full segment only
        constexpr int c = 3;
        const int adjB = tail30 ? S * 3 * half : 0, adjA = tail30 ? half * (PLA - 1) : 0;
        f32x4 af[2];
        float a2[2], b2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[i] = *reinterpret_cast<const f32x4 *>(A + 2 * c * PLA + i * 128);
          a2[i] = (A + 2 * c * PLA + i * 128 + 2)[-adjA];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) b2[j] = (P + bB[j] + S * (8 * c + 2))[-adjB];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float bf[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[j] = P[bB[j] + S * (8 * c + e)];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[i], b2[j], acc[i][j], 0, 0, 0);
        if (!tail30) {
          float bf[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[j] = P[bB[j] + S * (8 * c + 3)];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][3], bf[j], acc[i][j], 0, 0, 0);
        }
      }
#undef XM_WS_FULL
      __syncthreads();
      cur ^= 1;
    }
  }
#undef XM_WS_LOAD
#undef XM_WS_STORE
  float *out = a.out + (size_t)split * a.splitStride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = bn * BN + (wn * 2 + j) * 32 + l31;
    if (r >= a.R) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int m = bm * BM + (wm * 2 + i) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        if (m < a.M) out[(size_t)m * a.ldo + r] = acc[i][j][rr];
      }
  }
}

// ---- dgrad of a 5 x 5 / stride 2 layer (round 5: the student's conv2, 9 % of the default step) ------------------------------
// dX(c, y, x) = sum_k sum_{u = (y + pt) mod 2, v = (x + pl) mod 2 (step 2)} dY(k, (y + pt - u) / 2, (x + pl - v) / 2) F(u, v, c, k).
// conv_gemm_multi_kernel runs the four stride-parity classes as four masked implicit GEMMs (gathers tap by tap) and each class
// writes every other element of every other column: dword stores at an 8-byte stride, half-written cache lines (PMC: 2.1 x
// the algorithmic bytes written).  Here a wave owns ONE output column x and 32 consecutive row PAIRS (y = 2 m, 2 m + 1):
//   * both row parities in one wave (two pixel tiles x three 32-channel row tiles = 96 accumulator registers): filter rows
//     u = 1, 3 feed the even rows, u = 0, 2, 4 the odd ones -- every wave multiplies 5 filter rows per stage, so the waves
//     of a block stay in step -- and the epilogue stores (y = 2 m, 2 m + 1) of a channel as ONE 8-byte store: 32 lanes x
//     8 bytes = 256 contiguous bytes per instruction and channel, every line written once and whole;
//   * a stage = 8 filters (k) x one filter column v: the filter operand is a 15 KB tile that the preparation kernel has laid
//     out in LDS order [k-group][channel row][4] (a linear copy: 16-byte loads, no arithmetic), the dY patch of the block --
//     2 dY columns x 68 rows x 8 filters, every element loaded ONCE -- goes through registers into [k][column][row]; tap u
//     of pixel m reads patch[m + (1 + pt - u) / 2]: ds_read_b32 with immediate offsets, no masks, no VALU;
//   * 60 MFMAs per wave and stage behind one barrier, double-buffered LDS (2 x 20 KB, three blocks per CU), prefetch
//     distance 2 as in conv_wgrad_patch_s2_kernel.  Block = 2 columns of one column-parity class x 64 row pairs x 96 channels.
struct DgradS2Args {
  const float *dY, *Fd;             // Fd: [bm][class][v slot (3)][K / 8][10 planes][96 rows + 1][4]  (prep_dgrad_s2_kernel): the LDS image
  float *dX;
  unsigned dyBytes;
  int C, K, H, W, Ho, Wo;
  int pl;                           // left padding (the top padding is the template parameter)
  int nbm, nkg, nmt;                // channel blocks of 96, K / 8, blocks of 64 row pairs
  int ncg[2], xfirst[2], nv[2];     // per column-parity class: column pairs, first column, filter columns (3 / 2)
};

constexpr int kDgS2Rows = 96, kDgS2Planes = 10, kDgS2PLA = kDgS2Rows * 4 + 4;   // plane pitch: one float4 of padding (the two
constexpr int kDgS2Tile = kDgS2Planes * kDgS2PLA;                               // half-waves read neighbouring planes)

// Fd from the filter bank F (FH x FW x C x K, u fastest): plane p = 2 u + (kk >> 2) of stage (class, v slot, k-group)
__global__ void __launch_bounds__(256)
prep_dgrad_s2_kernel(const float *__restrict__ F, float *__restrict__ Fd, int C, int K, int nbm, int nkg, int pl) {
  const size_t total = (size_t)nbm * 2 * 3 * nkg * kDgS2Tile;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int q = (int)(i & 3);
    size_t r = i >> 2;
    const int row = (int)(r % (kDgS2Rows + 1));     // row 96 = the padding quad of a plane
    r /= kDgS2Rows + 1;
    const int p = (int)(r % kDgS2Planes);
    r /= kDgS2Planes;
    const int kg = (int)(r % nkg);
    r /= nkg;
    const int vs = (int)(r % 3);
    r /= 3;
    const int cls = (int)(r & 1), bm = (int)(r >> 1);
    const int u = p >> 1, k = 8 * kg + 4 * (p & 1) + q, v = cls + 2 * vs, c = bm * kDgS2Rows + row;
    Fd[i] = (row < kDgS2Rows && c < C && v < 5 && k < K) ? F[u + 5 * (v + 5 * (c + (size_t)C * k))] : 0.f;
  }
}

template <int PT>
__global__ void __launch_bounds__(256, 3)
conv_dgrad_s2_kernel(const DgradS2Args a) {
  static_assert(PT == 1, "row offsets of the five filter rows are written out for a top padding of 1");
  constexpr int PLA = kDgS2PLA;                           // plane pitch of the filter image (floats)
  constexpr int SA = kDgS2Tile;                           // 3880: the stage's filter tile is copied as it lies in memory
  constexpr int PR = 34, PCOL = 72, PK = 2 * PCOL;        // patch: 34 row pairs (rows m0 - 2 ... m0 + 65) per column, 2 columns per k
  constexpr int SP = 8 * PK, STG = SA + SP + 4;           // (+ 4 spare floats: staging threads without work)
  static_assert(STG % 4 == 0, "16-byte aligned stage buffers");
  constexpr int NLA = (kDgS2Tile / 4 + 255) / 256, NLB = (8 * 2 * PR + 255) / 256;   // 4, 3
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float smem[2 * STG];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  const int wc = wave & 1, wh = wave >> 1;
  // block -> (sample, channel block, row-pair block, class, column pair); the class with three filter columns first
  int b = blockIdx.x;
  const int percls0 = a.ncg[0] * a.nmt * a.nbm, percls1 = a.ncg[1] * a.nmt * a.nbm;
  const int n = b / (percls0 + percls1);
  b -= n * (percls0 + percls1);
  const int first = a.nv[0] >= a.nv[1] ? 0 : 1;
  int cls = first;
  if (b >= (first == 0 ? percls0 : percls1)) {
    b -= first == 0 ? percls0 : percls1;
    cls = first ^ 1;
  }
  const int cg = b % a.ncg[cls];
  b /= a.ncg[cls];
  const int mt = b % a.nmt, bm = b / a.nmt;
  const int m0 = 64 * mt;
  const int xa = a.xfirst[cls] + 4 * cg;                  // the block's columns: xa, xa + 2
  const int nv = a.nv[cls];
  const int nst = nv * a.nkg;
  const __amdgpu_buffer_rsrc_t dyrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dY, 0, a.dyBytes, 0x00020000);

  for (int i = t; i < 2 * STG / 4; i += 256) reinterpret_cast<f32x4 *>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging maps ----
  const f32x4 *fsrc = reinterpret_cast<const f32x4 *>(a.Fd + ((size_t)(bm * 2 + cls) * 3) * a.nkg * kDgS2Tile) + t;
  const int ldA3 = t + 768 < kDgS2Tile / 4 ? 4 * (t + 768) : STG - 4;   // the fourth quad of the threads that have one
  unsigned voB[NLB];
  int ldB[NLB], colB[NLB];
  bool rowOk[NLB];
#pragma unroll
  for (int j = 0; j < NLB; ++j) {
    const int idx = t + 256 * j;
    const int kk = idx / (2 * PR), rem = idx - kk * (2 * PR), col = rem / PR, pr = rem - col * PR;
    const bool ok = idx < 8 * 2 * PR;
    const int row = m0 - 2 + 2 * pr;
    rowOk[j] = ok && (unsigned)row < (unsigned)a.Ho;
    colB[j] = col;
    voB[j] = (unsigned)((((n * a.K + kk) * a.Wo + col) * a.Ho + row) * 4);   // + 8 kg filters and the stage's dY column below
    ldB[j] = ok ? SA + kk * PK + col * PCOL + 2 * pr : STG - 4;
  }
  // ---- fragment addresses ----
  const float *sAr = smem + half * PLA + l31 * 4;
  const float *sBr = smem + SA + 4 * half * PK + wc * PCOL + 32 * wh + l31 + 2;

  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][e][r] = 0.f;

  f32x4 ra[NLA];
  f32x2 rb[NLB];
#pragma unroll
  for (int j = 0; j < NLA; ++j) ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int lv = 0, lkg = 0;             // the next stage to request: filter column slot, k-group
#define XM_DG_LOAD()                                                                                 \
  {                                                                                                  \
    const int v_ = cls + 2 * lv, kg_ = lkg;                                                          \
    const f32x4 *fs_ = fsrc + (size_t)(lv * a.nkg + kg_) * (kDgS2Tile / 4);                         \
    if (++lkg == a.nkg) lkg = 0, ++lv;                                                               \
    const int ja_ = (xa + a.pl - v_) >> 1;               /* dY column of the block's first column (xa + pl - v is even) */ \
    const unsigned sB_ = (unsigned)(((8 * kg_) * a.Wo + ja_) * a.Ho * 4);                             \
    _Pragma("unroll") for (int j = 0; j < NLA; ++j)                                                  \
      if (j < NLA - 1 || t + 256 * j < kDgS2Tile / 4) ra[j] = fs_[256 * j];                           \
    _Pragma("unroll") for (int j = 0; j < NLB; ++j) {                                                \
      const bool ok_ = rowOk[j] && (unsigned)(ja_ + colB[j]) < (unsigned)a.Wo;                       \
      rb[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(dyrsrc, (int)(ok_ ? voB[j] + sB_ : 0xFFFFFFFFu), 0, 0)); \
    }                                                                                                \
  }
#define XM_DG_STORE(BUF)                                                                             \
  _Pragma("unroll") for (int j = 0; j < NLA; ++j)                                                    \
    *reinterpret_cast<f32x4 *>(smem + (BUF) * STG + (j < NLA - 1 ? 4 * (t + 256 * j) : ldA3)) = ra[j]; \
  _Pragma("unroll") for (int j = 0; j < NLB; ++j)                                                    \
    *reinterpret_cast<f32x2 *>(smem + (BUF) * STG + ldB[j]) = rb[j];
  // filter row U -> (row parity E = (U + PT) & 1, dY row offset (E + PT - U) / 2): u = 0: (1, +1), 1: (0, 0), 2: (1, 0), 3: (0, -1), 4: (1, -1)
#define XM_DG_TAP(U, E, DI)                                                                          \
  {                                                                                                  \
    f32x4 af[3];                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const f32x4 *>(A + 2 * (U) * PLA + i * 128); \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                  \
      const float bf = B[e * PK + (DI)];                                                             \
      _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                  \
        acc[i][E] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf, acc[i][E], 0, 0, 0);          \
    }                                                                                                \
  }

  __syncthreads();   // the zero fill
  if (nst > 0) {
    XM_DG_LOAD()
    XM_DG_STORE(0)
    __syncthreads();
    if (1 < nst) XM_DG_LOAD()
    int cur = 0;
    for (int s = 0; s < nst; ++s) {
      if (s + 1 < nst) {
        XM_DG_STORE(cur ^ 1)
      }
      if (s + 2 < nst) XM_DG_LOAD()
      const float *A = sAr + cur * STG;
      const float *B = sBr + cur * STG;
      XM_DG_TAP(1, 0, 0)
      XM_DG_TAP(0, 1, 1)
      XM_DG_TAP(3, 0, -1)
      XM_DG_TAP(2, 1, 0)
      XM_DG_TAP(4, 1, -1)
      __syncthreads();
      cur ^= 1;
    }
  }
#undef XM_DG_LOAD
#undef XM_DG_STORE
#undef XM_DG_TAP
  // ---- epilogue: (y = 2 m, 2 m + 1) of one channel = one 8-byte store; 32 lanes = 256 contiguous bytes ----
  const int x = xa + 2 * wc, m = m0 + 32 * wh + l31;
  if (x < a.W && 2 * m < a.H) {
    const bool pair = 2 * m + 1 < a.H;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = bm * kDgS2Rows + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c < a.C) {
          float *dst = a.dX + ((size_t)(n * a.C + c) * a.W + x) * a.H + 2 * m;
          if (pair) *reinterpret_cast<f32x2 *>(dst) = f32x2{acc[i][0][r], acc[i][1][r]};
          else *dst = acc[i][0][r];
        }
      }
  }
}

}  // namespace xm
