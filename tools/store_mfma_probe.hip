// Probe behind conv_stem_kernel: how well do the output stores of a store-bound GEMM tile (96 rows x
// 128 pixels, K = 56: 84 MFMAs per wave against 48 KB of stores per block) overlap with MFMAs, as a function of the
// number of co-resident blocks per CU (set through a dynamic LDS allocation), of the order inside a wave, of the OUTPUT
// LAYOUT and of the tile order?  No loads: the operands are register values that depend on the tile.
//   VAR 0: persistent; per tile 84 MFMAs, then 12 stores (the order of a GEMM with a separate epilogue)
//   VAR 1: persistent; per ROW TILE 28 MFMAs, then its 4 stores
//   VAR 2: stores only     VAR 3: MFMAs only (one store per tile)
//   LAYOUT 0: one [row][all pixels] matrix;  1: the real [sample][row][pixel] tensor (row pitch 37592 floats);
//          2: the same with rows padded to whole 128-byte lines;  3: the real tensor, XCD-contiguous tile ranges
// What it showed (boxes of the pool differ by +-20 % in their store behaviour; compare lines of ONE run): in layout 0
// the stores hide completely under the MFMAs (115 vs 114 us); in the real layout with the plain tile order they do not
// (142-166 us), row alignment is irrelevant (layout 2 = layout 1), XCD-contiguous ranges recovered the loss on one box
// (117 us) and not on another (162 us); twice the run length per row (probe_run: 64 pixels per wave) gains 4-6 %.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/store_mfma_probe.hip -o /tmp/smp && /tmp/smp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kM = 96, kBN = 128, kMfmaPerSub = 28;   // 3 row tiles x 28 k-steps = 84 MFMAs per wave and tile

template <int LAYOUT>
__device__ __forceinline__ void store_sub(float *y, long long NP, long long p0, int sub, const f32x16 &acc, int lane) {
  constexpr long long PIJ = 37592;   // 254 x 148 output pixels per spectrogram
  constexpr long long PITCH = LAYOUT == 2 ? 37600 : PIJ;   // LAYOUT 2: channel rows padded to whole 128-byte lines
  // the layout the GEMM epilogue produces after its in-register transpose: lane -> (row class, 4 consecutive pixels)
  const int half = lane >> 5, l31 = lane & 31, iq = l31 & 3;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int row = 32 * sub + 8 * g4 + 4 * half + iq;
    const long long p = p0 + (l31 & ~3);
    const long long n = p / PIJ, q = p - n * PIJ;
    float *dst = LAYOUT ? y + n * (kM * PITCH) + row * PITCH + q : y + (long long)row * NP + p;
    if (p + 3 < NP)
      *reinterpret_cast<f32x4 *>(dst) = f32x4{acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]};
  }
}

template <int VAR, int LAYOUT>
__global__ void __launch_bounds__(256) probe(float *y, long long NP, int ntiles, float a0, float b0) {
  extern __shared__ float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (a0 == 77.f) lds[t] = b0;   // keep the allocation
  f32x16 acc[3];
  const int per = (ntiles + 7) / 8, xb = (blockIdx.x & 7) * per, xe = min(ntiles, xb + per);
  for (int tile = LAYOUT == 3 ? xb + (int)(blockIdx.x >> 3) : (int)blockIdx.x; tile < (LAYOUT == 3 ? xe : ntiles); tile += LAYOUT == 3 ? (int)(gridDim.x >> 3) : (int)gridDim.x) {
    const long long p0 = (long long)tile * kBN + wave * 32;
    const float av = a0 + (float)(tile & 1023) * 1e-3f + lane * 1e-6f, bv = b0 + wave * 1e-3f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (VAR == 0 || VAR == 3) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < kMfmaPerSub; ++k) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av + (float)(i + 3 * k), bv, acc[i], 0, 0, 0);
      if (VAR == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) store_sub<LAYOUT>(y, NP, p0, i, acc[i], lane);
      } else {
        store_sub<LAYOUT>(y, NP, p0, 0, acc[0] + acc[1] + acc[2], lane);
      }
    } else if (VAR == 1) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < kMfmaPerSub; ++k) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av + (float)(i + 3 * k), bv, acc[i], 0, 0, 0);
        store_sub<LAYOUT>(y, NP, p0, i, acc[i], lane);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        acc[i][0] = av * (float)i;
        store_sub<LAYOUT>(y, NP, p0, i, acc[i], lane);
      }
    }
  }
}

template <int VAR, int LAYOUT = 0>
static void run(const char *name, float *y, long long NP, int per_cu) {
  const int ntiles = (int)((NP + kBN - 1) / kBN);
  const size_t lds = per_cu >= 8 ? 0 : (size_t)(160 * 1024 / per_cu - 1024);   // forces <= per_cu blocks per CU
  hipFuncSetAttribute((const void *)probe<VAR, LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = 256 * per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<VAR, LAYOUT>), dim3(grid), dim3(256), lds, 0, y, NP, ntiles, 1.f, 1e-3f);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<VAR, LAYOUT>), dim3(grid), dim3(256), lds, 0, y, NP, ntiles, 1.f, 1e-3f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%-44s %d blocks/CU  %6.1f us  store %5.0f GB/s  mfma %5.1f TFLOP/s\n", name, per_cu, ms * 1e3,
         (double)kM * NP * 4 / ms / 1e6, 2.0 * 96 * 56 * (double)ntiles * kBN / ms / 1e9);
}


// Run-length variant: a wave owns PW = 32 or 64 consecutive pixels of every row (a 96 x 128 or 96 x 256 block tile), real
// [sample][row][pixel] layout, XCD-contiguous ranges or plain order: does a longer run per row help the stores?
template <int PW, int XCD>
__global__ void __launch_bounds__(256) probe_run(float *y, long long NP, int ntiles, float a0, float b0, int mfma) {
  extern __shared__ float lds[];
  constexpr long long PIJ = 37592;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31, iq = l31 & 3;
  if (a0 == 77.f) lds[t] = b0;
  constexpr int NJ = PW / 32;
  f32x16 acc[3][NJ];
  const int per = (ntiles + 7) / 8, xb = (blockIdx.x & 7) * per, xe = min(ntiles, xb + per);
  for (int tile = XCD ? xb + (int)(blockIdx.x >> 3) : (int)blockIdx.x; tile < (XCD ? xe : ntiles);
       tile += XCD ? (int)(gridDim.x >> 3) : (int)gridDim.x) {
    const float av = a0 + (float)(tile & 1023) * 1e-3f + lane * 1e-6f, bv = b0 + wave * 1e-3f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (mfma)
#pragma unroll
          for (int k = 0; k < kMfmaPerSub; ++k)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av + (float)(i + 3 * k), bv + (float)j, acc[i][j], 0, 0, 0);
        else
          acc[i][j][0] = av * (float)(i + j);
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int row = 32 * i + 8 * g4 + 4 * half + iq;
          const long long p = (long long)tile * (4 * PW) + wave * PW + 32 * j + (l31 & ~3);
          const long long n = p / PIJ, q = p - n * PIJ;
          if (p + 3 < NP)
            *reinterpret_cast<f32x4 *>(y + n * (kM * PIJ) + row * PIJ + q) =
                f32x4{acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
        }
  }
}

template <int PW, int XCD>
static void run2(const char *name, float *y, long long NP, int mfma) {
  const int ntiles = (int)((NP + 4 * PW - 1) / (4 * PW));
  const size_t lds = (size_t)(160 * 1024 / 2 - 1024);
  hipFuncSetAttribute((const void *)probe_run<PW, XCD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe_run<PW, XCD>), dim3(512), dim3(256), lds, 0, y, NP, ntiles, 1.f, 1e-3f, mfma);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe_run<PW, XCD>), dim3(512), dim3(256), lds, 0, y, NP, ntiles, 1.f, 1e-3f, mfma);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%-44s pixels/wave %d %s  %6.1f us  store %5.0f GB/s\n", name, PW, XCD ? "XCD ranges" : "plain order", ms * 1e3,
         (double)kM * NP * 4 / ms / 1e6);
}

int main() {
  const long long NP = 1202944;   // conv1 of the student at 32 spectrograms: 96 x (32 * 254 * 148) floats = 462 MB
  float *y;
  hipMalloc(&y, (size_t)kM * (NP + 1024) * 4 + 4096);
  for (int pc : {2, 4}) {
    run<2>("stores only", y, NP, pc);
    run<3>("MFMAs only", y, NP, pc);
    run<0>("84 MFMAs then 12 stores", y, NP, pc);
    run<1>("3 x (28 MFMAs then 4 stores)", y, NP, pc);
    run<2, 1>("stores only, [sample][row][pixel] layout", y, NP, pc);
    run<0, 1>("84 MFMAs then 12 stores, [sample][row][pixel]", y, NP, pc);
    run<2, 2>("stores only, rows padded to 128 B", y, NP, pc);
    run<0, 2>("84 MFMAs then 12 stores, rows padded to 128 B", y, NP, pc);
    run<2, 3>("stores only, real layout, XCD-contiguous tiles", y, NP, pc);
    run<0, 3>("84 MFMAs + stores, real layout, XCD-contiguous", y, NP, pc);
  }
  run2<32, 1>("real layout, stores only", y, NP, 0);
  run2<64, 1>("real layout, stores only", y, NP, 0);
  run2<32, 1>("real layout, MFMAs + stores", y, NP, 1);
  run2<64, 1>("real layout, MFMAs + stores", y, NP, 1);
  run2<32, 0>("real layout, MFMAs + stores", y, NP, 1);
  run2<64, 0>("real layout, MFMAs + stores", y, NP, 1);
  hipFree(y);
  return 0;
}
