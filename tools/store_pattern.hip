// Store-pattern probe: how fast can the chip write an [M channels][NP pixels] fp32 matrix (pixel-contiguous rows, the
// layout of every convolution output) when the writes come in the shape of the GEMM epilogue -- a block owns a tile
// of M rows x BN pixels and every wave store instruction carries 8 rows x 128 bytes -- compared with plain streaming?
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VAR 0: tile = all M rows x BN pixels, one block per tile, tiles in blockIdx order
// VAR 1: same, consecutive tiles on the same XCD (block b runs on XCD b % 8)
// VAR 2: streaming: a block writes BN * M / 96 ... i.e. one row segment of 4096 pixels
template <int VAR, int BN>
__global__ void __launch_bounds__(256) store_probe(float *y, int M, long long NP, int ntiles) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
  if (VAR == 2) {
    // row-major streaming: block -> (row, 4096-pixel segment)
    const long long seg = blockIdx.x;
    const long long per_row = (NP + 4095) / 4096;
    const int row = (int)(seg / per_row);
    const long long p0 = (seg % per_row) * 4096;
    if (row >= M) return;
    for (int i = 0; i < 4; ++i) {
      long long p = p0 + (long long)(t + 256 * i) * 4;
      if (p + 3 < NP) *reinterpret_cast<f32x4 *>(y + (long long)row * NP + p) = f32x4{1.f, 2.f, 3.f, 4.f};
    }
    return;
  }
  int tile = blockIdx.x;
  if (VAR == 1) {
    const int nx = 8, q = ntiles / nx, r = ntiles % nx, xcd = tile % nx, idx = tile / nx;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int PW = BN / 4;   // pixels per wave
  const int iq = l31 & 3;
  for (int i = 0; i < M / 32; ++i)
    for (int g4 = 0; g4 < 4; ++g4) {
      const int row = 32 * i + 8 * g4 + 4 * half + iq;
      for (int j = 0; j < PW / 32; ++j) {
        const long long p = (long long)tile * BN + wave * PW + j * 32 + (l31 & ~3);
        if (p + 3 < NP) *reinterpret_cast<f32x4 *>(y + (long long)row * NP + p) = f32x4{1.f, 2.f, 3.f, (float)g4};
      }
    }
}

template <int VAR, int BN>
static void run(const char *name, float *y, int M, long long NP) {
  const int ntiles = (int)((NP + BN - 1) / BN);
  const int grid = VAR == 2 ? (int)(M * ((NP + 4095) / 4096)) : ntiles;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((store_probe<VAR, BN>), dim3(grid), dim3(256), 0, 0, y, M, NP, ntiles);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((store_probe<VAR, BN>), dim3(grid), dim3(256), 0, 0, y, M, NP, ntiles);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%-44s M %4d NP %8lld  %.3f ms  %6.0f GB/s\n", name, M, NP, ms, (double)M * NP * 4 / ms / 1e6);
}

int main() {
  // the student's conv1 output at 32 samples (each sample is its own [96][37592] matrix in memory; one matrix of the
  // same total size behaves the same for this probe) and the teacher's res2 64 -> 256 output
  struct { int M; long long NP; } cases[] = {{96, 1202944}, {256, 100352}, {384, 16320}};
  for (auto c : cases) {
    float *y;
    hipMalloc(&y, (size_t)c.M * c.NP * 4 + 4096);
    run<2, 128>("streaming rows (4096-pixel segments)", y, c.M, c.NP);
    run<0, 128>("epilogue tiles M x 128", y, c.M, c.NP);
    run<1, 128>("epilogue tiles M x 128, XCD-contiguous", y, c.M, c.NP);
    run<0, 256>("epilogue tiles M x 256", y, c.M, c.NP);
    run<1, 256>("epilogue tiles M x 256, XCD-contiguous", y, c.M, c.NP);
    hipFree(y);
  }
  return 0;
}
