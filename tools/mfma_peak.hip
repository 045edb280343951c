// Ceiling probe: sustained v_mfma_f32_32x32x2_f32 rate of the whole chip with nothing else going on
// (4 independent accumulators per wave, 2 blocks x 4 waves per CU), optionally with the LDS fragment
// reads / barrier of the convolution main loop added.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ void __launch_bounds__(256, 2) probe(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int t = threadIdx.x;
  for (int i = t; i < 8192; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    lds[i] = VAR >= 10 ? ((float)(h & 0xFFFFFF) / 8388608.f - 1.f) : (float)(i & 7) * 0.001f;
  }
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a0 = {1.f, 2.f, 3.f, 4.f}, a1 = a0, b0 = a0, b1 = a0;
  const f32x4 *lp = reinterpret_cast<const f32x4 *>(lds) + (t & 63);
  for (int it = 0; it < iters; ++it) {
    if (VAR >= 1) {
      a0 = lp[(it & 3) * 64];
      a1 = lp[(it & 3) * 64 + 256];
      b0 = lp[(it & 3) * 64 + 512];
      b1 = lp[(it & 3) * 64 + 768];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[3], 0, 0, 0);
    }
    if (VAR >= 2 && (it & 1)) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + t] = s;
}


// VARIANT B: NV independent VALU ops after every MFMA (+ NL buffer loads per 16 MFMAs from a
// 64 MB array), interleave enforced with sched_group_barrier like the convolution kernels.
__device__ unsigned long long g_cyc[2];
template <int NV, int NL, int NW>
__global__ void __launch_bounds__(256, 4) probe_mix(float *out, const float *src, unsigned srcBytes, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int t = threadIdx.x;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int i = t; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.001f;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a0, a1, b0, b1;
  const f32x4 *lp = reinterpret_cast<const f32x4 *>(lds) + (t & 63);
  f32x4 *lw = reinterpret_cast<f32x4 *>(lds) + 1024 + t;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, srcBytes, 0x00020000);
  unsigned off = (blockIdx.x * 256 + t) * 4u;
  unsigned x[8];
  for (int k = 0; k < 8; ++k) x[k] = t + k;
  float ld[NL > 0 ? NL : 1];
  for (int k = 0; k < (NL > 0 ? NL : 1); ++k) ld[k] = 0.f;
  float lsum = 0.f;
  for (int it = 0; it < iters; ++it) {
    a0 = lp[(it & 3) * 64];
    a1 = lp[(it & 3) * 64 + 256];
    b0 = lp[(it & 3) * 64 + 512];
    b1 = lp[(it & 3) * 64 + 768];
    for (int k = 0; k < NL; ++k) lsum += ld[k];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      ld[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off + k * 1048576u), 0, 0));
    }
    off = (off + 262144u) & 0x3FFFFFFu;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[3], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16 * NV; ++q) x[q & 7] = x[q & 7] * 3u + (unsigned)it;
    if (NW > 0) {
#pragma unroll
      for (int k = 0; k < NW; ++k) lw[k * 256] = f32x4{lsum, lsum, lsum, lsum};
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV * 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (it & 1) __syncthreads();
  }
  if (blockIdx.x == 0 && t == 0) {
    g_cyc[0] = c0;
    g_cyc[1] = __builtin_readcyclecounter();
  }
  float s = lsum;
  for (int k = 0; k < 8; ++k) s += (float)x[k];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + t] = s;
}

template <int NV, int NL, int NW>
void run_mix(const char *name, int blocks, int iters) {
  float *out, *src;
  const unsigned srcBytes = 64u << 20;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&src, srcBytes);
  hipMemset(src, 0, srcBytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe_mix<NV, NL, NW>), dim3(blocks), dim3(256), 0, 0, out, src, srcBytes, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe_mix<NV, NL, NW>), dim3(blocks), dim3(256), 0, 0, out, src, srcBytes, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * 16 * 4096.0;
  unsigned long long cyc[2];
  hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
  double c = (double)(cyc[1] - cyc[0]);
  // s_memtime ticks at a fixed 100 MHz on this part when read through readcyclecounter?  print both views
  printf("%-44s %8.3f ms  %7.1f TFLOP/s   block0: %.0f ticks = %.1f ticks/us, %.1f ticks per 16 mfma\n", name, ms,
         fl / ms / 1e9, c, c / (ms * 1e3), c / iters);
  hipFree(out);
  hipFree(src);
}

template <int VAR>
void run(const char *name, int blocks, int iters) {
  float *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<VAR>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<VAR>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("%-28s blocks %5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
  }
  hipFree(out);
}

template <int VAR>
void sustained(const char *name, int blocks, int iters, int launches) {
  float *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < launches; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<VAR>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 16 * 4096.0;
    if (rep % 20 == 0 || rep == launches - 1)
      printf("%-28s launch %4d  %8.3f ms  %7.1f TFLOP/s\n", name, rep, ms, fl / ms / 1e9);
  }
  hipFree(out);
}

int main(int argc, char **argv) {
  if (argc > 1) {
    sustained<11>("sustained random-data mfma", 512, 20000, 200);   // ~3.6 s of back-to-back MFMA
    return 0;
  }
  run<11>("random data: lds reads + mfma", 512, 20000);
  run<12>("random data: lds reads + barrier + mfma", 512, 20000);
  run<1>("same, tiny constant data", 512, 20000);
  run<0>("pure mfma", 512, 20000);
  run<0>("pure mfma (1 block/CU)", 256, 20000);
  run<0>("pure mfma (4 rounds)", 2048, 5000);
  run<1>("+ 4 ds_read_b128 / 16 mfma", 512, 20000);
  run<2>("+ barrier / 32 mfma", 512, 20000);
  run_mix<0, 0, 0>("mix: lds+barrier only", 512, 5000);
  run_mix<1, 0, 0>("mix: 1 VALU (mul+add=2 ops) / mfma", 512, 5000);
  run_mix<2, 0, 0>("mix: 4 VALU ops / mfma", 512, 5000);
  run_mix<3, 0, 0>("mix: 6 VALU ops / mfma", 512, 5000);
  run_mix<4, 0, 0>("mix: 8 VALU ops / mfma", 512, 5000);
  run_mix<0, 5, 0>("mix: 5 buffer loads / 16 mfma", 512, 5000);
  run_mix<0, 10, 0>("mix: 10 buffer loads / 16 mfma", 512, 5000);
  run_mix<0, 0, 2>("mix: 2 ds_write_b128 / 16 mfma", 512, 5000);
  run_mix<3, 5, 2>("mix: 6 VALU + 5 loads + 2 dsw", 512, 5000);
  run_mix<3, 5, 2>("mix: 6 VALU + 5 loads + 2 dsw (1 blk/CU)", 256, 5000);
  run_mix<3, 5, 2>("mix: 6 VALU + 5 loads + 2 dsw (3 blk/CU)", 768, 5000);
  run_mix<3, 5, 2>("mix: 6 VALU + 5 loads + 2 dsw (4 blk/CU)", 1024, 5000);
  run_mix<3, 3, 2>("mix: 6 VALU + 3 loads + 2 dsw (2 blk/CU)", 512, 5000);
  run_mix<3, 3, 2>("mix: 6 VALU + 3 loads + 2 dsw (4 blk/CU)", 1024, 5000);
  return 0;
}
