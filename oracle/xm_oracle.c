/*
 * xm_oracle.c -- CPU restatement of the MatConvNet / mcnExtraLayers operator
 * semantics that albanie/mcnCrossModalEmotions drives on its hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it, and only as the
 * checker.  The product path (libxmodal_hip.so) never links or calls it.
 *
 * PARITY UNPINNED.  The reference repo is MATLAB driver code only; every
 * operator below lives in un-vendored third-party modules that are absent from
 * /root/reference and carry no version pin there:
 *     vlfeat/matconvnet      (contemporaneous release v1.0-beta25)
 *     albanie/mcnExtraLayers (HEAD at install time, setup_mcnCrossModalEmotions.m:9)
 * The reference holds no tests / golden vectors for them, and neither MATLAB nor
 * Octave exists in the build image, so the functions here restate the PUBLISHED
 * algorithms (MatConvNet manual, "Convolutional blocks" / "Pooling" /
 * "Normalization" chapters and the nnconv/nnpooling/nnbnorm CPU implementations:
 * im2row + SGEMM per image, direct-loop pooling, two-pass bnorm) and are anchored
 * on the reference's call sites:
 *     conv/pool/bnorm/relu via dag.eval        emoVoxCeleb/fetch_emovoxceleb_imdb.m:129
 *                                              external/compute_audio_feats.m:126
 *     loss  SoftmaxCELoss(T=2,logitTargets)    emoVoxCeleb/emoVoxZoo.m:152
 *     loss  softmaxlog / classerror            emoVoxCeleb/emoVoxZoo.m:149,160
 *     softmaxt                                 emoVoxCeleb/student_stats.m:95
 *     spectrogram row normalisation            emoVoxCeleb/getBatchEmoVoxCeleb.m:164-169
 *     time2idx + logit aggregation             emoVoxCeleb/getBatchEmoVoxCeleb.m:145-158,179-188,210-214
 *     face normalisation                       emoVoxCeleb/fetch_emovoxceleb_imdb.m:176-193
 *     SGD-momentum (cnn_train_dag defaults)    emoVoxCeleb/run_distillation.m:170-182
 * The handful of numerical pins the reference *does* hold (pool6 bucket table,
 * time2idx closed form, audSamp, class order) are checked in tests/test_pins.py.
 *
 * Layout everywhere: MATLAB single, H x W x C x N, column-major (H fastest).
 * Every entry point has an `acc64` flag: 0 = fp32 arithmetic in MatConvNet's
 * algorithm shape (also the timed "MatConvNet-CPU-equivalent" baseline),
 * 1 = same maths with fp64 accumulators (error-budget ground truth).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define XI(h, w, c, n, H, W, C) \
  ((size_t)(h) + (size_t)(H) * ((size_t)(w) + (size_t)(W) * ((size_t)(c) + (size_t)(C) * (size_t)(n))))

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* one OpenMP thread per entry of `cpus`, thread i pinned to logical CPU cpus[i] (the timed CPU baseline of bench.py:
 * one thread per physical core; unpinned teams on a shared host moved the figure by 30 % between boxes).  Returns the
 * number of threads that could be pinned.  The calling thread is thread 0: the caller restores its affinity. */
int orc_pin_threads(const int *cpus, int n) {
  int pinned = 0;
#ifdef _OPENMP
  if (n <= 0) return 0;
  omp_set_num_threads(n);
#pragma omp parallel num_threads(n) reduction(+ : pinned)
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(cpus[omp_get_thread_num() % n], &set);
    if (sched_setaffinity(0, sizeof set, &set) == 0) pinned += 1;
  }
#else
  (void)cpus;
  (void)n;
#endif
  return pinned;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static int out_size(int in, int pa, int pb, int f, int dil, int s) {
  int feff = (f - 1) * dil + 1;
  int t = in + pa + pb - feff;
  if (t < 0) return 0;
  return t / s + 1;
}

int orc_conv_out_size(int in, int pad_a, int pad_b, int f, int dilate, int stride) {
  return out_size(in, pad_a, pad_b, f, dilate, stride);
}

/* ------------------------------------------------------------------ */
/* small blocked SGEMM pieces (no system BLAS in the image)            */
/* ------------------------------------------------------------------ */

/* Threads: conv forward / dgrad first build the im2row matrices of a CHUNK of images (as many as fit 1 GiB), then
 * run one flat parallel loop over (image, row block, column block) SGEMM work items -- MatConvNet's CPU path loops
 * over images and hands each SGEMM to a multithreaded BLAS; this is the same arithmetic (per output element an fp32
 * chain over k in the same order) with the cores kept busy for small images too.  No nested teams. */
/* grow-only scratch for the im2row matrices: a fresh 50+ MB malloc per call is an mmap + page-fault storm that
 * costs more than the SGEMM it feeds (not thread-safe across concurrent oracle calls; the callers are serial) */
static float *g_scratch = NULL;
static size_t g_scratch_floats = 0;
static float *scratch_get(size_t floats) {
  if (floats > g_scratch_floats) {
    free(g_scratch);
    g_scratch = (float *)aligned_alloc(4096, (sizeof(float) * floats + 4095) & ~(size_t)4095);
    g_scratch_floats = g_scratch ? floats : 0;
  }
  return g_scratch;
}

static size_t chunk_images(size_t per_image_floats, int N) {
  size_t cap = ((size_t)1 << 28) / (per_image_floats ? per_image_floats : 1); /* 2^28 floats = 1 GiB */
  if (cap < 1) cap = 1;
  return cap > (size_t)N ? (size_t)N : cap;
}

/* micro-kernel: acc[j][0..31] += A[0..31 + lda*k] * B[k*bks + j*bns], k = 0..K-1 (fp32 chain over k per element,
 * whatever the vector width: the clones differ in speed, not in rounding) */
enum { GMB = 32, GNB = 4 };
__attribute__((target_clones("avx512f", "default")))
static void gemm_nn_block(int K, const float *A, int lda, const float *B, size_t bks, size_t bns,
                          float acc[GNB][GMB]) {
  float c0[GMB], c1[GMB], c2[GMB], c3[GMB];
  for (int i = 0; i < GMB; ++i) {
    c0[i] = acc[0][i];
    c1[i] = acc[1][i];
    c2[i] = acc[2][i];
    c3[i] = acc[3][i];
  }
  for (int k = 0; k < K; ++k) {
    const float *a = A + (size_t)lda * k;
    const float b0 = B[k * bks], b1 = B[k * bks + bns], b2 = B[k * bks + 2 * bns], b3 = B[k * bks + 3 * bns];
    for (int i = 0; i < GMB; ++i) {
      float av = a[i];
      c0[i] += av * b0;
      c1[i] += av * b1;
      c2[i] += av * b2;
      c3[i] += av * b3;
    }
  }
  for (int i = 0; i < GMB; ++i) {
    acc[0][i] = c0[i];
    acc[1][i] = c1[i];
    acc[2][i] = c2[i];
    acc[3][i] = c3[i];
  }
}

/* C_i[m + ldc*n] (+)= sum_k A_i[m + lda*k] * B[k*bks + n*bns]   (fp32 chain over k) for `cnt` independent problems
 * that share B: A_i = A + i*strideA, C_i = C + i*strideC; one flat
 * parallel loop over (problem, row block, column block) */
static void gemm_nn_batched(int cnt, int M, int N, int K, const float *A, size_t strideA, int lda, const float *B,
                            size_t bks, size_t bns, float *C, size_t strideC, int ldc, int accumulate) {
  enum { MB = GMB, NB = GNB };
  int nmb = (M + MB - 1) / MB, nnb = (N + NB - 1) / NB;
#pragma omp parallel for collapse(3) schedule(static)
  for (int q = 0; q < cnt; ++q)
    for (int ib = 0; ib < nmb; ++ib)
      for (int jb = 0; jb < nnb; ++jb) {
        const float *Aq = A + strideA * q;
        float *Cq = C + strideC * q;
        int m0 = ib * MB, n0 = jb * NB;
        int mb = M - m0 < MB ? M - m0 : MB, nb = N - n0 < NB ? N - n0 : NB;
        float acc[NB][MB];
        for (int j = 0; j < NB; ++j)
          for (int i = 0; i < MB; ++i)
            acc[j][i] = (accumulate && j < nb && i < mb) ? Cq[m0 + i + (size_t)ldc * (n0 + j)] : 0.f;
        if (mb == MB && nb == NB) {
          gemm_nn_block(K, Aq + m0, lda, B + (size_t)n0 * bns, bks, bns, acc);
        } else {
          for (int k = 0; k < K; ++k) {
            const float *a = Aq + m0 + (size_t)lda * k;
            for (int j = 0; j < nb; ++j) {
              float bj = B[k * bks + (n0 + j) * bns];
              for (int i = 0; i < mb; ++i) acc[j][i] += a[i] * bj;
            }
          }
        }
        for (int j = 0; j < nb; ++j)
          for (int i = 0; i < mb; ++i) Cq[m0 + i + (size_t)ldc * (n0 + j)] = acc[j][i];
      }
}

/* C[r + ldc*n] += sum_p A[p + lda*r] * B[p + ldb*n]   (A^T B, reduction over contiguous p) */
static void gemm_tn_acc(int R, int N, int P, const float *A, int lda, const float *B, int ldb,
                        float *C, int ldc) {
  enum { RB = 4, NB = 2, V = 8 };
  int nrb = (R + RB - 1) / RB, nnb = (N + NB - 1) / NB;
#pragma omp parallel for collapse(2) schedule(static)
  for (int ib = 0; ib < nrb; ++ib)
    for (int jb = 0; jb < nnb; ++jb) {
      int r0 = ib * RB, n0 = jb * NB;
      int rb = R - r0 < RB ? R - r0 : RB, nb = N - n0 < NB ? N - n0 : NB;
      float acc[NB][RB][V];
      memset(acc, 0, sizeof acc);
      int p = 0;
      for (; p + V <= P; p += V)
        for (int j = 0; j < nb; ++j)
          for (int i = 0; i < rb; ++i) {
            const float *a = A + p + (size_t)lda * (r0 + i);
            const float *b = B + p + (size_t)ldb * (n0 + j);
            for (int v = 0; v < V; ++v) acc[j][i][v] += a[v] * b[v];
          }
      for (int j = 0; j < nb; ++j)
        for (int i = 0; i < rb; ++i) {
          float s = 0.f;
          for (int v = 0; v < V; ++v) s += acc[j][i][v];
          for (int q = p; q < P; ++q)
            s += A[q + (size_t)lda * (r0 + i)] * B[q + (size_t)ldb * (n0 + j)];
          C[r0 + i + (size_t)ldc * (n0 + j)] += s;
        }
    }
}

/* ------------------------------------------------------------------ */
/* fp64-accumulate helpers of vl_nnconv's acc64 path                    */
/* ------------------------------------------------------------------ */
/* The acc64 loops are the same sums as the fp32 path with double accumulators.  They are arranged so that the
 * innermost loop runs over the output ROW (ho) -- independent accumulators, contiguous memory -- instead of one
 * dependent chain per output element: the 256-spectrogram checks of tests/test_gpu_bench_sizes.py spent minutes
 * in a chain that retires one multiply-add per floating-point latency.  Forward and dX keep, per output element, exactly
 * the order of additions of the element-at-a-time loops they replace ((c, v, u) / (k, v, u) ascending; every product
 * of two floats is exact in double, so fused or separate multiply-add round alike): bit-identical results.  dF keeps
 * eight interleaved partial sums per filter element -- still fp64 throughout, the order is fixed. */

/* rows ho of an output column whose source row hi = ho * s + off lies in [0, H): [*lo, *hi] (empty if *lo > *hi) */
static void row_range(int off, int s, int H, int Ho, int *lo, int *hi) {
  *lo = off >= 0 ? 0 : (-off + s - 1) / s;
  int top = H - 1 - off;
  *hi = top < 0 ? -1 : top / s;
  if (*hi > Ho - 1) *hi = Ho - 1;
}

/* ------------------------------------------------------------------ */
/* vl_nnconv                                                           */
/* ------------------------------------------------------------------ */

/* im2row of one image / one filter group: col[p + P*(u + FH*(v + FW*c))] */
static void im2row(const float *x, int H, int W, int FC, int FH, int FW, int sy, int sx, int pt,
                   int pl, int dy, int dx, int Ho, int Wo, float *col) {
  size_t P = (size_t)Ho * Wo;
  int R = FH * FW * FC;
#pragma omp parallel for schedule(static)
  for (int r = 0; r < R; ++r) {
    int u = r % FH, v = (r / FH) % FW, c = r / (FH * FW);
    float *dst = col + P * r;
    for (int wo = 0; wo < Wo; ++wo) {
      int wi = wo * sx - pl + v * dx;
      for (int ho = 0; ho < Ho; ++ho) {
        int hi = ho * sy - pt + u * dy;
        float val = 0.f;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) val = x[XI(hi, wi, c, 0, H, W, FC)];
        dst[ho + (size_t)Ho * wo] = val;
      }
    }
  }
}

/* im2row of images [0, cnt) of one filter group: col_i = col + i*P*R, x_i = x + i*xstride */
static void im2row_batched(int cnt, const float *x, size_t xstride, int H, int W, int FC, int FH, int FW, int sy,
                           int sx, int pt, int pl, int dy, int dx, int Ho, int Wo, float *col) {
  size_t P = (size_t)Ho * Wo;
  int R = FH * FW * FC;
#pragma omp parallel for collapse(2) schedule(static)
  for (int q = 0; q < cnt; ++q)
    for (int r = 0; r < R; ++r) {
      const float *xq = x + xstride * q;
      int u = r % FH, v = (r / FH) % FW, c = r / (FH * FW);
      float *dst = col + P * ((size_t)R * q + r);
      for (int wo = 0; wo < Wo; ++wo) {
        int wi = wo * sx - pl + v * dx;
        for (int ho = 0; ho < Ho; ++ho) {
          int hi = ho * sy - pt + u * dy;
          float val = 0.f;
          if (hi >= 0 && hi < H && wi >= 0 && wi < W) val = xq[XI(hi, wi, c, 0, H, W, FC)];
          dst[ho + (size_t)Ho * wo] = val;
        }
      }
    }
}

/* row2im of images [0, cnt): scatter-add each dcol_i back into dx_i (taps of one channel stay in one thread) */
static void row2im_add_batched(int cnt, const float *col, int H, int W, int FC, int FH, int FW, int sy, int sx,
                               int pt, int pl, int dy, int dx, int Ho, int Wo, float *xg, size_t xstride) {
  size_t P = (size_t)Ho * Wo;
  int R = FH * FW * FC;
#pragma omp parallel for collapse(2) schedule(static)
  for (int q = 0; q < cnt; ++q)
    for (int c = 0; c < FC; ++c)
      for (int v = 0; v < FW; ++v)
        for (int u = 0; u < FH; ++u) {
          const float *src = col + P * ((size_t)R * q + (u + FH * (v + FW * c)));
          float *xq = xg + xstride * q;
          for (int wo = 0; wo < Wo; ++wo) {
            int wi = wo * sx - pl + v * dx;
            if (wi < 0 || wi >= W) continue;
            for (int ho = 0; ho < Ho; ++ho) {
              int hi = ho * sy - pt + u * dy;
              if (hi < 0 || hi >= H) continue;
              xq[XI(hi, wi, c, 0, H, W, FC)] += src[ho + (size_t)Ho * wo];
            }
          }
        }
}

/*
 * Y = vl_nnconv(X, F, B, 'stride', [sy sx], 'pad', [pt pb pl pr], 'dilate', [dy dx])
 * X: H x W x C x N, F: FH x FW x FC x K (groups G = C / FC), B: K (may be NULL).
 * Cross-correlation, zero padding.  Returns 0, or -1 on a shape error.
 */
int orc_nnconv_forward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                       int FC, int K, const float *b, float *y, int sy, int sx, int pt, int pb,
                       int pl, int pr, int dy, int dx, int acc64) {
  if (FC <= 0 || C % FC) return -1;
  int G = C / FC;
  if (K % G) return -1;
  int Kg = K / G;
  int Ho = out_size(H, pt, pb, FH, dy, sy), Wo = out_size(W, pl, pr, FW, dx, sx);
  if (Ho <= 0 || Wo <= 0) return -1;
  size_t P = (size_t)Ho * Wo;
  int R = FH * FW * FC;
  if (acc64) {
    int failed = 0;
    int lo_u[FH], hi_u[FH]; /* rows of an output column that filter row u reaches inside the image */
    for (int u = 0; u < FH; ++u) row_range(u * dy - pt, sy, H, Ho, &lo_u[u], &hi_u[u]);
    /* four filters of a group share every source row they read (one load + conversion, four multiply-adds) */
    const int KB = (Kg % 4 == 0) ? 4 : 1, nkb = K / KB;
#pragma omp parallel
    {
      double *acc = (double *)malloc(sizeof(double) * (size_t)Ho * 4);
      if (!acc) {
#pragma omp atomic write
        failed = 1;
      }
#pragma omp for collapse(2) schedule(static)
      for (int n = 0; n < N; ++n)
        for (int kb = 0; kb < nkb; ++kb) {
          if (!acc) continue;
          const int k0 = kb * KB, g = k0 / Kg;
          double *a0 = acc, *a1 = acc + Ho, *a2 = acc + 2 * (size_t)Ho, *a3 = acc + 3 * (size_t)Ho;
          for (int wo = 0; wo < Wo; ++wo) {
            for (int j = 0; j < KB; ++j) {
              const double b0 = b ? (double)b[k0 + j] : 0.0;
              for (int ho = 0; ho < Ho; ++ho) acc[(size_t)Ho * j + ho] = b0;
            }
            for (int c = 0; c < FC; ++c)
              for (int v = 0; v < FW; ++v) {
                int wi = wo * sx - pl + v * dx;
                if (wi < 0 || wi >= W) continue;
                const float *xc = x + XI(0, wi, g * FC + c, n, H, W, C);
                for (int u = 0; u < FH; ++u) {
                  const int off = u * dy - pt, lo = lo_u[u], hi = hi_u[u];
                  const size_t fi = (size_t)u + FH * ((size_t)v + FW * ((size_t)c + (size_t)FC * k0));
                  const double f0 = (double)f[fi];
                  if (KB == 4) {
                    const double f1 = (double)f[fi + (size_t)R], f2 = (double)f[fi + 2 * (size_t)R],
                                 f3 = (double)f[fi + 3 * (size_t)R];
                    for (int ho = lo; ho <= hi; ++ho) {
                      const double xv = (double)xc[ho * sy + off];
                      a0[ho] += xv * f0;
                      a1[ho] += xv * f1;
                      a2[ho] += xv * f2;
                      a3[ho] += xv * f3;
                    }
                  } else {
                    for (int ho = lo; ho <= hi; ++ho) a0[ho] += (double)xc[ho * sy + off] * f0;
                  }
                }
              }
            for (int j = 0; j < KB; ++j) {
              float *yc = y + XI(0, wo, k0 + j, n, Ho, Wo, K);
              for (int ho = 0; ho < Ho; ++ho) yc[ho] = (float)acc[(size_t)Ho * j + ho];
            }
          }
        }
      free(acc);
    }
    return failed ? -2 : 0;
  }
  size_t chunk = chunk_images(P * R, N);
  float *col = scratch_get(P * R * chunk);
  if (!col) return -2;
  const size_t xs = (size_t)H * W * C, ys = P * K;
  for (int n0 = 0; n0 < N; n0 += (int)chunk) {
    int cnt = N - n0 < (int)chunk ? N - n0 : (int)chunk;
    for (int g = 0; g < G; ++g) {
      im2row_batched(cnt, x + XI(0, 0, g * FC, n0, H, W, C), xs, H, W, FC, FH, FW, sy, sx, pt, pl, dy, dx, Ho, Wo,
                     col);
      float *yg = y + XI(0, 0, g * Kg, n0, Ho, Wo, K);
      /* bias first (MatConvNet: rank-1 GEMM with a ones vector), then accumulate */
#pragma omp parallel for collapse(2) schedule(static)
      for (int q = 0; q < cnt; ++q)
        for (int k = 0; k < Kg; ++k) {
          float bv = b ? b[g * Kg + k] : 0.f;
          float *yk = yg + ys * q + P * k;
          for (size_t p = 0; p < P; ++p) yk[p] = bv;
        }
      gemm_nn_batched(cnt, (int)P, Kg, R, col, P * R, (int)P, f + (size_t)R * g * Kg, 1, (size_t)R, yg, ys, (int)P,
                      1);
    }
  }
  return 0;
}

/*
 * [DX, DF, DB] = vl_nnconv(X, F, B, DZDY, ...).  Any of dxo/dfo/dbo may be NULL
 * (NoDerData / NoDerFilters / NoDerBiases).
 */
int orc_nnconv_backward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                        int FC, int K, const float *dzdy, float *dxo, float *dfo, float *dbo,
                        int sy, int sx, int pt, int pb, int pl, int pr, int dy, int dx,
                        int acc64) {
  if (FC <= 0 || C % FC) return -1;
  int G = C / FC;
  if (K % G) return -1;
  int Kg = K / G;
  int Ho = out_size(H, pt, pb, FH, dy, sy), Wo = out_size(W, pl, pr, FW, dx, sx);
  if (Ho <= 0 || Wo <= 0) return -1;
  size_t P = (size_t)Ho * Wo;
  int R = FH * FW * FC;
  if (dbo) {
    for (int k = 0; k < K; ++k) {
      if (acc64) {
        double s = 0;
        for (int n = 0; n < N; ++n) {
          const float *d = dzdy + XI(0, 0, k, n, Ho, Wo, K);
          for (size_t p = 0; p < P; ++p) s += d[p];
        }
        dbo[k] = (float)s;
      } else {
        float s = 0;
        for (int n = 0; n < N; ++n) {
          const float *d = dzdy + XI(0, 0, k, n, Ho, Wo, K);
          float si = 0;
          for (size_t p = 0; p < P; ++p) si += d[p];
          s += si;
        }
        dbo[k] = s;
      }
    }
  }
  if (acc64) {
    int lo_u[FH], hi_u[FH]; /* rows of an output column that filter row u reaches inside the image */
    for (int u = 0; u < FH; ++u) row_range(u * dy - pt, sy, H, Ho, &lo_u[u], &hi_u[u]);
    if (dfo) {
      /* per filter element eight interleaved partial sums run over ALL (sample, output column) rows and are added once at
         the end: lane j takes the rows' elements lo + j, lo + j + 8, ... -- a fixed order, fp64 throughout */
#pragma omp parallel for collapse(2) schedule(static)
      for (int k = 0; k < K; ++k)
        for (int c = 0; c < FC; ++c) {
          int g = k / Kg;
          for (int v = 0; v < FW; ++v) {
            double ps[FH][8];
            for (int u = 0; u < FH; ++u)
              for (int j = 0; j < 8; ++j) ps[u][j] = 0;
            for (int n = 0; n < N; ++n)
              for (int wo = 0; wo < Wo; ++wo) {
                int wi = wo * sx - pl + v * dx;
                if (wi < 0 || wi >= W) continue;
                const float *xc = x + XI(0, wi, g * FC + c, n, H, W, C);
                const float *dc = dzdy + XI(0, wo, k, n, Ho, Wo, K);
                for (int u = 0; u < FH; ++u) {
                  const int off = u * dy - pt, hi = hi_u[u];
                  double *pu = ps[u];
                  int h = lo_u[u];
                  if (sy == 1) {
                    for (; h + 7 <= hi; h += 8)
                      for (int j = 0; j < 8; ++j) pu[j] += (double)xc[h + j + off] * (double)dc[h + j];
                  } else {
                    for (; h + 7 <= hi; h += 8)
                      for (int j = 0; j < 8; ++j) pu[j] += (double)xc[(h + j) * sy + off] * (double)dc[h + j];
                  }
                  for (int j = 0; h <= hi; ++h, ++j) pu[j] += (double)xc[h * sy + off] * (double)dc[h];
                }
              }
            for (int u = 0; u < FH; ++u) {
              const double *q = ps[u];
              dfo[(size_t)u + FH * ((size_t)v + FW * ((size_t)c + (size_t)FC * k))] =
                  (float)(((q[0] + q[4]) + (q[2] + q[6])) + ((q[1] + q[5]) + (q[3] + q[7])));
            }
          }
        }
    }
    if (dxo) {
      int failed = 0;
      /* four input channels of a group share every derivative row they read */
      const int CB = (FC % 4 == 0) ? 4 : 1, ncb = C / CB;
      const size_t fcs = (size_t)FH * FW;   /* filter stride between input channels */
#pragma omp parallel
      {
        double *acc = (double *)malloc(sizeof(double) * (size_t)H * 4);
        if (!acc) {
#pragma omp atomic write
          failed = 1;
        }
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
          for (int cb = 0; cb < ncb; ++cb) {
            if (!acc) continue;
            const int ci0 = cb * CB, g = ci0 / FC, c = ci0 % FC;
            double *a0 = acc, *a1 = acc + H, *a2 = acc + 2 * (size_t)H, *a3 = acc + 3 * (size_t)H;
            for (int wi = 0; wi < W; ++wi) {
              for (size_t i = 0; i < (size_t)H * CB; ++i) acc[i] = 0;
              for (int kk = 0; kk < Kg; ++kk) {
                int k = g * Kg + kk;
                for (int v = 0; v < FW; ++v) {
                  int tw = wi + pl - v * dx;
                  if (tw < 0 || tw % sx) continue;
                  int wo = tw / sx;
                  if (wo >= Wo) continue;
                  const float *dc = dzdy + XI(0, wo, k, n, Ho, Wo, K);
                  for (int u = 0; u < FH; ++u) {
                    const int off = u * dy - pt, lo = lo_u[u], hi = hi_u[u];
                    const size_t fi = (size_t)u + FH * ((size_t)v + FW * ((size_t)c + (size_t)FC * k));
                    const double f0 = (double)f[fi];
                    if (CB == 4) {
                      const double f1 = (double)f[fi + fcs], f2 = (double)f[fi + 2 * fcs], f3 = (double)f[fi + 3 * fcs];
                      for (int ho = lo; ho <= hi; ++ho) {
                        const double dv = (double)dc[ho];
                        const int hi_ = ho * sy + off;
                        a0[hi_] += dv * f0;
                        a1[hi_] += dv * f1;
                        a2[hi_] += dv * f2;
                        a3[hi_] += dv * f3;
                      }
                    } else {
                      for (int ho = lo; ho <= hi; ++ho) a0[ho * sy + off] += (double)dc[ho] * f0;
                    }
                  }
                }
              }
              for (int j = 0; j < CB; ++j) {
                float *xo = dxo + XI(0, wi, ci0 + j, n, H, W, C);
                for (int hi = 0; hi < H; ++hi) xo[hi] = (float)acc[(size_t)H * j + hi];
              }
            }
          }
        free(acc);
      }
      if (failed) return -2;
    }
    return 0;
  }
  if (dfo) {
    /* dzdf accumulates over the images in image order (as MatConvNet does): images stay sequential, the threads
       work inside each im2row / SGEMM */
    float *col = scratch_get(P * R);
    if (!col) return -2;
    memset(dfo, 0, sizeof(float) * (size_t)R * K);
    for (int n = 0; n < N; ++n)
      for (int g = 0; g < G; ++g) {
        const float *dyg = dzdy + XI(0, 0, g * Kg, n, Ho, Wo, K);
        im2row(x + XI(0, 0, g * FC, n, H, W, C), H, W, FC, FH, FW, sy, sx, pt, pl, dy, dx, Ho, Wo, col);
        gemm_tn_acc(R, Kg, (int)P, col, (int)P, dyg, (int)P, dfo + (size_t)R * g * Kg, R);
      }
  }
  if (dxo) {
    memset(dxo, 0, sizeof(float) * (size_t)H * W * C * N);
    size_t chunk = chunk_images(P * R, N);
    float *col = scratch_get(P * R * chunk);
    if (!col) return -2;
    const size_t xs = (size_t)H * W * C, ys = P * K;
    for (int n0 = 0; n0 < N; n0 += (int)chunk) {
      int cnt = N - n0 < (int)chunk ? N - n0 : (int)chunk;
      for (int g = 0; g < G; ++g) {
        const float *dyg = dzdy + XI(0, 0, g * Kg, n0, Ho, Wo, K);
        /* dcol[p, r] = sum_k dy[p, k] * f[r, k] */
        gemm_nn_batched(cnt, (int)P, R, Kg, dyg, ys, (int)P, f + (size_t)R * g * Kg, (size_t)R, 1, col, P * R,
                        (int)P, 0);
        row2im_add_batched(cnt, col, H, W, FC, FH, FW, sy, sx, pt, pl, dy, dx, Ho, Wo,
                           dxo + XI(0, 0, g * FC, n0, H, W, C), xs);
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* vl_nnpool                                                           */
/* ------------------------------------------------------------------ */
/* method: 0 = max (padding = -inf), 1 = avg (divide by the CLIPPED window area).
 * Window scan order is column-major (w outer, h inner) as in MatConvNet's
 * pooling_cpu; max-backward routes to the FIRST maximum in that order. */
int orc_nnpool_forward(const float *x, int H, int W, int C, int N, int ph, int pw, int sy, int sx,
                       int pt, int pb, int pl, int pr, int method, float *y) {
  int Ho = out_size(H, pt, pb, ph, 1, sy), Wo = out_size(W, pl, pr, pw, 1, sx);
  if (Ho <= 0 || Wo <= 0) return -1;
#pragma omp parallel for schedule(static)
  for (long cn = 0; cn < (long)C * N; ++cn) {
    const float *xp = x + (size_t)H * W * cn;
    float *yp = y + (size_t)Ho * Wo * cn;
    for (int wo = 0; wo < Wo; ++wo)
      for (int ho = 0; ho < Ho; ++ho) {
        int w1 = wo * sx - pl, h1 = ho * sy - pt;
        int w2 = w1 + pw < W ? w1 + pw : W, h2 = h1 + ph < H ? h1 + ph : H;
        if (w1 < 0) w1 = 0;
        if (h1 < 0) h1 = 0;
        if (method == 0) {
          float m = -INFINITY;
          for (int w = w1; w < w2; ++w)
            for (int h = h1; h < h2; ++h) {
              float v = xp[h + (size_t)H * w];
              if (v > m) m = v;
            }
          yp[ho + (size_t)Ho * wo] = m;
        } else {
          float s = 0.f;
          for (int w = w1; w < w2; ++w)
            for (int h = h1; h < h2; ++h) s += xp[h + (size_t)H * w];
          yp[ho + (size_t)Ho * wo] = s * (1.0f / (float)((h2 - h1) * (w2 - w1)));
        }
      }
  }
  return 0;
}

int orc_nnpool_backward(const float *x, int H, int W, int C, int N, int ph, int pw, int sy, int sx,
                        int pt, int pb, int pl, int pr, int method, const float *dzdy, float *dxo) {
  int Ho = out_size(H, pt, pb, ph, 1, sy), Wo = out_size(W, pl, pr, pw, 1, sx);
  if (Ho <= 0 || Wo <= 0) return -1;
#pragma omp parallel for schedule(static)
  for (long cn = 0; cn < (long)C * N; ++cn) {
    const float *xp = x + (size_t)H * W * cn;
    const float *dp = dzdy + (size_t)Ho * Wo * cn;
    float *gp = dxo + (size_t)H * W * cn;
    memset(gp, 0, sizeof(float) * (size_t)H * W);
    for (int wo = 0; wo < Wo; ++wo)
      for (int ho = 0; ho < Ho; ++ho) {
        int w1 = wo * sx - pl, h1 = ho * sy - pt;
        int w2 = w1 + pw < W ? w1 + pw : W, h2 = h1 + ph < H ? h1 + ph : H;
        if (w1 < 0) w1 = 0;
        if (h1 < 0) h1 = 0;
        float d = dp[ho + (size_t)Ho * wo];
        if (method == 0) {
          float m = -INFINITY;
          long arg = -1;
          for (int w = w1; w < w2; ++w)
            for (int h = h1; h < h2; ++h) {
              float v = xp[h + (size_t)H * w];
              if (v > m) {
                m = v;
                arg = h + (long)H * w;
              }
            }
          if (arg >= 0) gp[arg] += d;
        } else {
          float sc = d * (1.0f / (float)((h2 - h1) * (w2 - w1)));
          for (int w = w1; w < w2; ++w)
            for (int h = h1; h < h2; ++h) gp[h + (size_t)H * w] += sc;
        }
      }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* vl_nnbnorm                                                          */
/* ------------------------------------------------------------------ */
/* moments: C x 2 column-major = [mean(0..C-1), sigma(0..C-1)], sigma = sqrt(var_biased + eps).
 * moments_in == NULL  -> train mode (batch moments, returned through moments_out if non-NULL)
 * moments_in != NULL  -> test mode  (stored moments used as constants) */
static void bn_moments(const float *x, size_t HW, int C, int N, float eps, int acc64, float *mom) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    double m = (double)HW * N;
    if (acc64) {
      double s = 0, ss = 0;
      for (int n = 0; n < N; ++n) {
        const float *p = x + HW * ((size_t)c + (size_t)C * n);
        for (size_t i = 0; i < HW; ++i) s += p[i];
      }
      double mu = s / m;
      for (int n = 0; n < N; ++n) {
        const float *p = x + HW * ((size_t)c + (size_t)C * n);
        for (size_t i = 0; i < HW; ++i) ss += ((double)p[i] - mu) * ((double)p[i] - mu);
      }
      mom[c] = (float)mu;
      mom[C + c] = (float)sqrt(ss / m + (double)eps);
    } else {
      /* MatConvNet CPU: accumulate sum and sum of squares in one pass, fp32 */
      float s = 0.f, ss = 0.f;
      for (int n = 0; n < N; ++n) {
        const float *p = x + HW * ((size_t)c + (size_t)C * n);
        float si = 0.f, ssi = 0.f;
        for (size_t i = 0; i < HW; ++i) {
          si += p[i];
          ssi += p[i] * p[i];
        }
        s += si;
        ss += ssi;
      }
      float mu = s / (float)m;
      float var = ss / (float)m - mu * mu;
      if (var < 0.f) var = 0.f;
      mom[c] = mu;
      mom[C + c] = sqrtf(var + eps);
    }
  }
}

int orc_nnbnorm_forward(const float *x, int H, int W, int C, int N, const float *g, const float *b,
                        float eps, const float *moments_in, float *y, float *moments_out,
                        int acc64) {
  size_t HW = (size_t)H * W;
  float *mom = (float *)malloc(sizeof(float) * 2 * C);
  if (moments_in)
    memcpy(mom, moments_in, sizeof(float) * 2 * C);
  else
    bn_moments(x, HW, C, N, eps, acc64, mom);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      const float *p = x + HW * ((size_t)c + (size_t)C * n);
      float *q = y + HW * ((size_t)c + (size_t)C * n);
      if (acc64) {
        double sc = (double)g[c] / (double)mom[C + c], mu = mom[c], bb = b[c];
        for (size_t i = 0; i < HW; ++i) q[i] = (float)(sc * ((double)p[i] - mu) + bb);
      } else {
        float sc = g[c] / mom[C + c], mu = mom[c], bb = b[c];
        for (size_t i = 0; i < HW; ++i) q[i] = sc * (p[i] - mu) + bb;
      }
    }
  if (moments_out) memcpy(moments_out, mom, sizeof(float) * 2 * C);
  free(mom);
  return 0;
}

int orc_nnbnorm_backward(const float *x, int H, int W, int C, int N, const float *g, const float *b,
                         const float *dzdy, float eps, const float *moments_in, float *dxo,
                         float *dgo, float *dbo, float *moments_out, int acc64) {
  (void)b;
  size_t HW = (size_t)H * W;
  double m = (double)HW * N;
  float *mom = (float *)malloc(sizeof(float) * 2 * C);
  if (moments_in)
    memcpy(mom, moments_in, sizeof(float) * 2 * C);
  else
    bn_moments(x, HW, C, N, eps, acc64, mom);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    double mu = mom[c], sg = mom[C + c];
    double sdy = 0, sdyx = 0;
    if (acc64) {
      for (int n = 0; n < N; ++n) {
        const float *p = x + HW * ((size_t)c + (size_t)C * n);
        const float *d = dzdy + HW * ((size_t)c + (size_t)C * n);
        for (size_t i = 0; i < HW; ++i) {
          sdy += d[i];
          sdyx += (double)d[i] * ((double)p[i] - mu);
        }
      }
    } else {
      float fs = 0.f, fsx = 0.f, fmu = mom[c];
      for (int n = 0; n < N; ++n) {
        const float *p = x + HW * ((size_t)c + (size_t)C * n);
        const float *d = dzdy + HW * ((size_t)c + (size_t)C * n);
        float a = 0.f, bx = 0.f;
        for (size_t i = 0; i < HW; ++i) {
          a += d[i];
          bx += d[i] * (p[i] - fmu);
        }
        fs += a;
        fsx += bx;
      }
      sdy = fs;
      sdyx = fsx;
    }
    double dg = sdyx / sg; /* sum dzdy * xhat */
    if (dgo) dgo[c] = (float)dg;
    if (dbo) dbo[c] = (float)sdy;
    if (dxo) {
      double gs = (double)g[c] / sg;
      for (int n = 0; n < N; ++n) {
        const float *p = x + HW * ((size_t)c + (size_t)C * n);
        const float *d = dzdy + HW * ((size_t)c + (size_t)C * n);
        float *q = dxo + HW * ((size_t)c + (size_t)C * n);
        if (moments_in) {
          for (size_t i = 0; i < HW; ++i) q[i] = (float)(gs * (double)d[i]);
        } else if (acc64) {
          for (size_t i = 0; i < HW; ++i) {
            double xh = ((double)p[i] - mu) / sg;
            q[i] = (float)(gs * ((double)d[i] - sdy / m - xh * dg / m));
          }
        } else {
          float fgs = (float)gs, c1 = (float)(sdy / m), c2 = (float)(dg / (m * sg)), fmu = mom[c];
          for (size_t i = 0; i < HW; ++i) q[i] = fgs * (d[i] - c1 - (p[i] - fmu) * c2);
        }
      }
    }
  }
  if (moments_out) memcpy(moments_out, mom, sizeof(float) * 2 * C);
  free(mom);
  return 0;
}

/* ------------------------------------------------------------------ */
/* elementwise: vl_nnrelu, vl_nnsigmoid, dagnn.Sum, SE scale / axpy     */
/* ------------------------------------------------------------------ */
void orc_nnrelu(const float *x, size_t n, float leak, const float *dzdy, float *y) {
  /* (elementwise passes run on all threads: at 256 spectrograms a serial pass over conv1's output is 0.9 G elements) */
  if (!dzdy) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : leak * x[i];
  } else {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? dzdy[i] : leak * dzdy[i];
  }
}

void orc_nnsigmoid(const float *x, size_t n, const float *dzdy, float *y) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) {
    float s = 1.f / (1.f + expf(-x[i]));
    y[i] = dzdy ? dzdy[i] * s * (1.f - s) : s;
  }
}

/* y = sum_i x_i  (dagnn.Sum with two inputs), optional fused relu */
void orc_sum2(const float *a, const float *b, size_t n, int relu, float *y) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) {
    float v = a[i] + b[i];
    y[i] = (relu && v < 0.f) ? 0.f : v;
  }
}

/* SE excite: y(h,w,c,n) = a(c,n) * x(h,w,c,n) [+ r(h,w,c,n)] [relu]   (mcnExtraLayers Scale / Axpy) */
void orc_scale_axpy(const float *x, size_t HW, size_t CN, const float *a, const float *r, int relu,
                    float *y) {
#pragma omp parallel for schedule(static)
  for (size_t j = 0; j < CN; ++j)
    for (size_t i = 0; i < HW; ++i) {
      float v = a[j] * x[HW * j + i] + (r ? r[HW * j + i] : 0.f);
      y[HW * j + i] = (relu && v < 0.f) ? 0.f : v;
    }
}

/* backward of y = a .* x (+ r): dx = a .* dy ; da(c,n) = sum_hw dy .* x ; dr = dy */
void orc_scale_backward(const float *x, size_t HW, size_t CN, const float *a, const float *dzdy,
                        float *dxo, float *dao) {
#pragma omp parallel for schedule(static)
  for (size_t j = 0; j < CN; ++j) {
    double s = 0;
    for (size_t i = 0; i < HW; ++i) {
      if (dxo) dxo[HW * j + i] = a[j] * dzdy[HW * j + i];
      s += (double)dzdy[HW * j + i] * (double)x[HW * j + i];
    }
    if (dao) dao[j] = (float)s;
  }
}

/* ------------------------------------------------------------------ */
/* softmax family (dim 3 = channels), losses                           */
/* ------------------------------------------------------------------ */
/* vl_nnsoftmaxt(X, 'temperature', T) along channels; x: HW x C x N viewed as [i + HW*(c + C*n)] */
void orc_nnsoftmaxt(const float *x, size_t HW, int C, int N, float T, float *y) {
  for (int n = 0; n < N; ++n)
    for (size_t i = 0; i < HW; ++i) {
      const float *p = x + i + HW * (size_t)C * n;
      float *q = y + i + HW * (size_t)C * n;
      double mx = -INFINITY, s = 0;
      for (int c = 0; c < C; ++c)
        if (p[HW * c] / T > mx) mx = p[HW * c] / T;
      for (int c = 0; c < C; ++c) s += exp((double)p[HW * c] / T - mx);
      for (int c = 0; c < C; ++c) q[HW * c] = (float)(exp((double)p[HW * c] / T - mx) / s);
    }
}

/*
 * mcnExtraLayers regression losses [EXT] -- dagnn.EuclideanLoss / dagnn.HuberLoss('sigma', 1) on
 * {'prediction', 'logitTarget', 'instanceWeights'} (emoVoxZoo.m:139-146; weights = ones(1,1,1,N),
 * getBatchEmoVoxCeleb.m:36-38).  X, T: E elements per sample, N samples; w: N weights (bsxfun over
 * the elements of a sample) or NULL.
 *   euclidean (kind 0): Y = 1/2 sum_n w_n sum_e (x - t)^2          dX = dzdy * w_n * (x - t)
 *   huber     (kind 1): d = x - t, s2 = sigma^2, linear <=> |d| > 1/s2
 *                       Y = sum_n w_n sum_e (linear ? |d| - 0.5/s2 : 0.5*s2*d^2)
 *                       dX = dzdy * w_n * (linear ? sign(d) : s2*d)
 */
void orc_nnregloss(const float *x, const float *t, size_t E, int N, int kind, float sigma, const float *w,
                   const float *dzdy, float *y /* 1 value or E*N grads */) {
  const double s2 = (double)sigma * sigma;
  double total = 0;
  for (int n = 0; n < N; ++n) {
    const double wn = w ? (double)w[n] : 1.0;
    for (size_t e = 0; e < E; ++e) {
      const double d = (double)x[E * n + e] - (double)t[E * n + e];
      const double a = fabs(d);
      const int lin = kind == 1 && a > 1.0 / s2;
      if (!dzdy) {
        total += wn * (kind == 0 ? 0.5 * d * d : (lin ? a - 0.5 / s2 : 0.5 * s2 * d * d));
      } else {
        const double g = kind == 0 ? d : (lin ? (d > 0 ? 1.0 : -1.0) : s2 * d);
        y[E * n + e] = (float)((double)dzdy[0] * wn * g);
      }
    }
  }
  if (!dzdy) y[0] = (float)total;
}

/* derivative of vl_nnsoftmaxt: y = softmax(x/T); dx = y .* (dzdy - sum_c dzdy.*y) / T
 * (MatConvNet vl_nnsoftmax backward [EXT]: Y .* bsxfun(@minus, dzdY, sum(dzdY .* Y, 3))) */
void orc_nnsoftmaxt_backward(const float *x, const float *dzdy, size_t HW, int C, int N, float T,
                             float *dx) {
  for (int n = 0; n < N; ++n)
    for (size_t i = 0; i < HW; ++i) {
      const float *p = x + i + HW * (size_t)C * n;
      const float *d = dzdy + i + HW * (size_t)C * n;
      float *q = dx + i + HW * (size_t)C * n;
      double mx = -INFINITY, s = 0, dot = 0;
      for (int c = 0; c < C; ++c)
        if (p[HW * c] / T > mx) mx = p[HW * c] / T;
      for (int c = 0; c < C; ++c) s += exp((double)p[HW * c] / T - mx);
      for (int c = 0; c < C; ++c) dot += (double)d[HW * c] * (exp((double)p[HW * c] / T - mx) / s);
      for (int c = 0; c < C; ++c) {
        double y = exp((double)p[HW * c] / T - mx) / s;
        q[HW * c] = (float)(y * ((double)d[HW * c] - dot) / T);
      }
    }
}

/*
 * vl_nnsoftmaxceloss(X, P, [DZDY], 'temperature', T, 'logitTargets', tf, 'instanceWeights', w)
 * X, P: 1 x 1 x C x N.   q = softmax(X/T);  p = logitTargets ? softmax(P/T) : P
 *   forward : Y = sum_n w_n * ( -sum_c p_c log q_c )          (summed over the batch)
 *   backward: dX = dzdy * w_n * (q - p * sum_c p_c... ) / T   -- with sum_c p_c = 1: (q - p)/T
 * (SURVEY Appendix A.6 decision: sum over batch, 1/T in the gradient, no T^2, log-sum-exp form)
 */
void orc_nnsoftmaxceloss(const float *x, const float *p, int C, int N, float T, int logit_targets,
                         const float *w, const float *dzdy, float *y /* 1 value or C*N grads */) {
  double total = 0;
  for (int n = 0; n < N; ++n) {
    const float *xn = x + (size_t)C * n, *pn = p + (size_t)C * n;
    double pt[64], mx = -INFINITY, s = 0, psum = 0;
    if (logit_targets) {
      double mp = -INFINITY, sp = 0;
      for (int c = 0; c < C; ++c)
        if (pn[c] / T > mp) mp = pn[c] / T;
      for (int c = 0; c < C; ++c) sp += exp((double)pn[c] / T - mp);
      for (int c = 0; c < C; ++c) pt[c] = exp((double)pn[c] / T - mp) / sp;
    } else
      for (int c = 0; c < C; ++c) pt[c] = pn[c];
    for (int c = 0; c < C; ++c) {
      if (xn[c] / T > mx) mx = xn[c] / T;
      psum += pt[c];
    }
    for (int c = 0; c < C; ++c) s += exp((double)xn[c] / T - mx);
    double lse = mx + log(s);
    double wn = w ? w[n] : 1.0;
    if (!dzdy) {
      double l = 0;
      for (int c = 0; c < C; ++c) l += pt[c] * (lse - (double)xn[c] / T);
      total += wn * l;
    } else {
      for (int c = 0; c < C; ++c) {
        double q = exp((double)xn[c] / T - lse);
        y[(size_t)C * n + c] = (float)((double)dzdy[0] * wn * (q * psum - pt[c]) / T);
      }
    }
  }
  if (!dzdy) y[0] = (float)total;
}

/* vl_nnloss(X, c, [DZDY], 'loss', 'softmaxlog' (0) | 'classerror' (1)); labels are 1-based */
void orc_nnloss(const float *x, const float *labels, int C, int N, int loss, const float *dzdy,
                float *y) {
  double total = 0;
  for (int n = 0; n < N; ++n) {
    const float *xn = x + (size_t)C * n;
    int c0 = (int)labels[n] - 1;
    double mx = -INFINITY, s = 0;
    int arg = 0;
    for (int c = 0; c < C; ++c)
      if (xn[c] > mx) {
        mx = xn[c];
        arg = c;
      }
    for (int c = 0; c < C; ++c) s += exp((double)xn[c] - mx);
    if (loss == 0) {
      if (!dzdy)
        total += mx + log(s) - (double)xn[c0];
      else
        for (int c = 0; c < C; ++c)
          y[(size_t)C * n + c] =
              (float)((double)dzdy[0] * (exp((double)xn[c] - mx) / s - (c == c0 ? 1.0 : 0.0)));
    } else {
      if (!dzdy)
        total += (arg != c0);
      else
        for (int c = 0; c < C; ++c) y[(size_t)C * n + c] = 0.f;
    }
  }
  if (!dzdy) y[0] = (float)total;
}

/* ------------------------------------------------------------------ */
/* cnn_train_dag accumulateGradients (solver = [], SGD with momentum)   */
/* ------------------------------------------------------------------ */
/* m <- mu*m - (wd*w + der/B) ;  w <- w + lr*m          (trainMethod 'gradient')
 * w <- (1-lr)*w + lr*der/nworkers                        (trainMethod 'average', BN moments) */
void orc_sgd_update(float *w, float *m, const float *der, size_t n, float lr, float momentum,
                    float wd, float batch) {
  for (size_t i = 0; i < n; ++i) {
    m[i] = momentum * m[i] - (wd * w[i] + der[i] / batch);
    w[i] = w[i] + lr * m[i];
  }
}

void orc_average_update(float *w, const float *der, size_t n, float lr, float nworkers) {
  for (size_t i = 0; i < n; ++i) w[i] = (1.f - lr) * w[i] + lr * (der[i] / nworkers);
}

/* ------------------------------------------------------------------ */
/* batch-provider maths (the parts of getBatchEmoVoxCeleb that are arithmetic) */
/* ------------------------------------------------------------------ */
/* getBatchEmoVoxCeleb.m:164-169: per-row (frequency) mean / UNBIASED std over time */
void orc_spec_rownorm(const float *spec, int H, int W, int N, float *out) {
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h) {
      const float *p = spec + h + (size_t)H * W * n;
      float *q = out + h + (size_t)H * W * n;
      double s = 0, ss = 0;
      for (int w = 0; w < W; ++w) s += p[(size_t)H * w];
      double mu = s / W;
      for (int w = 0; w < W; ++w) ss += (p[(size_t)H * w] - mu) * (p[(size_t)H * w] - mu);
      double sd = sqrt(ss / (W - 1));
      for (int w = 0; w < W; ++w) q[(size_t)H * w] = (float)(((double)p[(size_t)H * w] - mu) / sd);
    }
}

/* getBatchEmoVoxCeleb.m:210-214 */
int orc_time2idx(double t) {
  double v = t * 25.0 - 1.0;
  if (v < 0) v = 0;
  return (int)floor(v / 6.0) + 1;
}

/* getBatchEmoVoxCeleb.m:67-68 */
double orc_aud_samples(int width, double Tw_ms, double fs) {
  return (0.01 * width + 0.001 * Tw_ms - 0.001) * fs;
}

/* getBatchEmoVoxCeleb.m:179-188: aggregate F x E frame logits (column-major) over frames
 * [first,last] (1-based, inclusive) -> E values; agg 0 = max, 1 = mean */
void orc_aggregate_logits(const float *lg, int F, int E, int first, int last, int agg, float *out) {
  if (last > F) last = F;
  for (int e = 0; e < E; ++e) {
    double a = agg == 0 ? -INFINITY : 0;
    for (int fr = first - 1; fr < last; ++fr) {
      double v = lg[fr + (size_t)F * e];
      if (agg == 0)
        a = v > a ? v : a;
      else
        a += v;
    }
    out[e] = (float)(agg == 0 ? a : a / (last - first + 1));
  }
}

/* fetch_emovoxceleb_imdb.m:176-193: rgb2gray -> replicate x3 -> subtract averageImage(c).
 * rgb: H x W x 3 x N uint8-valued floats. rgb2gray weights 0.2989/0.5870/0.1140, rounded (uint8). */
/*
 * getImageBatch of fetch_emovoxceleb_imdb.m:152-193 from decoded frames: vl_imreadjpeg(..., 'CropSize',
 * 1/1.6, 'CropLocation', 'center', 'Interpolation', 'bilinear', 'Resize', imageSize) -> uint8 -> rgb2gray
 * -> x3 -> minus averageImage.  [EXT] vl_imreadjpeg's resampler is not in the reference tree; restated as:
 * crop window = centred (crop*Hin) x (crop*Win) box, output pixel (i, j) samples the window at its
 * pixel centre ((i + 0.5) * ch / Ho - 0.5 + h0), bilinear with edge clamping, value rounded to uint8.
 * src: Hin x Win x 3 x N, values 0..255 (float holding uint8); out: Ho x Wo x 3 x N.
 */
static float orc_bilin(const float *p, int H, int W, double y, double x) {
  if (y < 0) y = 0;
  if (x < 0) x = 0;
  if (y > H - 1) y = H - 1;
  if (x > W - 1) x = W - 1;
  int y0 = (int)floor(y), x0 = (int)floor(x);
  int y1 = y0 + 1 < H ? y0 + 1 : y0, x1 = x0 + 1 < W ? x0 + 1 : x0;
  double fy = y - y0, fx = x - x0;
  double v = (1 - fy) * ((1 - fx) * p[y0 + (size_t)H * x0] + fx * p[y0 + (size_t)H * x1]) +
             fy * ((1 - fx) * p[y1 + (size_t)H * x0] + fx * p[y1 + (size_t)H * x1]);
  v = floor(v + 0.5);
  return (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

void orc_crop_resize_face(const float *src, int Hin, int Win, int N, double crop, int Ho, int Wo,
                          const float *avg3, float *out) {
  const double ch = crop * Hin, cw = crop * Win;
  const double h0 = 0.5 * (Hin - ch), w0 = 0.5 * (Win - cw);
  const size_t HWi = (size_t)Hin * Win, HWo = (size_t)Ho * Wo;
  for (int n = 0; n < N; ++n)
    for (int j = 0; j < Wo; ++j)
      for (int i = 0; i < Ho; ++i) {
        double y = (i + 0.5) * ch / Ho - 0.5 + h0, x = (j + 0.5) * cw / Wo - 0.5 + w0;
        const float *p = src + HWi * 3 * n;
        float r = orc_bilin(p, Hin, Win, y, x), g = orc_bilin(p + HWi, Hin, Win, y, x),
              b = orc_bilin(p + 2 * HWi, Hin, Win, y, x);
        float gr = floorf(0.2989f * r + 0.5870f * g + 0.1140f * b + 0.5f);
        if (gr > 255.f) gr = 255.f;
        for (int c = 0; c < 3; ++c) out[i + (size_t)Ho * j + HWo * (c + 3 * (size_t)n)] = gr - avg3[c];
      }
}

void orc_normalize_face(const float *rgb, int H, int W, int N, const float *avg3, float *out) {
  size_t HW = (size_t)H * W;
  for (int n = 0; n < N; ++n)
    for (size_t i = 0; i < HW; ++i) {
      const float *p = rgb + i + HW * 3 * n;
      float gr = 0.2989f * p[0] + 0.5870f * p[HW] + 0.1140f * p[2 * HW];
      gr = floorf(gr + 0.5f);
      if (gr > 255.f) gr = 255.f;
      for (int c = 0; c < 3; ++c) out[i + HW * (c + 3 * (size_t)n)] = gr - avg3[c];
    }
}
