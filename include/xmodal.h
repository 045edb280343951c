/*
 * xmodal.h -- C ABI of libxmodal_hip.so, the MI355X (gfx950) replacement for the
 * MatConvNet / mcnExtraLayers operator MEX files that albanie/mcnCrossModalEmotions
 * drives on its distillation hot path.
 *
 * One entry point per MATLAB operator direction.  A MEX gateway (the mex/ sources,
 * INTEGRATION.md) or the ctypes mirror (mcncrossmodalemotions_amd/vl.py) maps the
 * MATLAB call 1:1 onto these.  The reference reaches them through dagnn.DagNN.eval:
 *     emoVoxCeleb/fetch_emovoxceleb_imdb.m:129   (teacher forward, test mode)
 *     external/compute_audio_feats.m:126         (student forward)
 *     emoVoxCeleb/run_distillation.m:170-182     (cnn_train_dag: fwd + bwd + update)
 *
 * Conventions
 *   - every tensor is MATLAB `single`, H x W x C x N, column-major (H fastest),
 *     densely packed, resident in device (HBM) memory; the caller owns all buffers,
 *     inputs are never written (MATLAB copy-on-write safe), outputs never alias inputs;
 *   - filters are FH x FW x FC x K (groups = C / FC), biases K x 1 (NULL = none);
 *   - pad = [top bottom left right], stride = [sy sx], dilate = [dy dx];
 *   - `stream` is a hipStream_t (NULL = the null stream, MatConvNet's behaviour);
 *     calls are asynchronous with respect to the host, ordered on that stream;
 *   - return value 0 = ok, otherwise an XM_E* code; xm_last_error() gives the text.
 *     Nothing throws, aborts or leaks across this boundary (a MEX gateway turns a
 *     non-zero code into mexErrMsgIdAndTxt).
 *   - re-entrant per device, not thread-safe (MATLAB calls MEX from one thread).
 *   - FINITE INPUTS.  Several kernels multiply a padded operand position by a zero weight instead of masking it
 *     (conv_stem3_kernel's eighth filter row, the zero-filled rows of the patch kernels, the ones / zero columns of the
 *     Gram route): an Inf or NaN just OUTSIDE an output's receptive field can turn that output into NaN there, where
 *     MatConvNet -- and this library's generic implicit-GEMM arm -- would stay finite.  Results are specified for finite
 *     X, F, DZDY; which arm runs is a function of (shape, tuning table, exec hint), see xm_set_exec_hint.
 */
#ifndef XMODAL_H
#define XMODAL_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
  XM_OK = 0,
  XM_EINVAL = 1,   /* bad shape / option combination (MATLAB side: "XM:invalidArgument") */
  XM_ENOMEM = 2,   /* workspace allocation failed */
  XM_EHIP = 3,     /* HIP runtime error (text has hipGetErrorString) */
  XM_ETOOBIG = 4,  /* a tensor has >= 2^31 elements */
  XM_ENOTSUP = 5   /* valid MatConvNet call that this build does not cover */
};

enum { XM_POOL_MAX = 0, XM_POOL_AVG = 1 };
enum { XM_LOSS_SOFTMAXLOG = 0, XM_LOSS_CLASSERROR = 1 };
enum { XM_AGG_MAX = 0, XM_AGG_MEAN = 1 };

/* fused-epilogue flags for xm_nnconv_forward_fused / xm_nnbnorm_forward_fused */
enum { XM_FUSE_RELU = 1, XM_BN_BATCH_MOMENTS = 2, XM_FUSE_SIGMOID = 4 };

/* ABI revision: 100 = round 1; 101 = xm_nnbnorm_relu_pool_backward gained `y_pool`, exchange entry points return
 * XM_EINVAL without a communicator; 102 = + xm_nnconv_forward_moments, xm_nnbnorm_backward_dxsum, xm_nnconv_forward_gated;
 * 103 = + xm_nnpool_global_avg_backward_accum; 104 = + xm_nnconv_backward_filter_bnrelupool, xm_nndropout_forward / _apply, xm_resample, xm_se_tail_backward_reduce / _apply, xm_se_squeeze_bn, xm_scale_axpy_bn; 105 = + xm_set_exec_hint / xm_get_exec_hint; 106 = + xm_nnconv_bnorm_relu_pool_forward, xm_stem_gram, xm_stem_gram_moments, xm_nnconv_backward_filter_bnrelupool_gram (additions never change the revision's meaning for older bindings).  A binding checks xm_version() >= the revision it was written against. */
int xm_version(void);
const char *xm_last_error(void);
/* Device memory for hosts that have no device-array type of their own (MATLAB's gpuArray is CUDA-only: on an
 * MI355X host the MEX layer keeps tensors in buffers obtained here and hands MATLAB an opaque handle -- mex/xm_mex.h,
 * mex/matlab/xmArray.m).  Plain hipMalloc / hipFree / hipMemcpy underneath; upload / download are synchronous with
 * respect to the host and ordered after everything enqueued on the null stream. */
int xm_device_alloc(void **ptr, size_t bytes);
int xm_device_free(void *ptr);
int xm_device_upload(void *dst_device, const void *src_host, size_t bytes);
int xm_device_download(void *dst_host, const void *src_device, size_t bytes);
int xm_device_synchronize(void);
/* Tile-configuration table of vl_nnconv ("find mode": the first time a (direction, geometry) is seen every tile
 * configuration is timed on the caller's stream and the winner kept; the device is drained first -- hipDeviceSynchronize --
 * so that the candidates run alone: do not meet a new shape while ANY stream of the process is being captured into a graph;
 * a stream that is itself being captured gets the analytic choice instead).  The table persists in a text file next to the
 * library (tune_gfx950.txt; $XM_TUNE_FILE overrides, XM_TUNE_FILE="" disables): loaded before the first lookup,
 * written by xm_tune_save.  With the shipped table the kernel and tile choice -- hence the summation order and the bits of
 * every result -- is a function of (shape, table, execution hint below) and of NOTHING else: the same in every process,
 * whatever it called before, on whatever streams; known shapes cost no timed launches on first use
 * (external/compute_audio_feats.m:116-136 walks ten width buckets).  XM_AUTOTUNE=0 uses the analytic model instead. */
/* Execution hint, an explicit statement of the host about HOW it calls (process-wide; default 0):
 *   XM_EXEC_SINGLE_STREAM  every operator call of this process arrives on ONE stream (MatConvNet's own sequence:
 *                          cnn_train_dag -> net.eval -> vl_nn* one after the other, run_distillation.m:170-182; what the
 *                          MEX binding does), so a kernel never shares the chip with another stream's kernel.  Kernels
 *                          that are faster alone but poor neighbours (conv_wgrad_patch_kernel: 48 KB of LDS x 3 blocks per
 *                          CU) become candidates.  A host that overlaps streams (dagnn.DagNN.wgradStream, bench.py's
 *                          default) leaves it 0.
 * Same results within the operator tolerance either way; another kernel is another summation order, so set the hint once,
 * before the first operator call, and identically on every worker.  Unknown bits: XM_EINVAL. */
enum { XM_EXEC_SINGLE_STREAM = 1 };
int xm_set_exec_hint(unsigned flags);
unsigned xm_get_exec_hint(void);
int xm_tune_load(const char *path);   /* NULL / "" = the default file; returns the number of entries read */
int xm_tune_save(const char *path);   /* merges with the file on disk, writes atomically */
int xm_tune_entries(int *total, int *unsaved);
/* optional: pre-size the internal scratch (split-K partials, filter transposes, tap tables). */
int xm_workspace_reserve(size_t bytes);
size_t xm_workspace_bytes(void);
/* Scratch is one grow-only buffer per stream; growing it frees the old buffer.  A HIP graph captured earlier still
 * holds the old address: reserve the largest need of a stream BEFORE capturing on it, and re-capture when the
 * generation counter (incremented by every growth, any stream) has moved since the capture. */
int xm_workspace_reserve_stream(size_t bytes, void *stream);
unsigned long long xm_workspace_generation(void);
/* output extent of conv / pool along one axis: floor((in + pa + pb - ((f-1)*d+1)) / s) + 1 */
int xm_out_size(int in, int pad_a, int pad_b, int f, int dilate, int stride);

/* ---- vl_nnconv  (MatConvNet matlab/vl_nnconv.m; dagnn.Conv at emoVoxZoo.m:118) -------------
 * Y = vl_nnconv(X, F, B, 'stride', [sy sx], 'pad', [t b l r], 'dilate', [dy dx]) */
int xm_nnconv_forward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                      int FC, int K, const float *b, float *y, int sy, int sx, int pt, int pb,
                      int pl, int pr, int dy, int dx, void *stream);
/* Extension (not a MatConvNet signature): same convolution with a fused epilogue
 *   y = act( (conv + b) .* scale_k + shift_k + residual ),  scale/shift/residual may be NULL;
 *   act = vl_nnrelu (XM_FUSE_RELU) or vl_nnsigmoid (XM_FUSE_SIGMOID: the SE gate fc2 -> sigmoid).
 * Used to fold test-mode vl_nnbnorm, dagnn.Sum and vl_nnrelu of the frozen teacher into the
 * producing convolution (fetch_emovoxceleb_imdb.m:107 sets dag.mode = 'test'). */
int xm_nnconv_forward_fused(const float *x, int H, int W, int C, int N, const float *f, int FH,
                            int FW, int FC, int K, const float *b, float *y, int sy, int sx,
                            int pt, int pb, int pl, int pr, int dy, int dx, const float *scale,
                            const float *shift, const float *residual, int flags, void *stream);
/* Extension: xm_nnconv_forward_fused with a per-(channel, sample) multiplier between the scale / shift and the residual:
 *   y = act( ((conv + b) .* scale_k + shift_k) .* gate(k, n) + residual ),   gate = 1 x 1 x K x N.
 * The SE block of senet50-ferplus (mcnExtraLayers dagnn.Axpy: out = a .* x + shortcut, teacher/ferPlusZoo.m:86-112;
 * fetch_emovoxceleb_imdb.m:98-136 runs it in test mode): x is the output of a bias-free 1 x 1 projection + test-mode
 * bnorm, i.e. affine in its input u, so the squeeze mean(x) = scale .* (F * mean(u)) + shift can be taken from the 4 x
 * narrower u BEFORE the projection runs; the gate a is then known when the projection's epilogue writes, and
 * relu(a .* x + shortcut) leaves the kernel directly -- x is never written, re-read for the squeeze, re-read and
 * re-written for the excite (three passes over the widest tensors of the network per block). */
int xm_nnconv_forward_gated(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                            int FC, int K, const float *b, float *y, int sy, int sx, int pt, int pb,
                            int pl, int pr, int dy, int dx, const float *scale, const float *shift,
                            const float *gate, const float *residual, int flags, void *stream);
/* Extension: Y = vl_nnconv(X, F, B, ...) AND the batch moments of Y that a train-mode vl_nnbnorm(Y, G, B) would
 * compute first -- moments_out (K x 2, column-major) = [mean_k, sqrt(var_k + epsilon)] over H x W x N, biased
 * variance (emoVoxZoo.m:118-123: every dagnn.Conv of the student is followed by dagnn.BatchNorm).  Hand them to
 * xm_nnbnorm_forward_fused / xm_nnbnorm_relu_pool_forward as `moments_in` and to the backward calls with
 * XM_BN_BATCH_MOMENTS / train = 1: same results as letting vl_nnbnorm compute them, minus one pass over Y (462 MB
 * for the student's first layer at 32 spectrograms) -- the per-channel sums ride in the GEMM epilogue. */
int xm_nnconv_forward_moments(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                              int FC, int K, const float *b, float *y, int sy, int sx, int pt, int pb,
                              int pl, int pr, int dy, int dx, float epsilon, float *moments_out, void *stream);
/* [DX, DF, DB] = vl_nnconv(X, F, B, DZDY, ...); dx_out / df_out / db_out may be NULL
 * (= 'NoDerData' / 'NoDerFilters' / 'NoDerBiases'). */
int xm_nnconv_backward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                       int FC, int K, const float *dzdy, float *dx_out, float *df_out,
                       float *db_out, int sy, int sx, int pt, int pb, int pl, int pr, int dy,
                       int dx, void *stream);
/* Extension: the DZDX GEMM reads the filter bank transposed (and split by stride parity); building that copy is
 * ~20 us per layer on the critical path of the backward pass.  xm_nnconv_prepare_backward builds it ahead of time
 * into a persistent buffer -- e.g. on a side stream while the forward pass of the same step runs -- and a later
 * xm_nnconv_backward with the same F pointer and geometry uses it (waiting on the device for it if it was built on
 * another stream).  Valid until the parameters change: xm_sgd_update / xm_average_update invalidate every prepared
 * operand; a host that updates parameters by other means calls xm_params_changed(). */
int xm_nnconv_prepare_backward(int H, int W, int C, int N, const float *f, int FH, int FW, int FC, int K,
                               int sy, int sx, int pt, int pb, int pl, int pr, int dy, int dx, void *stream);
int xm_params_changed(void);
/* Extension: DX = dgrad + dx_accum.  dagnn sums the derivatives that reach a variable from several
 * consumers (a ResNet block input: shortcut + branch2a); the sum rides in the dgrad epilogue instead of a
 * separate pass.  dx_accum has the size of X, must not alias dx_out, NULL = plain backward. */
int xm_nnconv_backward_accum(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                             int FC, int K, const float *dzdy, float *dx_out, float *df_out,
                             float *db_out, int sy, int sx, int pt, int pb, int pl, int pr, int dy,
                             int dx, const float *dx_accum, void *stream);
/* Extension: [DZDF, DZDB] of a FIRST-layer convolution together with [DG, DB] of the vl_nnbnorm behind it, when the
 * convolution's output Y feeds vl_nnbnorm -> vl_nnrelu -> vl_nnpool('max') and nothing else (the student's
 * conv1 -> bn1 -> relu1 -> pool1, emoVoxCeleb/emoVoxZoo.m:50-62) and the convolution's own input
 * needs no derivative.  Equivalent to
 *     xm_nnbnorm_relu_pool_backward(Y, ..., argmax, y_pool, dzdy_pool, DX, DG, DB, NULL)   then
 *     xm_nnconv_backward(X, ..., DZDY = DX, NULL, DZDF, DZDB)
 * but DX -- the widest tensor of the student's backward pass, 462 MB at 32 spectrograms -- is never written: the
 * filter-derivative kernel rebuilds it per element from the pooled derivative and the routing table.  Same decisions
 * (ReLU gates, routing) and the same per-element formula; the normalisation's constants are applied in fp32 with a
 * two-float mean instead of fp64 (1 ulp of DX).  Arguments: X / geometry of the convolution as for xm_nnconv_backward
 * (F itself is not needed), Y = its output, then the arguments of xm_nnbnorm_relu_pool_backward.  dzdb_out, dg_out,
 * db_out may be NULL.  Returns XM_ENOTSUP when the shapes are outside what the fused kernel covers (single input
 * channel, <= 96 filters, <= 8 x 7 taps, 3 x 3 / stride-2 unpadded pooling, y_pool given): make the two calls then. */
int xm_nnconv_backward_filter_bnrelupool(const float *x, int H, int W, int C, int N, int FH, int FW, int FC, int K,
                                         int sy, int sx, int pt, int pb, int pl, int pr, int dy, int dx,
                                         const float *y, const float *bn_g, const float *bn_b, const float *moments,
                                         int train, int ph, int pw, int psy, int psx, int ppt, int ppb, int ppl,
                                         int ppr, const unsigned char *argmax, const float *y_pool,
                                         const float *dzdy_pool, float *dzdf_out, float *dzdb_out, float *dg_out,
                                         float *db_out, void *stream);
/* Extension (ABI revision 106): the same derivatives WITHOUT a pass over the convolution's output.  A single-channel
 * first layer is Y[m][p] = sum_t F~[m][t] P~[p][t] over the im2col patches P~ of X (+ a column of ones for the bias), so
 * everything vl_nnbnorm's train-mode derivative needs from Y is a contraction with the (FH FW + 1)^2 Gram matrix
 * G = P~' P~ of the INPUT (xm_stem_gram, fp64 [64][64], row-major; row / column t = u + FH v, t = FH FW: the ones):
 *     A = DZ P~ (DZ: the pooled derivative routed through `argmax`, masked by y_pool > 0; built on chip),
 *     DG = (F~.A - mu A[:, ones]) / sigma, DB = A[:, ones],
 *     [DZDF, DZDB] = g/sigma (A - DB/P G[ones, :] - DG/(sigma P) ((F~ G) - mu G[ones, :]))          (train)
 * with [mu, sigma] = `moments` (the forward call's).  F and B (may be NULL) take Y's place in the argument list; `gram`
 * NULL = computed here; y_pool NULL = the table marks closed windows itself (code 255).  Same routing / ReLU decisions as the composition; the sums are another summation order
 * (fp32 MFMA chains per wave, fp64 across waves and in the closed form).  xm_stem_gram_moments gives the batch moments
 * of Y from G (mean = F~ G[:, ones] / P, E[y^2] = F~ G F~' / P; fp64), i.e. vl_nnbnorm's MOMENTS output for Y without Y.
 * XM_ENOTSUP outside: one input channel, <= 96 filters, 16 ... 63 taps in <= 8 x 7, stride 1 / 2, size(X,1) % 4 == 0,
 * 3 x 3 / stride-2 unpadded max pooling with >= 4 window rows. */
/* Extension (106): Y_POOL, ARGMAX, MOMENTS of
 *     vl_nnpool(vl_nnrelu(vl_nnbnorm(vl_nnconv(X, F, B), G, BB, 'epsilon', e)), [3 3], 'stride', 2, 'method', 'max')
 * for a single-channel first layer (7 x 7 / stride 2, K % 8 == 0) in ONE kernel: the convolution's output -- 3.7 GB at 256
 * spectrograms -- is never written.  Train mode (moments_in NULL): `gram` (caller-owned fp64 [64][64]) receives
 * xm_stem_gram(X), the batch moments come from it (xm_stem_gram_moments) before the kernel starts, and MOMENTS_OUT is
 * vl_nnbnorm's third output; test mode: moments_in as given.  The normalisation is folded into the MFMA operand
 * (g/sigma F, constant term on a spare reduction index), i.e. another rounding order than bnorm(conv(x)): 1e-6 relative.
 * ARGMAX: code = dh + 3 dw of the FIRST maximum in column-major scan order, or 255 where the window's maximum did not
 * pass the ReLU (y_pool == 0; the derivative of such a window is zero wherever it is routed) -- the form
 * xm_nnconv_backward_filter_bnrelupool_gram(..., y_pool = NULL, ...) takes. */
int xm_nnconv_bnorm_relu_pool_forward(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW, int FC,
                                      int K, const float *bias, int sy, int sx, int pt, int pb, int pl, int pr, int dy, int dx,
                                      const float *bn_g, const float *bn_b, float epsilon, const float *moments_in, int ph,
                                      int pw, int psy, int psx, int ppt, int ppb, int ppl, int ppr, double *gram,
                                      float *y_pool, unsigned char *argmax, float *moments_out, void *stream);
int xm_stem_gram(const float *x, int H, int W, int N, int FH, int FW, int sy, int sx, int pt, int pb, int pl, int pr,
                 double *gram, void *stream);
int xm_stem_gram_moments(const double *gram, const float *f, const float *b, int FH, int FW, int K, float epsilon,
                         float *moments_out, void *stream);
int xm_nnconv_backward_filter_bnrelupool_gram(const float *x, int H, int W, int C, int N, const float *f, int FH, int FW,
                                              int FC, int K, const float *bias, int sy, int sx, int pt, int pb, int pl,
                                              int pr, int dy, int dx, const float *bn_g, const float *moments, int train,
                                              int ph, int pw, int psy, int psx, int ppt, int ppb, int ppl, int ppr,
                                              const unsigned char *argmax, const float *y_pool, const float *dzdy_pool,
                                              const double *gram, float *df_out, float *dbias_out, float *dg_out,
                                              float *db_out, void *stream);

/* ---- vl_nnpool  (matlab/vl_nnpool.m; pool6 resized at emoVoxZoo.m:256-269) ------------------
 * Y = vl_nnpool(X, [ph pw], 'stride', .., 'pad', .., 'method', 'max'|'avg') */
int xm_nnpool_forward(const float *x, int H, int W, int C, int N, int ph, int pw, int sy, int sx,
                      int pt, int pb, int pl, int pr, int method, float *y, void *stream);
/* DX = vl_nnpool(X, [ph pw], DZDY, ...) */
int xm_nnpool_backward(const float *x, int H, int W, int C, int N, int ph, int pw, int sy, int sx,
                       int pt, int pb, int pl, int pr, int method, const float *dzdy, float *dx_out,
                       void *stream);

/* Extension (max pooling): the forward pass also records, per output, the position of the first
 * maximum inside its window (one byte: dh + ph * dw); the backward pass then routes DZDY from that
 * table without re-reading X.  Results are identical to xm_nnpool_forward / xm_nnpool_backward. */
int xm_nnpool_forward_argmax(const float *x, int H, int W, int C, int N, int ph, int pw, int sy,
                             int sx, int pt, int pb, int pl, int pr, float *y, unsigned char *argmax,
                             void *stream);
int xm_nnpool_backward_argmax(const unsigned char *argmax, int H, int W, int C, int N, int ph, int pw,
                              int sy, int sx, int pt, int pb, int pl, int pr, const float *dzdy,
                              float *dx_out, void *stream);
/* Global average pooling (mcnExtraLayers dagnn.GlobalPooling, the SE squeeze) backward where X has a second consumer
 * that already left its derivative (the SE excite: dagnn accumulates derivatives at forks): dx = accum + dzdy / (H W)
 * in one pass; bit-identical to xm_nnpool_backward followed by the sum. */
int xm_nnpool_global_avg_backward_accum(const float *dzdy, const float *accum, float *dx_out, int H, int W, int C, int N,
                                        void *stream);

/* ---- vl_nnbnorm  (matlab/vl_nnbnorm.m) -----------------------------------------------------
 * Y = vl_nnbnorm(X, G, B, 'epsilon', e [, 'moments', M]); M is C x 2 = [mean, sqrt(var+e)].
 * moments_in == NULL: train mode (batch moments, biased variance); they are written to
 * moments_out when it is non-NULL.  moments_in != NULL: test mode. */
int xm_nnbnorm_forward(const float *x, int H, int W, int C, int N, const float *g, const float *b,
                       float epsilon, const float *moments_in, float *y, float *moments_out,
                       void *stream);
/* Extension: vl_nnbnorm followed by vl_nnrelu in one pass (flags = XM_FUSE_RELU). */
int xm_nnbnorm_forward_fused(const float *x, int H, int W, int C, int N, const float *g,
                             const float *b, float epsilon, const float *moments_in, float *y,
                             float *moments_out, int flags, void *stream);
/* [DX, DG, DB, MOMENTS] = vl_nnbnorm(X, G, B, DZDY, ...) */
int xm_nnbnorm_backward(const float *x, int H, int W, int C, int N, const float *g, const float *b,
                        const float *dzdy, float epsilon, const float *moments_in, float *dx_out,
                        float *dg_out, float *db_out, float *moments_out, void *stream);
/* Extension: backward through relu(bnorm(x)): `y` is the fused forward output; dzdy is masked
 * by (y > 0) before the vl_nnbnorm backward formulas (flags & XM_FUSE_RELU).
 * flags & XM_BN_BATCH_MOMENTS: moments_in holds the BATCH moments the forward call returned for this
 * very X (train mode): they are not recomputed (one pass over X less), the train-mode formulas apply. */
int xm_nnbnorm_backward_fused(const float *x, const float *y, int H, int W, int C, int N,
                              const float *g, const float *b, const float *dzdy, float epsilon,
                              const float *moments_in, float *dx_out, float *dg_out, float *db_out,
                              float *moments_out, int flags, void *stream);

/* Extension: xm_nnbnorm_backward_fused that also returns dxsum_out (C x 1) = sum of DX over H x W x N per channel.  When
 * X is the output of a vl_nnconv with biases, that sum IS the convolution's DZDB (vl_nnconv: dzdb = sum of dzdy), so the
 * host skips the bias half of xm_nnconv_backward (db_out = NULL) -- one pass over DX and three launches less per layer
 * (emoVoxZoo.m:118-123: conv -> bnorm -> relu for every layer of the student).  dx_out must not be NULL. */
int xm_nnbnorm_backward_dxsum(const float *x, const float *y, int H, int W, int C, int N,
                              const float *g, const float *b, const float *dzdy, float epsilon,
                              const float *moments_in, float *dx_out, float *dg_out, float *db_out,
                              float *moments_out, float *dxsum_out, int flags, void *stream);

/* Extension: vl_nnbnorm -> vl_nnrelu -> vl_nnpool('max') as one fused operator pair.  The
 * normalised / rectified tensor is never materialised: forward = moments (train mode) + one pass
 * that normalises, rectifies and pools while recording the argmax table; backward = two passes
 * over X that rebuild the routed, ReLU-masked derivative on the fly.  Results are identical to the
 * three separate operators.  `moments` (backward) is what the forward returned; train != 0 applies
 * the batch-statistics terms of vl_nnbnorm's backward (train mode), 0 treats them as constants.
 * y_pool (backward, optional): the forward's pooled output; with it the two per-channel sums are formed from
 * the pooled tensors alone (x is read once instead of twice).
 * dxsum_out (optional, C floats): per-channel sum of DX = the DZDB of a vl_nnconv that produced X. */
int xm_nnbnorm_relu_pool_forward(const float *x, int H, int W, int C, int N, const float *g,
                                 const float *b, float epsilon, const float *moments_in, int ph, int pw,
                                 int sy, int sx, int pt, int pb, int pl, int pr, float *y_pool,
                                 unsigned char *argmax, float *moments_out, void *stream);
int xm_nnbnorm_relu_pool_backward(const float *x, int H, int W, int C, int N, const float *g,
                                  const float *b, const float *moments, int train, int ph, int pw,
                                  int sy, int sx, int pt, int pb, int pl, int pr,
                                  const unsigned char *argmax, const float *y_pool, const float *dzdy_pool,
                                  float *dx_out, float *dg_out, float *db_out, float *dxsum_out, void *stream);

/* ---- elementwise: vl_nnrelu, vl_nnsigmoid, dagnn.Sum, mcnExtraLayers Scale/Axpy ------------
 * dzdy == NULL: forward; otherwise y receives DZDX. */
int xm_nnrelu(const float *x, size_t n, float leak, const float *dzdy, float *y, void *stream);
int xm_nnsigmoid(const float *x, size_t n, const float *dzdy, float *y, void *stream);
/* y = a + b, optional fused relu (flags = XM_FUSE_RELU) -- dagnn.Sum (+ vl_nnrelu) */
int xm_sum2(const float *a, const float *b, size_t n, int flags, float *y, void *stream);
/* SE excite: y(:,:,c,n) = a(c,n) * x(:,:,c,n) [+ r(:,:,c,n)] [relu]; a is 1 x 1 x C x N */
int xm_scale_axpy(const float *x, int HW, int CN, const float *a, const float *r, int flags,
                  float *y, void *stream);
/* backward of y = a .* x: dx = a .* dzdy (NULL to skip), da(c,n) = sum_hw dzdy .* x */
int xm_scale_backward(const float *x, int HW, int CN, const float *a, const float *dzdy,
                      float *dx_out, float *da_out, void *stream);

/* ---- vl_nndropout (dagnn.DropOut, inserted behind fc6 / fc7 by emoVoxZoo.m:116-135,272-277 when opts.dropout > 0) ----
 * [Y, MASK] = vl_nndropout(X, 'rate', r):  MASK = (u >= r) / (1 - r), u ~ U[0, 1), Y = MASK .* X.  MATLAB's random
 * stream cannot be reproduced; the mask comes from a stateless Philox4x32-10 stream: key = seed, counter = element
 * index / 4 + offset (four elements per counter).  The host advances `offset` by ceil(n / 4) per call so that masks
 * never repeat within a run.  mask_out (size of X, single, values 0 or 1 / (1 - r)) may be NULL. */
int xm_nndropout_forward(const float *x, size_t n, float rate, unsigned long long seed, unsigned long long offset,
                         float *y, float *mask_out, void *stream);
/* Y = X .* MASK: DZDX = vl_nndropout(X, DZDY, 'mask', MASK), and the forward call with a given mask */
int xm_nndropout_apply(const float *x, const float *mask, size_t n, float *y, void *stream);

/* Extension: backward of the TAIL of an SE bottleneck block in training mode (the trainable SE-ResNet-50 teacher,
 * teacher/ferplus_baselines.m:140-141; BASELINE config 5):
 *     U -> vl_nnbnorm(G, B) -> X;   GP = mean_hw(X) -> fc1 -> relu -> fc2 -> sigmoid = A;   Y = vl_nnrelu(A .* X + S)
 * Two calls replace vl_nnrelu / Axpy / GlobalPooling / vl_nnbnorm backward (13 passes over block-sized tensors -> 8):
 *   xm_se_tail_backward_reduce  reads Y, DZDY, U;  leaves DA = dz/dA (1 x 1 x C x N, what the gate's backward consumes)
 *                               and three sums per (channel, sample) plane in `plane_sums` (caller-owned, 3 C N doubles);
 *   xm_se_tail_backward_apply   after the gate's backward has produced DGP = dz/dGP (1 x 1 x C x N): writes
 *                               DZ = [Y > 0] .* DZDY (the derivative of the shortcut S) and DU = dz/dU, and the bnorm's
 *                               DG / DB.  `moments` are the batch moments of U the forward pass used (train = 1) or
 *                               the stored ones (train = 0).  Same formulas as the separate operators (the bnorm's
 *                               per-element expression in fp64); X is recomputed from U where it is needed. */
/* ... and its forward without materialising X: GP = mean_hw(vl_nnbnorm(U)) and Y = [relu](A .* vl_nnbnorm(U) + S) straight
 * from U (the bnorm's own per-element expression: the same bits as vl_nnbnorm followed by vl_nnpool / Axpy), `moments`
 * as above.  With the two backward calls X is never needed, so the bnorm's apply pass disappears. */
int xm_se_squeeze_bn(const float *u, int H, int W, int C, int N, const float *g, const float *b, const float *moments,
                     float *gp_out, void *stream);
int xm_scale_axpy_bn(const float *u, int H, int W, int C, int N, const float *a, const float *r, const float *g,
                     const float *b, const float *moments, int flags, float *y, void *stream);
int xm_se_tail_backward_reduce(const float *y, const float *dzdy, const float *u, int H, int W, int C, int N,
                               const float *g, const float *b, const float *moments, float *da_out, double *plane_sums,
                               void *stream);
int xm_se_tail_backward_apply(const float *y, const float *dzdy, const float *u, int H, int W, int C, int N,
                              const float *gate, const float *dgp, const float *g, const float *moments, int train,
                              const double *plane_sums, float *dz_out, float *du_out, float *dg_out, float *db_out,
                              void *stream);

/* ---- losses --------------------------------------------------------------------------------
 * vl_nnsoftmaxt(X, 'temperature', T): softmax(X / T) along dim 3; X is HW x C x N */
int xm_nnsoftmaxt(const float *x, int HW, int C, int N, float temperature, float *y, void *stream);
/* DZDX = vl_nnsoftmax(X, DZDY) [EXT, SURVEY 8b] generalised to a temperature:
 * y = softmax(x/T) along C; dx = y .* (dzdy - sum_c(dzdy .* y)) / T.  Same indexing as above. */
int xm_nnsoftmaxt_backward(const float *x, const float *dzdy, int HW, int C, int N, float temperature,
                           float *dx, void *stream);
/* vl_nnsoftmaxceloss(X, P [, DZDY], 'temperature', T, 'logitTargets', tf, 'instanceWeights', w)
 * (dagnn.SoftmaxCELoss at emoVoxZoo.m:152, ferPlusZoo.m:244).  X, P: 1 x 1 x C x N, C <= 64.
 * forward (dzdy == NULL): y[0] = sum_n w_n * CE(softmax(P/T) or P, softmax(X/T));
 * backward: y (C*N floats) = dzdy[0] * w_n * (softmax(X/T) * sum(p) - p) / T.
 * dzdy is a DEVICE pointer to one float. */
int xm_nnsoftmaxceloss(const float *x, const float *p, int C, int N, float temperature,
                       int logit_targets, const float *instance_weights, const float *dzdy,
                       float *y, void *stream);
/* vl_nneuclideanloss(X, T [, DZDY], 'instanceWeights', w)  /  vl_nnhuberloss(X, T [, DZDY], 'sigma', s,
 * 'instanceWeights', w) -- mcnExtraLayers; dagnn.EuclideanLoss / dagnn.HuberLoss at emoVoxZoo.m:139-146.
 * X, T: E elements per sample x N samples; w: N weights or NULL (broadcast over a sample's elements).
 * forward (dzdy == NULL): y[0] = sum_n w_n sum_e l(x - t), l(d) = d^2/2 (euclidean) or smooth-L1 with
 * knee 1/sigma^2 (huber); backward: y (E*N floats) = dzdy[0] * w_n * l'(x - t).  dzdy: DEVICE pointer. */
#define XM_REGLOSS_EUCLIDEAN 0
#define XM_REGLOSS_HUBER 1
int xm_nnregloss(const float *x, const float *t, int E, int N, int kind, float sigma,
                 const float *instance_weights, const float *dzdy, float *y, void *stream);
/* vl_nnloss(X, c [, DZDY], 'loss', 'softmaxlog'|'classerror'); labels are 1-based floats */
int xm_nnloss(const float *x, const float *labels, int C, int N, int loss, const float *dzdy,
              float *y, void *stream);

/* ---- cnn_train_dag accumulateGradients + ParameterServer (run_distillation.m:88,170-182) ----
 * trainMethod 'gradient':  m <- momentum*m - (wd*w + der/batch);  w <- w + lr*m */
int xm_sgd_update(float *w, float *m, const float *der, size_t n, float lr, float momentum,
                  float weight_decay, float batch, void *stream);
/* trainMethod 'average' (BN moments):  w <- (1-lr)*w + lr*der/denom.
 * MatConvNet [EXT]: dagnn.BatchNorm hands back moments * (its worker's batch size), the workers' values are summed
 * and accumulateGradients divides by the GLOBAL batch size -- workers with ragged shards are weighted by their
 * sample counts.  Single worker: der = the batch moments, denom = 1. */
int xm_average_update(float *w, const float *der, size_t n, float lr, float denom,
                      void *stream);
/* x <- a * x (the worker-batch-size weighting of the moments before the exchange) */
int xm_scale_f32(float *x, size_t n, float a, void *stream);
/* ParameterServer.{start,push,sync,pull}: sum of `buf` over all workers, in place, via RCCL.
 * xm_comm_init takes the 128-byte ncclUniqueId produced by xm_comm_unique_id on rank 0 and
 * distributed by the host (MATLAB labBroadcast / torch.distributed broadcast).
 * CALL ORDER: create the communicator FIRST, before the first operator call / buffer upload of the process (in
 * cnn_train_dag terms: in startup, before net.move('gpu')).  Measured on ROCm 7.0 / RCCL 2.26: a communicator created
 * after the operator streams were in use makes every later step 12 % slower (the backward pass 0.8 ms longer at 32
 * pairs, even if no collective is ever issued); created first, the step time is that of a process without it. */
int xm_comm_unique_id(void *id128);
int xm_comm_init(const void *id128, int rank, int world);
int xm_allreduce_sum_f32(float *buf, size_t n, void *stream);   /* blocking-in-stream-order form: runs on `stream` */
/* Overlapped form (what cnn_train_dag's push-as-derivatives-become-ready / sync-before-update does with 'tmove'):
 *   xm_parserv_push(buf, n, producer)  after the kernels that wrote buf[0..n) were enqueued on `producer`: the sum over
 *                                      all workers starts when they finish and runs on the communicator's own HIP
 *                                      stream -- the producer stream goes on with the rest of the backward pass;
 *   xm_parserv_sync(consumer)          once per minibatch before accumulateGradients: `consumer` waits (on the
 *                                      device, not the host) for every push since the previous sync.
 * Every element must be pushed exactly once per minibatch.  After xm_comm_init(.., world == 1) both are no-ops;
 * WITHOUT a successful xm_comm_init they (and xm_allreduce_sum_f32) return XM_EINVAL -- a worker must never go on
 * training with derivatives that were silently not exchanged. */
int xm_parserv_push(float *buf, size_t n, void *producer_stream);
int xm_parserv_sync(void *consumer_stream);
/* ncclCommCount of the communicator (1 when no communicator exists): proof of the worker count */
int xm_comm_count(int *ranks);
int xm_comm_destroy(void);

/* ---- batch-provider arithmetic (device side of getBatchEmoVoxCeleb / getImageBatch) ---------
 * getBatchEmoVoxCeleb.m:164-169: per-frequency-row mean / unbiased std over time; H x W x 1 x N */
int xm_spec_rownorm(const float *spec, int H, int W, int N, float *out, void *stream);
/* z = resample(zo, p, q) of the speed-perturbation branch (getBatchEmoVoxCeleb.m:102-108, transformation 'S'; MATLAB
 * Signal Processing Toolbox [EXT]): y[j] = sum_k h[(j + delay) q - k p] x[k], j = 0 .. Ly - 1 -- upfirdn(x, h, p, q)
 * with the filter delay removed.  The host designs h (Kaiser-windowed ideal low-pass, batch.resample_design restates
 * the toolbox's recipe) and passes it with p, q already reduced by their gcd. */
int xm_resample(const float *x, int Lx, const float *h, int Lh, int p, int q, int delay, float *y, int Ly, void *stream);
/* |STFT| from the output of the framing convolution (runSpec of getBatchEmoVoxCeleb.m:162 [EXT VGGVox]):
 * reim is 1 x Wo x 2B x N (channel b = Re of bin b, channel B+b = Im), out is B x Wo x 1 x N with
 * out(b, j, 1, n) = sqrt(Re^2 + Im^2).  The framing/windowing/pre-emphasis/DFT itself is one
 * xm_nnconv_forward with a 1 x (Nw+1) x 1 x 2B filter bank and stride [1 Ns] (batch.runSpec). */
int xm_spec_magnitude(const float *reim, int Wo, int B, int N, float *out, void *stream);
/* getBatchEmoVoxCeleb.m:145-158,179-188: for sample n aggregate frame logits (F_total x E,
 * column-major, all wavs concatenated) over rows [first[n], last[n]] (1-based, inclusive)
 * -> out 1 x 1 x E x N and maxLabel (1-based argmax, getBatchEmoVoxCeleb.m:32) */
int xm_aggregate_logits(const float *frame_logits, int F_total, int E, const int *first,
                        const int *last, int N, int agg, float *out, float *max_label,
                        void *stream);
/* getBatchEmoVoxCeleb.m:32: [~, maxLabel] = max(lgo, [], 3); x is 1 x 1 x C x N, labels 1-based */
int xm_max_label(const float *x, int C, int N, float *labels, void *stream);
/* mcnExtraLayers dagnn.ErrorStats bookkeeping (emoVoxZoo.m:165-169, read by extractStats,
 * run_distillation.m:186-207): for every sample n with label c = labels[n] (1-based):
 * population[c-1] += 1, correct[c-1] += (argmax_c x(:, n) == c).  ACCUMULATES into the two
 * C-element device arrays (zero them at the start of an epoch); first maximum wins ties. */
int xm_class_stats(const float *x, const float *labels, int C, int N, float *correct,
                   float *population, void *stream);
/* fetch_emovoxceleb_imdb.m:176-193: rgb2gray -> replicate x3 -> minus averageImage(c).
 * avg3 is a HOST pointer to the three per-channel means (meta.normalization.averageImage). */
int xm_normalize_face(const float *rgb, int H, int W, int N, const float *avg3, float *out,
                      void *stream);
/* getImageBatch from decoded frames (fetch_emovoxceleb_imdb.m:152-193): centred crop of relative size
 * `crop` (1/1.6), bilinear resample to Ho x Wo (pixel-centre aligned, edge clamped, rounded to uint8 as
 * `uint8(data{1})` does), rgb2gray, replicate x3, minus averageImage -- one pass.
 * src: Hin x Win x 3 x N with values 0..255 (single holding the decoded uint8); avg3: HOST pointer. */
int xm_crop_resize_face(const float *src, int Hin, int Win, int N, float crop, int Ho, int Wo,
                        const float *avg3, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XMODAL_H */
