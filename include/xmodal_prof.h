/*
 * xmodal_prof.h -- measurement hooks of libxmodal_hip.so (NOT part of the drop-in operator ABI).
 * bench.py uses them to time each convolution kernel instantiation with HIP events recorded on
 * the stream the kernel is launched on, and to attach the algorithmic FLOPs of every launch
 * (2 * M * pixels * taps, un-padded), so that roofline.achieved is measured live.
 */
/*
 * Environment the library reads (nothing else): every name is read once.
 *   tuning table   XM_TUNE_FILE (path; "" disables), XM_AUTOTUNE=0 (analytic model only), XM_TUNE_REPS, XM_TUNE_VERBOSE,
 *                  XM_HALO_MARGIN (a challenger kernel must win by this fraction, default 0.04), XM_W8_MIN_TILES (the
 *                  eight-wave configuration is a candidate for launches of at least this many 128 x 128 tiles, default 1024),
 *                  XM_WGRAD_PATCH_SLOTS (conv_wgrad_patch_kernel's grid: blocks per round, default 768; WHETHER that kernel
 *                  is a candidate is the host's xm_set_exec_hint(XM_EXEC_SINGLE_STREAM), not an environment setting)
 *                  XM_WGRAD_PATCH_MIN_STAGES (conv_wgrad_patch_kernel without the one-stream hint: launches of at least this
 *                  many output columns, default 4096), XM_DGRAD_S2_MIN_BLOCKS (conv_dgrad_s2_kernel: launches of at least this
 *                  many blocks, default 6 x 768), XM_STEM3_WIDE=0 (conv_stem3_kernel with 128- instead of 256-pixel block tiles)
 *   workspace log  XM_WS_VERBOSE
 *   kernel-path selectors (csrc/xm_common.h `enum Path`, read in ONE place, csrc/context.cpp): XM_NO_HYBRID, XM_NO_HALO,
 *                  XM_NO_SKINNY, XM_NO_SKINNY4, XM_NO_STEM, XM_NO_STEM_WGRAD, XM_NO_DMA, XM_NO_FUSED_STATS,
 *                  XM_DGRAD_MERGE, XM_NO_FAST_TRANSPOSE, XM_NO_POOL_LDS, XM_NO_POOL_PATCH, XM_NO_POOL_POOLED,
 *                  XM_NO_W8, XM_NO_WGRAD_PATCH, XM_NO_WGRAD_PATCH_S2, XM_NO_DGRAD_S2, XM_NO_STEM3.
 *                  Each chooses between two complete, parity-tested implementations of the same operator (the operator tests
 *                  force both arms through xm_debug_set; tests/test_gpu_path_switches.py runs whole passes
 *                  with every selector set, in fresh processes, against the default; profiles/ holds the A/B lines);
 *                  none changes what is computed.
 */
#ifndef XMODAL_PROF_H
#define XMODAL_PROF_H
#ifdef __cplusplus
extern "C" {
#endif
/* on != 0: start recording (drops previous records); on == 0: stop recording, keep records */
int xm_prof_enable(int on);
/* after the stream has been synchronised: per-kernel totals; returns #distinct kernels */
int xm_prof_collect(int cap, int *keys, double *total_ms, double *total_flops, long long *launches);
/* algorithmic HBM bytes (x + f + y [+ residual], every operand once) of the recorded launches, per kernel */
int xm_prof_collect_bytes(int cap, int *keys, double *total_bytes);
/* name of a key, identical to the kernel name rocprofv3 --kernel-trace prints (sans namespace) */
int xm_prof_kernel_name(int key, char *buf, int len);
/* Test / tools switches: ONE entry (round 6; rounds 2-5 exported one xm_debug_force_* function per switch).  They select
 * between complete, parity-tested implementations of an operator (never what is computed) and are process-global: a host
 * that embeds the library has no reason to call this.  xm_debug_set returns the previous value, xm_debug_get the current
 * one; INT_MIN = unknown key.
 *   "conv_cfg"       tile configuration of every implicit-GEMM launch, 0 .. xm_debug_get("num_conv_cfgs") - 1; -1 = measured / modelled choice
 *   "conv_splits"    split-K factor of the implicit-GEMM launches (0 = automatic)
 *   "conv_halo"      1 + v: halo-patch kernel variant v (0: 128-row tiles, 1: 96-row tiles, 2: 96-row tiles / tall patch) wherever it
 *                    can run, another runnable variant otherwise; 0: never; -1: measured choice (default)
 *   "conv_stem"      1: the single-channel stem kernels (conv_stem_kernel, conv_stem_wgrad_*; also gates the Gram route's kernels)
 *                    wherever they can run; 0: never; -1: measured choice (default)
 *   "conv_stem3"     the same for the three-channel 7 x 7 / stride 2 stem kernel (conv_stem3_kernel: the teachers' first layer)
 *   "wgrad_patch"    1: conv_wgrad_patch_kernel (filter derivative of 3 x 3 / stride 1 / pad 1 layers) wherever it can run; 0: never; -1: default
 *   "wgrad_patch_s2" the same for conv_wgrad_patch_s2_kernel (5 x 5 / stride 2: the student's conv2)
 *   "dgrad_s2"       0: never conv_dgrad_s2_kernel (dgrad of 5 x 5 / stride 2 layers); 1 / -1: wherever it can run (default)
 *   "comm_single"    1: xm_comm_init with one worker creates a real one-rank RCCL communicator (tests of the exchange path)
 * (xm_debug_conv_cycles -- per-block shader-clock records for tools/conv_bench.py --cycles -- exists only in a library
 * built with XM_DEBUG_CYCLES=1.) */
int xm_debug_set(const char *key, int value);
int xm_debug_get(const char *key);
#ifdef __cplusplus
}
#endif
#endif
