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
 *                  force both arms through the xm_debug_force_* hooks; tests/test_gpu_path_switches.py runs whole passes
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
/* test hooks: force one tile configuration for every convolution launch (-1 = automatic) */
int xm_debug_force_conv_cfg(int cfg);
int xm_debug_num_conv_cfgs(void);
/* 1 + v: halo-patch kernel variant v (0: 128-row tiles, 1: 96-row tiles, 2: 96-row tiles / tall patch) wherever it can
 * run, another runnable variant otherwise; 0: never; -1: measured choice (default) */
int xm_debug_force_conv_halo(int on);
/* 1: the single-channel stem kernel (conv_stem_kernel) wherever it can run; 0: never; -1: measured choice (default) */
int xm_debug_force_conv_stem(int on);
/* the same for the three-channel 7 x 7 / stride 2 stem kernel (conv_stem3_kernel: the teachers' first layer) */
int xm_debug_force_conv_stem3(int on);
/* 1: the patch kernel for the filter derivative of 3 x 3 / stride 1 / pad 1 layers (conv_wgrad_patch_kernel) wherever it
 * can run; 0: never; -1: measured choice (default) */
int xm_debug_force_wgrad_patch(int on);
/* the same for the patch kernel of 5 x 5 / stride 2 layers (conv_wgrad_patch_s2_kernel: the student's conv2) */
int xm_debug_force_wgrad_patch_s2(int on);
/* 0: never conv_dgrad_s2_kernel (dgrad of 5 x 5 / stride 2 layers); 1 / -1: wherever it can run (default) */
int xm_debug_force_dgrad_s2(int on);
/* force the split-K factor of the implicit-GEMM launches (0 = automatic) */
int xm_debug_force_conv_splits(int splits);
/* on = 1: every block (< 4096) of every later conv_gemm launch stores {first shader clock, last shader clock, HW_ID,
   XCC_ID}; out[4 * nblocks] receives the records of the most recent launch (synchronise first).
   tools/conv_bench.py --cycles */
int xm_debug_conv_cycles(int on, unsigned long long *out, int nblocks);
#ifdef __cplusplus
}
#endif
#endif
