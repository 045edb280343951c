#!/usr/bin/env python
"""Micro-benchmark of vl_nnconv on the layer geometries of the hot path (per-GPU batch 32).
usage: python tools/conv_bench.py [--n 32] [--reps 20] [case ...]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl  # noqa: E402

# name: (H, W, C, FH, FW, K, stride, pad)
CASES = {
    "s_conv1": (512, 300, 1, 7, 7, 96, 2, 1),
    "s_conv2": (126, 73, 96, 5, 5, 256, 2, 1),
    "s_conv3": (30, 17, 256, 3, 3, 384, 1, 1),
    "s_conv4": (30, 17, 384, 3, 3, 256, 1, 1),
    "s_conv5": (30, 17, 256, 3, 3, 256, 1, 1),
    "s_fc6": (9, 8, 256, 9, 1, 4096, 1, 0),
    "s_fc7": (1, 1, 4096, 1, 1, 1024, 1, 0),
    "t_conv1": (224, 224, 3, 7, 7, 64, 2, 3),
    "t_res2_3x3": (56, 56, 64, 3, 3, 64, 1, 1),
    "t_res2_1x1a": (56, 56, 256, 1, 1, 64, 1, 0),
    "t_res2_1x1c": (56, 56, 64, 1, 1, 256, 1, 0),
    "t_res3_3x3": (28, 28, 128, 3, 3, 128, 1, 1),
    "t_res3_1x1c": (28, 28, 128, 1, 1, 512, 1, 0),
    "t_res4_3x3": (14, 14, 256, 3, 3, 256, 1, 1),
    "t_res4_1x1c": (14, 14, 256, 1, 1, 1024, 1, 0),
    "t_res5_3x3": (7, 7, 512, 3, 3, 512, 1, 1),
    "t_res5_1x1c": (7, 7, 512, 1, 1, 2048, 1, 0),
    # synthetic ceiling probes (9th entry = fixed N): 512 tiles of 128x128 = exactly two per CU, long reductions
    "x_fill3x3": (64, 64, 512, 3, 3, 512, 1, 1, 4),
    "x_fill1x1": (64, 64, 1024, 1, 1, 512, 1, 0, 4),
    "x_fill3x3_1": (64, 32, 512, 3, 3, 512, 1, 1, 4),     # 256 tiles: one per CU
}


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*")
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dirs", default="fwd,dgrad,wgrad")
    ap.add_argument("--scan", action="store_true", help="time every tile configuration (forced)")
    ap.add_argument("--cfg", type=int, default=-1, help="force ONE tile configuration")
    ap.add_argument("--cycles", action="store_true",
                    help="fwd only: shader clocks spent by block 0 (cycles per 16-k stage, effective clock)")
    args = ap.parse_args()
    names = args.cases or list(CASES)
    print("%-12s %-6s %9s %9s" % ("case", "dir", "ms", "TFLOP/s"))
    for nm in names:
        H, W, C, FH, FW, K, s, p = CASES[nm][:8]
        N = CASES[nm][8] if len(CASES[nm]) > 8 else args.n
        x = vl.from_numpy(np.random.default_rng(0).standard_normal((H, W, C, N)).astype(np.float32))
        f = vl.from_numpy(np.random.default_rng(1).standard_normal((FH, FW, C, K)).astype(np.float32))
        b = None if os.environ.get("CB_NOBIAS") else vl.from_numpy(np.zeros((K, 1), np.float32))
        y = vl.vl_nnconv(x, f, b, stride=s, pad=p)
        dz = vl.from_numpy(np.random.default_rng(2).standard_normal(tuple(y.shape)).astype(np.float32))
        Ho, Wo = int(y.shape[0]), int(y.shape[1])
        flops = 2.0 * Ho * Wo * N * K * FH * FW * C
        from mcncrossmodalemotions_amd import _lib
        L = _lib.load()
        cfgs = list(range(L.xm_debug_num_conv_cfgs())) if args.scan else [args.cfg]
        for d, cfg in [(dd, cc) for dd in args.dirs.split(",") for cc in cfgs]:
            L.xm_debug_force_conv_cfg(cfg)
            if d == "fwd":
                ms = timeit(lambda: vl.vl_nnconv(x, f, b, stride=s, pad=p), args.reps)
            elif d == "dgrad":
                ms = timeit(lambda: vl.vl_nnconv(x, f, None, dz, stride=s, pad=p, no_der_filters=True), args.reps)
            else:
                ms = timeit(lambda: vl.vl_nnconv(x, f, None, dz, stride=s, pad=p, no_der_data=True), args.reps)
            extra = ""
            if args.cycles and d == "fwd":
                import ctypes
                nb = 4096
                buf = (ctypes.c_ulonglong * (4 * nb))()
                L.xm_debug_conv_cycles(1, buf, 0)
                for _ in range(3):
                    vl.vl_nnconv(x, f, b, stride=s, pad=p)
                torch.cuda.synchronize()
                L.xm_debug_conv_cycles(0, buf, nb)
                rec = np.array(buf[:], dtype=np.uint64).reshape(nb, 4)
                rec = rec[rec[:, 1] > 0]
                dur = (rec[:, 1] - rec[:, 0]).astype(np.float64)
                stages = (FH * FW * C + 15) // 16
                xcc = rec[:, 3] & 0xF
                extra = "  %d blocks: clk/block min %.0f med %.0f max %.0f (%.0f clk/stage med)" % (
                    len(dur), dur.min(), np.median(dur), dur.max(), np.median(dur) / stages)
                for xc in sorted(set(xcc.tolist())):
                    m = xcc == xc
                    t0, t1 = rec[m, 0].astype(np.float64), rec[m, 1].astype(np.float64)
                    extra += "\n      xcc%d: %3d blocks, span %.0f clk (first start -> last end), starts within %.0f, ends within %.0f, dur med %.0f" % (
                        xc, m.sum(), t1.max() - t0.min(), t0.max() - t0.min(), t1.max() - t1.min(), np.median(t1 - t0))
            print("%-12s %-6s %9.3f %9.1f %s%s" % (nm, d, ms, flops / ms / 1e9, "" if cfg < 0 else "cfg%d" % cfg, extra),
                  flush=True)
        L.xm_debug_force_conv_cfg(-1)


if __name__ == "__main__":
    main()
