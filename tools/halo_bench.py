#!/usr/bin/env python
"""Unit-stride-gather layers of the hot path: best implicit-GEMM configuration (measured tile choice) against the
halo-patch kernel variants (1: 128-row tiles, 2: 96-row tiles, 3: 96-row tiles / tall patch), forward and dgrad.
usage: python tools/halo_bench.py [N=32] [case ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl, _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = _lib.load()
# name: (H, W, C, K, F, stride, pad)
CASES = {"s_conv2": (126, 73, 96, 256, 5, 2, 1), "s_conv3": (30, 17, 256, 384, 3, 1, 1), "s_conv4": (30, 17, 384, 256, 3, 1, 1),
         "s_conv5": (30, 17, 256, 256, 3, 1, 1), "t_res3": (28, 28, 128, 128, 3, 1, 1), "t_res4": (14, 14, 256, 256, 3, 1, 1),
         "t_res5": (7, 7, 512, 512, 3, 1, 1), "x_fill3x3": (64, 64, 512, 512, 3, 1, 1)}
if len(sys.argv) > 2:
    CASES = {k: v for k, v in CASES.items() if k in sys.argv[2:]}


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for name, (H, W, C, K, F, st, pad) in CASES.items():
    n = 4 if name.startswith("x_") else N
    x = torch.randn((n, C, W, H), device="cuda").permute(3, 2, 1, 0)
    f = (torch.randn((K, C, F, F), device="cuda") * 0.05).permute(3, 2, 1, 0)
    y = vl.vl_nnconv(x, f, None, stride=st, pad=pad)
    Ho, Wo = int(y.shape[0]), int(y.shape[1])
    dy = torch.randn((n, K, Wo, Ho), device="cuda").permute(3, 2, 1, 0)
    gf = 2.0 * Ho * Wo * n * K * F * F * C / 1e9
    row = "%-10s N=%-3d %6.1f GF" % (name, n, gf)
    t0_ = torch.cuda.Event(enable_timing=True); t1_ = torch.cuda.Event(enable_timing=True)
    t0_.record()
    for _ in range(200):                      # clocks to their steady state before the first timed variant
        vl.vl_nnconv(x, f, None, stride=st, pad=pad)
        t1_.record(); 
        if _ % 20 == 19:
            t1_.synchronize()
            if t0_.elapsed_time(t1_) > 400: break
    for h in (0, 1, 2, 3):
        L.xm_debug_force_conv_halo(h)
        tf = t(lambda: vl.vl_nnconv(x, f, None, stride=st, pad=pad))
        td = t(lambda: vl.vl_nnconv(x, f, None, dy, stride=st, pad=pad, no_der_filters=True))
        row += " | %s f %6.1f TF d %6.1f TF" % ("gemm" if h == 0 else "halo%d" % h, gf / tf * 1e3, gf / td * 1e3)
    L.xm_debug_force_conv_halo(-1)
    print(row)
