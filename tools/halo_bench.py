#!/usr/bin/env python
"""3 x 3 unit-stride layers of the hot path: best implicit-GEMM configuration (measured tile choice) against the
halo-patch kernel, forward and dgrad.  usage: python tools/halo_bench.py [N=32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl, _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = _lib.load()
# name: (H, W, C, K)
CASES = {"s_conv3": (30, 17, 256, 384), "s_conv4": (30, 17, 384, 256), "s_conv5": (30, 17, 256, 256),
         "t_res2": (56, 56, 64, 64), "t_res3": (28, 28, 128, 128), "t_res4": (14, 14, 256, 256), "t_res5": (7, 7, 512, 512)}
if len(sys.argv) > 2:
    CASES = {k: v for k, v in CASES.items() if k in sys.argv[2:]}
CASES["x_fill3x3"] = (64, 64, 512, 512)


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


for name, (H, W, C, K) in CASES.items():
    n = 4 if name.startswith("x_") else N
    x = torch.randn((n, C, W, H), device="cuda").permute(3, 2, 1, 0)
    f = (torch.randn((K, C, 3, 3), device="cuda") * 0.05).permute(3, 2, 1, 0)
    dy = torch.randn((n, K, W, H), device="cuda").permute(3, 2, 1, 0)
    gf = 2.0 * H * W * n * K * 9 * C / 1e9
    row = "%-10s N=%-3d %6.2f GF " % (name, n, gf)
    for h in (0, 1):
        L.xm_debug_force_conv_halo(h)
        tf = t(lambda: vl.vl_nnconv(x, f, None, pad=1))
        td = t(lambda: vl.vl_nnconv(x, f, None, dy, pad=1, no_der_filters=True))
        row += " | %s fwd %7.1f us %6.1f TF  dgrad %7.1f us %6.1f TF" % ("halo" if h else "gemm", tf, gf / tf * 1e3, td, gf / td * 1e3)
    L.xm_debug_force_conv_halo(-1)
    print(row)
