#!/usr/bin/env python
"""vl_nnbnorm backward (+ the producing convolution's bias derivative) at the student's layer shapes: the five-launch
chain (partial, finalize, apply, bias partial, bias finalize) against xm_nnbnorm_backward_dxsum (three launches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = {"conv3": (30, 17, 384), "conv4": (30, 17, 256), "conv5": (30, 17, 256), "fc6": (1, 8, 4096), "fc7": (1, 1, 1024),
          "conv2": (62, 36, 256)}


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for name, (H, W, C) in SHAPES.items():
    x = torch.randn((N, C, W, H), device="cuda").permute(3, 2, 1, 0)
    dz = torch.randn((N, C, W, H), device="cuda").permute(3, 2, 1, 0)
    g = vl.mat_empty(C, 1, device=x.device); g.fill_(1.0)
    b = vl.mat_empty(C, 1, device=x.device); b.fill_(0.0)
    y, m = vl.vl_nnbnorm(x, g, b, relu=True)
    f = torch.randn((C, 4, 1, 1), device="cuda").permute(3, 2, 1, 0)     # 1 x 1 x 4 x C filter bank: only dzdb is timed
    bb = vl.mat_empty(C, 1, device=x.device)
    dxs = vl.mat_empty(C, 1, device=x.device)

    def old():
        dx = vl.vl_nnbnorm(x, g, b, dz, relu=True, y=y, moments=m, batch_moments=True)[0]
        # bias half of the convolution's backward: sum of dx over pixels and samples
        vl._lib.check(vl._L().xm_nnconv_backward(None, H, W, 4, N, None, 1, 1, 4, C, vl._ptr(dx), None, None, vl._ptr(bb),
                                                 1, 1, 0, 0, 0, 0, 1, 1, vl._stream()))

    def new():
        vl.vl_nnbnorm(x, g, b, dz, relu=True, y=y, moments=m, batch_moments=True, dxsum_out=dxs)
    print("%-6s %dx%dx%dx%d  five launches %.1f us   dxsum path %.1f us" % (name, H, W, C, N, t(old), t(new)))
