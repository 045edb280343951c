#!/usr/bin/env python
"""Does a producer -> consumer pair keep its intermediate in the 256 MB Infinity Cache when it is run in sample chunks?

The student's bn1 + relu + pool1 backward writes dX (462 MB at 32 spectrograms) and conv1's filter derivative reads it
back once.  Unchunked both passes stream through HBM.  In chunks of n samples with ONE re-used scratch buffer for the
chunk's dX (n * 14.4 MB), the write and the read can stay on the die.  Timing only (chunk-local bnorm sums: the values are
not the step's).  usage: python tools/mall_chunk_bench.py [N=32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = 300


def t(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


x_in = torch.randn((N, 1, W, 512), device="cuda").permute(3, 2, 1, 0)
f = (torch.randn((96, 1, 7, 7), device="cuda") * 0.05).permute(3, 2, 1, 0)
bias = vl.mat_empty(96, 1, device=x_in.device); bias.fill_(0.1)
g = vl.mat_empty(96, 1, device=x_in.device); g.fill_(1.0)
b = vl.mat_empty(96, 1, device=x_in.device); b.fill_(0.0)
y = vl.vl_nnconv(x_in, f, bias, stride=2, pad=1)                       # 254 x 148 x 96 x N
yp, am, mo = vl.bnorm_relu_pool(y, g, b, [3, 3], stride=2, pad=0)
dzp = torch.randn(tuple(reversed(yp.shape)), device="cuda").permute(3, 2, 1, 0)
PER = am.numel() // N      # routing-table bytes per sample (flat, sample-major)


def pair(a, z):
    sl = slice(a, z)
    dx, _, _ = vl.bnorm_relu_pool_backward(y[..., sl], g, b, mo, am[a * PER:z * PER], dzp[..., sl], [3, 3], stride=2, pad=0,
                                           y_pool=yp[..., sl])
    vl.vl_nnconv(x_in[..., sl], f, bias, dx, stride=2, pad=1, no_der_data=True, no_der_biases=True)
    del dx        # the caching allocator hands the same block to the next chunk


def only_apply(a, z):
    sl = slice(a, z)
    dx, _, _ = vl.bnorm_relu_pool_backward(y[..., sl], g, b, mo, am[a * PER:z * PER], dzp[..., sl], [3, 3], stride=2, pad=0,
                                           y_pool=yp[..., sl])
    del dx


dxw = torch.randn((N, 96, 148, 254), device="cuda").permute(3, 2, 1, 0)


def only_wgrad(a, z):
    sl = slice(a, z)
    vl.vl_nnconv(x_in[..., sl], f, bias, dxw[..., sl], stride=2, pad=1, no_der_data=True, no_der_biases=True)


print("bn1+relu+pool1 backward -> conv1 wgrad at %d spectrograms (dX %.0f MB)" % (N, y.numel() * 4 / 1e6))
for n in (N, 16, 8, 4, 2):
    if n > N:
        continue
    ch = [(a, min(a + n, N)) for a in range(0, N, n)]
    ta = t(lambda: [only_apply(a, z) for a, z in ch])
    tw = t(lambda: [only_wgrad(a, z) for a, z in ch])
    tp = t(lambda: [pair(a, z) for a, z in ch])
    print("chunks of %2d samples (%3.0f MB of dX each): backward alone %.1f us, wgrad alone %.1f us, interleaved pairs %.1f us"
          % (n, n * y.numel() * 4 / N / 1e6, ta, tw, tp))
