#!/usr/bin/env python
"""The student's conv1 backward chain at full size: bn1 + relu1 + pool1 backward (writes DX) + conv1's filter derivative
(reads DX), against xm_nnconv_backward_filter_bnrelupool (DX rebuilt inside the filter-derivative kernel, never written).
usage: python tools/stem_bwd_bench.py [N=32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = 300


def t(fn, reps=30):
    for _ in range(6):
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
dxbuf = {}


def apply_only():
    dxbuf["dx"] = vl.bnorm_relu_pool_backward(y, g, b, mo, am, dzp, [3, 3], stride=2, pad=0, y_pool=yp)[0]


def wgrad_only():
    vl.vl_nnconv(x_in, f, bias, dxbuf["dx"], stride=2, pad=1, no_der_data=True, no_der_biases=True)


def fused():
    assert vl.conv_backward_filter_bnrelupool(x_in, (7, 7, 1, 96), y, g, b, mo, am, yp, dzp, [3, 3], stride=2, pad=1,
                                              pool_stride=2, pool_pad=0) is not None


def gram_only():
    return vl.stem_gram(x_in, (7, 7), stride=2, pad=1)


def fused_gram():
    assert vl.conv_backward_filter_bnrelupool_gram(x_in, f, bias, g, mo, am, yp, dzp, [3, 3], stride=2, pad=1, pool_stride=2,
                                                   pool_pad=0) is not None


gm = gram_only()


def fused_gram_given():
    assert vl.conv_backward_filter_bnrelupool_gram(x_in, f, bias, g, mo, am, yp, dzp, [3, 3], stride=2, pad=1, pool_stride=2,
                                                   pool_pad=0, gram=gm) is not None


if os.environ.get("ONLY_GRAM"):
    print("N=%d dbg=%s: Gram %.1f us, pool kernel + finalize with G given %.1f us" % (N, os.environ.get("XM_SP_DBG", "0"), t(gram_only), t(fused_gram_given)))
    sys.exit(0)
ta = t(apply_only)
tw = t(wgrad_only)
tf = t(fused)
print("conv1 backward chain at %d spectrograms (DX %.0f MB): bnorm+relu+pool backward %.1f us + filter derivative %.1f us "
      "= %.1f us;  fused (DX never written) %.1f us" % (N, y.numel() * 4 / 1e6, ta, tw, ta + tw, tf))
tg, tfg, tfgg = t(gram_only), t(fused_gram), t(fused_gram_given)
print("  Gram route (conv1's output not read): Gram matrix %.1f us, whole call %.1f us, with G given %.1f us" % (tg, tfg, tfgg))
