#!/usr/bin/env python
"""The student's conv1 -> bn1 -> relu1 -> pool1 chain at full size, forward and backward: the composed operators
(conv_stem_kernel + pool_fwd_lds_kernel; bnpool sums + conv_stem_wgrad_bnp_kernel) against the Gram route
(stem_gram_kernel + conv_stem_bnpool_fwd_kernel; conv_stem_wgrad_pool_kernel).   usage: python tools/stem_chain_bench.py [N=32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = 300


def t(fn, reps=20):
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


x_in = torch.randn((N, 1, W, 512), device="cuda").permute(3, 2, 1, 0)
f = (torch.randn((96, 1, 7, 7), device="cuda") * 0.05).permute(3, 2, 1, 0)
bias = vl.mat_empty(96, 1, device=x_in.device); bias.fill_(0.1)
g = vl.mat_empty(96, 1, device=x_in.device); g.fill_(1.0)
b = vl.mat_empty(96, 1, device=x_in.device); b.fill_(0.0)
mo0 = vl.mat_empty(96, 2, device=x_in.device)
st = {}


def fwd_old():
    y = vl.vl_nnconv(x_in, f, bias, stride=2, pad=1, moments_out=mo0)
    st["y"] = y
    st["old"] = vl.bnorm_relu_pool(y, g, b, [3, 3], stride=2, pad=0, moments=None, moments_out=None)


def fwd_new():
    st["new"] = vl.conv_bnorm_relu_pool(x_in, f, bias, g, b, [3, 3], stride=2, pad=1, pool_stride=2, pool_pad=0)
    assert st["new"] is not None


if os.environ.get("ONLY_FWD"):
    fwd_new()
    print("N=%d XM_SF_DBG=%s: fused forward incl. Gram %.1f us" % (N, os.environ.get("XM_SF_DBG", "0"), t(fwd_new)))
    sys.exit(0)
fwd_old(); fwd_new()
yp, am, mo = st["old"]
yp2, am2, mo2, gram = st["new"]
dzp = torch.randn(tuple(reversed(yp.shape)), device="cuda").permute(3, 2, 1, 0)
print("pooled outputs differ by %.2e (max |y| %.2f), moments by %.2e" % ((yp - yp2).abs().max().item(), yp.abs().max().item(),
                                                                         (mo - mo2).abs().max().item()))


def bwd_old():
    assert vl.conv_backward_filter_bnrelupool(x_in, (7, 7, 1, 96), st["y"], g, b, mo, am, yp, dzp, [3, 3], stride=2, pad=1,
                                              pool_stride=2, pool_pad=0) is not None


def bwd_gram_y():
    assert vl.conv_backward_filter_bnrelupool_gram(x_in, f, bias, g, mo, am, yp, dzp, [3, 3], stride=2, pad=1, pool_stride=2,
                                                   pool_pad=0, gram=gram) is not None


def bwd_gram():
    assert vl.conv_backward_filter_bnrelupool_gram(x_in, f, bias, g, mo2, am2, None, dzp, [3, 3], stride=2, pad=1, pool_stride=2,
                                                   pool_pad=0, gram=gram) is not None


def gram_only():
    vl.stem_gram(x_in, (7, 7), stride=2, pad=1)


r = dict(fo=t(fwd_old), fn=t(fwd_new), gr=t(gram_only), bo=t(bwd_old), by=t(bwd_gram_y), bn=t(bwd_gram))
print("N=%d dbg=%s forward: composed %.1f us, fused (incl. Gram %.1f) %.1f us | backward: composed %.1f us, Gram route with y_pool "
      "%.1f us, with the gated table %.1f us | chain %.1f -> %.1f us" % (N, os.environ.get("XM_SP_DBG", "0"), r["fo"], r["gr"], r["fn"],
                                                                        r["bo"], r["by"], r["bn"], r["fo"] + r["bo"], r["fn"] + r["bn"]))
