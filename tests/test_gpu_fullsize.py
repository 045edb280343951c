"""GPU, BASELINE.json's full layer sizes (where the CPU oracle would take minutes): parity through
size-independent properties of the operators --
  * adjoint identities  <dzdy, conv(x)> == <dx, x>  and  == <df, f> (+ <db, b>): they tie forward,
    dgrad and wgrad of the SAME geometry together, so a wrong tap / stride / class in any one breaks;
  * linearity of the convolution in x and in f;
  * batch-norm output statistics, pooling bounds, loss-gradient zero-sum, SGD closed form;
  * sample independence: running a sub-batch reproduces the corresponding slice (ragged last shard).
torch is used here only as the checker's reduction engine (fp64 dot products on the device)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (name, H, W, C, FH, FW, K, stride, pad) at N = 8 (full spatial size / channels of the hot path)
FULL = [
    ("student conv1", 512, 300, 1, 7, 7, 96, 2, 1),
    ("student conv2", 126, 73, 96, 5, 5, 256, 2, 1),
    ("student conv3", 30, 17, 256, 3, 3, 384, 1, 1),
    ("student fc6", 9, 8, 256, 9, 1, 4096, 1, 0),
    ("student fc7", 1, 1, 4096, 1, 1, 1024, 1, 0),
    ("teacher conv1", 224, 224, 3, 7, 7, 64, 2, 3),
    ("teacher res3 1x1/2", 56, 56, 256, 1, 1, 128, 2, 0),
    ("teacher res5 3x3", 7, 7, 512, 3, 3, 512, 1, 1),
]


def dot(a, b):
    return float((a.double() * b.double()).sum().item())


def rnd(shape, seed, dev, scale=1.0):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    t = torch.randn(tuple(reversed(shape)), generator=g, device=dev, dtype=torch.float32) * scale
    return t.permute(*reversed(range(len(shape))))


@pytest.mark.parametrize("case", FULL, ids=[c[0] for c in FULL])
def test_conv_adjoint_and_linearity_fullsize(gpu, case):
    from mcncrossmodalemotions_amd import vl
    _, H, W, C, FH, FW, K, s, p = case
    N = 8
    x, x2 = rnd((H, W, C, N), 1, gpu), rnd((H, W, C, N), 2, gpu)
    f = rnd((FH, FW, C, K), 3, gpu, scale=(2.0 / (FH * FW * C)) ** 0.5)
    b = rnd((K, 1), 4, gpu)
    y = vl.vl_nnconv(x, f, b, stride=s, pad=p)
    dz = rnd(tuple(y.shape), 5, gpu)
    dx, df, db = vl.vl_nnconv(x, f, b, dz, stride=s, pad=p)
    lhs = dot(dz, y)
    # <dz, conv(x; f, b)> = <dx, x> + <db, b>  (conv is linear in x given f) and = <df, f> + <db, b>
    tol = 2e-4 * max(1.0, (dot(dz, dz) * dot(y, y)) ** 0.5)
    assert abs(lhs - (dot(dx, x) + dot(db, b))) <= tol, ("dgrad adjoint", lhs, dot(dx, x) + dot(db, b))
    assert abs(lhs - (dot(df, f) + dot(db, b))) <= tol, ("wgrad adjoint", lhs, dot(df, f) + dot(db, b))
    # linearity in x: conv(x + x2) - b == (conv(x) - b) + (conv(x2) - b)
    y2 = vl.vl_nnconv(x2, f, b, stride=s, pad=p)
    y12 = vl.vl_nnconv(vl.sum2(x, x2), f, b, stride=s, pad=p)
    bb = b.reshape(1, 1, K, 1)
    err = ((y12 - bb) - ((y - bb) + (y2 - bb))).abs().max().item()
    assert err <= 1e-4 * max(1.0, y12.abs().max().item()), ("linearity", err)
    # sample independence / ragged shard: the last 3 samples alone reproduce their slice bit for bit
    xs = x[..., 5:].permute(3, 2, 1, 0).contiguous().permute(3, 2, 1, 0)
    ys = vl.vl_nnconv(xs, f, b, stride=s, pad=p)
    assert torch.equal(ys, y[..., 5:]) or (ys - y[..., 5:]).abs().max().item() <= 1e-5 * max(1.0, y.abs().max().item())


def test_bnorm_pool_chain_fullsize(gpu):
    """student bn1 -> relu1 -> mpool1 at full 254 x 148 x 96: fused operator == the three separate
    operators bit for bit; BN output statistics; backward adjoint of the fused pair."""
    from mcncrossmodalemotions_amd import vl
    H, W, C, N = 254, 148, 96, 8
    x = rnd((H, W, C, N), 11, gpu, 2.0) + 0.7
    g = rnd((C, 1), 12, gpu).abs() + 0.5
    b = rnd((C, 1), 13, gpu)
    yb, mom = vl.vl_nnbnorm(x, g, b)
    # zero mean / unit variance before the affine: (yb - b) / g
    z = (yb - b.reshape(1, 1, C, 1)) / g.reshape(1, 1, C, 1)
    m = z.double().mean(dim=(0, 1, 3))
    v = (z.double() ** 2).mean(dim=(0, 1, 3))
    assert m.abs().max().item() < 1e-4 and (v - 1).abs().max().item() < 2e-3  # eps = 1e-4 shifts var slightly
    yr = vl.vl_nnrelu(yb)
    yp, am = vl.vl_nnpool(yr, [3, 3], stride=2, method="max", want_argmax=True)
    yf, amf, mof = vl.bnorm_relu_pool(x, g, b, [3, 3], stride=2)
    assert torch.equal(yp, yf) and torch.equal(am, amf), "fused forward differs from the separate operators"
    assert (mom - mof).abs().max().item() == 0
    dz = rnd(tuple(yp.shape), 14, gpu)
    d1 = vl.vl_nnpool(yr, [3, 3], dz, stride=2, method="max", argmax=am)
    d2 = vl.vl_nnrelu(yb, d1)
    dx_ref, dg_ref, db_ref, _ = vl.vl_nnbnorm(x, g, b, d2)
    dx, dg, db = vl.bnorm_relu_pool_backward(x, g, b, mof, amf, dz, [3, 3], stride=2)
    sc = max(1.0, dx_ref.abs().max().item())
    assert (dx - dx_ref).abs().max().item() <= 1e-4 * sc
    assert (dg - dg_ref).abs().max().item() <= 1e-4 * max(1.0, dg_ref.abs().max().item())
    assert (db - db_ref).abs().max().item() <= 1e-4 * max(1.0, db_ref.abs().max().item())
    # the same with the per-channel sums taken from the pooled tensors (what dagnn's fused step passes)
    dx2, dg2, db2 = vl.bnorm_relu_pool_backward(x, g, b, mof, amf, dz, [3, 3], stride=2, y_pool=yf)
    assert (dx2 - dx_ref).abs().max().item() <= 1e-4 * sc
    assert (dg2 - dg_ref).abs().max().item() <= 1e-4 * max(1.0, dg_ref.abs().max().item())
    assert (db2 - db_ref).abs().max().item() <= 1e-4 * max(1.0, db_ref.abs().max().item())
    # pooling bounds and routing conservation: every output gradient lands on exactly one input
    assert (yp >= 0).all() and abs(dot(d1, torch.ones_like(d1)) - dot(dz, torch.ones_like(dz))) < 1e-2
    # train-mode BN backward output sums to zero per channel (batch statistics absorb the mean)
    assert dx_ref.double().sum(dim=(0, 1, 3)).abs().max().item() < 5e-2


def test_loss_and_sgd_properties(gpu):
    from mcncrossmodalemotions_amd import vl
    N = 256
    x, p = rnd((1, 1, 8, N), 21, gpu, 3.0), rnd((1, 1, 8, N), 22, gpu, 3.0)
    g = vl.vl_nnsoftmaxceloss(x, p, 1.0, temperature=2, logitTargets=True)
    # gradient of a softmax cross-entropy with normalised targets sums to zero over the classes
    assert g.double().sum(dim=2).abs().max().item() < 1e-6
    # identical prediction and target logits -> zero gradient, loss = entropy of the target
    g0 = vl.vl_nnsoftmaxceloss(p, p, 1.0, temperature=2, logitTargets=True)
    assert g0.abs().max().item() < 1e-6
    l0 = vl.vl_nnsoftmaxceloss(p, p, temperature=2, logitTargets=True).item()
    q = torch.softmax(p.double().reshape(8, N).t() / 2, 1)
    assert abs(l0 - float(-(q * q.log()).sum())) < 1e-3
    n = 16_649_928  # the student's parameter count: one fused update over the flat buffer
    w, m, d = rnd((n, 1), 23, gpu), rnd((n, 1), 24, gpu), rnd((n, 1), 25, gpu)
    w0, m0 = w.clone(), m.clone()
    vl.sgd_update(w, m, d, 1e-4, 0.9, 5e-4, 256)
    mref = 0.9 * m0 - (5e-4 * w0 + d / 256)
    assert (m - mref).abs().max().item() < 1e-6 and (w - (w0 + 1e-4 * mref)).abs().max().item() < 1e-6


def test_batch_provider_fullsize(gpu):
    from mcncrossmodalemotions_amd import vl
    spec = rnd((512, 300, 1, 32), 31, gpu).abs() * 5 + 0.1
    n = vl.spec_rownorm(spec)
    assert n.double().mean(dim=1).abs().max().item() < 1e-5
    assert (n.double().std(dim=1, unbiased=True) - 1).abs().max().item() < 1e-4
