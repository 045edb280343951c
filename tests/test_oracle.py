"""CPU: the oracle (oracle/xm_oracle.c) against (i) analytic known answers, (ii) an independent
second opinion (torch CPU ops -- NOT the reference), (iii) fp64 finite differences,
(iv) the committed golden fixtures.  Pure CPU; runs in well under a minute."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ops_small.npz")


def tn(a):  # MATLAB (H,W,C,N) -> torch (N,C,W,H) double
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a).transpose(3, 2, 1, 0))).double()


def fm(t):  # back
    return t.detach().numpy().transpose(3, 2, 1, 0)


def torch_conv(x, f, b, stride, pad, dil=(1, 1)):
    xt = tn(x).requires_grad_(True)
    ft = tn(f).requires_grad_(True)
    bt = None if b is None else torch.from_numpy(np.asarray(b, np.float64).ravel()).requires_grad_(True)
    pt, pb, pl, pr = pad
    y = TF.conv2d(TF.pad(xt, (pt, pb, pl, pr)), ft, bt, stride=(stride[1], stride[0]),
                  dilation=(dil[1], dil[0]), groups=x.shape[2] // f.shape[2])
    return xt, ft, bt, y


CASES = [
    ((12, 9, 5, 2), (3, 3, 5, 7), (1, 1), (1, 1, 1, 1), (1, 1)),
    ((21, 18, 6, 2), (5, 5, 6, 9), (2, 2), (1, 1, 1, 1), (1, 1)),
    ((11, 10, 4, 2), (3, 2, 4, 6), (2, 3), (0, 1, 2, 0), (1, 1)),
    ((13, 12, 4, 2), (3, 3, 4, 6), (1, 1), (2, 2, 2, 2), (2, 2)),
    ((10, 9, 8, 2), (3, 3, 4, 6), (1, 1), (1, 1, 1, 1), (1, 1)),  # groups
    ((9, 8, 16, 3), (9, 1, 16, 10), (1, 1), (0, 0, 0, 0), (1, 1)),  # fc6-like
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("acc64", [False, True])
def test_conv_vs_torch(case, acc64):
    xs, fs, stride, pad, dil = case
    rng = np.random.default_rng(1)
    x, f, b = O.F(rng.standard_normal(xs)), O.F(rng.standard_normal(fs)), O.F(rng.standard_normal(fs[3]))
    xt, ft, bt, yt = torch_conv(x, f, b, stride, pad, dil)
    y = O.vl_nnconv(x, f, b, stride=stride, pad=pad, dilate=dil, acc64=acc64)
    tol = 2e-6 if acc64 else 5e-5
    assert np.abs(y - fm(yt)).max() <= tol * max(1, np.abs(fm(yt)).max())
    dz = O.F(rng.standard_normal(y.shape))
    yt.backward(tn(dz))
    dx, df, db = O.vl_nnconv(x, f, b, dz, stride=stride, pad=pad, dilate=dil, acc64=acc64)
    for got, ref in ((dx, fm(xt.grad)), (df, fm(ft.grad)), (db, bt.grad.numpy())):
        assert np.abs(got - ref).max() <= tol * max(1, np.abs(ref).max())


def test_conv_known_answers():
    # identity 1x1 filter returns the input; a box filter sums the window (cross-correlation)
    x = O.F(np.arange(2 * 3 * 1 * 1).reshape(2, 3, 1, 1, order="F"))
    assert np.array_equal(O.vl_nnconv(x, O.F(np.ones((1, 1, 1, 1))), None), x)
    y = O.vl_nnconv(x, O.F(np.ones((2, 2, 1, 1))), O.F([10.0]))
    assert np.array_equal(y.ravel(order="F"), [0 + 1 + 2 + 3 + 10, 2 + 3 + 4 + 5 + 10])
    # no kernel flip: filter [1 0] picks the TOP element of each vertical pair
    f = O.F(np.array([1.0, 0.0]).reshape(2, 1, 1, 1))
    assert np.array_equal(O.vl_nnconv(x, f, None).ravel(order="F"), [0, 2, 4])
    # output size rule floor((H + pt + pb - FH) / sy) + 1
    assert O.conv_out_size(512, 1, 1, 7, 1, 2) == 254 and O.conv_out_size(300, 1, 1, 7, 1, 2) == 148


def test_pool_vs_torch_and_tie_rule():
    rng = np.random.default_rng(2)
    x = O.F(rng.standard_normal((13, 11, 5, 3)))
    y = O.vl_nnpool(x, [3, 3], stride=2, method="max")
    assert np.allclose(y, fm(TF.max_pool2d(tn(x), (3, 3), (2, 2))))
    ya = O.vl_nnpool(x, [3, 2], stride=(2, 1), pad=(1, 1, 0, 1), method="avg")
    ref = TF.avg_pool2d(TF.pad(tn(x), (1, 1, 0, 1)), (2, 3), (1, 2))  # count_include_pad would differ:
    # MatConvNet divides by the CLIPPED window area, so compare only interior outputs
    assert np.allclose(ya[1:-1, :-1], fm(ref)[1:-1, :-1], atol=1e-6)
    # clipped-area rule on the border: a window hanging over one padded row averages 2x2 of 3x2
    ones = O.F(np.ones((4, 4, 1, 1)))
    assert np.allclose(O.vl_nnpool(ones, [3, 2], stride=1, pad=(1, 1, 0, 0), method="avg"), 1.0)
    # max backward routes to the FIRST maximum in column-major scan order
    z = O.F(np.zeros((3, 3, 1, 1)))
    dx = O.vl_nnpool(z, [3, 3], O.F(np.ones((1, 1, 1, 1))), method="max")
    assert dx[0, 0, 0, 0] == 1 and dx.sum() == 1
    z[1, 0] = z[0, 1] = 5.0  # tie between (h=1,w=0) and (h=0,w=1): column 0 is scanned first
    dx = O.vl_nnpool(z, [3, 3], O.F(np.ones((1, 1, 1, 1))), method="max")
    assert dx[1, 0, 0, 0] == 1 and dx.sum() == 1


def test_bnorm_vs_torch():
    rng = np.random.default_rng(3)
    x = O.F(rng.standard_normal((7, 5, 4, 3)) * 2 + 1)
    g, b = O.F(rng.uniform(0.5, 1.5, 4)), O.F(rng.standard_normal(4))
    eps = 1e-4
    for acc64 in (False, True):
        y, mom = O.vl_nnbnorm(x, g, b, epsilon=eps, acc64=acc64)
        xt = tn(x).requires_grad_(True)
        gt = torch.from_numpy(g.astype(np.float64)).requires_grad_(True)
        bt = torch.from_numpy(b.astype(np.float64)).requires_grad_(True)
        yt = TF.batch_norm(xt, None, None, gt, bt, training=True, eps=eps)
        assert np.abs(y - fm(yt)).max() < 2e-5
        # moments = [mean, sqrt(biased var + eps)]
        assert np.allclose(mom[:, 0], x.mean((0, 1, 3)), atol=1e-5)
        assert np.allclose(mom[:, 1], np.sqrt(x.astype(np.float64).var((0, 1, 3)) + eps), atol=1e-5)
        dz = O.F(rng.standard_normal(x.shape))
        yt.backward(tn(dz))
        dx, dg, db, _ = O.vl_nnbnorm(x, g, b, dz, epsilon=eps, acc64=acc64)
        assert np.abs(dx - fm(xt.grad)).max() < 5e-5
        assert np.abs(dg - gt.grad.numpy()).max() < 5e-4 and np.abs(db - bt.grad.numpy()).max() < 5e-5
    # test mode: y = g * (x - M1) / M2 + b
    M = O.F(np.stack([rng.standard_normal(4), rng.uniform(0.5, 1.5, 4)], 1))
    yt, _ = O.vl_nnbnorm(x, g, b, moments=M, acc64=True)
    ref = g.reshape(1, 1, 4, 1) * (x - M[:, 0].reshape(1, 1, 4, 1)) / M[:, 1].reshape(1, 1, 4, 1) + b.reshape(1, 1, 4, 1)
    assert np.abs(yt - ref).max() < 1e-5


def test_distillation_loss_semantics_and_gradient():
    """SURVEY A.6 decisions: summed over the batch, 1/T in the gradient, no T^2."""
    rng = np.random.default_rng(4)
    x, p = O.F(rng.standard_normal((1, 1, 8, 6)) * 3), O.F(rng.standard_normal((1, 1, 8, 6)) * 3)
    T = 2.0
    xt = torch.from_numpy(x.reshape(8, 6, order="F").T.copy()).double().requires_grad_(True)
    pt = torch.from_numpy(p.reshape(8, 6, order="F").T.copy()).double()
    loss = -(TF.softmax(pt / T, 1) * TF.log_softmax(xt / T, 1)).sum()
    assert abs(O.vl_nnsoftmaxceloss(x, p, temperature=T, logit_targets=True) - loss.item()) < 1e-5
    loss.backward()
    dx = O.vl_nnsoftmaxceloss(x, p, np.ones(1, np.float32), temperature=T, logit_targets=True)
    assert np.abs(dx.reshape(8, 6, order="F").T - xt.grad.numpy()).max() < 1e-6
    # fp64 finite differences of the forward
    eps = 1e-3
    for (c, n) in [(0, 0), (3, 2), (7, 5)]:
        xp, xm = x.copy(), x.copy()
        xp[0, 0, c, n] += eps
        xm[0, 0, c, n] -= eps
        fd = (O.vl_nnsoftmaxceloss(xp, p, temperature=T, logit_targets=True) -
              O.vl_nnsoftmaxceloss(xm, p, temperature=T, logit_targets=True)) / (2 * eps)
        assert abs(fd - dx[0, 0, c, n]) < 2e-3
    # vl_nnloss
    lab = O.F(rng.integers(1, 9, (1, 1, 1, 6)))
    ref = TF.cross_entropy(torch.from_numpy(x.reshape(8, 6, order="F").T.copy()).double(),
                           torch.from_numpy(lab.ravel().astype(np.int64) - 1), reduction="sum")
    assert abs(O.vl_nnloss(x, lab, loss="softmaxlog") - ref.item()) < 1e-5
    err = (x.reshape(8, 6, order="F").argmax(0) + 1 != lab.ravel()).sum()
    assert O.vl_nnloss(x, lab, loss="classerror") == err


def test_elementwise_and_sgd():
    rng = np.random.default_rng(5)
    x, d = O.F(rng.standard_normal((4, 3, 2, 2))), O.F(rng.standard_normal((4, 3, 2, 2)))
    assert np.array_equal(O.vl_nnrelu(x), np.maximum(x, 0))
    assert np.array_equal(O.vl_nnrelu(x, d), d * (x > 0))
    assert np.allclose(O.vl_nnsigmoid(x), 1 / (1 + np.exp(-x)), atol=1e-6)
    a = O.F(rng.standard_normal((1, 1, 2, 2)))
    assert np.allclose(O.scale_axpy(x, a, d, relu=True), np.maximum(a * x + d, 0), atol=1e-6)
    dx, da = O.scale_backward(x, a, d)
    assert np.allclose(dx, a * d, atol=1e-6) and np.allclose(da, (d * x).sum((0, 1), keepdims=True), atol=1e-5)
    w, m, g = O.F(rng.standard_normal(10)), O.F(rng.standard_normal(10)), O.F(rng.standard_normal(10))
    w2, m2 = O.sgd_update(w, m, g, lr=0.1, momentum=0.9, wd=5e-4, batch=4)
    mref = 0.9 * m - (5e-4 * w + g / 4)
    assert np.allclose(m2, mref, atol=1e-6) and np.allclose(w2, w + 0.1 * mref, atol=1e-6)
    assert np.allclose(O.average_update(w, g, 0.1, 2), 0.9 * w + 0.1 * g / 2, atol=1e-6)


def test_batch_math():
    rng = np.random.default_rng(6)
    s = O.F(np.abs(rng.standard_normal((8, 20, 1, 2))) * 3)
    n = O.spec_rownorm(s)
    assert np.allclose(n.mean(1), 0, atol=1e-5) and np.allclose(n.std(1, ddof=1), 1, atol=1e-4)  # std(.,[],2): N-1
    lg = O.F(rng.standard_normal((17, 8)))
    assert np.allclose(O.aggregate_logits(lg, 3, 9, "max"), lg[2:9].max(0))
    assert np.allclose(O.aggregate_logits(lg, 3, 30, "mean"), lg[2:17].mean(0), atol=1e-6)  # clamp to F (:152)
    rgb = O.F(rng.integers(0, 256, (6, 5, 3, 2)))
    f = O.normalize_face(rgb, [131.1, 103.9, 91.5])
    grey = np.minimum(np.floor(0.2989 * rgb[:, :, 0] + 0.5870 * rgb[:, :, 1] + 0.1140 * rgb[:, :, 2] + 0.5), 255)
    assert np.allclose(f[:, :, 1], grey - 103.9, atol=1e-4)


def test_golden_fixtures():
    """the committed vectors (tests/golden/make_golden.py) still come out of the oracle."""
    G = np.load(GOLD)
    y = O.vl_nnconv(G["conv_x"], G["conv_f"], G["conv_b"], stride=2, pad=1, acc64=True)
    assert np.abs(y - G["conv_y"]).max() < 1e-6
    dx, df, db = O.vl_nnconv(G["conv_x"], G["conv_f"], G["conv_b"], G["conv_dzdy"], stride=2, pad=1, acc64=True)
    assert np.abs(dx - G["conv_dx"]).max() < 1e-6 and np.abs(df - G["conv_df"]).max() < 1e-5
    assert np.abs(db - G["conv_db"]).max() < 1e-5
    # fp32 path (MatConvNet algorithm shape) stays within 1e-4 of the fp64-accumulate vectors
    y32 = O.vl_nnconv(G["conv_x"], G["conv_f"], G["conv_b"], stride=2, pad=1, acc64=False)
    assert np.abs(y32 - G["conv_y"]).max() < 1e-4 * max(1, np.abs(G["conv_y"]).max())
    yb, mom = O.vl_nnbnorm(G["bn_x"], G["bn_g"], G["bn_b"], acc64=True)
    assert np.abs(yb - G["bn_y"]).max() < 1e-6 and np.abs(mom - G["bn_moments"]).max() < 1e-6
    dxb, dg, dbb, _ = O.vl_nnbnorm(G["bn_x"], G["bn_g"], G["bn_b"], G["bn_dzdy"], acc64=True)
    assert np.abs(dxb - G["bn_dx"]).max() < 1e-6 and np.abs(dg - G["bn_dg"]).max() < 1e-5
    assert np.array_equal(O.vl_nnpool(G["pool_x"], [3, 3], stride=2), G["pool_y"])
    assert np.array_equal(O.vl_nnpool(G["pool_x"], [3, 3], G["pool_dzdy"], stride=2), G["pool_dx"])
    assert abs(O.vl_nnsoftmaxceloss(G["loss_x"], G["loss_p"], temperature=2, logit_targets=True) - G["loss_y"]) < 1e-6
    assert np.abs(O.spec_rownorm(G["spec"]) - G["spec_norm"]).max() < 1e-6


def test_softmax_backward_matches_torch_autograd():
    """oracle vl_nnsoftmaxt backward (MatConvNet vl_nnsoftmax DZDY form) vs torch autograd."""
    import torch
    rng = np.random.default_rng(21)
    x = O.F(rng.standard_normal((2, 3, 8, 4)) * 2)
    d = O.F(rng.standard_normal((2, 3, 8, 4)))
    for T in (1.0, 2.0):
        xt = torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, requires_grad=True)
        y = torch.softmax(xt / T, dim=2)
        (y * torch.tensor(np.ascontiguousarray(d), dtype=torch.float64)).sum().backward()
        got = O.vl_nnsoftmaxt_backward(x, d, T)
        assert np.abs(got - xt.grad.numpy()).max() < 1e-6


def test_regression_losses_match_torch():
    """oracle vl_nneuclideanloss / vl_nnhuberloss ([EXT] mcnExtraLayers forms) vs torch + autograd."""
    import torch
    rng = np.random.default_rng(22)
    x = O.F(rng.standard_normal((1, 1, 8, 5)) * 2)
    t = O.F(rng.standard_normal((1, 1, 8, 5)) * 2)
    w = O.F(rng.uniform(0.5, 2, (1, 1, 1, 5)))
    wt = torch.tensor(w.ravel(), dtype=torch.float64)
    for kind, sigma in (("euclidean", 1.0), ("huber", 1.0), ("huber", 2.0)):
        xt = torch.tensor(x[0, 0].T.copy(), dtype=torch.float64, requires_grad=True)   # N x C
        tt = torch.tensor(t[0, 0].T.copy(), dtype=torch.float64)
        d = xt - tt
        if kind == "euclidean":
            per = 0.5 * d * d
        else:
            s2 = sigma * sigma
            per = torch.where(d.abs() > 1 / s2, d.abs() - 0.5 / s2, 0.5 * s2 * d * d)
        loss = (wt[:, None] * per).sum()
        loss.backward()
        assert abs(O.vl_nnregloss(x, t, kind=kind, sigma=sigma, instance_weights=w) - loss.item()) < 1e-5
        g = O.vl_nnregloss(x, t, np.ones(1, np.float32), kind=kind, sigma=sigma, instance_weights=w)
        assert np.abs(g[0, 0].T - xt.grad.numpy()).max() < 1e-6
        assert abs(O.vl_nnregloss(x, t, kind=kind, sigma=sigma) -
                   O.vl_nnregloss(x, t, kind=kind, sigma=sigma, instance_weights=np.ones(5))) < 1e-6


def test_run_spec_restatement_properties():
    """oracle runSpec [EXT]: a pure tone lands in the right bin, widths follow audSamp, pre-emphasis gain."""
    fs = 16000
    assert O.run_spec(np.zeros(int(O.aud_samples(300)))).shape == (512, 300, 1, 1)
    assert O.run_spec(np.zeros(int(O.aud_samples(400)))).shape == (512, 400, 1, 1)
    t = np.arange(8000) / fs
    f0 = 2000.0
    S = O.run_spec(np.sin(2 * np.pi * f0 * t))
    assert abs(int(S[:, 10, 0, 0].argmax()) - round(f0 / fs * 1024)) <= 1
    # DC is removed by the pre-emphasis up to 1 - alpha
    D = O.run_spec(np.ones(4000))
    win_sum = (0.54 - 0.46 * np.cos(2 * np.pi * np.arange(400) / 399)).sum()
    assert abs(D[0, 5, 0, 0] - 0.03 * win_sum) < 1e-3


def test_golden_extra_fixtures():
    """tests/golden/ops_extra.npz (regression losses, softmax derivative, runSpec, face crop/resize,
    logit aggregation) still comes out of the oracle."""
    E = np.load(os.path.join(os.path.dirname(GOLD), "ops_extra.npz"))
    one = np.ones(1, np.float32)
    assert abs(O.vl_nnregloss(E["reg_x"], E["reg_t"], kind="euclidean", instance_weights=E["reg_w"]) - E["euclid_y"]) < 1e-6
    assert np.abs(O.vl_nnregloss(E["reg_x"], E["reg_t"], one, kind="huber", instance_weights=E["reg_w"]) - E["huber_dx"]).max() < 1e-7
    assert np.abs(O.vl_nnsoftmaxt_backward(E["sm_x"], E["sm_dzdy"], 1.0) - E["sm_dx"]).max() < 1e-7
    assert np.abs(O.run_spec(E["wav"]) - E["wav_spec"]).max() < 1e-6
    assert np.array_equal(O.crop_resize_face(E["face_src"], (131.0912, 103.8827, 91.4953), (56, 56)), E["face_out"])
    for k, (a, b) in enumerate(zip(E["agg_first"], E["agg_last"])):
        assert np.array_equal(O.aggregate_logits(E["agg_logits"], int(a), int(b), "max"), E["agg_max_%d" % k])


def test_injected_gates_reproduce_the_oracles_own_backward():
    """oracle.graphs.backward(gates=...) -- the hook the full-size GPU tests use to re-run the oracle's backward pass
    with ANOTHER arithmetic's discrete decisions -- must be a no-op when it is handed the oracle's own decisions:
    routing tables from pool_argmax_codes (first maximum in column-major scan order, padding = -inf) and ReLU masks
    x > 0 give the derivatives of the plain backward pass, ties at exact zeros included."""
    from oracle import graphs as G
    rng = np.random.default_rng(3)
    for shape, pool, stride, pad in [((13, 11, 3, 2), (3, 3), (2, 2), (0, 0, 0, 0)),
                                     ((12, 12, 4, 2), (3, 3), (2, 2), (0, 1, 0, 1)),
                                     ((9, 8, 6, 2), (5, 3), (3, 2), (0, 0, 0, 0))]:
        x = O.F(np.maximum(rng.standard_normal(shape), 0))          # rectified: many exact ties at 0
        y = O.vl_nnpool(x, pool, stride=stride, pad=pad)
        dz = O.F(rng.standard_normal(y.shape))
        ref = O.vl_nnpool(x, pool, dz, stride=stride, pad=pad)
        code = G.pool_argmax_codes(x, pool, stride, pad)
        assert np.abs(G.pool_route(dz, code, x.shape, pool, stride, pad) - ref).max() < 1e-6
        pos = G.pool_positions(code, x.shape, pool, stride, pad)
        assert np.array_equal(x.ravel(order="F")[pos], y)
    # a whole (tiny) student: own gates injected == plain backward
    g = G.vggvox_student(100)
    if g is not None:
        P = G.make_params(g, 1)
        data, lgo, lab = G.spectrogram_batch(2, 100, 5)
        V = G.forward(g, {"data": data, "logitTarget": lgo, "maxLabel": lab}, P, mode="normal", acc64=True)
        gates = {}
        for l in g:
            if l.type == "relu":
                gates[l.name] = V[l.inputs[0]] > 0
            elif l.type == "pool" and l.attrs["method"] == "max":
                gates[l.name] = (G.pool_argmax_codes(V[l.inputs[0]], l.attrs["poolSize"], l.attrs["stride"],
                                                     l.attrs["pad"]), np.ones(V[l.outputs[0]].shape, bool))
        _, D0 = G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=True)
        _, D1 = G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=True, gates=gates)
        for k in D0:
            assert np.abs(D0[k] - D1[k]).max() <= 1e-6 * max(1.0, np.abs(D0[k]).max()), k


def test_dropout_restatement():
    """vl_nndropout = mask .* x both ways; the product's mask stream (Philox4x32-10, include/xmodal.h) is pinned on the
    generator's published known-answer vector (Random123 kat_vectors: counter 0, key 0), and the graph executor treats a
    dropout layer as the identity in test mode and as the given mask in training mode."""
    m = O.dropout_mask((4, 1, 1, 1), 0.0, 0, 0)          # rate 0: every element kept, scale 1
    assert np.array_equal(m.ravel(), np.ones(4, np.float32))
    # u = word >> 8: the first block of the stream for (seed 0, offset 0) is 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    words = np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], np.uint64)
    u = (words >> np.uint64(8)).astype(np.float32) / np.float32(16777216.0)
    for rate in (0.3, 0.5, 0.75, 0.9):
        want = np.where(u >= np.float32(rate), np.float32(1) / (np.float32(1) - np.float32(rate)), 0).astype(np.float32)
        assert np.array_equal(O.dropout_mask((4,), rate, 0, 0).ravel(), want), rate
    big = O.dropout_mask((64, 50, 3, 2), 0.5, 7, 0)
    assert set(np.unique(big)) == {0.0, 2.0} and abs(float((big > 0).mean()) - 0.5) < 0.02
    assert np.array_equal(O.dropout_mask((64, 50, 3, 2), 0.5, 7, 0), big)
    assert not np.array_equal(O.dropout_mask((64, 50, 3, 2), 0.5, 8, 0), big)
    # offset g advances the stream by whole blocks of four elements
    tail = O.dropout_mask((64 * 50 * 3 * 2 - 8,), 0.5, 7, 2)
    assert np.array_equal(tail, big.ravel(order="F")[8:])
    rng = np.random.default_rng(3)
    x = O.F(rng.standard_normal((64, 50, 3, 2)))
    assert np.array_equal(O.vl_nndropout(x, big), np.where(big > 0, 2 * x, 0))
    from oracle import graphs as G
    g = G.vggvox_student(100, dropout=0.5)
    P = G.make_params(g, 3)
    data, lgo, lab = G.spectrogram_batch(2, 100, 5)
    ins = {"data": data, "logitTarget": lgo, "maxLabel": lab}
    Vt = G.forward(g, ins, P, mode="test")
    V0 = G.forward(G.vggvox_student(100), ins, P, mode="test")
    assert np.array_equal(Vt["prediction"], V0["prediction"])
    shp = G.shapes(g, (512, 100, 1))
    masks = {"fc6_drop.mask": O.dropout_mask(shp["x_fc6"] + (2,), 0.5, 1), "fc7_drop.mask": O.dropout_mask(shp["x_fc7"] + (2,), 0.5, 2)}
    Vm = G.forward(g, dict(ins, **masks), P, mode="normal")
    assert np.array_equal(Vm["fc6_drop"], Vm["x_fc6"] * masks["fc6_drop.mask"])
    D, DP = G.backward(g, Vm, {"objective": np.float32(1)}, P, mode="normal")
    assert np.array_equal(D["x_fc7"], D["fc7_drop"] * masks["fc7_drop.mask"])


def test_resample_restatement_vs_scipy():
    """speed perturbation (getBatchEmoVoxCeleb.m:102-108): the restated resample(x, p, q) against scipy's
    resample_poly with the same Kaiser(5) window -- an independent implementation of the same polyphase recipe -- and
    against what a resampler must do: length ceil(Lx p / q), a band-limited tone comes out as the same tone on the new
    grid, speed factors 0.95 ... 1.05 as the provider draws them."""
    import scipy.signal
    rng = np.random.default_rng(4)
    for speed in (0.95, 0.9731, 1.0, 1.0417, 1.05):
        p, q = int(round(16000 / speed)), 16000
        x = rng.standard_normal(3000)
        y = O.resample(x, p, q)
        assert y.size == -(-x.size * p // q)
        import math
        g = math.gcd(p, q)
        ys = scipy.signal.resample_poly(x, p // g, q // g, window=("kaiser", 5.0))
        assert ys.size == y.size
        assert np.abs(y - ys).max() <= 2e-3 * np.abs(ys).max(), (speed, np.abs(y - ys).max())
        t = np.arange(4000)
        tone = np.sin(2 * np.pi * 0.05 * t)
        yt = O.resample(tone, p, q)
        want = np.sin(2 * np.pi * 0.05 * np.arange(yt.size) * q / p)
        assert np.abs(yt - want)[200:-200].max() < 2e-3, speed


def test_cpu_quota_reads_cgroup_v2_and_v1(tmp_path):
    """bench.py's cpu_baseline sizes its OpenMP team by min(affinity, cgroup CPU quota) (round-4 review: 128 pinned threads
    under a 16-core quota swung 2.4 x between boxes): cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`."""
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("1600000 100000\n")
    assert O.cpu_quota(str(v2)) == 16.0
    (v2 / "cpu.max").write_text("max 100000\n")
    assert O.cpu_quota(str(v2)) is None
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("250000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert O.cpu_quota(str(v1)) == 2.5
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert O.cpu_quota(str(v1)) is None
    assert O.cpu_quota(str(tmp_path / "nothing")) is None
    assert 1 <= len(O.baseline_cpus()) <= len(O.physical_core_cpus())
