"""GPU parity of the Gram route through the student's first layers (csrc/stem_pool_kernels.h, round 6):
conv1 -> bn1 -> relu1 -> pool1 (emoVoxCeleb/emoVoxZoo.m:50-62) backward WITHOUT a pass over conv1's output --
xm_stem_gram, xm_stem_gram_moments, xm_nnconv_backward_filter_bnrelupool_gram -- against the oracle's composition
vl_nnconv <- vl_nnbnorm <- vl_nnrelu <- vl_nnpool (fp64 accumulate), tolerance 1e-4 of the largest entry (north_star).
"""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import graphs as G

pytestmark = pytest.mark.gpu
TOL = 1e-4


def close(a, b, tol=TOL, what=""):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    err = float(np.abs(a - b).max()) if b.size else 0.0
    assert err <= tol * scale, "%s: max err %.3e > %.1e * %.3g" % (what, err, tol, scale)


def rnd(rng, *shape):
    return O.F(rng.standard_normal(shape))


def _kernels_run(L, fn):
    """names of the convolution kernels `fn` launched (profiler hooks of the library)"""
    import ctypes as C
    import torch
    L.xm_prof_enable(1)
    out = fn()
    torch.cuda.synchronize()
    L.xm_prof_enable(0)
    cap = 32
    keys, ms, fl, cnt = (C.c_int * cap)(), (C.c_double * cap)(), (C.c_double * cap)(), (C.c_longlong * cap)()
    n = L.xm_prof_collect(cap, keys, ms, fl, cnt)
    names = []
    for i in range(min(n, cap)):
        buf = C.create_string_buffer(128)
        L.xm_prof_kernel_name(keys[i], buf, 128)
        names.append(buf.value.decode())
    return out, names


def patches(x, FH, FW, stride, pad):
    """im2col of a single-channel input (+ a column of ones): [pixels of all samples][FH * FW + 1], float64"""
    H, W, _, N = x.shape
    sy, sx = (stride, stride) if np.isscalar(stride) else stride
    pt, pb, pl, pr = (pad,) * 4 if np.isscalar(pad) else pad
    xp = np.zeros((H + pt + pb, W + pl + pr, N), np.float64)
    xp[pt:pt + H, pl:pl + W] = x[:, :, 0, :]
    Ho, Wo = (H + pt + pb - FH) // sy + 1, (W + pl + pr - FW) // sx + 1
    cols = []
    for v in range(FW):
        for u in range(FH):
            cols.append(xp[u:u + sy * (Ho - 1) + 1:sy, v:v + sx * (Wo - 1) + 1:sx].reshape(-1))
    cols.append(np.ones(Ho * Wo * N))
    return np.stack(cols, 1), Ho, Wo


GRAM_CASES = [  # H, W, N, FH, FW, stride, pad
    (512, 60, 2, 7, 7, 2, 1),                    # the student's conv1, short clips
    (512, 42, 3, 7, 7, (2, 2), [1, 1, 1, 2]),    # 254-row columns: ragged last chunk, several samples
    (260, 37, 2, 5, 5, 1, 2),                    # unit stride, 25 taps (the ones column sits in the first column tile)
    (128, 40, 1, 8, 7, 2, 3),                    # 56 taps
]


@pytest.mark.parametrize("case", GRAM_CASES)
def test_stem_gram_and_moments(gpu, case):
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, N, FH, FW, stride, pad = case
    rng = np.random.default_rng(H + W + N + FH)
    x = rnd(rng, H, W, 1, N) + np.float32(0.3)
    P, Ho, Wo = patches(x, FH, FW, stride, pad)
    ref = P.T @ P
    R = FH * FW
    gram, names = _kernels_run(L, lambda: vl.stem_gram(vl.from_numpy(x), (FH, FW), stride=stride, pad=pad))
    assert gram is not None and any("stem_gram" in n for n in names), names
    g = gram.cpu().numpy().reshape(64, 64)
    assert np.array_equal(g, g.T)
    close(g[:R + 1, :R + 1], ref, 2e-6, what="gram")          # fp32 MFMA chains per wave, fp64 across
    assert g[R, R] == Ho * Wo * N
    assert not g[R + 1:].any() and not g[:, R + 1:].any()
    # batch moments of the convolution's output from G against vl_nnbnorm on the oracle's Y
    K = 24
    f, b = O.F(rng.standard_normal((FH, FW, 1, K)) * 0.2), rnd(rng, K)
    y = O.vl_nnconv(x, f, b, stride=stride, pad=pad, acc64=True)
    _, m_ref = O.vl_nnbnorm(y, O.F(np.ones(K)), O.F(np.zeros(K)), acc64=True)
    mo = vl.to_numpy(vl.stem_gram_moments(gram, vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1))))
    close(mo[:, 0], m_ref[:, 0], what="mean from G")
    assert np.abs(mo[:, 1] / m_ref[:, 1] - 1).max() <= 1e-5, "sigma from G"


STEM_POOL_CASES = [  # H, W, N, K, FH, FW, stride, pad, train, with conv bias
    (512, 60, 2, 96, 7, 7, 2, 1, True, True),                       # the student's conv1 -> bn1 -> relu1 -> pool1, short clips
    (512, 42, 3, 96, 7, 7, (2, 2), [1, 1, 1, 2], True, False),      # ragged last chunk, odd number of columns per sample
    (512, 44, 2, 80, 7, 7, 2, 1, False, True),                      # fewer filters than the 96-row tile, test-mode moments
    (260, 37, 2, 96, 5, 5, 1, 2, True, True),                       # unit stride: 260 x 37 outputs, odd pooled width
    (132, 20, 2, 40, 7, 7, 2, 1, True, True),                       # 64 rows: pooled columns of 31 (quads shifted back)
]


@pytest.mark.parametrize("case", STEM_POOL_CASES)
def test_conv_stem_wgrad_pool_gram(gpu, case):
    """Routing tables are the forward pass's own, with codes PLANTED at the first / last windows of the first / last
    planes (every corner of the window: the scatter touches both ends of the dz tile and of the pooled tensors), on a
    plateau (first maximum) and at dead windows (y_pool == 0: nothing is routed)."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, N, K, FH, FW, stride, pad, train, has_bias = case
    rng = np.random.default_rng(H + 5 * W + K + FH + int(train))
    x, f = rnd(rng, H, W, 1, N), O.F(rng.standard_normal((FH, FW, 1, K)) * 0.2)
    b = rnd(rng, K) if has_bias else None
    y = O.vl_nnconv(x, f, b, stride=stride, pad=pad, acc64=True)
    Ho, Wo = y.shape[:2]
    g, bb = O.F(rng.uniform(0.5, 1.5, K) * rng.choice([-1, 1], K)), rnd(rng, K)
    mom = None if train else O.F(np.stack([rng.standard_normal(K) * 0.3, rng.uniform(0.5, 1.5, K)], 1))
    yb, mref = O.vl_nnbnorm(y, g, bb, moments=mom, acc64=True)
    yr = np.maximum(yb, 0)
    yp = O.vl_nnpool(yr, [3, 3], stride=2, pad=0, method="max")
    code = G.pool_argmax_codes(yr, [3, 3], 2, 0)
    pHo, pWo = yp.shape[:2]
    # planted routes (the kernel takes the table as given): corners of the first and the last plane, all nine codes
    yp = yp.copy()
    for (c, n) in ((0, 0), (K - 1, N - 1)):
        for k, (hh, ww) in enumerate(((0, 0), (pHo - 1, pWo - 1), (pHo - 1, 0), (0, pWo - 1), (pHo // 2, pWo // 2),
                                      (15, 1), (16, 1), (pHo - 2, pWo - 1), (1, 0))):
            code[hh, ww, c, n] = (2 * k + 8) % 9
            yp[hh, ww, c, n] = 1.0
        code[16, 2:4, c, n] = 2                                    # rows 34 = 32 + 2: the element in front of a chunk
        code[15, 2:4, c, n] = 2                                    # row 32: first row of the next chunk
        yp[15:17, 2:4, c, n] = 1.0
        yp[3, min(3, pWo - 1), c, n] = 0.0                                       # a dead window next to live ones
    dz = rnd(rng, *yp.shape)
    dyr = G.pool_route(dz * (yp > 0), code, yr.shape, [3, 3], 2, 0)
    dx_ref, dg_ref, db_ref, _ = O.vl_nnbnorm(y, g, bb, dyr, moments=mom, acc64=True)
    _, df_ref, dbias_ref = O.vl_nnconv(x, f, b, dx_ref, stride=stride, pad=pad, acc64=True, no_der_data=True)

    xd, fd = vl.from_numpy(x), vl.from_numpy(f)
    bd = None if b is None else vl.from_numpy(b.reshape(K, 1))
    gd = vl.from_numpy(g.reshape(K, 1))
    mo = vl.from_numpy(mref if mom is None else mom)
    import torch
    am = torch.from_numpy(np.ascontiguousarray(code.ravel(order="F"))).cuda()
    ypd, dzd = vl.from_numpy(O.F(yp)), vl.from_numpy(dz)
    res, names = _kernels_run(L, lambda: vl.conv_backward_filter_bnrelupool_gram(
        xd, fd, bd, gd, mo, am, ypd, dzd, [3, 3], stride=stride, pad=pad, pool_stride=2, pool_pad=0, train=train))
    assert res is not None, "the fused kernel must cover this geometry"
    assert any("stem_wgrad_pool" in n for n in names), names
    assert any("stem_gram" in n for n in names) == train, names
    df, dbias, dg, db = res
    close(vl.to_numpy(df), df_ref, what="filter derivative")
    close(vl.to_numpy(dg).ravel(), dg_ref, what="dg")
    close(vl.to_numpy(db).ravel(), db_ref, what="db")
    if has_bias:
        # sum of DX: exactly zero in exact arithmetic for a train-mode bnorm -- compared on the scale of sum |DX|
        ref = dx_ref.astype(np.float64).sum((0, 1, 3))
        mag = np.abs(dx_ref.astype(np.float64)).sum((0, 1, 3)).max()
        assert np.abs(vl.to_numpy(dbias).ravel() - ref).max() <= 2e-6 * max(1.0, mag), "conv bias derivative"
    # a Gram matrix handed in gives the same bits as the one computed inside the call
    if train:
        gram = vl.stem_gram(xd, (FH, FW), stride=stride, pad=pad)
        df2, _, dg2, _ = vl.conv_backward_filter_bnrelupool_gram(
            xd, fd, bd, gd, mo, am, ypd, dzd, [3, 3], stride=stride, pad=pad, pool_stride=2, pool_pad=0, train=True, gram=gram)
        assert np.array_equal(vl.to_numpy(df2), vl.to_numpy(df)) and np.array_equal(vl.to_numpy(dg2), vl.to_numpy(dg))
    # two runs: bit-identical (the scatter's additions have a fixed order)
    res2 = vl.conv_backward_filter_bnrelupool_gram(xd, fd, bd, gd, mo, am, ypd, dzd, [3, 3], stride=stride, pad=pad,
                                                   pool_stride=2, pool_pad=0, train=train)
    assert np.array_equal(vl.to_numpy(res2[0]), vl.to_numpy(df))
    # shapes the kernel does not cover come back as None
    assert vl.conv_backward_filter_bnrelupool_gram(xd, fd, bd, gd, mo, am, ypd, dzd, [3, 3], stride=stride, pad=pad,
                                                   pool_stride=2, pool_pad=[0, 1, 0, 1], train=train) is None


FWD_CASES = [  # H, W, N, K, FH, pad, train, with conv bias
    (512, 60, 2, 96, 7, 1, True, True),                 # the student's conv1 -> bn1 -> relu1 -> pool1, short clips
    (512, 43, 3, 96, 7, [1, 1, 1, 2], True, False),     # odd number of output columns, three samples, no bias
    (512, 44, 2, 80, 7, 1, False, True),                # fewer filters than three row tiles, test-mode moments
    (256, 36, 2, 40, 6, [2, 3, 1, 1], True, True),      # one strip of window rows, six filter rows, other padding
    (512, 300, 1, 96, 7, 1, True, True),                # full-width spectrogram: many ring periods per segment
]


@pytest.mark.parametrize("case", FWD_CASES)
def test_conv_stem_bnorm_relu_pool_forward(gpu, case):
    """xm_nnconv_bnorm_relu_pool_forward (conv_stem_bnpool_fwd_kernel) against the oracle's composition: pooled output and
    moments at 1e-4; the routing table bit-exact wherever the oracle's decision is not within round-off of a tie, 255 exactly
    where the pooled value is zero; then the backward through that table (y_pool not given) against the oracle's."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, N, K, FH, pad, train, has_bias = case
    rng = np.random.default_rng(H + 3 * W + K + FH)
    x, f = rnd(rng, H, W, 1, N), O.F(rng.standard_normal((FH, 7, 1, K)) * 0.2)
    b = rnd(rng, K) if has_bias else None
    g, bb = O.F(rng.uniform(0.5, 1.5, K) * rng.choice([-1, 1], K)), rnd(rng, K)
    mom = None if train else O.F(np.stack([rng.standard_normal(K) * 0.3, rng.uniform(0.5, 1.5, K)], 1))
    y = O.vl_nnconv(x, f, b, stride=2, pad=pad, acc64=True)
    yb, mref = O.vl_nnbnorm(y, g, bb, moments=mom, acc64=True)
    yr = np.maximum(yb, 0)
    yp_ref = O.vl_nnpool(yr, [3, 3], stride=2, pad=0, method="max")
    code_ref = G.pool_argmax_codes(yr, [3, 3], 2, 0)
    xd, fd = vl.from_numpy(x), vl.from_numpy(f)
    bd = None if b is None else vl.from_numpy(b.reshape(K, 1))
    gd, bbd = vl.from_numpy(g.reshape(K, 1)), vl.from_numpy(bb.reshape(K, 1))
    res, names = _kernels_run(L, lambda: vl.conv_bnorm_relu_pool(xd, fd, bd, gd, bbd, [3, 3], stride=2, pad=pad, pool_stride=2,
                                                                 pool_pad=0, moments=None if mom is None else vl.from_numpy(mom)))
    assert res is not None, "the fused kernel must cover this geometry"
    assert any("bnpool_fwd" in n for n in names), names
    ypd, am, mo, gram = res
    yp = vl.to_numpy(ypd)
    close(yp, yp_ref, what="pooled output")
    m = vl.to_numpy(mo)
    close(m[:, 0], mref[:, 0], what="mean")
    assert np.abs(m[:, 1] / mref[:, 1] - 1).max() <= 1e-5
    code = am.cpu().numpy().reshape(yp_ref.shape, order="F")
    dead = yp == 0
    assert np.array_equal(code == 255, dead), "255 exactly at the closed windows"
    assert not (dead & (yp_ref > 1e-4 * max(1.0, float(np.abs(yb).max())))).any()
    # decisions: equal to the oracle's except within round-off of a tie (the value at the HIP choice is the window maximum)
    diff = (code != code_ref) & ~dead
    if diff.any():
        pos = G.pool_positions(np.where(dead, code_ref, code), yr.shape, [3, 3], 2, 0)
        at = yr.ravel(order="F")[pos]
        assert np.abs(at - yp_ref)[diff].max() <= 1e-4 * max(1.0, float(np.abs(yb).max())), "routing differs away from a tie"
        assert diff.mean() < 1e-3
    # backward through this table, y_pool not given
    dz = rnd(rng, *yp_ref.shape)
    code_use = np.where(dead, 0, code).astype(np.uint8)
    dyr = G.pool_route(dz * ~dead, code_use, yr.shape, [3, 3], 2, 0)
    dx_ref, dg_ref, db_ref, _ = O.vl_nnbnorm(y, g, bb, dyr, moments=mom, acc64=True)
    _, df_ref, _ = O.vl_nnconv(x, f, b, dx_ref, stride=2, pad=pad, acc64=True, no_der_data=True)
    out = vl.conv_backward_filter_bnrelupool_gram(xd, fd, bd, gd, mo, am, None, vl.from_numpy(dz), [3, 3], stride=2, pad=pad,
                                                  pool_stride=2, pool_pad=0, train=train, gram=gram)
    assert out is not None
    close(vl.to_numpy(out[0]), df_ref, what="filter derivative through the fused forward's table")
    close(vl.to_numpy(out[2]).ravel(), dg_ref, what="dg")
    close(vl.to_numpy(out[3]).ravel(), db_ref, what="db")
