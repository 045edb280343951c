"""GPU parity: every HIP operator (through the C ABI) against the CPU oracle on seeded inputs.

Tolerances (fp32 path, north_star: 1e-4): |hip - oracle| <= TOL * max(1, max|oracle|), where the
oracle is the fp64-accumulate restatement, so the bound covers the HIP kernel's own fp32
round-off (an fmaf chain in MFMA k-order) and nothing else.
"""
import numpy as np
import pytest

from oracle import oracle as O

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


# (H, W, C, N, FH, FW, FC, K, stride, pad, dilate)
CONV_CASES = [
    (12, 9, 5, 2, 3, 3, 5, 7, (1, 1), (1, 1, 1, 1), (1, 1)),       # 3x3 same
    (20, 17, 1, 3, 7, 7, 1, 10, (2, 2), (1, 1, 1, 1), (1, 1)),     # student conv1 shape (C=1)
    (21, 18, 6, 2, 5, 5, 6, 9, (2, 2), (1, 1, 1, 1), (1, 1)),      # student conv2 shape
    (9, 8, 16, 4, 9, 1, 16, 40, (1, 1), (0, 0, 0, 0), (1, 1)),     # fc6: 9x1 conv
    (1, 1, 48, 5, 1, 1, 48, 8, (1, 1), (0, 0, 0, 0), (1, 1)),      # fc7/fc8: 1x1 on 1x1
    (14, 14, 32, 2, 1, 1, 32, 64, (2, 2), (0, 0, 0, 0), (1, 1)),   # resnet 1x1 stride 2
    (15, 13, 3, 2, 7, 7, 3, 8, (2, 2), (3, 3, 3, 3), (1, 1)),      # teacher conv1
    (11, 10, 4, 2, 3, 2, 4, 6, (2, 3), (0, 1, 2, 0), (1, 1)),      # asymmetric everything
    (13, 12, 4, 2, 3, 3, 4, 6, (1, 1), (2, 2, 2, 2), (2, 2)),      # dilation
    (10, 9, 8, 2, 3, 3, 4, 6, (1, 1), (1, 1, 1, 1), (1, 1)),       # 2 filter groups
    (10, 11, 6, 2, 3, 3, 6, 5, (3, 2), (1, 0, 0, 1), (1, 1)),      # stride 3x2
    (40, 33, 20, 3, 3, 3, 20, 150, (1, 1), (1, 1, 1, 1), (1, 1)),  # multi-tile M and pixels
    (9, 8, 24, 4, 9, 1, 24, 80, (1, 1), (0, 0, 0, 0), (1, 1)),     # fc6 shape, K % 16 == 0: dgrad operand = plain transpose
    (1, 1, 100, 5, 1, 1, 100, 96, (1, 1), (0, 0, 0, 0), (1, 1)),   # fc7 shape, K % 16 == 0 (transpose_filter_kernel)
    (6, 6, 72, 2, 1, 1, 72, 64, (2, 2), (0, 0, 0, 0), (1, 1)),     # 1x1 stride 2, transposed operand + untouched pixels
    (1, 1, 512, 40, 1, 1, 512, 256, (1, 1), (0, 0, 0, 0), (1, 1)),  # fc7-like: forward AND dgrad through fc_skinny4_kernel
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_backward(gpu, case):
    from mcncrossmodalemotions_amd import vl
    H, W, C, N, FH, FW, FC, K, stride, pad, dil = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x, f, b = rnd(rng, H, W, C, N), rnd(rng, FH, FW, FC, K), rnd(rng, K)
    y_ref = O.vl_nnconv(x, f, b, stride=stride, pad=pad, dilate=dil, acc64=True)
    xd, fd, bd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1))
    y = vl.vl_nnconv(xd, fd, bd, stride=stride, pad=pad, dilate=dil)
    close(vl.to_numpy(y), y_ref, what="conv fwd")
    dzdy = rnd(rng, *y_ref.shape)
    dx_ref, df_ref, db_ref = O.vl_nnconv(x, f, b, dzdy, stride=stride, pad=pad, dilate=dil,
                                         acc64=True)
    dx, df, db = vl.vl_nnconv(xd, fd, bd, vl.from_numpy(dzdy), stride=stride, pad=pad, dilate=dil)
    close(vl.to_numpy(dx), dx_ref, what="conv dx")
    close(vl.to_numpy(df), df_ref, what="conv df")
    close(vl.to_numpy(db).ravel(), db_ref, what="conv db")
    # dX accumulated onto an existing derivative in the dgrad epilogue (xm_nnconv_backward_accum)
    acc = rnd(rng, H, W, C, N)
    dx2, _, _ = vl.vl_nnconv(xd, fd, bd, vl.from_numpy(dzdy), stride=stride, pad=pad, dilate=dil,
                             no_der_filters=True, dx_accum=vl.from_numpy(acc))
    close(vl.to_numpy(dx2), dx_ref + acc, what="conv dx + accum")


# (H, W, C, N, K, activation): 1x1 filters over few output values -> fc_skinny_kernel (SE gates, classifier, fc8)
SKINNY_CASES = [(1, 1, 2048, 32, 128, "relu"), (1, 1, 128, 32, 2048, "sigmoid"), (1, 1, 2048, 32, 8, None),
                (1, 8, 1024, 32, 8, None), (1, 1, 77, 5, 3, "relu"), (2, 3, 40, 7, 5, "sigmoid"),
                (1, 1, 16, 128, 256, "sigmoid"), (1, 1, 300, 33, 9, None),
                # four rows per block (fc_skinny4_kernel): the student's fc7, SE gates at 256 faces, a ragged pixel chunk
                (1, 1, 4096, 32, 1024, None), (1, 1, 512, 256, 32, "relu"), (1, 1, 64, 300, 128, "sigmoid")]


@pytest.mark.parametrize("case", SKINNY_CASES)
def test_conv_skinny_fc(gpu, case):
    """the skinny fully-connected path of vl_nnconv (+ fused vl_nnrelu / vl_nnsigmoid) against the oracle"""
    from mcncrossmodalemotions_amd import vl
    H, W, C, N, K, act = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x, f, b = rnd(rng, H, W, C, N), O.F(rng.standard_normal((1, 1, C, K)) / np.sqrt(C)), rnd(rng, K)
    y_ref = O.vl_nnconv(x, f, b, acc64=True)
    if act == "relu":
        y_ref = O.vl_nnrelu(y_ref)
    elif act == "sigmoid":
        y_ref = O.vl_nnsigmoid(y_ref)
    y = vl.vl_nnconv(vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1)), relu=act == "relu",
                     sigmoid=act == "sigmoid")
    close(vl.to_numpy(y), y_ref, 2e-6, what="skinny fc %s" % (act,))
    # no bias
    y0 = vl.vl_nnconv(vl.from_numpy(x), vl.from_numpy(f), None)
    close(vl.to_numpy(y0), O.vl_nnconv(x, f, None, acc64=True), 2e-6, what="skinny fc, no bias")
    # folded test-mode bnorm (scale / shift) on the skinny path: the SE squeeze taken through the projection
    sc, sh = O.F(rng.uniform(0.5, 1.5, K)), rnd(rng, K)
    ys = vl.vl_nnconv(vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1)),
                      scale=vl.from_numpy(sc.reshape(K, 1)), shift=vl.from_numpy(sh.reshape(K, 1)), relu=act == "relu")
    ref = O.vl_nnconv(x, f, b, acc64=True) * sc.reshape(1, 1, K, 1) + sh.reshape(1, 1, K, 1)
    close(vl.to_numpy(ys), np.maximum(ref, 0) if act == "relu" else ref, 3e-6, what="skinny fc, scale / shift")


def test_conv_fused_sigmoid_general_path(gpu):
    """XM_FUSE_SIGMOID on a geometry the skinny kernel does not take (3x3): convolution, then sigmoid in place"""
    from mcncrossmodalemotions_amd import vl
    rng = np.random.default_rng(77)
    x, f, b = rnd(rng, 9, 7, 6, 3), rnd(rng, 3, 3, 6, 20), rnd(rng, 20)
    y_ref = O.vl_nnsigmoid(O.vl_nnconv(x, f, b, pad=1, acc64=True))
    y = vl.vl_nnconv(vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(20, 1)), pad=1, sigmoid=True)
    close(vl.to_numpy(y), y_ref, 1e-5, what="conv + sigmoid")


WGRAD_PATCH_CASES = [  # W, C, N, K
    (17, 20, 5, 70),      # the student's 30-row grid; channels / filters that fill neither a tile nor a 9-tap group
    (17, 128, 8, 192),    # the narrow student's conv3: two filter-row tiles, nine column tiles, split over the columns
    (9, 16, 8, 130),      # ragged filter tile (130 = 128 + 2), few columns per sample
    (40, 3, 2, 8),        # wide, three channels (27 columns of the derivative)
]


@pytest.mark.parametrize("case", WGRAD_PATCH_CASES)
def test_conv_wgrad_patch_kernel(gpu, case):
    """Filter derivative of 3 x 3 / stride 1 / pad 1 layers over 30-row grids through conv_wgrad_patch_kernel (one output
    column per stage, the input patch under it staged once in LDS, the nine taps read from it) against the oracle, next to
    the generic kernel on the same operands; the profiler hooks prove which kernel ran."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    W, C, N, K = case
    rng = np.random.default_rng(W * 7 + C * 3 + N + K)
    x, f, b = rnd(rng, 30, W, C, N), rnd(rng, 3, 3, C, K), rnd(rng, K)
    dzdy = rnd(rng, 30, W, K, N)
    dzdy[rng.random(dzdy.shape) < 0.3] = 0                     # a ReLU mask's zeros
    _, df_ref, db_ref = O.vl_nnconv(x, f, b, dzdy, pad=1, acc64=True)
    xd, fd, bd, dd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1)), vl.from_numpy(dzdy)
    old = L.xm_debug_force_wgrad_patch(1)
    try:
        (_, df, db), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, dd, pad=1, no_der_data=True))
        assert ("conv_wgrad_patch_kernel<30>" in names) == (N * W >= 64), names
        close(vl.to_numpy(df), df_ref, what="patch wgrad")
        close(vl.to_numpy(db).ravel(), db_ref.ravel(), what="dzdb next to it")
        L.xm_debug_force_wgrad_patch(0)
        (_, df0, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, dd, pad=1, no_der_data=True))
        assert not any("patch" in n for n in names), names
        close(vl.to_numpy(df0), df_ref, what="generic wgrad")
        # other geometries never take it: 28 rows, stride 2, pad 0
        L.xm_debug_force_wgrad_patch(1)
        x2 = vl.from_numpy(rnd(rng, 28, W, C, N))
        y2 = vl.vl_nnconv(x2, fd, bd, pad=1)
        _, names = _kernels_run(L, lambda: vl.vl_nnconv(x2, fd, bd, vl.from_numpy(rnd(rng, *y2.shape)), pad=1, no_der_data=True))
        assert not any("patch" in n for n in names), names
    finally:
        L.xm_debug_force_wgrad_patch(old)


WGRAD_PATCH_S2_CASES = [  # H, W, C, N, K, pad (t b l r)
    (126, 25, 7, 3, 70, (1, 1, 1, 1)),      # the student's conv2 rows (62 output rows: segments of 32 + 30), ragged channels / filters
    (30, 21, 20, 8, 130, (1, 1, 1, 1)),     # 14 output rows: one segment, most of it zero padding of the reduction
    (126, 17, 96, 4, 256, (1, 1, 1, 1)),    # conv2's channels and filters: 19 column tiles, 2 filter tiles, channel groups of 7
    (64, 12, 5, 13, 33, (1, 2, 0, 1)),      # no left padding, other bottom / right padding
    (68, 11, 3, 8, 8, (2, 1, 3, 0)),        # two rows / three columns of padding in front
]


@pytest.mark.parametrize("case", WGRAD_PATCH_S2_CASES)
def test_conv_wgrad_patch_s2_kernel(gpu, case):
    """Filter derivative of 5 x 5 / stride 2 layers (the student's conv2) through conv_wgrad_patch_s2_kernel -- a stage is a
    32-row segment of one output column, the input patch under it is staged once in LDS and the 25 taps read from it --
    against the oracle, next to the generic kernel on the same operands; the profiler hooks prove which kernel ran."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, C, N, K, pad = case
    rng = np.random.default_rng(H * 7 + W * 3 + C + N + K)
    x, f, b = rnd(rng, H, W, C, N), rnd(rng, 5, 5, C, K), rnd(rng, K)
    y = O.vl_nnconv(x, f, b, stride=2, pad=pad)
    dzdy = rnd(rng, *y.shape)
    dzdy[rng.random(dzdy.shape) < 0.3] = 0                     # a ReLU mask's zeros
    _, df_ref, db_ref = O.vl_nnconv(x, f, b, dzdy, stride=2, pad=pad, acc64=True)
    xd, fd, bd, dd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1)), vl.from_numpy(dzdy)
    old = L.xm_debug_force_wgrad_patch_s2(1)
    try:
        (_, df, db), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, dd, stride=2, pad=pad, no_der_data=True))
        assert "conv_wgrad_patch_s2_kernel<5, 2>" in names, names
        close(vl.to_numpy(df), df_ref, what="patch wgrad (5 x 5 / 2)")
        close(vl.to_numpy(db).ravel(), db_ref.ravel(), what="dzdb next to it")
        L.xm_debug_force_wgrad_patch_s2(0)
        (_, df0, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, dd, stride=2, pad=pad, no_der_data=True))
        assert not any("patch" in n for n in names), names
        close(vl.to_numpy(df0), df_ref, what="generic wgrad")
        # other geometries never take it: odd row count, stride 1
        L.xm_debug_force_wgrad_patch_s2(1)
        x2 = vl.from_numpy(rnd(rng, H + 1, W, C, N))
        y2 = vl.vl_nnconv(x2, fd, bd, stride=2, pad=pad)
        _, names = _kernels_run(L, lambda: vl.vl_nnconv(x2, fd, bd, vl.from_numpy(rnd(rng, *y2.shape)), stride=2, pad=pad,
                                                        no_der_data=True))
        assert not any("patch" in n for n in names), names
    finally:
        L.xm_debug_force_wgrad_patch_s2(old)


DGRAD_S2_CASES = [  # H, W, C, N, K, pad (t b l r)
    (126, 13, 96, 2, 16, (1, 1, 1, 1)),     # the student's conv2 rows and channels; columns of both parities, a lone last column
    (30, 21, 20, 3, 24, (1, 1, 1, 1)),      # 15 row pairs of 64, 20 of 96 channel rows
    (126, 9, 7, 2, 8, (1, 1, 0, 1)),        # no left padding: the column classes swap
    (62, 12, 100, 2, 16, (1, 1, 2, 1)),     # 100 channels: two channel blocks; two columns of padding in front
    (134, 10, 5, 1, 8, (1, 1, 1, 1)),       # 67 row pairs: two row blocks
]


@pytest.mark.parametrize("case", DGRAD_S2_CASES)
def test_conv_dgrad_s2_kernel(gpu, case):
    """dgrad of 5 x 5 / stride 2 layers (the student's conv2) through conv_dgrad_s2_kernel -- a wave owns one output column and
    32 row PAIRS, both row parities accumulate in it and leave as 8-byte stores -- against the oracle, next to the merged
    stride-parity launch on the same operands; the profiler hooks prove which kernel ran."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, C, N, K, pad = case
    rng = np.random.default_rng(H * 7 + W * 3 + C + N + K)
    x, f = rnd(rng, H, W, C, N), rnd(rng, 5, 5, C, K)
    y = O.vl_nnconv(x, f, None, stride=2, pad=pad)
    dzdy = rnd(rng, *y.shape)
    dx_ref, _, _ = O.vl_nnconv(x, f, None, dzdy, stride=2, pad=pad, acc64=True)
    xd, fd, dd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(dzdy)
    old = L.xm_debug_force_dgrad_s2(1)
    try:
        (dx, _, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, None, dd, stride=2, pad=pad, no_der_filters=True))
        assert "conv_dgrad_s2_kernel<1>" in names, names
        close(vl.to_numpy(dx), dx_ref, what="dgrad (5 x 5 / 2), both row parities per wave")
        L.xm_debug_force_dgrad_s2(0)
        (dx0, _, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, None, dd, stride=2, pad=pad, no_der_filters=True))
        assert not any("dgrad_s2" in n for n in names), names
        close(vl.to_numpy(dx0), dx_ref, what="merged stride-parity dgrad")
        # the accumulating epilogue (derivative sums at forks) keeps the implicit-GEMM path
        L.xm_debug_force_dgrad_s2(1)
        acc = rnd(rng, H, W, C, N)
        (dx2, _, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, None, dd, stride=2, pad=pad, no_der_filters=True,
                                                                  dx_accum=vl.from_numpy(acc)))
        assert not any("dgrad_s2" in n for n in names), names
        close(vl.to_numpy(dx2), dx_ref + acc, what="dgrad + accum")
    finally:
        L.xm_debug_force_dgrad_s2(old)


# (H, W, N, K, pad): 7x7 / stride 2 on one channel (the student's first layer)
STEM_CASES = [(512, 60, 2, 96, (1, 1, 1, 1)), (131, 45, 3, 96, (1, 1, 1, 1)), (64, 33, 2, 64, (3, 3, 3, 3)),
              (40, 41, 2, 33, (0, 0, 0, 0)), (29, 23, 1, 7, (2, 1, 0, 3)), (300, 18, 1, 96, (1, 0, 1, 0))]


@pytest.mark.parametrize("case", STEM_CASES)
def test_conv_stem7(gpu, case):
    """the student's first layer geometry (7x7 / 2, C = 1, K = 49 padded to 64): plain and fused epilogue, odd and
    even output heights, ragged tiles in both directions, K below / at the 96-row tile -- against the oracle"""
    from mcncrossmodalemotions_amd import vl
    H, W, N, K, pad = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x, f, b = rnd(rng, H, W, 1, N), O.F(rng.standard_normal((7, 7, 1, K)) * 0.2), rnd(rng, K)
    sc, sh = O.F(rng.uniform(0.5, 1.5, K)), rnd(rng, K)
    y_ref = O.vl_nnconv(x, f, b, stride=2, pad=pad, acc64=True)
    xd, fd, bd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(-1, 1))
    close(vl.to_numpy(vl.vl_nnconv(xd, fd, bd, stride=2, pad=pad)), y_ref, 1e-5, "stem")
    close(vl.to_numpy(vl.vl_nnconv(xd, fd, None, stride=2, pad=pad)), O.vl_nnconv(x, f, None, stride=2, pad=pad, acc64=True),
          1e-5, "stem, no bias")
    yf = vl.vl_nnconv(xd, fd, bd, stride=2, pad=pad, scale=vl.from_numpy(sc.reshape(-1, 1)),
                      shift=vl.from_numpy(sh.reshape(-1, 1)), relu=True)
    close(vl.to_numpy(yf), np.maximum(y_ref * sc.reshape(1, 1, K, 1) + sh.reshape(1, 1, K, 1), 0), 1e-5, "stem fused")


# enough pixels per stride-parity class for the merged single-launch dgrad (conv_gemm_multi_kernel)
MERGED_DGRAD_CASES = [
    (200, 200, 8, 4, 3, 3, 8, 8, (2, 2), (1, 1, 1, 1), (1, 1)),     # 4 classes, 3x3
    (182, 150, 6, 5, 5, 5, 6, 12, (2, 2), (1, 1, 1, 1), (1, 1)),    # the student's conv2 pattern (5x5 / 2)
    (260, 140, 4, 4, 3, 3, 4, 6, (2, 1), (1, 0, 1, 1), (1, 1)),     # 2 classes (stride 2 x 1)
    (150, 150, 5, 3, 4, 4, 5, 7, (3, 3), (1, 2, 0, 1), (1, 1)),     # 9 classes -> per-class launches
]


@pytest.mark.parametrize("case", MERGED_DGRAD_CASES)
def test_conv_strided_dgrad_large(gpu, case):
    test_conv_forward_backward(gpu, case)


def test_conv_strided_dgrad_merged_all_configs(gpu):
    """the merged strided dgrad (all stride-parity classes in one launch, conv_gemm_multi_kernel) with every register-staged
    tile configuration forced, incl. the eight-wave one, against the oracle"""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, C, N, FH, FW, FC, K, stride, pad, _ = MERGED_DGRAD_CASES[1]
    rng = np.random.default_rng(5)
    x, f = rnd(rng, H, W, C, N), rnd(rng, FH, FW, FC, K)
    y_ref = O.vl_nnconv(x, f, None, stride=stride, pad=pad, acc64=True)
    dzdy = rnd(rng, *y_ref.shape)
    dx_ref, _, _ = O.vl_nnconv(x, f, None, dzdy, stride=stride, pad=pad, acc64=True, no_der_filters=True)
    xd, fd, dd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(dzdy)
    old_h = L.xm_debug_force_conv_halo(0)
    try:
        for cfg in range(L.xm_debug_num_conv_cfgs() - 4):
            L.xm_debug_force_conv_cfg(cfg)
            (dx, _, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, None, dd, stride=stride, pad=pad, no_der_filters=True))
            assert any(n.startswith("conv_gemm_multi_kernel") for n in names), (cfg, names)
            close(vl.to_numpy(dx), dx_ref, what="merged dgrad, cfg %d" % cfg)
    finally:
        L.xm_debug_force_conv_cfg(-1)
        L.xm_debug_force_conv_halo(old_h)


def test_conv_random_geometries(gpu):
    """seeded fuzz over filter size / stride / padding / dilation / groups / odd sizes: forward, dX, dF, dB
    against the fp64-accumulate oracle."""
    from mcncrossmodalemotions_amd import vl
    rng = np.random.default_rng(20260928)
    done = 0
    while done < 28:
        FH, FW = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        sy, sx = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        dy, dx = (int(rng.integers(1, 3)), int(rng.integers(1, 3))) if rng.random() < 0.25 else (1, 1)
        pad = tuple(int(v) for v in rng.integers(0, 3, 4))
        G = int(rng.choice([1, 1, 1, 2]))
        FC = int(rng.integers(1, 9))
        C, K = FC * G, int(rng.integers(1, 6)) * G
        H, W, N = int(rng.integers(6, 40)), int(rng.integers(6, 40)), int(rng.integers(1, 4))
        if H + pad[0] + pad[1] < (FH - 1) * dy + 1 or W + pad[2] + pad[3] < (FW - 1) * dx + 1:
            continue
        if FH * FW > 63:
            continue
        case = (H, W, C, N, FH, FW, FC, K, (sy, sx), pad, (dy, dx))
        x, f, b = rnd(rng, H, W, C, N), rnd(rng, FH, FW, FC, K), rnd(rng, K)
        y_ref = O.vl_nnconv(x, f, b, stride=(sy, sx), pad=pad, dilate=(dy, dx), acc64=True)
        dzdy = rnd(rng, *y_ref.shape)
        dx_ref, df_ref, db_ref = O.vl_nnconv(x, f, b, dzdy, stride=(sy, sx), pad=pad, dilate=(dy, dx), acc64=True)
        xd, fd, bd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1))
        close(vl.to_numpy(vl.vl_nnconv(xd, fd, bd, stride=(sy, sx), pad=pad, dilate=(dy, dx))), y_ref,
              what="fuzz fwd %s" % (case,))
        gx, gf, gb = vl.vl_nnconv(xd, fd, bd, vl.from_numpy(dzdy), stride=(sy, sx), pad=pad, dilate=(dy, dx))
        close(vl.to_numpy(gx), dx_ref, what="fuzz dx %s" % (case,))
        close(vl.to_numpy(gf), df_ref, what="fuzz df %s" % (case,))
        close(vl.to_numpy(gb).ravel(), db_ref, what="fuzz db %s" % (case,))
        done += 1


def test_conv_all_tile_configs(gpu):
    """every templated tile configuration must give the same answer."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    rng = np.random.default_rng(7)
    H, W, C, N, K = 19, 15, 12, 3, 100
    x, f, b = rnd(rng, H, W, C, N), rnd(rng, 3, 3, C, K), rnd(rng, K)
    y_ref = O.vl_nnconv(x, f, b, pad=1, acc64=True)
    dzdy = rnd(rng, *y_ref.shape)
    dx_ref, df_ref, _ = O.vl_nnconv(x, f, b, dzdy, pad=1, acc64=True)
    xd, fd, bd, dd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1)), vl.from_numpy(dzdy)
    try:
        for cfg in range(L.xm_debug_num_conv_cfgs()):
            L.xm_debug_force_conv_cfg(cfg)
            close(vl.to_numpy(vl.vl_nnconv(xd, fd, bd, pad=1)), y_ref, what="cfg%d fwd" % cfg)
            dx, df, _ = vl.vl_nnconv(xd, fd, bd, dd, pad=1)
            close(vl.to_numpy(dx), dx_ref, what="cfg%d dx" % cfg)
            close(vl.to_numpy(df), df_ref, what="cfg%d df" % cfg)
    finally:
        L.xm_debug_force_conv_cfg(-1)


# (H, W, C, N, FH, K, stride, pad, bias offset): Ho*Wo % 4 == 0 -> partial sums in the GEMM epilogue; otherwise (and
# with split-K / groups / skinny layers) the moments come from a pass over Y -- same contract either way
MOMENT_CASES = [(19, 15, 12, 3, 3, 100, 1, 1, 0.0),      # 285 pixels / sample: scalar-store epilogue -> fallback pass
                (20, 16, 12, 3, 3, 100, 1, 1, 0.0),      # 320 pixels / sample: fused partial sums, M and pixel tails
                (20, 16, 12, 3, 3, 100, 1, 1, 25.0),     # mean >> sigma: cancellation in E[y^2] - mean^2
                (41, 32, 1, 3, 7, 24, 2, 1, 0.0),        # student conv1 shape (C = 1, stride 2)
                (8, 8, 64, 4, 1, 48, 1, 0, 0.0),         # 1x1 (LDS-DMA eligible: must take the register-staged kernel)
                (1, 8, 256, 4, 1, 40, 1, 0, 0.0),        # FC-shaped, few pixels
                (1, 1, 64, 6, 1, 5, 1, 0, 0.0)]          # skinny FC -> fallback pass


@pytest.mark.parametrize("case", MOMENT_CASES)
def test_conv_forward_with_batch_moments(gpu, case):
    """xm_nnconv_forward_moments: Y = vl_nnconv(X, F, B) plus the batch moments vl_nnbnorm(Y, ...) computes first
    ([mean, sqrt(var + eps)], biased variance), for every tile configuration and for a forced split-K launch."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, C, N, FH, K, stride, pad, off = case
    rng = np.random.default_rng(H * 131 + W * 17 + K)
    x, f = rnd(rng, H, W, C, N), rnd(rng, FH, FH, C, K)
    b = O.F(rng.standard_normal(K) + off)
    y_ref = O.vl_nnconv(x, f, b, stride=stride, pad=pad, acc64=True)
    _, m_ref = O.vl_nnbnorm(y_ref, O.F(np.ones(K)), O.F(np.zeros(K)), acc64=True)
    xd, fd, bd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1))

    def check(tag):
        mo = vl.mat_empty(K, 2, device=xd.device)
        mo.fill_(float("nan"))
        y = vl.vl_nnconv(xd, fd, bd, stride=stride, pad=pad, moments_out=mo)
        close(vl.to_numpy(y), y_ref, what="conv fwd " + tag)
        m = vl.to_numpy(mo)
        close(m[:, 0], m_ref[:, 0], what="mean " + tag)
        # sigma relative to ITS OWN size (a mean of 25 must not hide a wrong sigma of 1)
        assert np.abs(m[:, 1] / m_ref[:, 1] - 1).max() <= 1e-4, (tag, float(np.abs(m[:, 1] / m_ref[:, 1] - 1).max()))
        # identical to what vl_nnbnorm computes from Y itself, to fp32 round-off
        _, m2 = vl.vl_nnbnorm(y, vl.from_numpy(O.F(np.ones((K, 1)))), vl.from_numpy(O.F(np.zeros((K, 1)))))
        assert np.abs(m / vl.to_numpy(m2) - 1)[:, 1].max() <= 2e-5, tag
    try:
        for cfg in range(L.xm_debug_num_conv_cfgs()):
            L.xm_debug_force_conv_cfg(cfg)
            check("cfg%d" % cfg)
        L.xm_debug_force_conv_cfg(-1)
        check("auto")
        old = L.xm_debug_force_conv_splits(3)
        try:
            check("split-K")
        finally:
            L.xm_debug_force_conv_splits(old)
    finally:
        L.xm_debug_force_conv_cfg(-1)


# (H, W, C, N, K, pad): 3 x 3, unit stride -> conv_halo_kernel (forward: C % 8 == 0; dgrad: K % 8 == 0)
HALO_CASES = [(30, 17, 16, 3, 40, (1, 1, 1, 1)),      # student conv3-5 geometry, M tail
              (14, 14, 8, 5, 136, (1, 1, 1, 1)),      # tiles straddle samples (196 pixels each)
              (7, 7, 24, 9, 64, (1, 1, 1, 1)),        # a 128-pixel tile covers three samples
              (56, 56, 8, 2, 64, (1, 1, 1, 1)),       # long columns
              (12, 20, 16, 2, 24, (0, 0, 0, 0)),      # no padding: output smaller than the input
              (13, 11, 8, 3, 16, (2, 0, 1, 2)),       # asymmetric padding
              (9, 6, 32, 4, 8, (1, 1, 1, 1)),
              (7, 7, 128, 9, 64, (1, 1, 1, 1)),       # few tiles, 16 stages: split-K slabs + combine kernel
              (14, 14, 64, 3, 136, (1, 1, 1, 1)),
              (30, 17, 96, 2, 96, (1, 1, 1, 1)),      # 96 rows both ways: the 3 x 1 wave-tile variant
              (15, 9, 192, 3, 192, (1, 1, 1, 1))]


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


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv_halo_kernel(gpu, case):
    """3 x 3 unit-stride convolutions through the halo-patch kernel (input patch of 8 channels staged once in LDS, the
    nine taps read from it): forward incl. the fused epilogue and the batch moments, and dgrad, against the oracle; the
    profiler hooks prove which kernel ran."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, C, N, K, pad = case
    rng = np.random.default_rng(H * 7 + W * 3 + C + K)
    x, f, b = rnd(rng, H, W, C, N), rnd(rng, 3, 3, C, K), rnd(rng, K)
    y_ref = O.vl_nnconv(x, f, b, pad=pad, acc64=True)
    dzdy = rnd(rng, *y_ref.shape)
    dx_ref, df_ref, _ = O.vl_nnconv(x, f, b, dzdy, pad=pad, acc64=True)
    xd, fd, bd, dd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1)), vl.from_numpy(dzdy)
    old = L.xm_debug_force_conv_halo(1 + (H + K) % 2)      # 128-row / 96-row variant (the latter only where M % 96 == 0)
    try:
        y, names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, pad=pad))
        assert any("halo" in n for n in names), names
        close(vl.to_numpy(y), y_ref, what="halo fwd")
        (dx, df, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, dd, pad=pad))
        assert any("halo" in n for n in names) == (K % 8 == 0), names
        close(vl.to_numpy(dx), dx_ref, what="halo dgrad")
        close(vl.to_numpy(df), df_ref, what="wgrad next to halo dgrad")
        # accumulate-into-dx epilogue
        acc = rnd(rng, H, W, C, N)
        dx2, _, _ = vl.vl_nnconv(xd, fd, bd, dd, pad=pad, no_der_filters=True, dx_accum=vl.from_numpy(acc))
        close(vl.to_numpy(dx2), dx_ref + acc, what="halo dgrad + accum")
        # fused epilogue (test-mode bnorm fold + residual + relu) and batch moments
        sc, sh = O.F(rng.uniform(0.5, 1.5, K)), rnd(rng, K)
        res = rnd(rng, *y_ref.shape)
        ref = np.maximum(y_ref * sc.reshape(1, 1, K, 1) + sh.reshape(1, 1, K, 1) + res, 0)
        yf = vl.vl_nnconv(xd, fd, bd, pad=pad, scale=vl.from_numpy(sc.reshape(K, 1)), shift=vl.from_numpy(sh.reshape(K, 1)),
                          residual=vl.from_numpy(res), relu=True)
        close(vl.to_numpy(yf), ref, what="halo fused epilogue")
        mo = vl.mat_empty(K, 2, device=xd.device)
        ym = vl.vl_nnconv(xd, fd, bd, pad=pad, moments_out=mo)
        _, m_ref = O.vl_nnbnorm(y_ref, O.F(np.ones(K)), O.F(np.zeros(K)), acc64=True)
        close(vl.to_numpy(ym), y_ref, what="halo fwd + moments")
        m = vl.to_numpy(mo)
        close(m[:, 0], m_ref[:, 0], what="halo mean")
        assert np.abs(m[:, 1] / m_ref[:, 1] - 1).max() <= 1e-4
    finally:
        L.xm_debug_force_conv_halo(old)


STEM_CASES = [  # H, W, N, K, FH, FW, stride, pad
    (512, 60, 2, 96, 7, 7, 2, 1),             # the student's conv1 on short spectrograms (254 x 28 outputs)
    (512, 42, 3, 96, 7, 7, (2, 2), [1, 1, 1, 2]),   # tiles that straddle columns AND samples, ragged last tile
    (256, 20, 2, 64, 5, 5, 1, 2),             # unit stride, fewer filters than the 96-row tile, fewer taps than 8 x 7
    (504, 31, 2, 96, 8, 6, (2, 1), [2, 2, 1, 1]),   # 8 filter rows, stride 2 x 1, bottom padding read from zero rows
]


@pytest.mark.parametrize("case", STEM_CASES)
def test_conv_stem_kernel(gpu, case):
    """single-channel stem (conv_stem_kernel: persistent blocks, filter bank resident in LDS, source columns staged as
    a patch, stores draining under the next tile's MFMAs) against the oracle and against the implicit-GEMM kernel it
    replaces, incl. the batch moments from its epilogue; the profiler hook proves which kernel ran."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, N, K, FH, FW, stride, pad = case
    rng = np.random.default_rng(H + 3 * W + K + FH)
    x, f, b = rnd(rng, H, W, 1, N), rnd(rng, FH, FW, 1, K), rnd(rng, K)
    y_ref = O.vl_nnconv(x, f, b, stride=stride, pad=pad, acc64=True)
    assert (y_ref.shape[0] * y_ref.shape[1]) % 4 == 0
    xd, fd, bd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1))
    old = L.xm_debug_force_conv_stem(1)
    try:
        y, names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, stride=stride, pad=pad))
        assert any("stem" in n for n in names), names
        close(vl.to_numpy(y), y_ref, what="stem fwd")
        mo = vl.mat_empty(K, 2, device=xd.device)
        ym, names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, stride=stride, pad=pad, moments_out=mo))
        assert any("stem" in n for n in names), names
        _, m_ref = O.vl_nnbnorm(y_ref, O.F(np.ones(K)), O.F(np.zeros(K)), acc64=True)
        close(vl.to_numpy(ym), y_ref, what="stem fwd + moments")
        m = vl.to_numpy(mo)
        close(m[:, 0], m_ref[:, 0], what="stem mean")
        assert np.abs(m[:, 1] / m_ref[:, 1] - 1).max() <= 1e-4
        # filter derivative through conv_stem_wgrad_kernel (no input derivative: the stem is the first layer)
        dzdy = rnd(rng, *y_ref.shape)
        _, df_ref, _ = O.vl_nnconv(x, f, b, dzdy, stride=stride, pad=pad, acc64=True, no_der_data=True)
        (_, df, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, vl.from_numpy(dzdy), stride=stride, pad=pad,
                                                                  no_der_data=True))
        assert any("stem_wgrad" in n for n in names), names
        close(vl.to_numpy(df), df_ref, what="stem wgrad")
        L.xm_debug_force_conv_stem(0)
        (_, df0, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, vl.from_numpy(dzdy), stride=stride, pad=pad,
                                                                   no_der_data=True))
        assert not any("stem" in n for n in names), names
        close(vl.to_numpy(df0), df_ref, what="generic wgrad of the stem")
        y0, names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, stride=stride, pad=pad))
        assert not any("stem" in n for n in names), names
        d = np.abs(vl.to_numpy(y) - vl.to_numpy(y0)).max()
        assert d <= 2e-5 * max(1.0, float(np.abs(y_ref).max())), d
    finally:
        L.xm_debug_force_conv_stem(old)


STEM_BNP_CASES = [  # H, W, N, K, FH, FW, stride, pad, train, with conv bias
    (512, 60, 2, 96, 7, 7, 2, 1, True, True),                       # the student's conv1 -> bn1 -> relu1 -> pool1, short clips
    (512, 42, 3, 96, 7, 7, (2, 2), [1, 1, 1, 2], True, False),      # tiles straddle columns AND samples, ragged last tile
    (512, 44, 2, 80, 7, 7, 2, 1, False, True),                      # fewer filters than the 96-row tile, test-mode moments
    (260, 37, 2, 96, 5, 5, 1, 2, True, True),                       # unit stride: 260 x 37 outputs, odd pooled width
]


@pytest.mark.parametrize("case", STEM_BNP_CASES)
def test_conv_stem_wgrad_through_bnorm_relu_pool(gpu, case):
    """xm_nnconv_backward_filter_bnrelupool (conv_stem_wgrad_bnp_kernel): the first layer's filter / bias derivative and
    the bnorm's dg / db straight from the POOLED derivative -- the bnorm's DZDX is rebuilt per element inside the kernel
    and never written.  Against the oracle's composition vl_nnconv <- vl_nnbnorm <- vl_nnrelu <- vl_nnpool (fp64
    accumulate) and against the two separate HIP calls.  The inputs plant window maxima on the first / last rows and
    columns of the first / last planes (the loads of the routing operands touch both ends of the pooled tensors), and
    on equal values (first-maximum routing)."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, N, K, FH, FW, stride, pad, train, has_bias = case
    rng = np.random.default_rng(H + 5 * W + K + FH + int(train))
    x, f, b = rnd(rng, H, W, 1, N), O.F(rng.standard_normal((FH, FW, 1, K)) * 0.2), rnd(rng, K)
    if not has_bias:
        b = O.F(np.zeros(K))
    y = O.vl_nnconv(x, f, b, stride=stride, pad=pad, acc64=True)
    Ho, Wo = y.shape[:2]
    assert Ho % 2 == 0 and (Ho * Wo) % 4 == 0 and Ho >= 128
    # planted maxima: corners of the first and of the last plane, last covered row / column, and a plateau
    big = float(np.abs(y).max()) * 4 + 1
    pHo, pWo = (Ho - 3) // 2 + 1, (Wo - 3) // 2 + 1
    for (c, n) in ((0, 0), (K - 1, N - 1)):
        y[0, 0, c, n] = big
        y[2 * (pHo - 1) + 2, 2 * (pWo - 1) + 2, c, n] = big      # last row / column any window covers
        y[2 * (pHo - 1) + 2, 0, c, n] = big
        y[0, 2 * (pWo - 1) + 2, c, n] = big
        y[10:13, 6:9, c, n] = big * 0.5                           # plateau: the FIRST maximum of each window wins
    # window (0, 0) routed to pixel (1, 1): rows 0, 1 of an odd column belong to the quad that straddles in from the column
    # before it, whose pooled triple for the window column in FRONT of column 0 would start before the tensor in the
    # first plane of every row group (a load that starts in front of the buffer is dropped as a whole)
    y[1, 1, 8, 0] = big
    y[0, 1, 16, 0] = big
    g, bb = O.F(rng.uniform(0.5, 1.5, K) * rng.choice([-1, 1], K)), rnd(rng, K)
    mom = None if train else O.F(np.stack([rng.standard_normal(K) * 0.3, rng.uniform(0.5, 1.5, K)], 1))
    yb, mref = O.vl_nnbnorm(y, g, bb, moments=mom, acc64=True)
    yr = np.maximum(yb, 0)
    yp = O.vl_nnpool(yr, [3, 3], stride=2, pad=0, method="max")
    dz = rnd(rng, *yp.shape)
    dyr = O.vl_nnpool(yr, [3, 3], dz, stride=2, pad=0, method="max")
    dx_ref, dg_ref, db_ref, _ = O.vl_nnbnorm(y, g, bb, dyr * (yb > 0), moments=mom, acc64=True)
    _, df_ref, dbias_ref = O.vl_nnconv(x, f, b, dx_ref, stride=stride, pad=pad, acc64=True, no_der_data=True)

    xd, yd = vl.from_numpy(x), vl.from_numpy(y)
    gd, bd = vl.from_numpy(g.reshape(K, 1)), vl.from_numpy(bb.reshape(K, 1))
    md = None if mom is None else vl.from_numpy(mom)
    ypd, am, mo = vl.bnorm_relu_pool(yd, gd, bd, [3, 3], stride=2, pad=0, moments=md)
    close(vl.to_numpy(ypd), yp, what="pooled forward")
    dzd = vl.from_numpy(dz)
    res, names = _kernels_run(L, lambda: vl.conv_backward_filter_bnrelupool(
        xd, (FH, FW, 1, K), yd, gd, bd, mo, am, ypd, dzd, [3, 3], stride=stride, pad=pad, pool_stride=2, pool_pad=0,
        train=train, has_bias=has_bias))
    assert res is not None, "the fused kernel must cover this geometry"
    assert any("stem_wgrad_bnp" in n for n in names), names
    df, dbias, dg, db = res
    # the two separate calls (bnorm + relu + pool backward writes DX, the convolution's backward reads it)
    dx2, dg2, db2 = vl.bnorm_relu_pool_backward(yd, gd, bd, mo, am, dzd, [3, 3], stride=2, pad=0, train=train, y_pool=ypd)
    close(vl.to_numpy(dx2), dx_ref, what="unfused dx")
    _, df2, dbias2 = vl.vl_nnconv(xd, vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1)), dx2, stride=stride, pad=pad,
                                  no_der_data=True)
    scale = max(1.0, float(np.abs(df_ref).max()))
    close(vl.to_numpy(df2), df_ref, what="unfused filter derivative")
    close(vl.to_numpy(df), df_ref, what="fused filter derivative")
    assert np.abs(vl.to_numpy(df) - vl.to_numpy(df2)).max() <= 2e-5 * scale      # same decisions, same formula
    close(vl.to_numpy(dg).ravel(), dg_ref, what="fused dg")
    close(vl.to_numpy(db).ravel(), db_ref, what="fused db")
    assert np.array_equal(vl.to_numpy(dg), vl.to_numpy(dg2)) and np.array_equal(vl.to_numpy(db), vl.to_numpy(db2))
    if has_bias:
        # sum of DX over pixels and samples: exactly zero in exact arithmetic for a train-mode bnorm, i.e. pure round-off
        # of ~1e5 ... 1e6 terms in both implementations -- compared on the scale of sum |DX|
        ref = dx_ref.astype(np.float64).sum((0, 1, 3))
        mag = np.abs(dx_ref.astype(np.float64)).sum((0, 1, 3)).max()
        assert np.abs(vl.to_numpy(dbias).ravel() - ref).max() <= 2e-6 * max(1.0, mag), "fused conv bias derivative"
        close(vl.to_numpy(dbias2).ravel(), dbias_ref, 2e-4, what="unfused conv bias derivative")
    # shapes the fused kernel does not cover come back as None (the caller makes the two calls)
    assert vl.conv_backward_filter_bnrelupool(xd, (FH, FW, 1, K), yd, gd, bd, mo, am, ypd, dzd, [3, 3], stride=stride,
                                              pad=pad, pool_stride=2, pool_pad=[0, 1, 0, 1], train=train) is None


@pytest.mark.parametrize("N,variant", [(3, 1), (3, 3), (16, 3), (16, 1)])
def test_conv_halo_strided_dgrad(gpu, N, variant):
    """dgrad of a 5 x 5 / stride-2 convolution (the student's conv2): four stride-parity classes with 3x3, 3x2, 2x3 and
    2x2 taps, each a unit-stride gather in dY space -> the halo-patch kernel with T = 9 / 6 / 4 taps, per class (few
    tiles) or all classes in one launch (conv_halo_multi_kernel), 128-row and 96-row / tall-patch variants."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, C, K = 126, 73, 96, 8
    rng = np.random.default_rng(N * 10 + variant)
    x, f = rnd(rng, H, W, C, N), rnd(rng, 5, 5, C, K)
    y_ref = O.vl_nnconv(x, f, None, stride=2, pad=1, acc64=True)
    dzdy = rnd(rng, *y_ref.shape)
    dx_ref, _, _ = O.vl_nnconv(x, f, None, dzdy, stride=2, pad=1, acc64=True, no_der_filters=True)
    xd, fd, dd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(dzdy)
    old = L.xm_debug_force_conv_halo(variant)
    try:
        (dx, _, _), names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, None, dd, stride=2, pad=1, no_der_filters=True))
        assert any("halo" in n for n in names), names
        assert any("multi" in n for n in names) == (N == 16), names
        close(vl.to_numpy(dx), dx_ref, what="strided dgrad through the halo-patch kernel")
    finally:
        L.xm_debug_force_conv_halo(old)


def test_conv_fused_epilogue(gpu):
    from mcncrossmodalemotions_amd import vl
    rng = np.random.default_rng(11)
    H, W, C, N, K = 14, 14, 24, 3, 40
    x, f, b = rnd(rng, H, W, C, N), rnd(rng, 3, 3, C, K), rnd(rng, K)
    sc, sh = O.F(rng.uniform(0.5, 1.5, K)), rnd(rng, K)
    res = rnd(rng, H, W, K, N)
    y0 = O.vl_nnconv(x, f, b, pad=1, acc64=True)
    ref = np.maximum(y0 * sc.reshape(1, 1, K, 1) + sh.reshape(1, 1, K, 1) + res, 0)
    y = vl.vl_nnconv(vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1)), pad=1,
                     scale=vl.from_numpy(sc.reshape(K, 1)), shift=vl.from_numpy(sh.reshape(K, 1)),
                     residual=vl.from_numpy(res), relu=True)
    close(vl.to_numpy(y), ref, what="fused conv")


@pytest.mark.parametrize("case", [(8, 8, 16, 5, 40), (7, 7, 24, 9, 70), (14, 14, 64, 3, 256), (2, 2, 512, 6, 128)])
def test_conv_gated_epilogue(gpu, case):
    """xm_nnconv_forward_gated: y = relu(((conv + b) .* scale + shift) .* gate(k, n) + residual) for every tile
    configuration (vector and scalar stores, LDS-DMA kernel, split-K combine) -- the SE excite folded into the 1 x 1
    projection (mcnExtraLayers dagnn.Axpy: out = a .* x + shortcut)."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, C, N, K = case
    rng = np.random.default_rng(H + C + K)
    x, f, b = rnd(rng, H, W, C, N), rnd(rng, 1, 1, C, K), rnd(rng, K)
    sc, sh = O.F(rng.uniform(0.5, 1.5, K)), rnd(rng, K)
    gate = O.F(rng.uniform(0.0, 1.0, (1, 1, K, N)))
    res = rnd(rng, H, W, K, N)
    y0 = O.vl_nnconv(x, f, b, acc64=True)
    ref = np.maximum((y0 * sc.reshape(1, 1, K, 1) + sh.reshape(1, 1, K, 1)) * gate + res, 0)
    xd, fd, bd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1))
    kw = dict(scale=vl.from_numpy(sc.reshape(K, 1)), shift=vl.from_numpy(sh.reshape(K, 1)), gate=vl.from_numpy(gate),
              residual=vl.from_numpy(res), relu=True)
    try:
        for cfg in list(range(L.xm_debug_num_conv_cfgs())) + [-1]:
            L.xm_debug_force_conv_cfg(cfg)
            close(vl.to_numpy(vl.vl_nnconv(xd, fd, bd, **kw)), ref, what="gated cfg %d" % cfg)
        L.xm_debug_force_conv_cfg(-1)
        old = L.xm_debug_force_conv_splits(2)
        try:
            close(vl.to_numpy(vl.vl_nnconv(xd, fd, bd, **kw)), ref, what="gated split-K")
        finally:
            L.xm_debug_force_conv_splits(old)
    finally:
        L.xm_debug_force_conv_cfg(-1)


@pytest.mark.parametrize("geom", [(40, 35, 256, 32, 256, 1), (15, 13, 32, 230, 128, 3)])
def test_conv_hybrid_schedule(gpu, geom):
    """launches of more than one round of the chip with a partly filled last round: whole tiles for the full rounds, the
    remaining tiles split along the reduction and combined by conv_splitk_epilogue_kernel over their pixel range
    (ConvGemmArgs::hyS) -- every register-staged tile configuration, plain and fused epilogue, vector and scalar stores."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, C, N, K, F = geom
    rng = np.random.default_rng(H + K)
    pad = F // 2
    x, f, b = rnd(rng, H, W, C, N), O.F(rng.standard_normal((F, F, C, K)) / np.sqrt(F * F * C)), rnd(rng, K)
    y_ref = O.vl_nnconv(x, f, b, pad=pad, acc64=True)
    sc, sh = O.F(rng.uniform(0.5, 1.5, K)), rnd(rng, K)
    res = rnd(rng, *y_ref.shape)
    ref2 = np.maximum(y_ref * sc.reshape(1, 1, K, 1) + sh.reshape(1, 1, K, 1) + res, 0)
    xd, fd, bd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1))
    kw = dict(scale=vl.from_numpy(sc.reshape(K, 1)), shift=vl.from_numpy(sh.reshape(K, 1)), residual=vl.from_numpy(res), relu=True)
    dzdy = rnd(rng, *y_ref.shape)
    dx_ref, _, _ = O.vl_nnconv(x, f, b, dzdy, pad=pad, acc64=True, no_der_filters=True)
    _, m_ref = O.vl_nnbnorm(y_ref, O.F(np.ones(K)), O.F(np.zeros(K)), acc64=True)
    old = L.xm_debug_force_conv_halo(0)
    try:
        for cfg in range(L.xm_debug_num_conv_cfgs() - 4):     # the register-staged ones (the last four are LDS-DMA)
            L.xm_debug_force_conv_cfg(cfg)
            close(vl.to_numpy(vl.vl_nnconv(xd, fd, bd, pad=pad)), y_ref, what="hybrid cfg %d" % cfg)
            close(vl.to_numpy(vl.vl_nnconv(xd, fd, bd, pad=pad, **kw)), ref2, what="hybrid cfg %d, fused epilogue" % cfg)
            dx, _, _ = vl.vl_nnconv(xd, fd, bd, vl.from_numpy(dzdy), pad=pad, no_der_filters=True)
            close(vl.to_numpy(dx), dx_ref, what="hybrid cfg %d dgrad" % cfg)
            # batch moments: per-tile partial sums from the full rounds + the combine kernel's chunks for the remainder
            mo = vl.mat_empty(K, 2, device=xd.device)
            ym = vl.vl_nnconv(xd, fd, bd, pad=pad, moments_out=mo)
            close(vl.to_numpy(ym), y_ref, what="hybrid cfg %d + moments" % cfg)
            m = vl.to_numpy(mo)
            close(m[:, 0], m_ref[:, 0], what="hybrid cfg %d mean" % cfg)
            assert np.abs(m[:, 1] / m_ref[:, 1] - 1).max() <= 1e-4, cfg
    finally:
        L.xm_debug_force_conv_cfg(-1)
        L.xm_debug_force_conv_halo(old)


@pytest.mark.parametrize("shape", [(56, 56, 16, 3), (7, 7, 33, 5), (14, 14, 8, 2)])
def test_global_avg_pool_backward_at_a_fork(gpu, shape):
    """SE squeeze backward where X has a second consumer: dx = accum + dzdy / (H W) in one pass
    (xm_nnpool_global_avg_backward_accum) is bit-identical to the broadcast pass followed by the sum pass."""
    import torch
    from mcncrossmodalemotions_amd import vl
    H, W, C, N = shape
    rng = np.random.default_rng(H + C)
    x, acc, dz = rnd(rng, H, W, C, N), rnd(rng, H, W, C, N), rnd(rng, 1, 1, C, N)
    xd, ad, dd = vl.from_numpy(x), vl.from_numpy(acc), vl.from_numpy(dz)
    fused = vl.vl_nnpool(xd, [H, W], dd, method="avg", dx_accum=ad)
    plain = vl.sum2(ad, vl.vl_nnpool(xd, [H, W], dd, method="avg"))
    assert torch.equal(fused, plain)
    close(vl.to_numpy(fused), acc + O.vl_nnpool(x, [H, W], dz, method="avg"), what="global avg backward + accum")
    with pytest.raises(ValueError):
        vl.vl_nnpool(xd, [2, 2], vl.from_numpy(rnd(rng, H // 2, W // 2, C, N)), stride=2, method="avg", dx_accum=ad)


def test_conv_no_der_flags_and_errors(gpu):
    from mcncrossmodalemotions_amd import vl, _lib
    rng = np.random.default_rng(3)
    x, f = rnd(rng, 8, 8, 4, 2), rnd(rng, 3, 3, 4, 5)
    xd, fd = vl.from_numpy(x), vl.from_numpy(f)
    y = vl.vl_nnconv(xd, fd, None)
    close(vl.to_numpy(y), O.vl_nnconv(x, f, None, acc64=True), what="no bias")
    dz = vl.from_numpy(rnd(rng, 6, 6, 5, 2))
    dx, df, db = vl.vl_nnconv(xd, fd, None, dz, no_der_data=True)
    assert dx is None and db is None and df is not None
    with pytest.raises(_lib.XmError):  # filter larger than the padded input
        vl.vl_nnconv(xd, vl.from_numpy(rnd(rng, 9, 9, 4, 5)), None)
    with pytest.raises(_lib.XmError):  # channel mismatch
        vl.vl_nnconv(xd, vl.from_numpy(rnd(rng, 3, 3, 3, 5)), None)


POOL_CASES = [
    (13, 11, 5, 3, (3, 3), (2, 2), (0, 0, 0, 0), "max"),     # student mpool1/2
    (9, 8, 6, 2, (5, 3), (3, 2), (0, 0, 0, 0), "max"),       # student mpool5
    (12, 12, 4, 2, (3, 3), (2, 2), (0, 1, 0, 1), "max"),     # resnet pool1 (Caffe ceil pad)
    (1, 8, 16, 3, (1, 8), (1, 1), (0, 0, 0, 0), "avg"),      # student pool6
    (7, 7, 10, 2, (7, 7), (1, 1), (0, 0, 0, 0), "avg"),      # resnet pool5 / SE global
    (10, 9, 3, 2, (3, 2), (2, 1), (1, 1, 0, 1), "avg"),      # avg with padding (clipped area)
    (10, 9, 3, 2, (2, 2), (1, 1), (1, 0, 1, 0), "max"),
]


@pytest.mark.parametrize("case", POOL_CASES)
def test_pool(gpu, case):
    from mcncrossmodalemotions_amd import vl
    H, W, C, N, pool, stride, pad, method = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rnd(rng, H, W, C, N)
    x = np.maximum(x, 0)  # post-ReLU inputs: plenty of exact ties for the max-backward rule
    y_ref = O.vl_nnpool(x, pool, stride=stride, pad=pad, method=method)
    xd = vl.from_numpy(x)
    y = vl.vl_nnpool(xd, pool, stride=stride, pad=pad, method=method)
    close(vl.to_numpy(y), y_ref, 1e-6, "pool fwd")
    dzdy = rnd(rng, *y_ref.shape)
    dx_ref = O.vl_nnpool(x, pool, dzdy, stride=stride, pad=pad, method=method)
    dx = vl.vl_nnpool(xd, pool, vl.from_numpy(dzdy), stride=stride, pad=pad, method=method)
    close(vl.to_numpy(dx), dx_ref, 1e-5, "pool bwd")
    # extension: forward records the first-maximum routing table, backward consumes it
    y2, am = vl.vl_nnpool(xd, pool, stride=stride, pad=pad, method=method, want_argmax=True)
    close(vl.to_numpy(y2), y_ref, 1e-6, "pool fwd (argmax variant)")
    dx2 = vl.vl_nnpool(xd, pool, vl.from_numpy(dzdy), stride=stride, pad=pad, method=method, argmax=am)
    close(vl.to_numpy(dx2), dx_ref, 1e-5, "pool bwd (argmax variant)")


BN_CASES = [(13, 7, 5, 4), (8, 8, 16, 3), (1, 1, 32, 6), (1, 8, 24, 5), (30, 17, 3, 2)]


@pytest.mark.parametrize("shape", BN_CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_bnorm(gpu, shape, relu):
    from mcncrossmodalemotions_amd import vl
    H, W, C, N = shape
    rng = np.random.default_rng(H * 1000 + W * 100 + C * 10 + N)
    x = O.F(rng.standard_normal(shape) * 2.0 + 3.0)  # non-zero mean: stresses the variance pass
    g, b = O.F(rng.uniform(0.5, 1.5, C)), rnd(rng, C)
    y_ref, m_ref = O.vl_nnbnorm(x, g, b, acc64=True)
    if relu:
        y_ref = np.maximum(y_ref, 0)
    xd, gd, bd = vl.from_numpy(x), vl.from_numpy(g.reshape(C, 1)), vl.from_numpy(b.reshape(C, 1))
    y, m = vl.vl_nnbnorm(xd, gd, bd, relu=relu)
    close(vl.to_numpy(y), y_ref, what="bn fwd")
    close(vl.to_numpy(m), m_ref, what="bn moments")
    dzdy = rnd(rng, *shape)
    dz_eff = dzdy * (y_ref > 0) if relu else dzdy
    dx_ref, dg_ref, db_ref, _ = O.vl_nnbnorm(x, g, b, dz_eff, acc64=True)
    dx, dg, db, _ = vl.vl_nnbnorm(xd, gd, bd, vl.from_numpy(dzdy), relu=relu, y=y if relu else None)
    close(vl.to_numpy(dx), dx_ref, what="bn dx")
    close(vl.to_numpy(dg).ravel(), dg_ref, what="bn dg")
    close(vl.to_numpy(db).ravel(), db_ref, what="bn db")
    # train-mode backward handed the forward's batch moments (XM_BN_BATCH_MOMENTS): same bits, one pass less
    dx2, dg2, db2, m2 = vl.vl_nnbnorm(xd, gd, bd, vl.from_numpy(dzdy), relu=relu, y=y if relu else None,
                                      moments=m, batch_moments=True)
    close(vl.to_numpy(dx2), vl.to_numpy(dx), 0, "bn dx with forward moments")
    close(vl.to_numpy(dg2), vl.to_numpy(dg), 0, "bn dg with forward moments")
    close(vl.to_numpy(db2), vl.to_numpy(db), 0, "bn db with forward moments")
    close(vl.to_numpy(m2), vl.to_numpy(m), 0, "moments passed through")
    # xm_nnbnorm_backward_dxsum: same dx / dg / db from the per-channel apply kernel, plus sum(dx) per channel (= dzdb
    # of a convolution that produced x; ~0 in train mode, so it is compared on the scale of |dx| * sqrt(count))
    dxs = vl.mat_empty(C, 1, device=xd.device)
    dx3, dg3, db3, _ = vl.vl_nnbnorm(xd, gd, bd, vl.from_numpy(dzdy), relu=relu, y=y if relu else None,
                                     moments=m, batch_moments=True, dxsum_out=dxs)
    close(vl.to_numpy(dx3), dx_ref, what="bn dx (dxsum path)")
    close(vl.to_numpy(dg3).ravel(), dg_ref, what="bn dg (dxsum path)")
    close(vl.to_numpy(db3).ravel(), db_ref, what="bn db (dxsum path)")
    sref = dx_ref.astype(np.float64).sum(axis=(0, 1, 3))
    got = vl.to_numpy(dxs).ravel().astype(np.float64)
    assert np.abs(got - vl.to_numpy(dx3).astype(np.float64).sum(axis=(0, 1, 3))).max() <= 1e-6 * max(1.0, np.abs(dx_ref).max()) * np.sqrt(H * W * N)
    assert np.abs(got - sref).max() <= 1e-5 * max(1.0, np.abs(dx_ref).max()) * np.sqrt(H * W * N)
    # test mode with stored moments
    mom = O.F(np.stack([rng.standard_normal(C), rng.uniform(0.5, 1.5, C)], 1))
    yt_ref, _ = O.vl_nnbnorm(x, g, b, moments=mom, acc64=True)
    yt, _ = vl.vl_nnbnorm(xd, gd, bd, moments=vl.from_numpy(mom))
    close(vl.to_numpy(yt), yt_ref, what="bn test-mode")
    dxt_ref, dgt_ref, dbt_ref, _ = O.vl_nnbnorm(x, g, b, dzdy, moments=mom, acc64=True)
    dxt, dgt, dbt, _ = vl.vl_nnbnorm(xd, gd, bd, vl.from_numpy(dzdy), moments=vl.from_numpy(mom))
    close(vl.to_numpy(dxt), dxt_ref, what="bn test-mode dx")
    dxs2 = vl.mat_empty(C, 1, device=xd.device)
    dxt2, _, _, _ = vl.vl_nnbnorm(xd, gd, bd, vl.from_numpy(dzdy), moments=vl.from_numpy(mom), dxsum_out=dxs2)
    close(vl.to_numpy(dxt2), dxt_ref, what="bn test-mode dx (dxsum path)")
    close(vl.to_numpy(dxs2).ravel(), dxt_ref.astype(np.float64).sum(axis=(0, 1, 3)), 2e-5, "bn test-mode sum(dx)")
    close(vl.to_numpy(dgt).ravel(), dgt_ref, what="bn test-mode dg")


@pytest.mark.parametrize("case", [(13, 11, 5, 3, (3, 3), (2, 2), (0, 0, 0, 0)), (9, 8, 6, 2, (5, 3), (3, 2), (0, 0, 0, 0)),
                                  (12, 12, 4, 2, (3, 3), (2, 2), (0, 1, 0, 1)), (10, 9, 3, 2, (2, 2), (1, 1), (1, 0, 1, 0))])
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("pooled", [False, True])
def test_fused_bnorm_relu_pool(gpu, case, train, pooled):
    """extension op == vl_nnpool(vl_nnrelu(vl_nnbnorm(x))) forward and backward (oracle composition)."""
    from mcncrossmodalemotions_amd import vl
    H, W, C, N, pool, stride, pad = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + int(train))
    x = O.F(rng.standard_normal((H, W, C, N)) * 1.5 + 0.3)
    g, b = O.F(rng.uniform(0.5, 1.5, C) * rng.choice([-1, 1], C)), rnd(rng, C)  # negative gains too
    g[0] = 1e-4 * np.sign(g[0])       # |b| > 100 |g|: the pooled-domain sums must gather x for this channel
    b[0] = 0.5
    mom = None if train else O.F(np.stack([rng.standard_normal(C) * 0.3, rng.uniform(0.5, 1.5, C)], 1))
    yb, mref = O.vl_nnbnorm(x, g, b, moments=mom, acc64=True)
    yr = np.maximum(yb, 0)
    yp = O.vl_nnpool(yr, pool, stride=stride, pad=pad, method="max")
    dz = rnd(rng, *yp.shape)
    dyr = O.vl_nnpool(yr, pool, dz, stride=stride, pad=pad, method="max")
    dyb = dyr * (yb > 0)
    dx_ref, dg_ref, db_ref, _ = O.vl_nnbnorm(x, g, b, dyb, moments=mom, acc64=True)
    xd, gd, bd = vl.from_numpy(x), vl.from_numpy(g.reshape(C, 1)), vl.from_numpy(b.reshape(C, 1))
    md = None if mom is None else vl.from_numpy(mom)
    y, am, mo = vl.bnorm_relu_pool(xd, gd, bd, pool, stride=stride, pad=pad, moments=md)
    close(vl.to_numpy(y), yp, what="fused fwd")
    close(vl.to_numpy(mo), mref, what="fused moments")
    dxs = vl.mat_zeros(C, 1)
    dx, dg, db = vl.bnorm_relu_pool_backward(xd, gd, bd, mo, am, vl.from_numpy(dz), pool, stride=stride,
                                             pad=pad, train=train, dxsum_out=dxs, y_pool=y if pooled else None)
    close(vl.to_numpy(dx), dx_ref, what="fused dx")
    # dxsum = dzdb of a convolution that produced x (sum of dx over pixels and samples)
    close(vl.to_numpy(dxs).ravel(), dx_ref.astype(np.float64).sum((0, 1, 3)), 2e-4, what="fused dxsum")
    close(vl.to_numpy(dg).ravel(), dg_ref, what="fused dg")
    close(vl.to_numpy(db).ravel(), db_ref, what="fused db")


@pytest.mark.parametrize("shape", [(56, 56, 24, 3), (7, 7, 40, 5), (14, 13, 9, 2)])
@pytest.mark.parametrize("train", [True, False])
def test_se_tail_backward(gpu, shape, train):
    """xm_se_tail_backward_reduce / _apply (training-mode SE block tail, ferplus_baselines.m:140-141) against the
    oracle's composition of the separate operators: relu mask, Axpy backward (scale), GlobalPooling backward added at
    the fork, vl_nnbnorm backward -- da, dz (the shortcut's derivative), du, dg, db."""
    from mcncrossmodalemotions_amd import vl
    H, W, C, N = shape
    rng = np.random.default_rng(H * 7 + C + int(train))
    u = O.F(rng.standard_normal(shape) * 1.3 + 0.4)
    g, b = O.F(rng.uniform(0.5, 1.5, C) * rng.choice([-1, 1], C)), rnd(rng, C)
    mom = None if train else O.F(np.stack([rng.standard_normal(C) * 0.3, rng.uniform(0.5, 1.5, C)], 1))
    x, mref = O.vl_nnbnorm(u, g, b, moments=mom, acc64=True)
    a = O.F(rng.uniform(0.05, 0.95, (1, 1, C, N)))
    s = rnd(rng, *shape)
    y = O.scale_axpy(x, a, s, relu=True)
    dzdy, dgp = rnd(rng, *shape), rnd(rng, 1, 1, C, N)
    dz_ref = O.vl_nnrelu(y, dzdy)                 # y > 0 <=> pre-activation > 0
    dx1, da_ref = O.scale_backward(x, a, dz_ref)
    dx = dx1 + O.vl_nnpool(x, (H, W), dgp, method="avg")
    du_ref, dg_ref, db_ref, _ = O.vl_nnbnorm(u, g, b, O.F(dx), moments=mom, acc64=True)
    d = vl.from_numpy
    gd, bd = d(g.reshape(C, 1)), d(b.reshape(C, 1))
    md = d(mref)
    # forward halves: squeeze and excite straight from u (the bnorm's output is never materialised)
    gp = vl.se_squeeze_bn(d(u), gd, bd, md)
    close(vl.to_numpy(gp), O.vl_nnpool(x, (H, W), method="avg"), 2e-5, "squeeze of bnorm(u)")
    yd = vl.scale_axpy_bn(d(u), d(a), d(s), gd, bd, md, relu=True)
    close(vl.to_numpy(yd), y, 2e-5, "excite of bnorm(u)")
    x_hip, _ = vl.vl_nnbnorm(d(u), gd, bd, moments=md)
    assert np.array_equal(vl.to_numpy(yd), vl.to_numpy(vl.scale_axpy(x_hip, d(a), d(s), relu=True)))   # the same bits
    da, sums = vl.se_tail_backward_reduce(d(y), d(dzdy), d(u), gd, bd, md)
    close(vl.to_numpy(da), da_ref, what="da")
    dz, du, dg, db = vl.se_tail_backward_apply(d(y), d(dzdy), d(u), d(a), d(dgp), gd, md, sums, train=train)
    assert np.array_equal(vl.to_numpy(dz), dz_ref)
    close(vl.to_numpy(du), du_ref, what="du")
    close(vl.to_numpy(dg).ravel(), dg_ref, what="dg")
    close(vl.to_numpy(db).ravel(), db_ref, what="db")


def test_elementwise(gpu):
    from mcncrossmodalemotions_amd import vl
    rng = np.random.default_rng(5)
    for shape in [(7, 5, 3, 2), (16, 16, 8, 4), (1, 1, 9, 7)]:
        x, d, r = rnd(rng, *shape), rnd(rng, *shape), rnd(rng, *shape)
        xd, dd, rd = vl.from_numpy(x), vl.from_numpy(d), vl.from_numpy(r)
        close(vl.to_numpy(vl.vl_nnrelu(xd)), O.vl_nnrelu(x), 0, "relu")
        close(vl.to_numpy(vl.vl_nnrelu(xd, dd)), O.vl_nnrelu(x, d), 0, "relu bwd")
        close(vl.to_numpy(vl.vl_nnrelu(xd, leak=0.1)), O.vl_nnrelu(x, leak=0.1), 1e-7, "leaky")
        close(vl.to_numpy(vl.vl_nnsigmoid(xd)), O.vl_nnsigmoid(x), 1e-6, "sigmoid")
        close(vl.to_numpy(vl.vl_nnsigmoid(xd, dd)), O.vl_nnsigmoid(x, d), 1e-6, "sigmoid bwd")
        close(vl.to_numpy(vl.sum2(xd, rd, relu=True)), O.sum2(x, r, relu=True), 0, "sum relu")
        close(vl.to_numpy(vl.sum2(xd, rd)), O.sum2(x, r), 0, "sum")
        a = rnd(rng, 1, 1, shape[2], shape[3])
        ad = vl.from_numpy(a)
        close(vl.to_numpy(vl.scale_axpy(xd, ad, rd, relu=True)), O.scale_axpy(x, a, r, relu=True),
              1e-6, "axpy")
        close(vl.to_numpy(vl.scale_axpy(xd, ad)), O.scale_axpy(x, a), 1e-6, "scale")
        dx_ref, da_ref = O.scale_backward(x, a, d)
        dx, da = vl.scale_backward(xd, ad, dd)
        close(vl.to_numpy(dx), dx_ref, 1e-6, "scale dx")
        close(vl.to_numpy(da), da_ref, 1e-5, "scale da")


def test_dropout(gpu):
    """vl_nndropout (dagnn.DropOut of emoVoxZoo.m:116-135): the mask is bit for bit the documented Philox stream
    (oracle.dropout_mask; the generator itself is pinned on its published known-answer vector in tests/test_oracle.py),
    Y = MASK .* X and DZDX = MASK .* DZDY exactly, for sizes that are / are not multiples of four and for several
    rates, seeds and counter offsets; a given mask is applied as is."""
    from mcncrossmodalemotions_amd import vl
    rng = np.random.default_rng(11)
    for shape, rate, seed, off in (((9, 8, 16, 4), 0.5, 1, 0), ((7, 3, 5, 3), 0.3, 2 ** 40 + 17, 5), ((1, 1, 1024, 32), 0.75, 7, 10 ** 6),
                                   ((5,), 0.0, 3, 0)):
        x = rnd(rng, *shape)
        y, m = vl.vl_nndropout(vl.from_numpy(x), rate=rate, seed=seed, offset=off)
        mref = O.dropout_mask(shape, rate, seed, off)
        assert np.array_equal(vl.to_numpy(m), mref), (shape, rate)
        assert np.array_equal(vl.to_numpy(y), O.vl_nndropout(x, mref))
        dz = rnd(rng, *shape)
        dx = vl.vl_nndropout(vl.from_numpy(x), vl.from_numpy(dz), mask=m)
        assert np.array_equal(vl.to_numpy(dx), O.vl_nndropout(dz, mref))
        y2, _ = vl.vl_nndropout(vl.from_numpy(x), mask=vl.from_numpy(mref))
        assert np.array_equal(vl.to_numpy(y2), vl.to_numpy(y))
    with pytest.raises(Exception):
        vl.vl_nndropout(vl.from_numpy(rnd(rng, 4, 4)), rate=1.0)


def test_losses(gpu):
    from mcncrossmodalemotions_amd import vl
    rng = np.random.default_rng(9)
    for N in (1, 5, 64, 300):
        x, p = rnd(rng, 1, 1, 8, N) * 3, rnd(rng, 1, 1, 8, N) * 3
        xd, pd = vl.from_numpy(x), vl.from_numpy(p)
        # the distillation loss as configured at emoVoxZoo.m:152
        l = vl.vl_nnsoftmaxceloss(xd, pd, temperature=2, logitTargets=True)
        close(vl.to_numpy(l).ravel()[0], O.vl_nnsoftmaxceloss(x, p, temperature=2, logit_targets=True),
              1e-6, "softmaxce fwd")
        g = vl.vl_nnsoftmaxceloss(xd, pd, 1.0, temperature=2, logitTargets=True)
        close(vl.to_numpy(g), O.vl_nnsoftmaxceloss(x, p, np.ones(1), temperature=2, logit_targets=True),
              1e-6, "softmaxce bwd")
        # probability targets + instance weights
        pr = O.vl_nnsoftmaxt(p, 1.0)
        w = O.F(rng.uniform(0.5, 2, (1, 1, 1, N)))
        l2 = vl.vl_nnsoftmaxceloss(xd, vl.from_numpy(pr), instanceWeights=vl.from_numpy(w))
        close(vl.to_numpy(l2).ravel()[0], O.vl_nnsoftmaxceloss(x, pr, instance_weights=w), 1e-6, "ce w")
        lab = O.F(rng.integers(1, 9, (1, 1, 1, N)))
        ld = vl.from_numpy(lab)
        for loss in ("softmaxlog", "classerror"):
            close(vl.to_numpy(vl.vl_nnloss(xd, ld, loss=loss)).ravel()[0], O.vl_nnloss(x, lab, loss=loss),
                  1e-6, loss)
            close(vl.to_numpy(vl.vl_nnloss(xd, ld, 1.0, loss=loss)), O.vl_nnloss(x, lab, np.ones(1), loss=loss),
                  1e-6, loss + " bwd")
        close(vl.to_numpy(vl.vl_nnsoftmaxt(xd, temperature=2.0)), O.vl_nnsoftmaxt(x, 2.0), 1e-6, "softmaxt")
        # regression losses of emoVoxZoo.m:138-147
        w1 = O.F(rng.uniform(0.5, 2, (1, 1, 1, N)))
        for wt in (None, w1):
            wd = None if wt is None else vl.from_numpy(wt)
            close(vl.to_numpy(vl.vl_nneuclideanloss(xd, pd, instanceWeights=wd)).ravel()[0],
                  O.vl_nnregloss(x, p, kind="euclidean", instance_weights=wt), 1e-6, "euclid fwd")
            close(vl.to_numpy(vl.vl_nneuclideanloss(xd, pd, 0.5, instanceWeights=wd)),
                  O.vl_nnregloss(x, p, np.full(1, 0.5, np.float32), kind="euclidean", instance_weights=wt),
                  1e-6, "euclid bwd")
            for sg in (1.0, 0.7):
                close(vl.to_numpy(vl.vl_nnhuberloss(xd, pd, sigma=sg, instanceWeights=wd)).ravel()[0],
                      O.vl_nnregloss(x, p, kind="huber", sigma=sg, instance_weights=wt), 1e-6, "huber fwd")
                close(vl.to_numpy(vl.vl_nnhuberloss(xd, pd, 1.0, sigma=sg, instanceWeights=wd)),
                      O.vl_nnregloss(x, p, np.ones(1, np.float32), kind="huber", sigma=sg, instance_weights=wt),
                      1e-6, "huber bwd")
        with pytest.raises(ValueError):
            vl.vl_nnhuberloss(xd, pd, sigma=0.0)
        # vl_nnsoftmax(X, DZDY) / vl_nnsoftmaxt backward, spatial tensor and 'dim', 2 (student_stats.m:95)
        xs = O.F(rng.standard_normal((3, 5, 8, N)) * 2)
        ds = O.F(rng.standard_normal((3, 5, 8, N)))
        close(vl.to_numpy(vl.vl_nnsoftmax(vl.from_numpy(xs), vl.from_numpy(ds))),
              O.vl_nnsoftmaxt_backward(xs, ds, 1.0), 1e-6, "softmax bwd")
        close(vl.to_numpy(vl.vl_nnsoftmaxt(vl.from_numpy(xs), vl.from_numpy(ds), temperature=2.0)),
              O.vl_nnsoftmaxt_backward(xs, ds, 2.0), 1e-6, "softmaxt bwd")
        m = O.F(rng.standard_normal((6, 8)))
        got = vl.to_numpy(vl.vl_nnsoftmaxt(vl.from_numpy(m), temperature=2.0, dim=2))
        e = np.exp(m.astype(np.float64) / 2.0)
        close(got, (e / e.sum(1, keepdims=True)).astype(np.float32), 1e-6, "softmaxt dim 2")


def test_run_spec_front_end(gpu):
    """batch.runSpec (STFT as one strided convolution + magnitude kernel) vs the float64 FFT restatement;
    widths follow audSamp (getBatchEmoVoxCeleb.m:67-68): 48384 samples -> 512 x 300."""
    from mcncrossmodalemotions_amd import batch as xbatch, vl
    rng = np.random.default_rng(51)
    for L, N in ((int(xbatch.aud_samples(100)), 3), (48384, 2), (400, 1), (1000, 2)):
        z = O.F(rng.standard_normal((L, N)) * 0.1)
        ref = O.run_spec(z)
        got = vl.to_numpy(xbatch.runSpec(vl.from_numpy(z)))
        assert got.shape == ref.shape and ref.shape[0] == 512
        close(got, ref, 1e-4, "runSpec L=%d" % L)
    assert O.run_spec(np.zeros(48384)).shape[1] == 300
    # the whole student front-end: crop -> runSpec -> per-row normalisation
    z = O.F(rng.standard_normal((48384, 2)) * 0.1)
    got = vl.to_numpy(vl.spec_rownorm(xbatch.runSpec(vl.from_numpy(z))))
    close(got, O.spec_rownorm(O.run_spec(z)), 1e-3, "rownorm(runSpec)")


def test_crop_resize_face(gpu):
    """xm_crop_resize_face (getImageBatch from decoded frames, fetch_emovoxceleb_imdb.m:152-193) vs oracle:
    bit-exact (integer grey levels minus the mean)."""
    from mcncrossmodalemotions_amd import batch as xbatch, vl
    rng = np.random.default_rng(61)
    avg = (131.0912, 103.8827, 91.4953)
    for (Hin, Win, N, out) in ((358, 358, 2, (224, 224)), (300, 420, 3, (224, 224)), (64, 48, 1, (32, 40)),
                               (224, 224, 2, (224, 224))):
        src = O.F(rng.integers(0, 256, (Hin, Win, 3, N)))
        got = vl.to_numpy(vl.crop_resize_face(vl.from_numpy(src), avg, out))
        ref = O.crop_resize_face(src, avg, out)
        assert got.shape == ref.shape
        close(got, ref, 0, "crop+resize %dx%d" % (Hin, Win))
    # identity geometry (crop 1, same size) reduces to normalize_face
    src = O.F(rng.integers(0, 256, (40, 30, 3, 2)))
    a = vl.to_numpy(vl.crop_resize_face(vl.from_numpy(src), avg, (40, 30), crop=1.0))
    close(a, O.normalize_face(src, avg), 0, "identity geometry")
    f = vl.to_numpy(xbatch.getImageBatch(2, frameSize=(358, 358)))
    assert f.shape == (224, 224, 3, 2)


def test_class_stats(gpu):
    """xm_class_stats (dagnn.ErrorStats bookkeeping): accumulates per-class hits / population."""
    from mcncrossmodalemotions_amd import vl
    rng = np.random.default_rng(31)
    C = 8
    correct = vl.mat_zeros(C, 1)
    pop = vl.mat_zeros(C, 1)
    ref_c, ref_p = np.zeros(C), np.zeros(C)
    for N in (1, 7, 300, 64):
        x = rnd(rng, 1, 1, C, N)
        lab = O.F(rng.integers(1, C + 1, (1, 1, 1, N)))
        vl.class_stats(vl.from_numpy(x), vl.from_numpy(lab), correct, pop)
        pred = x[0, 0].argmax(0) + 1
        for c in range(1, C + 1):
            ref_p[c - 1] += np.sum(lab.ravel() == c)
            ref_c[c - 1] += np.sum((lab.ravel() == c) & (pred == c))
        close(vl.to_numpy(correct).ravel(), ref_c.astype(np.float32), 0, "correct")
        close(vl.to_numpy(pop).ravel(), ref_p.astype(np.float32), 0, "population")


def test_sgd_and_batch_math(gpu):
    from mcncrossmodalemotions_amd import vl
    import torch
    rng = np.random.default_rng(13)
    for n in (5, 1024, 4099):
        w, m, d = rnd(rng, n, 1), rnd(rng, n, 1), rnd(rng, n, 1)
        w_ref, m_ref = O.sgd_update(w, m, d, 1e-4, 0.9, 5e-4, 64)
        wd, md = vl.from_numpy(w), vl.from_numpy(m)
        vl.sgd_update(wd, md, vl.from_numpy(d), 1e-4, 0.9, 5e-4, 64)
        close(vl.to_numpy(wd), w_ref, 1e-7, "sgd w")
        close(vl.to_numpy(md), m_ref, 1e-7, "sgd m")
        wa = vl.from_numpy(w)
        vl.average_update(wa, vl.from_numpy(d), 0.1, 2)
        close(vl.to_numpy(wa), O.average_update(w, d, 0.1, 2), 1e-7, "avg update")
    spec = O.F(np.abs(rng.standard_normal((64, 30, 1, 3))) * 5 + 1)
    close(vl.to_numpy(vl.spec_rownorm(vl.from_numpy(spec))), O.spec_rownorm(spec), 1e-5, "rownorm")
    lg = rnd(rng, 40, 8)
    first = np.array([1, 3, 30, 40], np.int32)
    last = np.array([13, 3, 45, 40], np.int32)
    for agg in ("max", "mean"):
        out, lab = vl.aggregate_logits(vl.from_numpy(lg), torch.from_numpy(first).cuda(),
                                       torch.from_numpy(last).cuda(), agg)
        ref = np.stack([O.aggregate_logits(lg, f, l, agg) for f, l in zip(first, last)], 1)
        close(vl.to_numpy(out).reshape(8, 4), ref, 1e-6, "aggregate " + agg)
        close(vl.to_numpy(lab).ravel(), ref.argmax(0) + 1, 0, "maxLabel")
    rgb = O.F(rng.integers(0, 256, (12, 10, 3, 2)))
    avg = [131.1, 103.9, 91.5]
    close(vl.to_numpy(vl.normalize_face(vl.from_numpy(rgb), avg)), O.normalize_face(rgb, avg), 1e-6, "face")


def test_device_memory_entry_points(gpu):
    """xm_device_alloc / upload / download / free: the device-array substitute a MATLAB host uses,
    round trip + one operator on the raw buffers."""
    import ctypes as C
    from mcncrossmodalemotions_amd import _lib
    L = _lib.load()
    x = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
    px, py = C.c_void_p(), C.c_void_p()
    _lib.check(L.xm_device_alloc(C.byref(px), x.nbytes))
    _lib.check(L.xm_device_alloc(C.byref(py), x.nbytes))
    _lib.check(L.xm_device_upload(px, x.ctypes.data_as(C.c_void_p), x.nbytes))
    _lib.check(L.xm_nnrelu(px, x.size, 0.0, None, py, None))
    _lib.check(L.xm_device_synchronize())
    y = np.empty_like(x)
    _lib.check(L.xm_device_download(y.ctypes.data_as(C.c_void_p), py, x.nbytes))
    assert np.array_equal(y, np.maximum(x, 0))
    _lib.check(L.xm_device_free(px))
    _lib.check(L.xm_device_free(py))
    assert L.xm_device_alloc(None, 16) != 0 and b"NULL" in L.xm_last_error()


def test_tuning_table_persists(gpu, tmp_path):
    """xm_tune_save / xm_tune_load: a shape measured in this process lands in the file; loading the file in a fresh
    table reproduces the choice (the shipped tune_gfx950.txt makes tile choices identical across processes)."""
    import ctypes as C
    from mcncrossmodalemotions_amd import _lib, vl
    L = _lib.load()
    rng = np.random.default_rng(1)
    x = vl.from_numpy(rng.standard_normal((19, 23, 8, 3)).astype(np.float32))      # a geometry no table contains
    f = vl.from_numpy(rng.standard_normal((3, 3, 8, 24)).astype(np.float32))
    y0 = vl.to_numpy(vl.vl_nnconv(x, f, None, pad=1))
    path = str(tmp_path / "tune.txt").encode()
    _lib.check(L.xm_tune_save(path))
    tot, new = C.c_int(), C.c_int()
    _lib.check(L.xm_tune_entries(C.byref(tot), C.byref(new)))
    assert tot.value >= 1 and new.value == 0
    lines = open(path.decode()).read().splitlines()
    assert lines[0].startswith("xmodal-tune 1 rev=") and len(lines) - 1 == tot.value
    assert any(l.split()[:3] == ["0", "24", str(19 * 23 * 3)] for l in lines[1:])
    assert L.xm_tune_load(path) == 0          # everything in the file is already in the table
    assert np.array_equal(vl.to_numpy(vl.vl_nnconv(x, f, None, pad=1)), y0)


@pytest.mark.parametrize("geom", [(28, 28, 64, 48, 3), (14, 14, 160, 256, 5), (56, 8, 32, 200, 2), (4, 4, 1024, 96, 7)])
def test_lds_dma_conv_configs(gpu, geom):
    """conv_gemm_dma_kernel (both operands global -> LDS with buffer_load_dwordx4 ... lds): every DMA tile
    configuration, forced, on 1x1 unit-stride layers with ragged M / pixel counts -- plain, fused epilogue
    (bias + bnorm fold + residual + ReLU), and with a forced split-K -- against the oracle (fp64 accumulate)."""
    from mcncrossmodalemotions_amd import _lib, vl
    L = _lib.load()
    H, W, C, K, N = geom
    rng = np.random.default_rng(H * 1000 + C)
    x = O.F(rng.standard_normal((H, W, C, N)))
    f = O.F(rng.standard_normal((1, 1, C, K)) * 0.2)
    b = O.F(rng.standard_normal(K))
    sc, sh = O.F(rng.uniform(0.5, 1.5, K)), O.F(rng.standard_normal(K))
    y_ref = O.vl_nnconv(x, f, b, acc64=True)
    res = O.F(rng.standard_normal(y_ref.shape))
    yf_ref = np.maximum((y_ref * sc.reshape(1, 1, K, 1) + sh.reshape(1, 1, K, 1)) + res, 0)
    xd, fd, bd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(-1, 1))
    scd, shd, rd = vl.from_numpy(sc.reshape(-1, 1)), vl.from_numpy(sh.reshape(-1, 1)), vl.from_numpy(res)
    ncfg = L.xm_debug_num_conv_cfgs()
    assert ncfg >= 12
    try:
        for cfg in range(ncfg - 4, ncfg):
            for splits in (0, 3):
                L.xm_debug_force_conv_cfg(cfg)
                L.xm_debug_force_conv_splits(splits)
                y = vl.to_numpy(vl.vl_nnconv(xd, fd, bd))
                close(y, y_ref, 1e-5, "dma cfg %d splits %d" % (cfg, splits))
                yf = vl.to_numpy(vl.vl_nnconv(xd, fd, bd, scale=scd, shift=shd, residual=rd, relu=True))
                close(yf, yf_ref, 1e-5, "dma fused cfg %d splits %d" % (cfg, splits))
    finally:
        L.xm_debug_force_conv_cfg(-1)
        L.xm_debug_force_conv_splits(0)


@pytest.mark.parametrize("case", [(224, 224, 2, (3, 3, 3, 3)), (134, 256, 1, (3, 3, 3, 3)), (224, 224, 1, (3, 2, 2, 3))])
def test_conv_stem3_kernel(gpu, case):
    """The teachers' first layer (7 x 7 / stride 2 over three channels, 64 filters) through conv_stem3_kernel -- the input patch
    of a 128-pixel tile staged once in LDS, the 147 taps read from it with immediate offsets, filter rows padded to 8 --
    against the oracle: plain (bias) and with the folded bnorm + relu epilogue of the frozen teacher, next to the
    implicit-GEMM kernel on the same operands; the profiler hooks prove which kernel ran."""
    from mcncrossmodalemotions_amd import vl, _lib
    L = _lib.load()
    H, W, N, pad = case
    K = 64
    rng = np.random.default_rng(H + W + N)
    x, f, b = rnd(rng, H, W, 3, N), O.F(rng.standard_normal((7, 7, 3, K)) * 0.1), rnd(rng, K)
    sc, sh = O.F(rng.uniform(0.5, 1.5, K)), rnd(rng, K)
    y_ref = O.vl_nnconv(x, f, b, stride=2, pad=pad, acc64=True)
    yf_ref = np.maximum(y_ref * sc.reshape(1, 1, K, 1) + sh.reshape(1, 1, K, 1), 0)
    xd, fd, bd = vl.from_numpy(x), vl.from_numpy(f), vl.from_numpy(b.reshape(K, 1))
    scd, shd = vl.from_numpy(sc.reshape(K, 1)), vl.from_numpy(sh.reshape(K, 1))
    old = L.xm_debug_force_conv_stem3(1)
    try:
        y, names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, stride=2, pad=pad))
        assert any(n.startswith("conv_stem3_kernel") for n in names), names
        close(vl.to_numpy(y), y_ref, what="stem3 fwd")
        yf, names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, stride=2, pad=pad, scale=scd, shift=shd, relu=True))
        assert any(n.startswith("conv_stem3_kernel") for n in names), names
        close(vl.to_numpy(yf), yf_ref, what="stem3 fwd + folded bnorm + relu")
        L.xm_debug_force_conv_stem3(0)
        y0, names = _kernels_run(L, lambda: vl.vl_nnconv(xd, fd, bd, stride=2, pad=pad))
        assert not any(n.startswith("conv_stem3_kernel") for n in names), names
        close(vl.to_numpy(y0), y_ref, what="implicit GEMM")
        # 63 filters, or a pixel count that is not a multiple of 128 per sample, keep the implicit GEMM
        L.xm_debug_force_conv_stem3(1)
        _, names = _kernels_run(L, lambda: vl.vl_nnconv(xd, vl.from_numpy(f[..., :63].copy(order="F")), None, stride=2, pad=pad))
        assert not any(n.startswith("conv_stem3_kernel") for n in names), names
    finally:
        L.xm_debug_force_conv_stem3(old)
