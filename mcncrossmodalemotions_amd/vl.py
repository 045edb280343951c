"""Host-side mirror of the MatConvNet / mcnExtraLayers operator API, over the HIP C ABI.

Same names, argument meaning and error behaviour as the MATLAB operators the reference's
graphs execute:

    y            = vl_nnconv(x, f, b, stride=.., pad=.., dilate=..)
    dx, df, db   = vl_nnconv(x, f, b, dzdy, ...)           # backward when dzdy is given
    y            = vl_nnpool(x, pool, stride=.., pad=.., method='max'|'avg')
    y[, moments] = vl_nnbnorm(x, g, b, epsilon=1e-4, moments=M)
    ...

Tensors are torch float32 CUDA tensors in MATLAB layout: logical shape (H, W, C, N) with
column-major strides (H fastest) -- create them with `mat_empty/mat_zeros/from_numpy`.
torch is used for device memory and streams only; every operator below is one or more
hand-written HIP kernels from libxmodal_hip.so.  There is no CPU / eager fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

# --------------------------------------------------------------------------------------------
# MATLAB-layout tensor helpers
# --------------------------------------------------------------------------------------------


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


XM_ENOTSUP = 5   # include/xmodal.h


def mat_empty(*shape, device=None):
    """uninitialised `single` array of MATLAB shape `shape` (column-major)."""
    shape = tuple(int(s) for s in (shape[0] if len(shape) == 1 and not np.isscalar(shape[0]) else shape))
    t = torch.empty(tuple(reversed(shape)), dtype=torch.float32, device=device or _dev())
    return t.permute(*reversed(range(len(shape))))


def mat_zeros(*shape, device=None):
    t = mat_empty(*shape, device=device)
    t.zero_()
    return t


def from_numpy(a, device=None):
    """numpy array (any order) -> device tensor with MATLAB (column-major) layout."""
    a = np.asarray(a, dtype=np.float32)
    ct = np.ascontiguousarray(a.transpose(*reversed(range(a.ndim))))
    t = torch.from_numpy(ct).to(device or _dev())
    return t.permute(*reversed(range(a.ndim)))


def to_numpy(t):
    """device tensor in MATLAB layout -> numpy Fortran-ordered array of the same shape."""
    c = t.permute(*reversed(range(t.dim()))).contiguous().cpu().numpy()
    return np.asfortranarray(c.transpose(*reversed(range(c.ndim))))


def is_mat(t):
    return t.permute(*reversed(range(t.dim()))).is_contiguous()


def _chk(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s: expected a torch tensor" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s: expected single precision (float32), got %s" % (name, t.dtype))
    if not t.is_cuda:
        raise RuntimeError("%s: tensor is not on the GPU; this build has no CPU path" % name)
    if not is_mat(t):
        raise ValueError("%s: tensor is not in MATLAB column-major layout" % name)
    return t


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _shape4(t):
    s = tuple(t.shape) + (1,) * (4 - t.dim())
    if len(s) != 4:
        raise ValueError("expected an array with at most 4 dimensions, got %r" % (tuple(t.shape),))
    return [int(v) for v in s]


def _pair(v, name):
    if np.isscalar(v):
        return int(v), int(v)
    v = list(v)
    if len(v) == 1:
        return int(v[0]), int(v[0])
    if len(v) != 2:
        raise ValueError("%s must have 1 or 2 elements" % name)
    return int(v[0]), int(v[1])


def _pad4(pad):
    if np.isscalar(pad):
        return (int(pad),) * 4
    pad = list(pad)
    if len(pad) == 1:
        return (int(pad[0]),) * 4
    if len(pad) == 2:
        return int(pad[0]), int(pad[0]), int(pad[1]), int(pad[1])
    if len(pad) != 4:
        raise ValueError("PAD must have 1, 2 or 4 elements")
    return tuple(int(v) for v in pad)


def _L():
    return _lib.load()


def tune_save(path=None):
    """write the measured tile-configuration table next to the library (or to `path`); returns (total, new)"""
    tot, new = C.c_int(0), C.c_int(0)
    _lib.check(_L().xm_tune_entries(C.byref(tot), C.byref(new)))
    _lib.check(_L().xm_tune_save(path.encode() if path else None))
    return int(tot.value), int(new.value)


EXEC_SINGLE_STREAM = 1    # include/xmodal.h XM_EXEC_SINGLE_STREAM


def set_exec_hint(flags):
    """xm_set_exec_hint: tell the library HOW this host calls it (EXEC_SINGLE_STREAM: every operator call on one stream,
    MatConvNet's own sequence) -- kernel choice is a function of (shape, table, this hint), never of the call history.
    Returns the previous value."""
    old = int(_L().xm_get_exec_hint())
    _lib.check(_L().xm_set_exec_hint(int(flags)))
    return old


def out_size(n, pa, pb, f, d, s):
    return _L().xm_out_size(n, pa, pb, f, d, s)


# --------------------------------------------------------------------------------------------
# vl_nnconv
# --------------------------------------------------------------------------------------------


def vl_nnconv(x, f, b=None, dzdy=None, stride=1, pad=0, dilate=1, no_der_data=False,
              no_der_filters=False, no_der_biases=False, scale=None, shift=None, residual=None,
              relu=False, df_out=None, db_out=None, dx_accum=None, sigmoid=False, moments_out=None, epsilon=1e-4,
              gate=None):
    """Y = VL_NNCONV(X, F, B) / [DX, DF, DB] = VL_NNCONV(X, F, B, DZDY).

    `gate` (forward, extension; 1 x 1 x K x N): per-(channel, sample) multiplier between scale / shift and the residual --
    the SE excite folded into the projection that produces its operand (xm_nnconv_forward_gated).
    `moments_out` (forward, extension; a K x 2 device matrix): also receives the batch moments [mean, sqrt(var +
    epsilon)] of Y -- the statistics pass of the train-mode vl_nnbnorm that follows (xm_nnconv_forward_moments).
    `scale/shift/residual/relu/sigmoid` select the fused forward epilogue (extension; see xmodal.h);
    `dx_accum` (backward, extension): DX = dgrad + dx_accum in the dgrad epilogue."""
    x, f = _chk(x, "X"), _chk(f, "F")
    H, W, Cc, N = _shape4(x)
    FH, FW, FC, K = _shape4(f)
    sy, sx = _pair(stride, "STRIDE")
    dy, dx = _pair(dilate, "DILATE")
    pt, pb, pl, pr = _pad4(pad)
    bb = None
    if b is not None and b.numel() > 0:
        bb = _chk(b, "B")
        if bb.numel() != K:
            raise ValueError("vl_nnconv: B has %d elements, expected %d" % (bb.numel(), K))
    L = _L()
    Ho = L.xm_out_size(H, pt, pb, FH, dy, sy)
    Wo = L.xm_out_size(W, pl, pr, FW, dx, sx)
    if dzdy is None:
        y = mat_empty(max(Ho, 0), max(Wo, 0), K, N, device=x.device)
        fused = scale is not None or residual is not None or relu or sigmoid
        if gate is not None:
            gt = _chk(gate, "GATE")
            if gt.numel() != K * N:
                raise ValueError("vl_nnconv: GATE must be 1 x 1 x %d x %d" % (K, N))
            if residual is not None and _shape4(_chk(residual, "RESIDUAL")) != [Ho, Wo, K, N]:
                raise ValueError("vl_nnconv: residual shape mismatch")
            _lib.check(L.xm_nnconv_forward_gated(
                _ptr(x), H, W, Cc, N, _ptr(f), FH, FW, FC, K, _ptr(bb), _ptr(y), sy, sx, pt, pb, pl, pr, dy, dx,
                _ptr(scale), _ptr(shift), _ptr(gt), _ptr(residual), (1 if relu else 0) | (4 if sigmoid else 0), _stream()))
        elif moments_out is not None:
            if fused:
                raise ValueError("vl_nnconv: moments_out cannot be combined with a fused epilogue")
            mo = _chk(moments_out, "MOMENTS")
            if mo.numel() != 2 * K:
                raise ValueError("vl_nnconv: moments_out must be %d x 2" % K)
            _lib.check(L.xm_nnconv_forward_moments(_ptr(x), H, W, Cc, N, _ptr(f), FH, FW, FC, K, _ptr(bb), _ptr(y),
                                                   sy, sx, pt, pb, pl, pr, dy, dx, float(epsilon), _ptr(mo),
                                                   _stream()))
        elif fused:
            if residual is not None:
                _chk(residual, "RESIDUAL")
                if _shape4(residual) != [Ho, Wo, K, N]:
                    raise ValueError("vl_nnconv: residual shape mismatch")
            _lib.check(L.xm_nnconv_forward_fused(
                _ptr(x), H, W, Cc, N, _ptr(f), FH, FW, FC, K, _ptr(bb), _ptr(y), sy, sx, pt, pb,
                pl, pr, dy, dx, _ptr(scale), _ptr(shift), _ptr(residual), (1 if relu else 0) | (4 if sigmoid else 0),
                _stream()))
        else:
            _lib.check(L.xm_nnconv_forward(_ptr(x), H, W, Cc, N, _ptr(f), FH, FW, FC, K, _ptr(bb),
                                           _ptr(y), sy, sx, pt, pb, pl, pr, dy, dx, _stream()))
        return y
    dzdy = _chk(dzdy, "DZDY")
    if _shape4(dzdy) != [Ho, Wo, K, N]:
        raise ValueError("vl_nnconv: DZDY is %r, expected %r" % (tuple(dzdy.shape), (Ho, Wo, K, N)))
    dxo = None if no_der_data else mat_empty(H, W, Cc, N, device=x.device)
    # df_out / db_out: caller-owned destinations (e.g. views of the flat gradient buffer)
    dfo = None if no_der_filters else (df_out if df_out is not None else mat_empty(FH, FW, FC, K, device=x.device))
    dbo = None if (no_der_biases or bb is None) else (db_out if db_out is not None else mat_empty(K, 1, device=x.device))
    acc = None
    if dx_accum is not None and dxo is not None:
        acc = _chk(dx_accum, "DX_ACCUM")
        if _shape4(acc) != [H, W, Cc, N]:
            raise ValueError("vl_nnconv: dx_accum must have the size of X")
    _lib.check(L.xm_nnconv_backward_accum(_ptr(x), H, W, Cc, N, _ptr(f), FH, FW, FC, K, _ptr(dzdy),
                                          _ptr(dxo), _ptr(dfo), _ptr(dbo), sy, sx, pt, pb, pl, pr, dy, dx,
                                          _ptr(acc), _stream()))
    return dxo, dfo, dbo


def conv_prepare_backward(x, f, stride=1, pad=0, dilate=1):
    """Extension: build the transposed filter operand of the DZDX GEMM now, on the current stream (see xmodal.h);
    the backward call of the same layer finds it as long as the parameters have not been updated in between."""
    H, W, Cc, N = _shape4(x)
    FH, FW, FC, K = _shape4(f)
    sy, sx = _pair(stride, "STRIDE")
    dy, dx = _pair(dilate, "DILATE")
    pt, pb, pl, pr = _pad4(pad)
    _lib.check(_L().xm_nnconv_prepare_backward(H, W, Cc, N, _ptr(f), FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx,
                                               _stream()))


# --------------------------------------------------------------------------------------------
# vl_nnpool
# --------------------------------------------------------------------------------------------
_METHOD = {"max": 0, "avg": 1}


def vl_nnpool(x, pool, dzdy=None, stride=1, pad=0, method="max", argmax=None, want_argmax=False, dx_accum=None):
    """Y = VL_NNPOOL(X, POOL) / DX = VL_NNPOOL(X, POOL, DZDY).

    Extension for max pooling: `want_argmax=True` (forward) also returns the uint8 routing table
    of first maxima; pass it back as `argmax=` (backward) to skip the recomputation from X.
    Extension for global average pooling (POOL = the whole plane, the SE squeeze): `dx_accum` = the derivative another
    consumer of X already left; the result is DX + dx_accum in one pass (xm_nnpool_global_avg_backward_accum)."""
    x = _chk(x, "X")
    if method not in _METHOD:
        raise ValueError("vl_nnpool: unknown METHOD '%s'" % method)
    H, W, Cc, N = _shape4(x)
    ph, pw = _pair(pool, "POOL")
    sy, sx = _pair(stride, "STRIDE")
    pt, pb, pl, pr = _pad4(pad)
    L = _L()
    Ho = L.xm_out_size(H, pt, pb, ph, 1, sy)
    Wo = L.xm_out_size(W, pl, pr, pw, 1, sx)
    if dzdy is None:
        y = mat_empty(max(Ho, 0), max(Wo, 0), Cc, N, device=x.device)
        if want_argmax and method == "max":
            am = torch.empty(max(Ho, 0) * max(Wo, 0) * Cc * N, dtype=torch.uint8, device=x.device)
            _lib.check(L.xm_nnpool_forward_argmax(_ptr(x), H, W, Cc, N, ph, pw, sy, sx, pt, pb, pl, pr,
                                                  _ptr(y), C.c_void_p(am.data_ptr()), _stream()))
            return y, am
        _lib.check(L.xm_nnpool_forward(_ptr(x), H, W, Cc, N, ph, pw, sy, sx, pt, pb, pl, pr,
                                       _METHOD[method], _ptr(y), _stream()))
        return (y, None) if want_argmax else y
    dzdy = _chk(dzdy, "DZDY")
    if _shape4(dzdy) != [Ho, Wo, Cc, N]:
        raise ValueError("vl_nnpool: DZDY is %r, expected %r" % (tuple(dzdy.shape), (Ho, Wo, Cc, N)))
    dxo = mat_empty(H, W, Cc, N, device=x.device)
    if dx_accum is not None:
        if method != "avg" or (ph, pw) != (H, W) or (pt | pb | pl | pr) or _shape4(dx_accum) != [H, W, Cc, N]:
            raise ValueError("vl_nnpool: dx_accum is built for global average pooling only")
        _lib.check(L.xm_nnpool_global_avg_backward_accum(_ptr(dzdy), _ptr(_chk(dx_accum, "DX_ACCUM")), _ptr(dxo), H, W, Cc, N,
                                                         _stream()))
        return dxo
    if argmax is not None and method == "max":
        _lib.check(L.xm_nnpool_backward_argmax(C.c_void_p(argmax.data_ptr()), H, W, Cc, N, ph, pw, sy,
                                               sx, pt, pb, pl, pr, _ptr(dzdy), _ptr(dxo), _stream()))
    else:
        _lib.check(L.xm_nnpool_backward(_ptr(x), H, W, Cc, N, ph, pw, sy, sx, pt, pb, pl, pr,
                                        _METHOD[method], _ptr(dzdy), _ptr(dxo), _stream()))
    return dxo


# --------------------------------------------------------------------------------------------
# vl_nnbnorm
# --------------------------------------------------------------------------------------------


def vl_nnbnorm(x, g, b, dzdy=None, epsilon=1e-4, moments=None, relu=False, y=None, dg_out=None,
               db_out=None, moments_out=None, batch_moments=False, dxsum_out=None):
    """forward:  Y, MOMENTS = VL_NNBNORM(X, G, B);  backward: DX, DG, DB, MOMENTS = (..., DZDY).

    MOMENTS is C x 2 = [mean, sqrt(var + epsilon)].  `relu=True` fuses vl_nnrelu (forward) /
    its mask (backward; pass the fused forward output as `y`).  `batch_moments=True` (backward,
    extension): `moments` are the batch moments the forward call returned for this X -- train-mode
    derivative without recomputing them.  `dxsum_out` (backward, extension; C x 1): receives sum(DX) per channel =
    the DZDB of the vl_nnconv that produced X (xm_nnbnorm_backward_dxsum)."""
    x, g, b = _chk(x, "X"), _chk(g, "G"), _chk(b, "B")
    H, W, Cc, N = _shape4(x)
    if g.numel() != Cc or b.numel() != Cc:
        raise ValueError("vl_nnbnorm: G and B must have %d elements" % Cc)
    mi = None
    if moments is not None:
        mi = _chk(moments, "MOMENTS")
        if mi.numel() != 2 * Cc:
            raise ValueError("vl_nnbnorm: MOMENTS must be %d x 2" % Cc)
    L = _L()
    mo = moments_out if moments_out is not None else mat_empty(Cc, 2, device=x.device)
    if dzdy is None:
        yo = mat_empty(H, W, Cc, N, device=x.device)
        _lib.check(L.xm_nnbnorm_forward_fused(_ptr(x), H, W, Cc, N, _ptr(g), _ptr(b),
                                              float(epsilon), _ptr(mi), _ptr(yo), _ptr(mo),
                                              1 if relu else 0, _stream()))
        return yo, mo
    dzdy = _chk(dzdy, "DZDY")
    if _shape4(dzdy) != [H, W, Cc, N]:
        raise ValueError("vl_nnbnorm: DZDY shape mismatch")
    dxo = mat_empty(H, W, Cc, N, device=x.device)
    dg = dg_out if dg_out is not None else mat_empty(Cc, 1, device=x.device)
    db = db_out if db_out is not None else mat_empty(Cc, 1, device=x.device)
    if batch_moments and mi is None:
        raise ValueError("vl_nnbnorm: batch_moments needs the moments of the forward call")
    if dxsum_out is not None:
        if relu and y is None:
            raise ValueError("vl_nnbnorm: fused backward needs the forward output y")
        if _chk(dxsum_out, "DXSUM").numel() != Cc:
            raise ValueError("vl_nnbnorm: dxsum_out must have %d elements" % Cc)
        _lib.check(L.xm_nnbnorm_backward_dxsum(_ptr(x), _ptr(_chk(y, "Y")) if relu else None, H, W, Cc, N,
                                               _ptr(g), _ptr(b), _ptr(dzdy), float(epsilon), _ptr(mi),
                                               _ptr(dxo), _ptr(dg), _ptr(db), _ptr(mo), _ptr(dxsum_out),
                                               (1 if relu else 0) | (2 if batch_moments else 0), _stream()))
    elif relu or batch_moments:
        if relu and y is None:
            raise ValueError("vl_nnbnorm: fused backward needs the forward output y")
        _lib.check(L.xm_nnbnorm_backward_fused(_ptr(x), _ptr(_chk(y, "Y")) if relu else None, H, W, Cc, N,
                                               _ptr(g), _ptr(b), _ptr(dzdy), float(epsilon), _ptr(mi),
                                               _ptr(dxo), _ptr(dg), _ptr(db), _ptr(mo),
                                               (1 if relu else 0) | (2 if batch_moments else 0), _stream()))
    else:
        _lib.check(L.xm_nnbnorm_backward(_ptr(x), H, W, Cc, N, _ptr(g), _ptr(b), _ptr(dzdy),
                                         float(epsilon), _ptr(mi), _ptr(dxo), _ptr(dg), _ptr(db),
                                         _ptr(mo), _stream()))
    return dxo, dg, db, mo


def bnorm_relu_pool(x, g, b, pool, stride=1, pad=0, epsilon=1e-4, moments=None, moments_out=None):
    """Extension: vl_nnpool(vl_nnrelu(vl_nnbnorm(x, g, b)), pool, 'method', 'max') in one fused pass.
    Returns (y_pool, argmax_table, moments)."""
    x, g, b = _chk(x, "X"), _chk(g, "G"), _chk(b, "B")
    H, W, Cc, N = _shape4(x)
    ph, pw = _pair(pool, "POOL")
    sy, sx = _pair(stride, "STRIDE")
    pt, pb, pl, pr = _pad4(pad)
    L = _L()
    Ho, Wo = L.xm_out_size(H, pt, pb, ph, 1, sy), L.xm_out_size(W, pl, pr, pw, 1, sx)
    y = mat_empty(max(Ho, 0), max(Wo, 0), Cc, N, device=x.device)
    am = torch.empty(max(Ho, 0) * max(Wo, 0) * Cc * N, dtype=torch.uint8, device=x.device)
    mo = moments_out if moments_out is not None else mat_empty(Cc, 2, device=x.device)
    mi = None if moments is None else _chk(moments, "MOMENTS")
    _lib.check(L.xm_nnbnorm_relu_pool_forward(_ptr(x), H, W, Cc, N, _ptr(g), _ptr(b), float(epsilon),
                                              _ptr(mi), ph, pw, sy, sx, pt, pb, pl, pr, _ptr(y),
                                              C.c_void_p(am.data_ptr()), _ptr(mo), _stream()))
    return y, am, mo


def bnorm_relu_pool_backward(x, g, b, moments, argmax, dzdy, pool, stride=1, pad=0, train=True,
                             dg_out=None, db_out=None, need_dx=True, dxsum_out=None, y_pool=None):
    """Backward of bnorm_relu_pool: returns (dx, dg, db).  dxsum_out (optional, C x 1): receives
    sum(dx) per channel = the bias derivative of the convolution that produced x.  y_pool (optional): the
    forward's pooled output -- the per-channel sums then come from the pooled tensors alone."""
    x, g, b, dzdy = _chk(x, "X"), _chk(g, "G"), _chk(b, "B"), _chk(dzdy, "DZDY")
    H, W, Cc, N = _shape4(x)
    ph, pw = _pair(pool, "POOL")
    sy, sx = _pair(stride, "STRIDE")
    pt, pb, pl, pr = _pad4(pad)
    dx = mat_empty(H, W, Cc, N, device=x.device) if need_dx else None
    dg = dg_out if dg_out is not None else mat_empty(Cc, 1, device=x.device)
    db = db_out if db_out is not None else mat_empty(Cc, 1, device=x.device)
    _lib.check(_L().xm_nnbnorm_relu_pool_backward(
        _ptr(x), H, W, Cc, N, _ptr(g), _ptr(b), _ptr(_chk(moments, "MOMENTS")), 1 if train else 0, ph, pw,
        sy, sx, pt, pb, pl, pr, C.c_void_p(argmax.data_ptr()),
        None if y_pool is None else _ptr(_chk(y_pool, "Y_POOL")), _ptr(dzdy), _ptr(dx), _ptr(dg), _ptr(db),
        _ptr(dxsum_out), _stream()))
    return dx, dg, db


def conv_backward_filter_bnrelupool(x, filter_shape, y, g, b, moments, argmax, y_pool, dzdy, pool, stride=1, pad=0,
                                    dilate=1, pool_stride=1, pool_pad=0, train=True, df_out=None, dbias_out=None,
                                    dg_out=None, db_out=None, has_bias=True):
    """Extension (xm_nnconv_backward_filter_bnrelupool): [DZDF, DZDB] of the first-layer convolution Y = vl_nnconv(X, F, B)
    and [DG, DB] of the bnorm in  vl_nnpool(vl_nnrelu(vl_nnbnorm(Y, G, B)))  from the POOLED derivative `dzdy`: the
    bnorm's DZDX is never materialised.  Returns (df, dbias, dg, db), or None when the shapes are outside what the fused
    kernel covers (the caller then runs bnorm_relu_pool_backward + vl_nnconv backward)."""
    x, y, g, b, dzdy = _chk(x, "X"), _chk(y, "Y"), _chk(g, "G"), _chk(b, "B"), _chk(dzdy, "DZDY")
    H, W, Cc, N = _shape4(x)
    FH, FW, FC, K = (int(v) for v in filter_shape)
    sy, sx = _pair(stride, "STRIDE")
    dy, dx = _pair(dilate, "DILATE")
    pt, pb, pl, pr = _pad4(pad)
    ph, pw = _pair(pool, "POOL")
    psy, psx = _pair(pool_stride, "STRIDE")
    ppt, ppb, ppl, ppr = _pad4(pool_pad)
    df = df_out if df_out is not None else mat_empty(FH, FW, FC, K, device=x.device)
    dbias = (dbias_out if dbias_out is not None else mat_empty(K, 1, device=x.device)) if has_bias else None
    dg = dg_out if dg_out is not None else mat_empty(K, 1, device=x.device)
    db = db_out if db_out is not None else mat_empty(K, 1, device=x.device)
    rc = _L().xm_nnconv_backward_filter_bnrelupool(
        _ptr(x), H, W, Cc, N, FH, FW, FC, K, sy, sx, pt, pb, pl, pr, dy, dx, _ptr(y), _ptr(g), _ptr(b),
        _ptr(_chk(moments, "MOMENTS")), 1 if train else 0, ph, pw, psy, psx, ppt, ppb, ppl, ppr,
        C.c_void_p(argmax.data_ptr()), None if y_pool is None else _ptr(_chk(y_pool, "Y_POOL")), _ptr(dzdy), _ptr(df),
        _ptr(dbias), _ptr(dg), _ptr(db), _stream())
    if rc == XM_ENOTSUP:
        return None
    _lib.check(rc)
    return df, dbias, dg, db


def stem_gram(x, filter_shape, stride=1, pad=0):
    """Extension (xm_stem_gram): fp64 [64][64] Gram matrix of the im2col patches (+ ones) of a single-channel first
    layer; None when the geometry is not covered."""
    x = _chk(x, "X")
    H, W, Cc, N = _shape4(x)
    FH, FW = int(filter_shape[0]), int(filter_shape[1])
    sy, sx = _pair(stride, "STRIDE")
    pt, pb, pl, pr = _pad4(pad)
    if Cc != 1:
        return None
    gram = torch.empty(64 * 64, dtype=torch.float64, device=x.device)
    rc = _L().xm_stem_gram(_ptr(x), H, W, N, FH, FW, sy, sx, pt, pb, pl, pr, C.c_void_p(gram.data_ptr()), _stream())
    if rc == XM_ENOTSUP:
        return None
    _lib.check(rc)
    return gram


def stem_gram_moments(gram, f, b, epsilon=1e-4, moments_out=None):
    """Extension (xm_stem_gram_moments): vl_nnbnorm's MOMENTS of Y = vl_nnconv(X, F, B) from the Gram matrix of X."""
    f = _chk(f, "F")
    FH, FW, FC, K = _shape4(f)
    mo = moments_out if moments_out is not None else mat_empty(K, 2, device=f.device)
    _lib.check(_L().xm_stem_gram_moments(C.c_void_p(gram.data_ptr()), _ptr(f), _ptr(None if b is None else _chk(b, "B")),
                                         FH, FW, K, float(epsilon), _ptr(mo), _stream()))
    return mo


def conv_bnorm_relu_pool(x, f, bias, g, b, pool, stride=1, pad=0, dilate=1, pool_stride=1, pool_pad=0, epsilon=1e-4,
                         moments=None, moments_out=None, gram=None):
    """Extension (xm_nnconv_bnorm_relu_pool_forward): vl_nnpool(vl_nnrelu(vl_nnbnorm(vl_nnconv(x, f, bias), g, b))) for a
    single-channel first layer in one kernel -- the convolution's output is never written.  Returns
    (y_pool, argmax_table, moments, gram) or None when the shapes are not covered.  `moments` given = test mode (no Gram
    matrix); the table marks closed windows with 255 (include/xmodal.h)."""
    x, f, g, b = _chk(x, "X"), _chk(f, "F"), _chk(g, "G"), _chk(b, "B")
    H, W, Cc, N = _shape4(x)
    FH, FW, FC, K = _shape4(f)
    sy, sx = _pair(stride, "STRIDE")
    dy, dx = _pair(dilate, "DILATE")
    pt, pb, pl, pr = _pad4(pad)
    ph, pw = _pair(pool, "POOL")
    psy, psx = _pair(pool_stride, "STRIDE")
    ppt, ppb, ppl, ppr = _pad4(pool_pad)
    L = _L()
    Ho, Wo = L.xm_out_size(H, pt, pb, FH, dy, sy), L.xm_out_size(W, pl, pr, FW, dx, sx)
    pHo, pWo = L.xm_out_size(Ho, ppt, ppb, ph, 1, psy), L.xm_out_size(Wo, ppl, ppr, pw, 1, psx)
    if Ho <= 0 or Wo <= 0 or pHo <= 0 or pWo <= 0:
        return None
    y = mat_empty(pHo, pWo, K, N, device=x.device)
    am = torch.empty(pHo * pWo * K * N, dtype=torch.uint8, device=x.device)
    mi = None if moments is None else _chk(moments, "MOMENTS")
    mo = None
    if mi is None:
        mo = moments_out if moments_out is not None else mat_empty(K, 2, device=x.device)
        if gram is None:
            gram = torch.empty(64 * 64, dtype=torch.float64, device=x.device)
    rc = L.xm_nnconv_bnorm_relu_pool_forward(
        _ptr(x), H, W, Cc, N, _ptr(f), FH, FW, FC, K, None if bias is None else _ptr(_chk(bias, "B")), sy, sx, pt, pb, pl, pr,
        dy, dx, _ptr(g), _ptr(b), float(epsilon), _ptr(mi), ph, pw, psy, psx, ppt, ppb, ppl, ppr,
        None if gram is None else C.c_void_p(gram.data_ptr()), _ptr(y), C.c_void_p(am.data_ptr()), _ptr(mo), _stream())
    if rc == XM_ENOTSUP:
        return None
    _lib.check(rc)
    return y, am, (mi if mo is None else mo), gram


def conv_backward_filter_bnrelupool_gram(x, f, bias, g, moments, argmax, y_pool, dzdy, pool, stride=1, pad=0, dilate=1,
                                         pool_stride=1, pool_pad=0, train=True, gram=None, df_out=None, dbias_out=None,
                                         dg_out=None, db_out=None):
    """Extension (xm_nnconv_backward_filter_bnrelupool_gram): conv_backward_filter_bnrelupool without the convolution's
    output -- F / B take its place, the bnorm's sums and the normalisation's correction terms come from the Gram matrix of
    the input patches (include/xmodal.h).  Returns (df, dbias, dg, db) or None when the shapes are not covered."""
    x, f, g, dzdy = _chk(x, "X"), _chk(f, "F"), _chk(g, "G"), _chk(dzdy, "DZDY")
    H, W, Cc, N = _shape4(x)
    FH, FW, FC, K = _shape4(f)
    sy, sx = _pair(stride, "STRIDE")
    dy, dx = _pair(dilate, "DILATE")
    pt, pb, pl, pr = _pad4(pad)
    ph, pw = _pair(pool, "POOL")
    psy, psx = _pair(pool_stride, "STRIDE")
    ppt, ppb, ppl, ppr = _pad4(pool_pad)
    has_bias = bias is not None
    df = df_out if df_out is not None else mat_empty(FH, FW, FC, K, device=x.device)
    dbias = (dbias_out if dbias_out is not None else mat_empty(K, 1, device=x.device)) if has_bias else None
    dg = dg_out if dg_out is not None else mat_empty(K, 1, device=x.device)
    db = db_out if db_out is not None else mat_empty(K, 1, device=x.device)
    rc = _L().xm_nnconv_backward_filter_bnrelupool_gram(
        _ptr(x), H, W, Cc, N, _ptr(f), FH, FW, FC, K, _ptr(_chk(bias, "B")) if has_bias else None, sy, sx, pt, pb, pl, pr,
        dy, dx, _ptr(g), _ptr(_chk(moments, "MOMENTS")), 1 if train else 0, ph, pw, psy, psx, ppt, ppb, ppl, ppr,
        C.c_void_p(argmax.data_ptr()), None if y_pool is None else _ptr(_chk(y_pool, "Y_POOL")), _ptr(dzdy),
        None if gram is None else C.c_void_p(gram.data_ptr()), _ptr(df), _ptr(dbias), _ptr(dg), _ptr(db), _stream())
    if rc == XM_ENOTSUP:
        return None
    _lib.check(rc)
    return df, dbias, dg, db


# --------------------------------------------------------------------------------------------
# elementwise
# --------------------------------------------------------------------------------------------


def vl_nnrelu(x, dzdy=None, leak=0.0):
    x = _chk(x, "X")
    y = mat_empty(*x.shape, device=x.device)
    d = None if dzdy is None else _chk(dzdy, "DZDY")
    _lib.check(_L().xm_nnrelu(_ptr(x), x.numel(), float(leak), _ptr(d), _ptr(y), _stream()))
    return y


def vl_nndropout(x, dzdy=None, rate=0.5, mask=None, seed=0, offset=0):
    """[Y, MASK] = vl_nndropout(X, 'rate', r) / Y = vl_nndropout(X, 'mask', M) / DZDX = vl_nndropout(X, DZDY, 'mask', M).
    Without a mask one is drawn from the library's stateless Philox stream (seed, offset: include/xmodal.h)."""
    x = _chk(x, "X")
    y = mat_empty(*x.shape, device=x.device)
    if dzdy is not None:
        if mask is None:
            raise ValueError("vl_nndropout: the backward call needs the forward call's mask")
        _lib.check(_L().xm_nndropout_apply(_ptr(_chk(dzdy, "DZDY")), _ptr(_chk(mask, "MASK")), x.numel(), _ptr(y), _stream()))
        return y
    if mask is not None:
        _lib.check(_L().xm_nndropout_apply(_ptr(x), _ptr(_chk(mask, "MASK")), x.numel(), _ptr(y), _stream()))
        return y, mask
    mask = mat_empty(*x.shape, device=x.device)
    _lib.check(_L().xm_nndropout_forward(_ptr(x), x.numel(), float(rate), int(seed), int(offset), _ptr(y), _ptr(mask),
                                         _stream()))
    return y, mask


def vl_nnsigmoid(x, dzdy=None):
    x = _chk(x, "X")
    y = mat_empty(*x.shape, device=x.device)
    d = None if dzdy is None else _chk(dzdy, "DZDY")
    _lib.check(_L().xm_nnsigmoid(_ptr(x), x.numel(), _ptr(d), _ptr(y), _stream()))
    return y


def sum2(a, b, relu=False):
    """dagnn.Sum over two inputs (+ optional fused vl_nnrelu)."""
    a, b = _chk(a, "A"), _chk(b, "B")
    if a.shape != b.shape:
        raise ValueError("dagnn.Sum: input sizes differ")
    y = mat_empty(*a.shape, device=a.device)
    _lib.check(_L().xm_sum2(_ptr(a), _ptr(b), a.numel(), 1 if relu else 0, _ptr(y), _stream()))
    return y


def scale_axpy(x, a, r=None, relu=False):
    """y = a .* x (+ r) (relu): the SE-block excite + residual of SENet50 (mcnExtraLayers)."""
    x, a = _chk(x, "X"), _chk(a, "A")
    H, W, Cc, N = _shape4(x)
    if a.numel() != Cc * N:
        raise ValueError("scale: A must be 1 x 1 x %d x %d" % (Cc, N))
    rr = None if r is None else _chk(r, "R")
    y = mat_empty(H, W, Cc, N, device=x.device)
    _lib.check(_L().xm_scale_axpy(_ptr(x), H * W, Cc * N, _ptr(a), _ptr(rr), 1 if relu else 0,
                                  _ptr(y), _stream()))
    return y


def se_squeeze_bn(u, g, b, moments):
    """gp = mean_hw(vl_nnbnorm(u, g, b, 'moments', moments)) without materialising the bnorm's output (xm_se_squeeze_bn)"""
    u = _chk(u, "U")
    H, W, Cc, N = _shape4(u)
    gp = mat_empty(1, 1, Cc, N, device=u.device)
    _lib.check(_L().xm_se_squeeze_bn(_ptr(u), H, W, Cc, N, _ptr(_chk(g, "G")), _ptr(_chk(b, "B")),
                                     _ptr(_chk(moments, "MOMENTS")), _ptr(gp), _stream()))
    return gp


def scale_axpy_bn(u, a, r, g, b, moments, relu=False):
    """y = [relu](a .* vl_nnbnorm(u, g, b, 'moments', moments) + r) (xm_scale_axpy_bn)"""
    u, a = _chk(u, "U"), _chk(a, "A")
    H, W, Cc, N = _shape4(u)
    if a.numel() != Cc * N:
        raise ValueError("scale: A must be 1 x 1 x %d x %d" % (Cc, N))
    y = mat_empty(H, W, Cc, N, device=u.device)
    _lib.check(_L().xm_scale_axpy_bn(_ptr(u), H, W, Cc, N, _ptr(a), _ptr(None if r is None else _chk(r, "R")),
                                     _ptr(_chk(g, "G")), _ptr(_chk(b, "B")), _ptr(_chk(moments, "MOMENTS")),
                                     1 if relu else 0, _ptr(y), _stream()))
    return y


def se_tail_backward_reduce(y, dzdy, u, g, b, moments):
    """first half of the fused SE-tail backward (xm_se_tail_backward_reduce): returns (da 1 x 1 x C x N, plane sums)"""
    y, dzdy, u = _chk(y, "Y"), _chk(dzdy, "DZDY"), _chk(u, "U")
    H, W, Cc, N = _shape4(u)
    da = mat_empty(1, 1, Cc, N, device=u.device)
    sums = torch.empty(3 * Cc * N, dtype=torch.float64, device=u.device)
    _lib.check(_L().xm_se_tail_backward_reduce(_ptr(y), _ptr(dzdy), _ptr(u), H, W, Cc, N, _ptr(_chk(g, "G")), _ptr(_chk(b, "B")),
                                               _ptr(_chk(moments, "MOMENTS")), _ptr(da), C.c_void_p(sums.data_ptr()), _stream()))
    return da, sums


def se_tail_backward_apply(y, dzdy, u, gate, dgp, g, moments, sums, train=True, dg_out=None, db_out=None):
    """second half (xm_se_tail_backward_apply): returns (dz = the shortcut's derivative, du, dg, db)"""
    y, dzdy, u = _chk(y, "Y"), _chk(dzdy, "DZDY"), _chk(u, "U")
    H, W, Cc, N = _shape4(u)
    dz = mat_empty(H, W, Cc, N, device=u.device)
    du = mat_empty(H, W, Cc, N, device=u.device)
    dg = dg_out if dg_out is not None else mat_empty(Cc, 1, device=u.device)
    db = db_out if db_out is not None else mat_empty(Cc, 1, device=u.device)
    _lib.check(_L().xm_se_tail_backward_apply(_ptr(y), _ptr(dzdy), _ptr(u), H, W, Cc, N, _ptr(_chk(gate, "A")),
                                              _ptr(_chk(dgp, "DGP")), _ptr(_chk(g, "G")), _ptr(_chk(moments, "MOMENTS")),
                                              1 if train else 0, C.c_void_p(sums.data_ptr()), _ptr(dz), _ptr(du), _ptr(dg),
                                              _ptr(db), _stream()))
    return dz, du, dg, db


def scale_backward(x, a, dzdy, need_dx=True):
    x, a, dzdy = _chk(x, "X"), _chk(a, "A"), _chk(dzdy, "DZDY")
    H, W, Cc, N = _shape4(x)
    dx = mat_empty(H, W, Cc, N, device=x.device) if need_dx else None
    da = mat_empty(1, 1, Cc, N, device=x.device)
    _lib.check(_L().xm_scale_backward(_ptr(x), H * W, Cc * N, _ptr(a), _ptr(dzdy), _ptr(dx),
                                      _ptr(da), _stream()))
    return dx, da


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------


def vl_nnsoftmaxt(x, dzdy=None, temperature=1.0, dim=3):
    """Y = VL_NNSOFTMAXT(X, 'temperature', T, 'dim', d); DZDX = VL_NNSOFTMAXT(X, DZDY, ...).
    dim is 1-based as in MATLAB (student_stats.m:95 uses 'dim', 2)."""
    x = _chk(x, "X")
    shp = [int(s) for s in x.shape] + [1] * (4 - x.dim())
    d = int(dim) - 1
    HW = int(np.prod(shp[:d])) if d > 0 else 1
    Cc = shp[d]
    N = int(np.prod(shp[d + 1:])) if d < 3 else 1
    y = mat_empty(*x.shape, device=x.device)
    if dzdy is None:
        _lib.check(_L().xm_nnsoftmaxt(_ptr(x), HW, Cc, N, float(temperature), _ptr(y), _stream()))
        return y
    dzdy = _chk(dzdy, "DZDY")
    if tuple(dzdy.shape) != tuple(x.shape):
        raise ValueError("vl_nnsoftmaxt: DZDY must have the size of X")
    _lib.check(_L().xm_nnsoftmaxt_backward(_ptr(x), _ptr(dzdy), HW, Cc, N, float(temperature), _ptr(y),
                                           _stream()))
    return y


def vl_nnsoftmax(x, dzdy=None):
    """Y = VL_NNSOFTMAX(X); DZDX = VL_NNSOFTMAX(X, DZDY) -- channel softmax along dim 3."""
    return vl_nnsoftmaxt(x, dzdy, 1.0, 3)


def vl_nnsoftmaxceloss(x, p, dzdy=None, temperature=1.0, logitTargets=False, instanceWeights=None):
    """VL_NNSOFTMAXCELOSS(X, P [, DZDY], 'temperature', T, 'logitTargets', tf, ...)."""
    x, p = _chk(x, "X"), _chk(p, "P")
    H, W, Cc, N = _shape4(x)
    if H != 1 or W != 1:
        raise ValueError("vl_nnsoftmaxceloss: X must be 1 x 1 x C x N")
    if _shape4(p) != [1, 1, Cc, N]:
        raise ValueError("vl_nnsoftmaxceloss: P must have the size of X")
    w = None if instanceWeights is None else _chk(instanceWeights, "instanceWeights")
    if dzdy is None:
        y = mat_empty(1, 1, device=x.device)
        _lib.check(_L().xm_nnsoftmaxceloss(_ptr(x), _ptr(p), Cc, N, float(temperature),
                                           1 if logitTargets else 0, _ptr(w), None, _ptr(y),
                                           _stream()))
        return y
    if not isinstance(dzdy, torch.Tensor):
        dzdy = from_numpy(np.array([[float(dzdy)]], np.float32), device=x.device)
    y = mat_empty(1, 1, Cc, N, device=x.device)
    _lib.check(_L().xm_nnsoftmaxceloss(_ptr(x), _ptr(p), Cc, N, float(temperature),
                                       1 if logitTargets else 0, _ptr(w), _ptr(dzdy), _ptr(y),
                                       _stream()))
    return y


_LOSS = {"softmaxlog": 0, "classerror": 1}


def vl_nnloss(x, c, dzdy=None, loss="softmaxlog"):
    """VL_NNLOSS(X, c [, DZDY], 'loss', 'softmaxlog' | 'classerror'); c holds 1-based labels."""
    x, c = _chk(x, "X"), _chk(c, "C")
    H, W, Cc, N = _shape4(x)
    if H != 1 or W != 1:
        raise ValueError("vl_nnloss: X must be 1 x 1 x C x N")
    if loss not in _LOSS:
        raise ValueError("vl_nnloss: unknown loss '%s'" % loss)
    if c.numel() != N:
        raise ValueError("vl_nnloss: need one label per sample")
    if dzdy is None:
        y = mat_empty(1, 1, device=x.device)
        _lib.check(_L().xm_nnloss(_ptr(x), _ptr(c), Cc, N, _LOSS[loss], None, _ptr(y), _stream()))
        return y
    if not isinstance(dzdy, torch.Tensor):
        dzdy = from_numpy(np.array([[float(dzdy)]], np.float32), device=x.device)
    y = mat_empty(1, 1, Cc, N, device=x.device)
    _lib.check(_L().xm_nnloss(_ptr(x), _ptr(c), Cc, N, _LOSS[loss], _ptr(dzdy), _ptr(y), _stream()))
    return y


def _regloss(x, t, dzdy, kind, sigma, instanceWeights, name):
    x, t = _chk(x, "X"), _chk(t, "T")
    if tuple(x.shape) != tuple(t.shape):
        raise ValueError("%s: X and T must have the same size" % name)
    N = int(x.shape[3]) if x.dim() > 3 else 1
    E = int(x.numel()) // N
    w = None if instanceWeights is None else _chk(instanceWeights, "instanceWeights")
    if w is not None and int(w.numel()) != N:
        raise ValueError("%s: need one instance weight per sample" % name)
    if dzdy is None:
        y = mat_empty(1, 1, device=x.device)
        _lib.check(_L().xm_nnregloss(_ptr(x), _ptr(t), E, N, kind, float(sigma), _ptr(w), None, _ptr(y),
                                     _stream()))
        return y
    if not isinstance(dzdy, torch.Tensor):
        dzdy = from_numpy(np.array([[float(dzdy)]], np.float32), device=x.device)
    y = mat_empty(*x.shape, device=x.device)
    _lib.check(_L().xm_nnregloss(_ptr(x), _ptr(t), E, N, kind, float(sigma), _ptr(w), _ptr(dzdy), _ptr(y),
                                 _stream()))
    return y


def vl_nneuclideanloss(x, t, dzdy=None, instanceWeights=None):
    """VL_NNEUCLIDEANLOSS(X, T [, DZDY], 'instanceWeights', w) -- mcnExtraLayers (emoVoxZoo.m:139)."""
    return _regloss(x, t, dzdy, 0, 1.0, instanceWeights, "vl_nneuclideanloss")


def vl_nnhuberloss(x, t, dzdy=None, sigma=1.0, instanceWeights=None):
    """VL_NNHUBERLOSS(X, T [, DZDY], 'sigma', s, 'instanceWeights', w) -- mcnExtraLayers (emoVoxZoo.m:147)."""
    if not sigma > 0:
        raise ValueError("vl_nnhuberloss: sigma must be positive")
    return _regloss(x, t, dzdy, 1, sigma, instanceWeights, "vl_nnhuberloss")


# --------------------------------------------------------------------------------------------
# optimiser / parameter server
# --------------------------------------------------------------------------------------------


def sgd_update(w, m, der, lr, momentum=0.9, weight_decay=5e-4, batch=1.0):
    """in-place accumulateGradients step of cnn_train_dag (trainMethod 'gradient')."""
    _lib.check(_L().xm_sgd_update(_ptr(w), _ptr(m), _ptr(der), w.numel(), float(lr),
                                  float(momentum), float(weight_decay), float(batch), _stream()))


def scale_(x, a):
    """x <- a * x in place (worker-batch weighting of the BN moments before the ParameterServer exchange)."""
    _lib.check(_L().xm_scale_f32(_ptr(x), x.numel(), float(a), _stream()))


def average_update(w, der, lr, nworkers=1.0):
    """in-place trainMethod 'average' update (BN moments): w <- (1-lr) w + lr der / nworkers (denominator: 1 for a
    single worker, the GLOBAL batch size when der = sum over workers of moments * worker batch size)."""
    _lib.check(_L().xm_average_update(_ptr(w), _ptr(der), w.numel(), float(lr), float(nworkers),
                                      _stream()))


# --------------------------------------------------------------------------------------------
# batch-provider arithmetic
# --------------------------------------------------------------------------------------------


def spec_rownorm(spec):
    """getBatchEmoVoxCeleb.m:164-169 on the device; spec is H x W x 1 x N."""
    spec = _chk(spec, "SPEC")
    H, W, Cc, N = _shape4(spec)
    out = mat_empty(*spec.shape, device=spec.device)
    _lib.check(_L().xm_spec_rownorm(_ptr(spec), H, W, Cc * N, _ptr(out), _stream()))
    return out


def spec_magnitude(reim):
    """|Re + i Im| of the framing convolution's output: 1 x Wo x 2B x N -> B x Wo x 1 x N."""
    reim = _chk(reim, "REIM")
    H, Wo, C2, N = _shape4(reim)
    if H != 1 or C2 % 2:
        raise ValueError("spec_magnitude: expected a 1 x Wo x 2B x N tensor")
    out = mat_empty(C2 // 2, Wo, 1, N, device=reim.device)
    _lib.check(_L().xm_spec_magnitude(_ptr(reim), Wo, C2 // 2, N, _ptr(out), _stream()))
    return out


def resample(x, h, p, q, delay, Ly):
    """y = upfirdn(x, h, p, q) without the filter delay, Ly samples (xm_resample); x, h: 1-D device tensors."""
    y = torch.empty(int(Ly), dtype=torch.float32, device=x.device)
    _lib.check(_L().xm_resample(_ptr(x), int(x.numel()), _ptr(h), int(h.numel()), int(p), int(q), int(delay), _ptr(y),
                                int(Ly), _stream()))
    return y


def aggregate_logits(frame_logits, first, last, agg="max"):
    """frame_logits F x E (column-major), first/last int32 device vectors (1-based, inclusive).
    Returns (logitTarget 1 x 1 x E x N, maxLabel 1 x 1 x 1 x N)."""
    fl = _chk(frame_logits, "LOGITS")
    Fr, E = int(fl.shape[0]), int(fl.shape[1])
    N = int(first.numel())
    out = mat_empty(1, 1, E, N, device=fl.device)
    lab = mat_empty(1, 1, 1, N, device=fl.device)
    if agg not in ("max", "mean"):
        raise ValueError("unreccognised aggregator %s" % agg)
    _lib.check(_L().xm_aggregate_logits(_ptr(fl), Fr, E, C.c_void_p(first.data_ptr()),
                                        C.c_void_p(last.data_ptr()), N, 0 if agg == "max" else 1,
                                        _ptr(out), _ptr(lab), _stream()))
    return out, lab


def max_label(lgo):
    """[~, maxLabel] = max(lgo, [], 3) -- getBatchEmoVoxCeleb.m:32; lgo is 1 x 1 x C x N."""
    lgo = _chk(lgo, "LGO")
    H, W, Cc, N = _shape4(lgo)
    lab = mat_empty(1, 1, 1, N, device=lgo.device)
    _lib.check(_L().xm_max_label(_ptr(lgo), Cc, N, _ptr(lab), _stream()))
    return lab


def class_stats(x, labels, correct, population):
    """dagnn.ErrorStats bookkeeping: correct[c] / population[c] += ... for the samples of this batch
    (x: 1 x 1 x C x N scores, labels: 1-based, correct / population: C-element device arrays)."""
    x, labels = _chk(x, "X"), _chk(labels, "LABELS")
    H, W, Cc, N = _shape4(x)
    if H != 1 or W != 1 or labels.numel() != N:
        raise ValueError("class_stats: X must be 1 x 1 x C x N with one label per sample")
    if correct.numel() != Cc or population.numel() != Cc:
        raise ValueError("class_stats: counters must have C elements")
    _lib.check(_L().xm_class_stats(_ptr(x), _ptr(labels), Cc, N, _ptr(correct), _ptr(population), _stream()))


def crop_resize_face(src, average_image, image_size=(224, 224), crop=1 / 1.6):
    """getImageBatch of fetch_emovoxceleb_imdb.m:152-193 from decoded frames (Hin x Win x 3 x N, values
    0..255): centre crop 1/1.6 -> bilinear resize -> uint8 -> grey -> x3 -> minus averageImage, fused."""
    src = _chk(src, "SRC")
    Hin, Win, c3, N = _shape4(src)
    if c3 != 3:
        raise ValueError("crop_resize_face: expected Hin x Win x 3 x N")
    avg = (C.c_float * 3)(*[float(v) for v in np.ravel(average_image)[:3]])
    out = mat_empty(int(image_size[0]), int(image_size[1]), 3, N, device=src.device)
    _lib.check(_L().xm_crop_resize_face(_ptr(src), Hin, Win, N, float(crop), int(image_size[0]),
                                        int(image_size[1]), avg, _ptr(out), _stream()))
    return out


def normalize_face(rgb, average_image):
    """fetch_emovoxceleb_imdb.m:176-193: grey -> x3 -> minus averageImage; rgb H x W x 3 x N."""
    rgb = _chk(rgb, "RGB")
    H, W, c3, N = _shape4(rgb)
    if c3 != 3:
        raise ValueError("normalize_face: expected H x W x 3 x N")
    avg = (C.c_float * 3)(*[float(v) for v in np.ravel(average_image)[:3]])
    out = mat_empty(H, W, 3, N, device=rgb.device)
    _lib.check(_L().xm_normalize_face(_ptr(rgb), H, W, N, avg, _ptr(out), _stream()))
    return out
