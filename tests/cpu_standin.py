"""TEST-ONLY CPU stand-in for the HIP operators (never imported by the product).

The multi-GPU logic of the product -- train.train_step / process_epoch / GradBuckets / ParameterServer, the
dagnn executor, flat parameter packing -- is host code around the C ABI.  To run THAT code end to end on the CPU
(gloo, world_size 2) the functions of `mcncrossmodalemotions_amd.vl` are replaced, inside the test process only,
by thin wrappers over the CPU oracle (fp64-accumulate path) working on CPU torch tensors in MATLAB layout, and
DagNN.move('gpu') is redirected to the CPU.  Nothing here is a fallback of the product: without `install()` the
package raises as soon as an operator is touched on a machine without the HIP library / a GPU.
"""
import numpy as np
import torch

from oracle import oracle as O


def _np(t):
    """CPU tensor in MATLAB layout -> Fortran-ordered numpy"""
    if t is None:
        return None
    c = t.permute(*reversed(range(t.dim()))).contiguous().numpy()
    return np.asfortranarray(c.transpose(*reversed(range(c.ndim))))


def _mat(a, device=None):
    a = np.asarray(a, dtype=np.float32)
    ct = np.ascontiguousarray(a.transpose(*reversed(range(a.ndim))))
    return torch.from_numpy(ct.copy()).permute(*reversed(range(a.ndim)))


def _mat_empty(*shape, device=None):
    shape = tuple(int(s) for s in (shape[0] if len(shape) == 1 and not np.isscalar(shape[0]) else shape))
    return torch.zeros(tuple(reversed(shape)), dtype=torch.float32).permute(*reversed(range(len(shape))))


def _into(dst, val):
    """write numpy `val` into a caller-owned destination tensor (flat-buffer view) and return it"""
    if dst is None:
        return _mat(val)
    dst.copy_(_mat(np.reshape(val, tuple(dst.shape), order="F")))
    return dst


def vl_nnconv(x, f, b=None, dzdy=None, stride=1, pad=0, dilate=1, no_der_data=False, no_der_filters=False,
              no_der_biases=False, scale=None, shift=None, residual=None, relu=False, df_out=None, db_out=None,
              dx_accum=None):
    if dzdy is None:
        assert scale is None and residual is None and not relu, "stand-in: run the net with fuse=False"
        return _mat(O.vl_nnconv(_np(x), _np(f), _np(b), stride=stride, pad=pad, dilate=dilate, acc64=True))
    dx, df, db = O.vl_nnconv(_np(x), _np(f), _np(b), _np(dzdy), stride=stride, pad=pad, dilate=dilate, acc64=True,
                             no_der_data=no_der_data, no_der_filters=no_der_filters,
                             no_der_biases=no_der_biases or b is None)
    if dx is not None and dx_accum is not None:
        dx = dx + _np(dx_accum)
    return (None if dx is None else _mat(dx), None if df is None else _into(df_out, df),
            None if db is None else _into(db_out, db.reshape(-1, 1)))


def vl_nnbnorm(x, g, b, dzdy=None, epsilon=1e-4, moments=None, relu=False, y=None, dg_out=None, db_out=None,
               moments_out=None, batch_moments=False, dxsum_out=None):
    assert not relu and dxsum_out is None, "stand-in: run the net with fuse=False"
    if dzdy is None:
        yy, mom = O.vl_nnbnorm(_np(x), _np(g), _np(b), epsilon=epsilon, moments=_np(moments), acc64=True)
        return _mat(yy), _into(moments_out, mom)
    mi = None if batch_moments else _np(moments)       # batch moments: the oracle recomputes them (same values)
    dx, dg, db, mom = O.vl_nnbnorm(_np(x), _np(g), _np(b), _np(dzdy), epsilon=epsilon, moments=mi, acc64=True)
    return _mat(dx), _into(dg_out, dg.reshape(-1, 1)), _into(db_out, db.reshape(-1, 1)), _into(moments_out, mom)


def vl_nnrelu(x, dzdy=None, leak=0.0):
    return _mat(O.vl_nnrelu(_np(x), _np(dzdy), leak=leak))


def vl_nnpool(x, pool, dzdy=None, stride=1, pad=0, method="max", argmax=None, want_argmax=False, dx_accum=None):
    if dzdy is None:
        y = _mat(O.vl_nnpool(_np(x), pool, stride=stride, pad=pad, method=method))
        return (y, None) if want_argmax else y
    dx = O.vl_nnpool(_np(x), pool, _np(dzdy), stride=stride, pad=pad, method=method)
    return _mat(dx if dx_accum is None else dx + _np(dx_accum))


def sum2(a, b, relu=False):
    return _mat(O.sum2(_np(a), _np(b), relu=relu))


def vl_nnsoftmaxceloss(x, p, dzdy=None, temperature=1.0, logitTargets=False, instanceWeights=None):
    if dzdy is None:
        return _mat(np.array([[O.vl_nnsoftmaxceloss(_np(x), _np(p), temperature=temperature,
                                                    logit_targets=logitTargets)]], np.float32))
    return _mat(O.vl_nnsoftmaxceloss(_np(x), _np(p), _np(dzdy).ravel(), temperature=temperature,
                                     logit_targets=logitTargets))


def vl_nnloss(x, c, dzdy=None, loss="softmaxlog"):
    if dzdy is None:
        return _mat(np.array([[O.vl_nnloss(_np(x), _np(c), loss=loss)]], np.float32))
    return _mat(O.vl_nnloss(_np(x), _np(c), _np(dzdy).ravel(), loss=loss))


def class_stats(x, labels, correct, population):
    xs, lab = _np(x).reshape(x.shape[2], -1, order="F"), _np(labels).ravel().astype(int)
    for n, c in enumerate(lab):
        population.reshape(-1)[c - 1] += 1
        correct.reshape(-1)[c - 1] += float(int(xs[:, n].argmax()) + 1 == c)


def max_label(lgo):
    a = _np(lgo)
    return _mat(a.reshape(a.shape[2], -1, order="F").argmax(0).reshape(1, 1, 1, -1).astype(np.float32) + 1)


def sgd_update(w, m, der, lr, momentum=0.9, weight_decay=5e-4, batch=1.0):
    # cnn_train_dag accumulateGradients, trainMethod 'gradient' (oracle: orc_sgd_update)
    wn, mn = O.sgd_update(w.numpy().copy(), m.numpy().copy(), der.numpy(), lr, momentum, weight_decay, batch)
    w.copy_(torch.from_numpy(np.ascontiguousarray(wn)))
    m.copy_(torch.from_numpy(np.ascontiguousarray(mn)))


def average_update(w, der, lr, nworkers=1.0):
    w.copy_(torch.from_numpy(np.ascontiguousarray(O.average_update(w.numpy().copy(), der.numpy(), lr, nworkers))))


def scale_(x, a):
    x.mul_(float(a))


def install():
    """patch mcncrossmodalemotions_amd.vl / dagnn for this process; returns an undo function"""
    from mcncrossmodalemotions_amd import dagnn, vl
    names = ["vl_nnconv", "vl_nnbnorm", "vl_nnrelu", "vl_nnpool", "sum2", "vl_nnsoftmaxceloss", "vl_nnloss",
             "class_stats", "max_label", "sgd_update", "average_update", "scale_"]
    saved = {n: getattr(vl, n) for n in names}
    saved.update(from_numpy=vl.from_numpy, to_numpy=vl.to_numpy, mat_zeros=vl.mat_zeros, mat_empty=vl.mat_empty)
    g = globals()
    for n in names:
        setattr(vl, n, g[n])
    vl.from_numpy = _mat
    vl.to_numpy = _np
    vl.mat_zeros = _mat_empty
    vl.mat_empty = _mat_empty
    old_move = dagnn.DagNN.move

    def move(self, device="gpu"):
        self.device = torch.device("cpu")
        for p in self.params.values():
            if p.value is not None and not isinstance(p.value, torch.Tensor):
                p.value = _mat(p.value)
        return self

    dagnn.DagNN.move = move

    def undo():
        for n, fn in saved.items():
            setattr(vl, n, fn)
        dagnn.DagNN.move = old_move

    return undo
