"""Oracle-side executor of a dagnn graph (TEST INFRASTRUCTURE ONLY, see xm_oracle.c): walks the same
mcncrossmodalemotions_amd.dagnn.DagNN structure with the CPU oracle's operators, so end-to-end
HIP results can be compared against the restated MatConvNet semantics on identical graphs."""
import numpy as np

from mcncrossmodalemotions_amd import dagnn
from oracle import oracle as O


def host_params(net):
    out = {}
    for k, p in net.params.items():
        v = p.value
        if not isinstance(v, np.ndarray):
            from mcncrossmodalemotions_amd import vl
            v = vl.to_numpy(v)
        out[k] = np.asfortranarray(v, dtype=np.float32)
    return out


def forward(net, inputs, params=None, acc64=True, mode=None):
    """returns dict of all variable values (numpy, MATLAB layout)."""
    P = params or host_params(net)
    mode = mode or net.mode
    V = dict(inputs)
    aux = {}
    for l in net.layers:
        b = l.block
        ins = [V.get(v) for v in l.inputs]
        prm = [P[p] for p in l.params]
        if isinstance(b, dagnn.Conv):
            y = O.vl_nnconv(ins[0], prm[0], prm[1] if b.hasBias else None, stride=b.stride, pad=b.pad,
                            dilate=b.dilate, acc64=acc64)
        elif isinstance(b, dagnn.BatchNorm):
            y, mom = O.vl_nnbnorm(ins[0], prm[0], prm[1], epsilon=b.epsilon,
                                  moments=prm[2] if mode == "test" else None, acc64=acc64)
            aux[l.name] = mom
        elif isinstance(b, dagnn.ReLU):
            y = O.vl_nnrelu(ins[0], leak=b.leak)
        elif isinstance(b, dagnn.Sigmoid):
            y = O.vl_nnsigmoid(ins[0])
        elif isinstance(b, dagnn.GlobalPooling):
            y = O.vl_nnpool(ins[0], ins[0].shape[:2], method=b.method)
        elif isinstance(b, dagnn.Pooling):
            y = O.vl_nnpool(ins[0], b.poolSize, stride=b.stride, pad=b.pad, method=b.method)
        elif isinstance(b, dagnn.Sum):
            y = O.sum2(ins[0], ins[1])
        elif isinstance(b, dagnn.Axpy):
            y = O.scale_axpy(ins[1], ins[0], ins[2])
        elif isinstance(b, dagnn.Scale):
            y = O.scale_axpy(ins[0], ins[1])
        elif isinstance(b, dagnn.SoftMax):
            y = O.vl_nnsoftmaxt(ins[0], 1.0)
        elif isinstance(b, dagnn.SoftmaxCELoss):
            if ins[0] is None or ins[1] is None:
                continue
            y = np.float32(O.vl_nnsoftmaxceloss(ins[0], ins[1], temperature=b.temperature,
                                                logit_targets=b.logitTargets))
        elif isinstance(b, (dagnn.EuclideanLoss, dagnn.HuberLoss)):
            if ins[0] is None or ins[1] is None:
                continue
            kind = "euclidean" if isinstance(b, dagnn.EuclideanLoss) else "huber"
            y = np.float32(O.vl_nnregloss(ins[0], ins[1], kind=kind, sigma=getattr(b, "sigma", 1.0),
                                          instance_weights=ins[2] if len(ins) > 2 else None))
        elif isinstance(b, (dagnn.Loss, dagnn.ErrorStats)):
            if ins[0] is None or ins[1] is None:
                continue
            loss = b.loss if isinstance(b, dagnn.Loss) else "classerror"
            y = np.float32(O.vl_nnloss(ins[0], ins[1], loss=loss))
        elif isinstance(b, dagnn.DropOut):
            # vl_nndropout(X, 'mask', M): the mask is an input of the pass ('<layer>.mask'); test mode / no mask = identity
            m = None if mode == "test" else V.get(l.name + ".mask")
            y = ins[0] if m is None else O.vl_nndropout(ins[0], m)
        else:
            raise NotImplementedError(type(b))
        V[l.outputs[0]] = y
    V["__aux__"] = aux
    return V


def backward(net, V, der_outputs, params=None, acc64=True, mode=None):
    """returns (dict var -> der, dict param -> der)."""
    P = params or host_params(net)
    mode = mode or net.mode
    D = dict(der_outputs)
    DP = {}

    def add(name, d):
        if d is None:
            return
        D[name] = d if name not in D else D[name] + d

    for l in reversed(net.layers):
        b = l.block
        dz = D.get(l.outputs[0])
        if dz is None:
            continue
        ins = [V.get(v) for v in l.inputs]
        prm = [P[p] for p in l.params]
        if isinstance(b, dagnn.Conv):
            dx, df, db = O.vl_nnconv(ins[0], prm[0], prm[1] if b.hasBias else None, dz, stride=b.stride,
                                     pad=b.pad, dilate=b.dilate, acc64=acc64)
            add(l.inputs[0], dx)
            DP[l.params[0]] = df
            if b.hasBias:
                DP[l.params[1]] = db
        elif isinstance(b, dagnn.BatchNorm):
            dx, dg, db, mom = O.vl_nnbnorm(ins[0], prm[0], prm[1], dz, epsilon=b.epsilon,
                                           moments=prm[2] if mode == "test" else None, acc64=acc64)
            add(l.inputs[0], dx)
            DP[l.params[0]], DP[l.params[1]], DP[l.params[2]] = dg, db, mom
        elif isinstance(b, dagnn.ReLU):
            add(l.inputs[0], O.vl_nnrelu(ins[0], dz, leak=b.leak))
        elif isinstance(b, dagnn.Sigmoid):
            add(l.inputs[0], O.vl_nnsigmoid(ins[0], dz))
        elif isinstance(b, dagnn.GlobalPooling):
            add(l.inputs[0], O.vl_nnpool(ins[0], ins[0].shape[:2], dz, method=b.method))
        elif isinstance(b, dagnn.Pooling):
            add(l.inputs[0], O.vl_nnpool(ins[0], b.poolSize, dz, stride=b.stride, pad=b.pad, method=b.method))
        elif isinstance(b, dagnn.Sum):
            add(l.inputs[0], dz)
            add(l.inputs[1], dz)
        elif isinstance(b, dagnn.Axpy):
            dx, da = O.scale_backward(ins[1], ins[0], dz)
            add(l.inputs[0], da)
            add(l.inputs[1], dx)
            add(l.inputs[2], dz)
        elif isinstance(b, dagnn.SoftmaxCELoss):
            add(l.inputs[0], O.vl_nnsoftmaxceloss(ins[0], ins[1], np.asarray(dz, np.float32).ravel(),
                                                  temperature=b.temperature, logit_targets=b.logitTargets))
        elif isinstance(b, (dagnn.EuclideanLoss, dagnn.HuberLoss)):
            kind = "euclidean" if isinstance(b, dagnn.EuclideanLoss) else "huber"
            add(l.inputs[0], O.vl_nnregloss(ins[0], ins[1], np.asarray(dz, np.float32).ravel(), kind=kind,
                                            sigma=getattr(b, "sigma", 1.0),
                                            instance_weights=ins[2] if len(ins) > 2 else None))
        elif isinstance(b, dagnn.Loss):
            add(l.inputs[0], O.vl_nnloss(ins[0], ins[1], np.asarray(dz, np.float32).ravel(), loss=b.loss))
        elif isinstance(b, dagnn.SoftMax):
            add(l.inputs[0], O.vl_nnsoftmaxt_backward(ins[0], dz, 1.0))
        elif isinstance(b, dagnn.DropOut):
            m = None if mode == "test" else V.get(l.name + ".mask")
            add(l.inputs[0], dz if m is None else O.vl_nndropout(dz, m))
        else:
            raise NotImplementedError(type(b))
    return D, DP
