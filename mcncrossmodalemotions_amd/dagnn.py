"""Host-side mirror of MatConvNet's dagnn.DagNN / dagnn.Layer for the reference's hot path.

The reference never touches an operator directly: it builds / loads a `dagnn.DagNN`
(emoVoxCeleb/emoVoxZoo.m:44,199; teacher/ferPlusZoo.m:100,145), edits it with
addLayer / removeLayer / renameVar / initParams, and calls `dag.eval(inputs)` or hands it to
cnn_train_dag, which calls `net.eval(inputs, derOutputs)`
(emoVoxCeleb/fetch_emovoxceleb_imdb.m:129, external/compute_audio_feats.m:126,
emoVoxCeleb/run_distillation.m:170-182).  This module keeps those names and semantics; each
block's forward/backward is a thin wrapper over the HIP operators in vl.py, exactly as
dagnn.Conv.forward wraps vl_nnconv.

MI355X-specific additions (all optional, results identical):
  * peephole fusion at eval time: BatchNorm -> ReLU pairs run as one kernel (fwd and bwd);
    in test mode Conv -> BatchNorm [-> Sum] [-> ReLU] chains fold into the conv epilogue;
  * pack_params(): all parameters / derivatives / momenta live in three flat HBM buffers
    (grouped by learning-rate / weight-decay multipliers) so the optimiser is a handful of
    launches and the ParameterServer all-reduce is one RCCL call per bucket.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import vl


class Param:
    def __init__(self, name, value=None, learningRate=1.0, weightDecay=1.0, trainMethod="gradient"):
        self.name = name
        self.value = value
        self.der = None
        self.learningRate = learningRate
        self.weightDecay = weightDecay
        self.trainMethod = trainMethod
        self.fanout = 0


class Var:
    def __init__(self, name):
        self.name = name
        self.value = None
        self.der = None
        self.precious = False
        self.fanin = 0
        self.fanout = 0


# ---------------------------------------------------------------------------------------------
# blocks (dagnn.Layer subclasses)
# ---------------------------------------------------------------------------------------------
class Layer:
    """dagnn.Layer: forward(inputs, params) -> outputs; backward(...) -> derInputs, derParams."""

    def __init__(self):
        self.net = None
        self.layerIndex = -1

    def forward(self, inputs, params):
        raise NotImplementedError

    def backward(self, inputs, params, derOutputs):
        raise NotImplementedError

    def getParams(self):
        return []

    def initParams(self, rng):
        return []


class Conv(Layer):
    def __init__(self, size, hasBias=True, stride=(1, 1), pad=(0, 0, 0, 0), dilate=(1, 1)):
        super().__init__()
        self.size = tuple(int(s) for s in size)  # [FH FW FC K]
        self.hasBias = hasBias
        self.stride = tuple(stride) if not np.isscalar(stride) else (stride, stride)
        self.pad = pad
        self.dilate = tuple(dilate) if not np.isscalar(dilate) else (dilate, dilate)

    def forward(self, inputs, params):
        b = params[1] if self.hasBias else None
        return [vl.vl_nnconv(inputs[0], params[0], b, stride=self.stride, pad=self.pad,
                             dilate=self.dilate)]

    def backward(self, inputs, params, derOutputs, need_dx=True, der_out=None, skip_db=False,
                 need_df=True, dx_accum=None):
        b = params[1] if self.hasBias else None
        dfo = der_out[0] if der_out else None
        dbo = der_out[1] if (der_out and self.hasBias) else None
        dx, df, db = vl.vl_nnconv(inputs[0], params[0], b, derOutputs[0], stride=self.stride,
                                  pad=self.pad, dilate=self.dilate, no_der_data=not need_dx,
                                  no_der_filters=not need_df, no_der_biases=skip_db or not need_df,
                                  df_out=dfo if need_df else None, db_out=dbo if need_df else None,
                                  dx_accum=dx_accum)
        return [dx], ([df, db] if self.hasBias else [df])

    def initParams(self, rng):
        # dagnn.Conv.initParams [EXT]: "Xavier improved", FAN-IN: sc = sqrt(2 / prod(size(1:3)));
        # filters randn * sc, biases 0 (the fan-out variant is the commented-out line upstream)
        FH, FW, FC, K = self.size
        sc = np.sqrt(2.0 / (FH * FW * FC))
        p = [np.asfortranarray(rng.standard_normal(self.size).astype(np.float32) * np.float32(sc))]
        if self.hasBias:
            p.append(np.zeros((K, 1), np.float32))
        return p


class BatchNorm(Layer):
    def __init__(self, numChannels, epsilon=1e-4):
        super().__init__()
        self.numChannels = int(numChannels)
        self.epsilon = float(epsilon)
        self.moments = None  # last batch moments (train mode), the 'der' of the moments param
        self._pre_moments = None  # batch moments of the input, left by the producing convolution's epilogue

    def take_pre_moments(self):
        """batch moments the producing Conv step already computed for this eval (vl_nnconv moments_out), or None"""
        pre, self._pre_moments = self._pre_moments, None
        return pre

    def forward(self, inputs, params, relu=False):
        test = self.net is not None and self.net.mode == "test"
        pre = None if test else self.take_pre_moments()
        y, mom = vl.vl_nnbnorm(inputs[0], params[0], params[1], epsilon=self.epsilon,
                               moments=params[2] if test else pre, relu=relu, moments_out=pre)
        self.moments = None if test else mom
        return [y]

    def backward(self, inputs, params, derOutputs, relu=False, y=None, der_out=None, dxsum_out=None):
        test = self.net is not None and self.net.mode == "test"
        do = der_out or [None, None, None]
        # train mode: the forward pass of this eval left the batch moments of this very input in
        # self.moments -- hand them back instead of re-reading X to recompute them
        saved = None if test else self.moments
        dx, dg, db, mom = vl.vl_nnbnorm(inputs[0], params[0], params[1], derOutputs[0],
                                        epsilon=self.epsilon,
                                        moments=params[2] if test else saved,
                                        relu=relu, y=y, dg_out=do[0], db_out=do[1], moments_out=do[2],
                                        batch_moments=(not test) and saved is not None, dxsum_out=dxsum_out)
        # dagnn.BatchNorm: derParams{3} = the batch moments (consumed by trainMethod 'average')
        return [dx], [dg, db, mom]

    def initParams(self, rng):
        C = self.numChannels
        mom = np.zeros((C, 2), np.float32, order="F")
        mom[:, 1] = 1.0
        return [np.ones((C, 1), np.float32), np.zeros((C, 1), np.float32), mom]


class ReLU(Layer):
    def __init__(self, leak=0.0):
        super().__init__()
        self.leak = leak

    def forward(self, inputs, params):
        return [vl.vl_nnrelu(inputs[0], leak=self.leak)]

    def backward(self, inputs, params, derOutputs):
        return [vl.vl_nnrelu(inputs[0], derOutputs[0], leak=self.leak)], []


class Sigmoid(Layer):
    def forward(self, inputs, params):
        return [vl.vl_nnsigmoid(inputs[0])]

    def backward(self, inputs, params, derOutputs):
        return [vl.vl_nnsigmoid(inputs[0], derOutputs[0])], []


class Pooling(Layer):
    def __init__(self, poolSize, stride=(1, 1), pad=(0, 0, 0, 0), method="max"):
        super().__init__()
        self.poolSize = list(poolSize)
        self.stride = stride
        self.pad = pad
        self.method = method
        self._argmax = None  # routing table kept between forward and backward (training)

    def forward(self, inputs, params):
        training = self.net is not None and self.net.mode != "test" and self.method == "max"
        if training:
            y, self._argmax = vl.vl_nnpool(inputs[0], self.poolSize, stride=self.stride, pad=self.pad,
                                           method=self.method, want_argmax=True)
            return [y]
        self._argmax = None
        return [vl.vl_nnpool(inputs[0], self.poolSize, stride=self.stride, pad=self.pad,
                             method=self.method)]

    def backward(self, inputs, params, derOutputs, outputs=None):
        am, self._argmax = self._argmax, None
        return [vl.vl_nnpool(inputs[0], self.poolSize, derOutputs[0], stride=self.stride,
                             pad=self.pad, method=self.method, argmax=am)], []


class GlobalPooling(Pooling):
    """mcnExtraLayers dagnn.GlobalPooling (SE squeeze): window = the whole H x W plane."""

    def __init__(self, method="avg"):
        super().__init__([1, 1], method=method)

    def forward(self, inputs, params):
        self.poolSize = [int(inputs[0].shape[0]), int(inputs[0].shape[1])]
        return super().forward(inputs, params)

    def backward(self, inputs, params, derOutputs, outputs=None):
        self.poolSize = [int(inputs[0].shape[0]), int(inputs[0].shape[1])]
        return super().backward(inputs, params, derOutputs, outputs)


class Sum(Layer):
    def forward(self, inputs, params, relu=False):
        y = vl.sum2(inputs[0], inputs[1], relu=relu and len(inputs) == 2)
        for extra in inputs[2:]:
            y = vl.sum2(y, extra)
        return [y]

    def backward(self, inputs, params, derOutputs):
        return [derOutputs[0] for _ in inputs], []


class Scale(Layer):
    """mcnExtraLayers dagnn.Scale as used by SE blocks: inputs {x, a}; y = a(1,1,c,n) .* x."""

    def forward(self, inputs, params):
        return [vl.scale_axpy(inputs[0], inputs[1])]

    def backward(self, inputs, params, derOutputs):
        dx, da = vl.scale_backward(inputs[0], inputs[1], derOutputs[0])
        return [dx, da], []


class Axpy(Layer):
    """mcnExtraLayers dagnn.Axpy (SENet50 Caffe import): inputs {a, x, y}; out = a .* x + y."""

    def forward(self, inputs, params, relu=False):
        return [vl.scale_axpy(inputs[1], inputs[0], inputs[2], relu=relu)]

    def backward(self, inputs, params, derOutputs):
        dx, da = vl.scale_backward(inputs[1], inputs[0], derOutputs[0])
        return [da, dx, derOutputs[0]], []


class SoftMax(Layer):
    def forward(self, inputs, params):
        return [vl.vl_nnsoftmax(inputs[0])]

    def backward(self, inputs, params, derOutputs):
        return [vl.vl_nnsoftmax(inputs[0], derOutputs[0])], []


class DropOut(Layer):
    """dagnn.DropOut [EXT]: identity in test mode; in training mode Y = MASK .* X with a fresh mask per call
    (vl_nndropout), the backward pass multiplies by the same mask.  emoVoxZoo.m:116-135,272-277 inserts it behind fc6
    and fc7 when opts.dropout > 0.  Masks come from the library's stateless Philox stream: `seed` per layer, the
    counter offset advances by ceil(numel / 4) per call.  The offset is training state: train.save_checkpoint stores it and
    `cont` restores it, so a resumed run continues the mask sequence instead of repeating it; data-parallel workers fold
    their rank into the seed (DagNN.workerRank, set by train.process_epoch) so that the shards do not share masks."""

    def __init__(self, rate=0.5, seed=0):
        super().__init__()
        self.rate = rate
        self.seed = seed
        self.frozen = False
        self.mask = None
        self._offset = 0

    def forward(self, inputs, params):
        if self.net is None or self.net.mode == "test" or self.frozen or self.rate <= 0:
            self.mask = None
            return [inputs[0]]
        seed = (self.seed + 0x9E3779B1 * int(getattr(self.net, "workerRank", 0))) & 0x7FFFFFFF
        y, self.mask = vl.vl_nndropout(inputs[0], rate=self.rate, seed=seed, offset=self._offset)
        self._offset += (int(inputs[0].numel()) + 3) // 4
        return [y]

    def backward(self, inputs, params, derOutputs):
        if self.mask is None:
            return [derOutputs[0]], []
        return [vl.vl_nndropout(inputs[0], derOutputs[0], mask=self.mask)], []


class LossBase(Layer):
    """dagnn.Loss bookkeeping [EXT]: `average` = running per-sample mean of the (batch-summed) loss,
    `numAveraged` = samples seen since reset().  The running sum stays on the device (one tiny add
    per minibatch); reading `average` is the only host synchronisation."""

    ignoreAverage = False

    def __init__(self):
        super().__init__()
        self.reset()

    def reset(self):
        self._sum = None
        self._pending = []
        self.numAveraged = 0
        self.lastValue = None
        self.lastN = 0

    def _fold(self):
        if self._pending:
            t = torch.stack([v.reshape(-1)[0] for v in self._pending]).sum()
            self._sum = t if self._sum is None else self._sum + t
            self._pending = []

    def _accumulate(self, value, n):
        # the batch-summed loss values are kept as device scalars and folded 64 at a time: no extra
        # launch in the step itself
        self.lastValue = value
        self.lastN = n
        self._pending.append(value)
        if len(self._pending) >= 64:
            self._fold()
        self.numAveraged += n

    @property
    def average(self):
        self._fold()
        if self._sum is None or self.numAveraged == 0:
            return 0.0
        return float(self._sum.item()) / self.numAveraged


class SoftmaxCELoss(LossBase):
    """mcnExtraLayers dagnn.SoftmaxCELoss('temperature', T, 'logitTargets', tf)."""

    def __init__(self, temperature=1.0, logitTargets=False):
        super().__init__()
        self.temperature = temperature
        self.logitTargets = logitTargets

    def forward(self, inputs, params):
        w = inputs[2] if len(inputs) > 2 else None
        y = vl.vl_nnsoftmaxceloss(inputs[0], inputs[1], temperature=self.temperature,
                                  logitTargets=self.logitTargets, instanceWeights=w)
        self._accumulate(y, int(inputs[0].shape[3]) if inputs[0].dim() > 3 else 1)
        return [y]

    def backward(self, inputs, params, derOutputs):
        w = inputs[2] if len(inputs) > 2 else None
        dx = vl.vl_nnsoftmaxceloss(inputs[0], inputs[1], derOutputs[0],
                                   temperature=self.temperature, logitTargets=self.logitTargets,
                                   instanceWeights=w)
        return [dx] + [None] * (len(inputs) - 1), []


class EuclideanLoss(LossBase):
    """mcnExtraLayers dagnn.EuclideanLoss on {prediction, target, instanceWeights} (emoVoxZoo.m:139-140)."""

    def forward(self, inputs, params):
        w = inputs[2] if len(inputs) > 2 else None
        y = vl.vl_nneuclideanloss(inputs[0], inputs[1], instanceWeights=w)
        self._accumulate(y, int(inputs[0].shape[3]) if inputs[0].dim() > 3 else 1)
        return [y]

    def backward(self, inputs, params, derOutputs):
        w = inputs[2] if len(inputs) > 2 else None
        dx = vl.vl_nneuclideanloss(inputs[0], inputs[1], derOutputs[0], instanceWeights=w)
        return [dx] + [None] * (len(inputs) - 1), []


class HuberLoss(LossBase):
    """mcnExtraLayers dagnn.HuberLoss('sigma', s) (emoVoxZoo.m:146-147)."""

    def __init__(self, sigma=1.0):
        super().__init__()
        self.sigma = float(sigma)

    def forward(self, inputs, params):
        w = inputs[2] if len(inputs) > 2 else None
        y = vl.vl_nnhuberloss(inputs[0], inputs[1], sigma=self.sigma, instanceWeights=w)
        self._accumulate(y, int(inputs[0].shape[3]) if inputs[0].dim() > 3 else 1)
        return [y]

    def backward(self, inputs, params, derOutputs):
        w = inputs[2] if len(inputs) > 2 else None
        dx = vl.vl_nnhuberloss(inputs[0], inputs[1], derOutputs[0], sigma=self.sigma, instanceWeights=w)
        return [dx] + [None] * (len(inputs) - 1), []


class Loss(LossBase):
    """dagnn.Loss('loss', 'softmaxlog' | 'classerror')."""

    def __init__(self, loss="softmaxlog"):
        super().__init__()
        self.loss = loss

    def forward(self, inputs, params):
        y = vl.vl_nnloss(inputs[0], inputs[1], loss=self.loss)
        self._accumulate(y, int(inputs[0].shape[3]) if inputs[0].dim() > 3 else 1)
        return [y]

    def backward(self, inputs, params, derOutputs):
        return [vl.vl_nnloss(inputs[0], inputs[1], derOutputs[0], loss=self.loss), None], []


class VerboseLoss(Loss):
    """mcnExtraLayers dagnn.VerboseLoss: a dagnn.Loss that also prints; same arithmetic."""


class ErrorStats(LossBase):
    """mcnExtraLayers dagnn.ErrorStats('numClasses', C) on {prediction, maxLabel} (emoVoxZoo.m:165-169):
    per-class accuracy and class population, read back by extractStats (run_distillation.m:186-207)
    as `block.average` (C accuracies) and `block.classDist` (C counts).  The counters live on the
    device (xm_class_stats); a class without samples reports accuracy 0 [EXT]."""

    def __init__(self, numClasses=8):
        self.numClasses = int(numClasses)
        self._correct = None
        self._population = None
        super().__init__()

    def reset(self):
        super().reset()
        if self._correct is not None:
            self._correct.zero_()
            self._population.zero_()

    def forward(self, inputs, params):
        y = vl.vl_nnloss(inputs[0], inputs[1], loss="classerror")
        if self._correct is None:
            self._correct = vl.mat_zeros(self.numClasses, 1, device=inputs[0].device)
            self._population = vl.mat_zeros(self.numClasses, 1, device=inputs[0].device)
        vl.class_stats(inputs[0], inputs[1], self._correct, self._population)
        self.lastValue = y
        self.lastN = int(inputs[0].shape[3]) if inputs[0].dim() > 3 else 1
        self.numAveraged += self.lastN
        return [y]

    @property
    def classDist(self):
        if self._population is None:
            return np.zeros(self.numClasses)
        return vl.to_numpy(self._population).ravel().astype(np.float64)

    @property
    def average(self):
        if self._correct is None:
            return np.zeros(self.numClasses)
        c = vl.to_numpy(self._correct).ravel().astype(np.float64)
        return c / np.maximum(self.classDist, 1.0)

    def backward(self, inputs, params, derOutputs):
        return [None, None], []


# ---------------------------------------------------------------------------------------------
# the graph
# ---------------------------------------------------------------------------------------------
class _LayerRec:
    def __init__(self, name, block, inputs, outputs, params):
        self.name = name
        self.block = block
        self.inputs = list(inputs)
        self.outputs = list(outputs)
        self.params = list(params)


class DagNN:
    """dagnn.DagNN subset: addLayer, removeLayer, renameVar, getVarIndex, getParamIndex,
    getLayerIndex, getInputs, initParams, rebuild, move, eval, mode, vars, params, layers, meta."""

    def __init__(self):
        self.layers = []
        self.vars = OrderedDict()
        self.params = OrderedDict()
        self.meta = {}
        self.mode = "normal"
        self.conserveMemory = True
        self.accumulateParamDers = False
        self.fuse = True  # MI355X peephole fusion (results identical)
        self.fuseStats = os.environ.get("XM_NO_FUSED_STATS") is None  # bnorm batch moments from the conv epilogue
        self.wgradAfterDgrad = os.environ.get("XM_WGRAD_AFTER_DGRAD") is not None
        self.fuseBiasDer = os.environ.get("XM_NO_FUSED_BIASDER") is None   # conv dzdb = sum(dx) from the bnorm backward
        # first-layer Conv -> BatchNorm -> ReLU -> Pooling: the bnorm's DZDX is rebuilt inside the convolution's
        # filter-derivative kernel instead of being written and read back (vl.conv_backward_filter_bnrelupool)
        self.fuseStemBackward = os.environ.get("XM_NO_FUSED_STEM_BWD") is None
        # ... and without reading the convolution's output either (vl.conv_backward_filter_bnrelupool_gram, round 6)
        self.fuseStemGram = os.environ.get("XM_NO_STEM_GRAM") is None
        # ... and the forward pass of conv -> bnorm -> relu -> pool as one kernel (vl.conv_bnorm_relu_pool)
        self.fuseStemForward = os.environ.get("XM_NO_STEM_FWD") is None
        self.fuseForkSums = os.environ.get("XM_NO_FORK_SUMS") is None   # global-avg backward adds the fork's other derivative
        self.fuseSE = os.environ.get("XM_NO_FUSED_SE") is None   # test mode: SE squeeze from the projection's input, excite in its epilogue
        self.foldSEReduce = os.environ.get("XM_NO_SE_FOLD_FC") is None   # ... and the projection folded into the SE reduction layer
        # training plans: relu mask + excite + squeeze + bnorm backward of an SE block's tail in two fused calls
        self.fuseSETrain = os.environ.get("XM_NO_FUSED_SE_BWD") is None
        self.wgradStream = None  # optional side HIP stream for the filter / bias derivatives
        self.gradHook = None     # callable(layer name): called right after a conv layer's parameter
                                 # derivatives were enqueued, on the stream they were enqueued on
        self.markHook = None     # diagnostics: callable(label) at the phase boundaries of eval (bench.py XM_BENCH_MARKS)
        self.bwdHooks = {}       # {layer name: callable()} run right before that layer's backward step is enqueued
                                 # (a host can hang stream events there, e.g. to start a prefetch -- bench.py)
        self._side_pending = False
        self._training = False
        self.prepareBackward = os.environ.get("XM_NO_PREPARE") is None   # dgrad filter transposition during forward
        self.device = None
        self._flat = None
        # bumped whenever parameter VALUES change behind torch's back (raw-pointer HIP updates: xm_sgd_update /
        # xm_average_update on the flat buffers, checkpoint loads, pack_params): cached derived quantities
        # (folded test-mode bnorm scale / shift) are keyed on it.  Shared with replica()s (same parameters).
        self._gen = [0]

    @property
    def paramGeneration(self):
        return self._gen[0]

    @paramGeneration.setter
    def paramGeneration(self, v):
        self._gen[0] = int(v)

    # ---- construction -------------------------------------------------------------------
    def addLayer(self, name, block, inputs, outputs, params=()):
        if isinstance(inputs, str):
            inputs = [inputs]
        if isinstance(outputs, str):
            outputs = [outputs]
        if isinstance(params, str):
            params = [params]
        if any(l.name == name for l in self.layers):
            raise ValueError("There is already a layer with name '%s'." % name)
        block.net = self
        self.layers.append(_LayerRec(name, block, inputs, outputs, params))
        for p in params:
            if p not in self.params:
                self.params[p] = Param(p)
        self.rebuild()
        return self

    def insertLayerAfter(self, prev, name, block, inputs, outputs, params=()):
        """addLayer, placed right behind layer `prev` (the layer list is the execution order here; MatConvNet's
        dagnn re-derives the order from the graph, so net.addLayer + setLayerInputs suffice there)"""
        self.addLayer(name, block, inputs, outputs, params)
        rec = self.layers.pop()
        self.layers.insert(self.getLayerIndex(prev) + 1, rec)
        self.rebuild()
        return self

    def setLayerInputs(self, name, inputs):
        self.layers[self.getLayerIndex(name)].inputs = [inputs] if isinstance(inputs, str) else list(inputs)
        self.rebuild()
        return self

    def removeLayer(self, name):
        names = [name] if isinstance(name, str) else list(name)
        self.layers = [l for l in self.layers if l.name not in names]
        self.rebuild()

    def renameVar(self, old, new):
        for l in self.layers:
            l.inputs = [new if v == old else v for v in l.inputs]
            l.outputs = [new if v == old else v for v in l.outputs]
        self.rebuild()

    def rebuild(self):
        old = self.vars
        self.vars = OrderedDict()
        for p in self.params.values():
            p.fanout = 0
        for i, l in enumerate(self.layers):
            l.block.layerIndex = i
            l.block.net = self
            for v in l.inputs:
                self.vars.setdefault(v, old.get(v) or Var(v))
            for v in l.outputs:
                self.vars.setdefault(v, old.get(v) or Var(v))
        for v in self.vars.values():
            v.fanin = v.fanout = 0
        for l in self.layers:
            for v in l.inputs:
                self.vars[v].fanout += 1
            for v in l.outputs:
                self.vars[v].fanin += 1
            for p in l.params:
                self.params[p].fanout += 1
        used = {p for l in self.layers for p in l.params}
        for p in list(self.params):
            if p not in used:
                del self.params[p]

    def replica(self):
        """A second evaluation context over the SAME parameters: own variables, layer blocks and
        execution plan, shared Param records.  Replicas can evaluate different batches concurrently
        on different HIP streams (forward-only use; derivatives would collide in the shared Params)."""
        import copy
        r = DagNN()
        r.meta, r.mode, r.fuse, r.device = self.meta, self.mode, self.fuse, self.device
        r.conserveMemory = self.conserveMemory
        for l in self.layers:
            r.layers.append(_LayerRec(l.name, copy.copy(l.block), l.inputs, l.outputs, l.params))
        r.params = self.params
        r._gen = self._gen
        r.rebuild()
        for name, v in self.vars.items():
            r.vars[name].precious = v.precious
        return r

    def getLayerIndex(self, name):
        for i, l in enumerate(self.layers):
            if l.name == name:
                return i
        return None

    def getLayer(self, name):
        i = self.getLayerIndex(name)
        return None if i is None else self.layers[i]

    def getVarIndex(self, name):
        return list(self.vars).index(name) if name in self.vars else None

    def getParamIndex(self, name):
        return list(self.params).index(name) if name in self.params else None

    def getInputs(self):
        return [v.name for v in self.vars.values() if v.fanin == 0]

    def getOutputs(self):
        return [v.name for v in self.vars.values() if v.fanout == 0]

    def initParams(self, seed=0):
        """dag.initParams(): re-randomise every parameter (emoVoxZoo.m:54)."""
        rng = np.random.default_rng(seed)
        for l in self.layers:
            vals = l.block.initParams(rng)
            for pname, v in zip(l.params, vals):
                self.params[pname].value = v  # host array until move('gpu')
                if isinstance(l.block, BatchNorm):
                    k = l.params.index(pname)
                    p = self.params[pname]
                    # [EXT] MatConvNet convention for BN params: g lr 2, b lr 1, moments 'average' 0.1
                    p.weightDecay = 0.0
                    p.learningRate = (2.0, 1.0, 0.1)[k]
                    p.trainMethod = "average" if k == 2 else "gradient"
                elif isinstance(l.block, Conv) and l.params.index(pname) == 1:
                    self.params[pname].weightDecay = 0.0
                    self.params[pname].learningRate = 2.0
        self._flat = None
        self.paramGeneration += 1

    def move(self, device="gpu"):
        """dag.move('gpu'): upload parameters (fetch_emovoxceleb_imdb.m:108)."""
        if device != "gpu":
            raise RuntimeError("this build has no CPU path (dag.move('cpu'))")
        self.device = torch.device("cuda", torch.cuda.current_device())
        for p in self.params.values():
            if p.value is not None and not isinstance(p.value, torch.Tensor):
                p.value = vl.from_numpy(p.value, self.device)
        return self

    # ---- flat parameter storage -----------------------------------------------------------
    def pack_params(self):
        """Re-home all parameters into flat value / der / momentum buffers grouped by
        (trainMethod, lr multiplier, wd multiplier).  Returns the FlatParams record."""
        self.move("gpu")
        groups = OrderedDict()
        for p in self.params.values():
            key = (p.trainMethod, float(p.learningRate), float(p.weightDecay))
            groups.setdefault(key, []).append(p)
        total = sum(int(p.value.numel() + 3) // 4 * 4 for ps in groups.values() for p in ps)
        val = torch.zeros(total, dtype=torch.float32, device=self.device)
        der = torch.zeros(total, dtype=torch.float32, device=self.device)
        mom = torch.zeros(total, dtype=torch.float32, device=self.device)
        off = 0
        segs = []
        for key, ps in groups.items():
            start = off
            for p in ps:
                n = int(p.value.numel())
                shp = tuple(int(s) for s in p.value.shape)
                view = val[off:off + n].view(tuple(reversed(shp))).permute(*reversed(range(len(shp))))
                view.copy_(p.value)
                p.value = view
                p.der = der[off:off + n].view(tuple(reversed(shp))).permute(*reversed(range(len(shp))))
                p._flat_off = off
                off += (n + 3) // 4 * 4
            segs.append((key, start, off))
        self._flat = FlatParams(val, der, mom, segs)
        self.paramGeneration += 1
        return self._flat

    # ---- evaluation -----------------------------------------------------------------------
    def eval(self, inputs, derOutputs=None, input_events=None):
        """net.eval(inputs) / net.eval(inputs, derOutputs).

        inputs:     ['name', tensor, ...] (the cell array of getBatch) or a dict
        derOutputs: ['objective', 1] or a dict; None -> forward only.
        input_events (extension): {input name: torch.cuda.Event} -- the current stream waits for the
        event right before the first layer that consumes that input (lets a producer running on
        another stream, e.g. the frozen teacher, overlap with the layers that do not need it)."""
        if not isinstance(inputs, dict):
            inputs = {inputs[i]: inputs[i + 1] for i in range(0, len(inputs), 2)}
        if derOutputs is not None and not isinstance(derOutputs, dict):
            derOutputs = {derOutputs[i]: derOutputs[i + 1] for i in range(0, len(derOutputs), 2)}
        for v in self.vars.values():
            if not v.precious:
                v.value = None
            v.der = None
        for k, t in inputs.items():
            if k not in self.vars:
                continue  # MatConvNet ignores unused inputs with a warning
            self.vars[k].value = t
        plan = self._plan(derOutputs is not None)
        self._training = derOutputs is not None
        if self._training and self.wgradStream is not None:
            # the side stream reads the parameters (filter transposition for dgrad, vl.conv_prepare_backward) and
            # overwrites the transposed copies the previous step's dgrad read: both happened on THIS stream (the
            # previous step's xm_sgd_update / backward pass), so the side stream must not start before them
            self.wgradStream.wait_stream(torch.cuda.current_stream())
        pending = dict(input_events or {})
        mark = self.markHook or (lambda label: None)
        mark("fwd0")
        for step in plan:
            if pending:
                recs = [step.rec] + [getattr(step, n) for n in ("relu_rec", "pool_rec", "bn_rec", "sum_rec")
                                     if getattr(step, n, None) is not None]
                for rec in recs:
                    for v in rec.inputs:
                        ev = pending.pop(v, None)
                        if ev is not None:
                            torch.cuda.current_stream().wait_event(ev)
            step.forward(self)
        mark("fwd1")
        if derOutputs is None:
            return
        for k, d in derOutputs.items():
            if not isinstance(d, torch.Tensor):
                # cached device scalar: a fresh host->device copy here would block the host until the
                # queued forward pass has drained
                cache = self.__dict__.setdefault("_der_consts", {})
                key = (float(d), str(self.device))
                if key not in cache:
                    cache[key] = vl.from_numpy(np.array([[float(d)]], np.float32), self.device)
                d = cache[key]
            self.vars[k].der = d
        if not self.accumulateParamDers and self._flat is None:
            for p in self.params.values():
                p.der = None
        self._pending_param_ders = {}
        hooks = self.bwdHooks
        for step in reversed(plan):
            if hooks:
                h = hooks.get(step.rec.name)
                if h is not None:
                    h()
            step.backward(self)
        mark("bwd1")
        if self._side_pending:
            torch.cuda.current_stream().wait_stream(self.wgradStream)
            self._side_pending = False
        mark("join")

    # helpers used by the plan steps
    def _set_var_der(self, name, d):
        if d is None:
            return
        v = self.vars[name]
        if v.der is None:
            v.der = d
        else:
            v.der = vl.sum2(v.der, d)  # fan-out > 1: derivatives add (dagnn accumulates)

    def _direct_der(self, rec):
        """flat storage: let the kernels write parameter derivatives straight into the flat
        gradient buffer (only when no parameter of the layer is shared or accumulated)."""
        if self._flat is None or self.accumulateParamDers:
            return None
        ps = [self.params[p] for p in rec.params]
        if any(p.fanout > 1 or p.der is None for p in ps):
            return None
        return [p.der for p in ps]

    def _set_param_der(self, name, d):
        if d is None:
            return
        p = self.params[name]
        if self._flat is not None and d is p.der:
            return  # already written in place by the kernel
        seen = self._pending_param_ders.get(name, 0)
        self._pending_param_ders[name] = seen + 1
        if self._flat is not None:
            if seen == 0 and not self.accumulateParamDers:
                p.der.copy_(d)
            else:
                p.der.add_(d)  # shared parameter (fanout > 1) -- not on the built path
        else:
            p.der = d if (p.der is None or seen == 0 and not self.accumulateParamDers) else vl.sum2(p.der, d)

    def _plan(self, training):
        # the set of precious variables is part of the key: a fused step never materialises the intermediate
        # variables it swallows, so marking one precious after a plan was built must rebuild the plan
        key = (training, self.mode, self.fuse, len(self.layers),
               tuple(n for n, v in self.vars.items() if v.precious))
        if getattr(self, "_plan_key", None) == key and getattr(self, "_plan_layers", None) == [
                id(l) for l in self.layers]:
            return self._plan_cache
        steps = build_plan(self, training)
        self._plan_key, self._plan_cache = key, steps
        self._plan_layers = [id(l) for l in self.layers]
        return steps


class FlatParams:
    def __init__(self, val, der, mom, segments):
        self.val, self.der, self.mom, self.segments = val, der, mom, segments


# ---------------------------------------------------------------------------------------------
# execution plan with peephole fusion
# ---------------------------------------------------------------------------------------------
class _Step:
    """plain step: one dagnn layer."""

    def __init__(self, rec):
        self.rec = rec
        self.bias_from = None  # fused consumer step that already produced this conv's bias derivative
        self.moments_for = None  # _LayerRec of the train-mode BatchNorm this conv step computes the batch moments for
        self.bias_conv = None    # bnorm steps: _Step of the biased Conv that produced the input (its dzdb = sum of our dx)
        self.bias_conv_done = False
        self.se_bn = None        # training plans: the _SEBnTrainStep whose fused backward covers this step (its squeeze / excite)
        self.stem_into = None    # training plans: the _BnReluPoolStep that can run this (first-layer) convolution inside its own kernel

    def _bias_der_slot(self, net):
        """flat derivative slot of the producing convolution's bias when this bnorm step may fill it (sum of dx)"""
        self.bias_conv_done = False
        if self.bias_conv is None or net._flat is None or net.accumulateParamDers or not net.fuseBiasDer:
            return None
        if net.wgradStream is not None and type(self) is not _BnReluPoolStep:
            # measured (default step, three A/B pairs): with a side stream the convolution's own dzdb pass runs there,
            # next to the main stream's dgrad, for free; pulling it into the bnorm backward lengthens the main stream
            # (the critical path) and cost 1 % -- although it removes 20 us of kernel time per layer.  One-stream hosts
            # (the MEX binding) take the shorter chain.
            return None
        bp = net.params[self.bias_conv.rec.params[1]]
        if bp.fanout == 1 and bp.der is not None:
            self.bias_conv_done = True
            return bp.der
        return None

    def _params(self, net):
        return [net.params[p].value for p in self.rec.params]

    def forward(self, net):
        r = self.rec
        if self.se_bn is not None and self.se_bn.fwd is not None and isinstance(r.block, GlobalPooling):
            f = self.se_bn.fwd      # the squeeze of a fused SE tail: straight from the bnorm's input
            net.vars[r.outputs[0]].value = vl.se_squeeze_bn(f["u"], f["g"], f["b"], f["moments"])
            return
        if self.stem_into is not None and self.stem_into.stem_forward(net, self):
            return      # conv -> bnorm -> relu -> pool ran as one kernel; this convolution's output is never written
        ins = [net.vars[v].value for v in r.inputs]
        if net._training and net.wgradStream is not None and isinstance(r.block, Conv) and \
                net.vars[r.inputs[0]].fanin > 0 and net.prepareBackward:
            # the backward pass of this layer will need the transposed filters: build them now on the idle side
            # stream instead of on the critical path of the backward pass (vl.conv_prepare_backward)
            with torch.cuda.stream(net.wgradStream):
                vl.conv_prepare_backward(ins[0], self._params(net)[0], stride=r.block.stride, pad=r.block.pad,
                                         dilate=r.block.dilate)
        bn = self.moments_for
        if bn is not None and net._training and net.mode != "test" and net.fuseStats:
            # the train-mode bnorm that consumes this convolution needs the batch moments of its output: the GEMM
            # epilogue leaves them (straight in the flat derivative slot of the moments parameter when there is one)
            do = net._direct_der(bn)
            mo = do[2] if do else vl.mat_empty(bn.block.numChannels, 2, device=ins[0].device)
            prm = self._params(net)
            blk = r.block
            outs = [vl.vl_nnconv(ins[0], prm[0], prm[1] if blk.hasBias else None, stride=blk.stride, pad=blk.pad,
                                 dilate=blk.dilate, moments_out=mo, epsilon=bn.block.epsilon)]
            bn.block._pre_moments = mo
        else:
            outs = r.block.forward(ins, self._params(net))
        for v, t in zip(r.outputs, outs):
            net.vars[v].value = t

    def backward(self, net):
        r = self.rec
        douts = [net.vars[v].der for v in r.outputs]
        if all(d is None for d in douts):
            return
        ins = [net.vars[v].value for v in r.inputs]
        if isinstance(r.block, Conv):
            need_dx = net.vars[r.inputs[0]].fanin > 0  # network inputs need no derivative
            skip_db = self.bias_from is not None and self.bias_from.bias_conv_done
            side = net.wgradStream if need_dx else None
            if side is not None and net._flat is not None and net._direct_der(r) is None:
                side = None  # the derivative would need a copy on this stream
            # fork: another consumer of the input already left its derivative -> add it in the dgrad
            # epilogue (dx = dgrad + existing) instead of a separate sum2 pass
            xin = net.vars[r.inputs[0]]
            accum = xin.der if need_dx else None
            if accum is not None:
                xin.der = None   # replaced by the accumulated result below
            if side is None:
                dins, dpar = r.block.backward(ins, self._params(net), douts, need_dx=need_dx,
                                              der_out=net._direct_der(r), skip_db=skip_db, dx_accum=accum)
                if net.gradHook is not None:
                    if net._side_pending:
                        # the hook may push a bucket that also covers layers whose filter derivatives were
                        # enqueued on the side stream: the push is ordered against THIS stream only
                        torch.cuda.current_stream().wait_stream(net.wgradStream)
                    net.gradHook(r.name)
            else:
                # dzdw / dzdb are off the critical path of the backward pass: they run on the side
                # stream next to the (HBM-bound) bnorm / pooling derivatives of the layers below
                main = torch.cuda.current_stream()
                defer = net.wgradAfterDgrad
                if defer:
                    # the dgrad of this layer first, alone on the chip; its wgrad starts when it has finished and
                    # then runs next to the HBM-bound bnorm / pooling derivatives of the layers below (two
                    # MFMA-bound kernels side by side only slow each other down)
                    dins, _ = r.block.backward(ins, self._params(net), douts, need_dx=True, need_df=False,
                                               dx_accum=accum)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    _, dpar = r.block.backward(ins, self._params(net), douts, need_dx=False,
                                               der_out=net._direct_der(r), skip_db=skip_db)
                    if net.gradHook is not None:
                        net.gradHook(r.name)
                douts[0].record_stream(side)
                ins[0].record_stream(side)
                net._side_pending = True
                if not defer:
                    dins, _ = r.block.backward(ins, self._params(net), douts, need_dx=True, need_df=False,
                                               dx_accum=accum)
            if skip_db:
                dpar = [dpar[0], net.params[r.params[1]].der]
        elif isinstance(r.block, BatchNorm):
            dins, dpar = r.block.backward(ins, self._params(net), douts, der_out=net._direct_der(r),
                                          dxsum_out=self._bias_der_slot(net))
        elif isinstance(r.block, GlobalPooling) and self.se_bn is not None and self.se_bn.stash is not None:
            # the squeeze of a fused SE tail: its derivative is a per-plane constant that the bnorm's fused backward adds
            self.se_bn.stash["dgp"] = douts[0]
            if net.conserveMemory:
                for v in r.outputs:
                    if not net.vars[v].precious:
                        net.vars[v].der = None
            return
        elif isinstance(r.block, GlobalPooling) and r.block.method == "avg" and net.vars[r.inputs[0]].der is not None \
                and net.fuseForkSums:
            # fork: the other consumer of X (the SE excite) already left its derivative -> one pass instead of a
            # broadcast pass + a sum pass over the widest tensors of the block
            xin = net.vars[r.inputs[0]]
            acc, xin.der = xin.der, None
            dins, dpar = [vl.vl_nnpool(ins[0], [int(ins[0].shape[0]), int(ins[0].shape[1])], douts[0], method="avg",
                                       dx_accum=acc)], []
        elif isinstance(r.block, Pooling):
            outs = [net.vars[v].value for v in r.outputs]
            dins, dpar = r.block.backward(ins, self._params(net), douts,
                                          outputs=outs if outs[0] is not None else None)
        else:
            dins, dpar = r.block.backward(ins, self._params(net), douts)
        for v, d in zip(r.inputs, dins):
            net._set_var_der(v, d)
        for p, d in zip(r.params, dpar):
            net._set_param_der(p, d)
        if net.conserveMemory:
            for v in r.outputs:
                if not net.vars[v].precious:
                    net.vars[v].der = None


class _BnReluStep(_Step):
    """BatchNorm -> ReLU executed as one kernel each way (train or test mode)."""

    def __init__(self, bn_rec, relu_rec):
        super().__init__(bn_rec)
        self.relu_rec = relu_rec

    def forward(self, net):
        r = self.rec
        ins = [net.vars[v].value for v in r.inputs]
        y = r.block.forward(ins, self._params(net), relu=True)[0]
        net.vars[self.relu_rec.outputs[0]].value = y

    def backward(self, net):
        r = self.rec
        out = net.vars[self.relu_rec.outputs[0]]
        if out.der is None:
            return
        ins = [net.vars[v].value for v in r.inputs]
        dins, dpar = r.block.backward(ins, self._params(net), [out.der], relu=True, y=out.value,
                                      der_out=net._direct_der(r), dxsum_out=self._bias_der_slot(net))
        net._set_var_der(r.inputs[0], dins[0])
        for p, d in zip(r.params, dpar):
            net._set_param_der(p, d)
        if net.conserveMemory and not out.precious:
            out.der = None


class _AddReluStep(_Step):
    """dagnn.Sum / dagnn.Axpy followed by ReLU: the rectification rides in the producing kernel
    (y = relu(a + b) or relu(s .* x + r)); backward masks dzdy by (y > 0) first."""

    def __init__(self, rec, relu_rec):
        super().__init__(rec)
        self.relu_rec = relu_rec

    def forward(self, net):
        r = self.rec
        ins = [net.vars[v].value for v in r.inputs]
        if self.se_bn is not None and self.se_bn.fwd is not None:
            f = self.se_bn.fwd      # the excite of a fused SE tail: a .* bnorm(u) + shortcut, relu
            y = vl.scale_axpy_bn(f["u"], ins[0], ins[2], f["g"], f["b"], f["moments"], relu=True)
        else:
            y = r.block.forward(ins, [], relu=True)[0]
        net.vars[self.relu_rec.outputs[0]].value = y

    def backward(self, net):
        r = self.rec
        out = net.vars[self.relu_rec.outputs[0]]
        if out.der is None:
            return
        if self.se_bn is not None and self.se_bn.begin(net, self, out):
            return      # the fused SE-tail backward: da is out, everything else follows at the bnorm's position
        dz = vl.vl_nnrelu(out.value, out.der)  # y = relu(.) > 0  <=>  pre-activation > 0
        ins = [net.vars[v].value for v in r.inputs]
        dins, _ = r.block.backward(ins, [], [dz])
        for v, d in zip(r.inputs, dins):
            net._set_var_der(v, d)
        if net.conserveMemory and not out.precious:
            out.der = None


class _SEBnTrainStep(_Step):
    """Training plans: the plain BatchNorm whose output X feeds exactly the squeeze (GlobalPooling 'avg') and the
    excite (Axpy(gate, X, shortcut) -> ReLU) of an SE block.  Forward: unchanged.  Backward: two fused calls
    (vl.se_tail_backward_reduce at the Axpy's position -> the gate's derivative; vl.se_tail_backward_apply here, once
    the gate's backward has produced the squeeze's derivative) instead of relu mask, scale backward, squeeze backward
    and bnorm backward over the block's widest tensors."""

    def __init__(self, bn_rec, axpy_step, gp_step):
        super().__init__(bn_rec)
        self.axpy_step, self.gp_step = axpy_step, gp_step
        self.stash = None
        self.fwd = None
        axpy_step.se_bn = self
        gp_step.se_bn = self

    def forward(self, net):
        """X has exactly two readers, the squeeze and the excite, and the fused backward never reads it: both recompute
        it from U (vl.se_squeeze_bn, vl.scale_axpy_bn -- the bnorm's own per-element expression), so the bnorm's apply
        pass is skipped and X is never written.  Needs the batch moments the producing convolution's epilogue left."""
        r = self.rec
        self.fwd = None
        xv = net.vars[r.outputs[0]]
        if net.fuseSETrain and net._training and not xv.precious:
            g, b, mom = self._params(net)
            test = net.mode == "test"
            moments = mom if test else r.block.take_pre_moments()
            if moments is not None:
                r.block.moments = None if test else moments
                self.fwd = {"u": net.vars[r.inputs[0]].value, "g": g, "b": b, "moments": moments}
                xv.value = None
                return
        super().forward(net)

    def begin(self, net, axpy_step, out):
        """called by the Axpy+ReLU step's backward; False = run the separate operators"""
        self.stash = None
        # the fused backward writes no bias-derivative slot: a value left from an unfused pass (fuseSETrain toggled
        # between steps) must not make the producing Conv skip its own dzdb (round-4 advisor)
        self.bias_conv_done = False
        if not net.fuseSETrain:
            if self.fwd is not None:
                raise RuntimeError("fused SE tail: fuseSETrain was switched off between the forward and the backward pass")
            return False
        r, ax = self.rec, axpy_step.rec
        gate_v, x_v, sc_v = (net.vars[v] for v in ax.inputs)
        if x_v.precious or net.vars[r.outputs[0]] is not x_v:
            if self.fwd is not None:
                raise RuntimeError("fused SE tail: the forward pass skipped X but the fused backward cannot run (%s)" % r.name)
            return False
        g, b, mom = self._params(net)
        test = net.mode == "test"
        moments = mom if test else r.block.moments
        if moments is None:
            return False
        u = net.vars[r.inputs[0]].value
        da, sums = vl.se_tail_backward_reduce(out.value, out.der, u, g, b, moments)
        net._set_var_der(ax.inputs[0], da)
        self.stash = {"y": out.value, "dzdy": out.der, "u": u, "sums": sums, "gate": gate_v.value, "moments": moments,
                      "test": test, "shortcut": ax.inputs[2], "dgp": None}
        if net.conserveMemory and not out.precious:
            out.der = None
        return True

    def backward(self, net):
        st, self.stash = self.stash, None
        if st is None:
            return super().backward(net)
        r = self.rec
        if st["dgp"] is None:
            raise RuntimeError("fused SE tail: the squeeze's derivative never arrived (layer %s)" % r.name)
        g, b, mom = self._params(net)
        do = net._direct_der(r)
        dz, du, dg, db = vl.se_tail_backward_apply(st["y"], st["dzdy"], st["u"], st["gate"], st["dgp"], g, st["moments"],
                                                   st["sums"], train=not st["test"], dg_out=do[0] if do else None,
                                                   db_out=do[1] if do else None)
        net._set_var_der(st["shortcut"], dz)
        net._set_var_der(r.inputs[0], du)
        for p, d in zip(r.params, [dg, db, None if st["test"] else st["moments"]]):
            net._set_param_der(p, d)


class _BnReluPoolStep(_Step):
    """BatchNorm -> ReLU -> Pooling('max') as one fused operator pair (train or test mode): the
    normalised / rectified tensor is never written to HBM (vl.bnorm_relu_pool)."""

    def __init__(self, bn_rec, relu_rec, pool_rec, bias_conv=None):
        super().__init__(bn_rec)
        self.relu_rec, self.pool_rec = relu_rec, pool_rec
        self._saved = None
        self.bias_conv = bias_conv        # _Step of the biased Conv feeding this BN (or None)
        self.bias_conv_done = False
        self.producer_conv = None         # _Step of the Conv whose ONLY consumer is this BN (training plans, build_plan)
        self._stem_fused = True           # cleared when the library reports the shapes as not covered
        self._stem_fwd_ok = True          # ... the fused forward (conv + bnorm + relu + pool in one kernel)
        self._stem_fwd = None             # set by stem_forward: {"gram": G or None} -- the convolution's output does not exist
        if bias_conv is not None:
            bias_conv.bias_from = self

    def _stem_conditions(self, net):
        """what both fused directions need from the plan: the producing convolution is a first layer that feeds only this
        step, and every derivative has its own slot in the flat buffer"""
        cs = self.producer_conv
        if cs is None or net._flat is None or net.accumulateParamDers:
            return None
        cr, r = cs.rec, self.rec
        if net.vars[cr.inputs[0]].fanin > 0 or net.vars[cr.outputs[0]].precious:
            return None
        do, cdo = net._direct_der(r), net._direct_der(cr)
        if do is None or cdo is None or any(net.params[p].fanout != 1 for p in cr.params):
            return None
        return cs, do, cdo

    def stem_forward(self, net, cs):
        """Called by the producing convolution's step in its place (round 6): conv -> bnorm -> relu -> pool as ONE kernel
        from the convolution's input; the batch moments come from the Gram matrix of the input patches, which the backward
        call takes as well (vl.conv_bnorm_relu_pool).  False = not applicable, the convolution runs as usual."""
        self._stem_fwd = None
        if not (self._stem_fwd_ok and self._stem_fused and net.fuseStemForward and net.fuseStemGram and
                net.fuseStemBackward and net._training and net.fuseStats):
            return False
        ok = self._stem_conditions(net)
        if ok is None or ok[0] is not cs:
            return False
        _, do, _ = ok
        r, pb, blk = self.rec, self.pool_rec.block, cs.rec.block
        g, b, mom = self._params(net)
        test = net.mode == "test"
        cpar = cs._params(net)
        res = vl.conv_bnorm_relu_pool(net.vars[cs.rec.inputs[0]].value, cpar[0], cpar[1] if blk.hasBias else None, g, b,
                                      pb.poolSize, stride=blk.stride, pad=blk.pad, dilate=blk.dilate, pool_stride=pb.stride,
                                      pool_pad=pb.pad, epsilon=r.block.epsilon, moments=mom if test else None,
                                      moments_out=None if test else do[2])
        if res is None:
            self._stem_fwd_ok = False
            return False
        y, am, mo, gram = res
        r.block.moments = None if test else mo
        self._saved = (am, mo)
        self._stem_fwd = {"gram": gram}
        net.vars[self.pool_rec.outputs[0]].value = y
        return True

    @staticmethod
    def eligible(pool_block):
        ph, pw = pool_block.poolSize
        sy, sx = vl._pair(pool_block.stride, "STRIDE")
        return pool_block.method == "max" and -(-ph // sy) <= 2 and -(-pw // sx) <= 2 and ph * pw <= 255

    def forward(self, net):
        if self._stem_fwd is not None:
            return      # stem_forward did it all at the convolution's position
        r, pb = self.rec, self.pool_rec.block
        x = net.vars[r.inputs[0]].value
        g, b, mom = self._params(net)
        test = net.mode == "test"
        do = net._direct_der(r) if not test else None
        pre = None if test else r.block.take_pre_moments()
        y, am, mo = vl.bnorm_relu_pool(x, g, b, pb.poolSize, stride=pb.stride, pad=pb.pad,
                                       epsilon=r.block.epsilon, moments=mom if test else pre,
                                       moments_out=pre if pre is not None else (do[2] if do else None))
        r.block.moments = None if test else mo
        self._saved = (am, mo)
        net.vars[self.pool_rec.outputs[0]].value = y

    def backward(self, net):
        r, pb = self.rec, self.pool_rec.block
        out = net.vars[self.pool_rec.outputs[0]]
        if out.der is None:
            return
        am, mo = self._saved
        self._saved = None
        x = net.vars[r.inputs[0]].value
        g, b, mom = self._params(net)
        test = net.mode == "test"
        do = net._direct_der(r)
        if self._stem_backward(net, x, g, b, mom if test else mo, am, out, do, test):
            net._set_param_der(r.params[2], mo)
            if net.conserveMemory and not out.precious:
                out.der = None
            return
        # the producing convolution's bias derivative (= per-channel sum of this dx) comes for free
        bias_der = None
        if self.bias_conv is not None and net._flat is not None and not net.accumulateParamDers:
            bp = net.params[self.bias_conv.rec.params[1]]
            if bp.fanout == 1 and bp.der is not None:
                bias_der = bp.der
        self.bias_conv_done = bias_der is not None
        dx, dg, db = vl.bnorm_relu_pool_backward(x, g, b, mom if test else mo, am, out.der, pb.poolSize,
                                                 stride=pb.stride, pad=pb.pad, train=not test,
                                                 dg_out=do[0] if do else None, db_out=do[1] if do else None,
                                                 dxsum_out=bias_der, y_pool=out.value)
        net._set_var_der(r.inputs[0], dx)
        for p, d in zip(r.params, [dg, db, mo]):
            net._set_param_der(p, d)
        if net.conserveMemory and not out.precious:
            out.der = None


    def _stem_backward(self, net, x, g, b, moments, am, out, do, test):
        """The producing convolution is a first layer (its input needs no derivative) and feeds only this step: one
        fused call leaves its filter / bias derivative and this bnorm's dg / db; the bnorm's DZDX is never written.
        Runs on the side stream when there is one (nothing on the main stream depends on it).  False = not applicable,
        the two separate steps run."""
        fwd, self._stem_fwd = self._stem_fwd, None
        ok = self._stem_conditions(net) if (self._stem_fused and net.fuseStemBackward and do is not None) else None
        if ok is None:
            if fwd is not None:
                raise RuntimeError("fused stem: the forward pass skipped the convolution's output but the fused backward "
                                   "cannot run (%s)" % self.rec.name)
            return False
        cs, _, cdo = ok
        cr = cs.rec
        blk, pb = cr.block, self.pool_rec.block
        xin = net.vars[cr.inputs[0]].value
        side = net.wgradStream
        main = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(main)
        with torch.cuda.stream(side if side is not None else main):
            res = None
            if net.fuseStemGram or fwd is not None:
                # round 6: no pass over the convolution's output at all (the Gram matrix of the input patches); after the
                # fused forward the table marks the closed windows itself and y_pool is not read either
                cpar = cs._params(net)
                res = vl.conv_backward_filter_bnrelupool_gram(
                    xin, cpar[0], cpar[1] if blk.hasBias else None, g, moments, am, None if fwd is not None else out.value,
                    out.der, pb.poolSize, stride=blk.stride, pad=blk.pad, dilate=blk.dilate, pool_stride=pb.stride,
                    pool_pad=pb.pad, train=not test, gram=fwd["gram"] if fwd is not None else None, df_out=cdo[0],
                    dbias_out=cdo[1] if blk.hasBias else None, dg_out=do[0], db_out=do[1])
                if res is None and fwd is not None:
                    raise RuntimeError("fused stem: the library took the forward call but not the backward one (%s)" % cr.name)
            if res is None:
                res = vl.conv_backward_filter_bnrelupool(
                    xin, blk.size, x, g, b, moments, am, out.value, out.der, pb.poolSize, stride=blk.stride, pad=blk.pad,
                    dilate=blk.dilate, pool_stride=pb.stride, pool_pad=pb.pad, train=not test, df_out=cdo[0],
                    dbias_out=cdo[1] if blk.hasBias else None, dg_out=do[0], db_out=do[1], has_bias=blk.hasBias)
            if res is not None and net.gradHook is not None:
                if side is None and net._side_pending:
                    main.wait_stream(net.wgradStream)
                net.gradHook(cr.name)
        if res is None:
            self._stem_fused = False
            return False
        if side is not None:
            for t in (xin, x, am, out.value, out.der, moments) + ((fwd["gram"],) if fwd is not None else ()):
                if t is not None:
                    t.record_stream(side)
            net._side_pending = True
        self.bias_conv_done = True
        return True


class _ConvActStep(_Step):
    """forward-only plans: Conv -> ReLU / Sigmoid with the activation in the producing kernel (the SE gate's
    fc1 -> relu and fc2 -> sigmoid: four launches become two skinny-FC kernels, see xm_nnconv_forward_fused)."""

    def __init__(self, conv_rec, act_rec):
        super().__init__(conv_rec)
        self.act_rec = act_rec

    def forward(self, net):
        r, blk = self.rec, self.rec.block
        par = self._params(net)
        sig = isinstance(self.act_rec.block, Sigmoid)
        y = vl.vl_nnconv(net.vars[r.inputs[0]].value, par[0], par[1] if blk.hasBias else None, stride=blk.stride,
                         pad=blk.pad, dilate=blk.dilate, relu=not sig, sigmoid=sig)
        net.vars[self.act_rec.outputs[0]].value = y

    def backward(self, net):
        raise RuntimeError("dagnn: a forward-only plan was asked for derivatives")


class _ConvFoldStep(_Step):
    """test mode only: Conv -> BatchNorm [-> Sum(shortcut)] [-> ReLU] in the conv epilogue.
    scale_k = g_k / sigma_k, shift_k = b_k - mu_k * scale_k  (frozen moments)."""

    def __init__(self, conv_rec, bn_rec, sum_rec, relu_rec, out_name):
        super().__init__(conv_rec)
        self.bn_rec, self.sum_rec, self.relu_rec, self.out_name = bn_rec, sum_rec, relu_rec, out_name
        self._folded = None

    def _fold(self, net):
        g, b, mom = [net.params[p].value for p in self.bn_rec.params]
        key = (g.data_ptr(), b.data_ptr(), mom.data_ptr(), g._version, b._version, mom._version,
               net.paramGeneration)
        if self._folded is None or self._folded[0] != key:
            gn, bn_, mn = vl.to_numpy(g).ravel(), vl.to_numpy(b).ravel(), vl.to_numpy(mom)
            sc = (gn / mn[:, 1]).astype(np.float32)
            sh = (bn_ - mn[:, 0] * sc).astype(np.float32)
            self._folded = (key, vl.from_numpy(sc.reshape(-1, 1), g.device),
                            vl.from_numpy(sh.reshape(-1, 1), g.device))
        return self._folded[1], self._folded[2]

    def forward(self, net):
        r = self.rec
        blk = r.block
        x = net.vars[r.inputs[0]].value
        prm = self._params(net)
        sc, sh = self._fold(net)
        resid = None
        if self.sum_rec is not None:
            other = [v for v in self.sum_rec.inputs if v != self.bn_rec.outputs[0]][0]
            resid = net.vars[other].value
        y = vl.vl_nnconv(x, prm[0], prm[1] if blk.hasBias else None, stride=blk.stride, pad=blk.pad,
                         dilate=blk.dilate, scale=sc, shift=sh, residual=resid,
                         relu=self.relu_rec is not None)
        net.vars[self.out_name].value = y

    def backward(self, net):
        raise RuntimeError("folded conv+bn steps exist only in forward-only test-mode plans")


class _SEFoldStep(_ConvFoldStep):
    """test mode only: the whole SE tail of a bottleneck as TWO passes over narrow tensors + ONE over the wide one.

        x  = bnorm(conv1x1(u))                 (projection C/4 -> C, frozen moments: affine in u)
        a  = sigmoid(fc2(relu(fc1(mean_hw(x)))))
        y  = relu(a .* x + shortcut)           (mcnExtraLayers dagnn.Axpy + ReLU)

    mean_hw(x) = scale .* (F * mean_hw(u) + b) + shift by linearity, so the gate a is computed from the 4 x narrower u
    BEFORE the projection runs, and the projection's epilogue writes relu(a .* x + shortcut) directly
    (vl_nnconv(..., gate=a, residual=shortcut)): x is never materialised -- no write, no squeeze read, no excite
    read / write of the widest tensors of the network."""

    def __init__(self, conv_rec, bn_rec, se):
        super().__init__(conv_rec, bn_rec, None, se["relu"], se["out"])
        self.se = se
        self._reduce = None

    def _fold_reduce(self, net, F, bias, sc, sh, F1, b1):
        """(filter [1 1 C/4 C/16], bias [C/16 1]) of the reduction layer applied straight to the pooled projection input;
        float64 on the host, once per parameter set (keyed like _fold)"""
        ts = [t for t in (F, bias, sc, sh, F1, b1) if t is not None]
        key = tuple(t.data_ptr() for t in ts) + tuple(t._version for t in ts) + (net.paramGeneration,)
        if self._reduce is None or self._reduce[0] != key:
            Kp, Cc = int(F.shape[2]), int(F.shape[3])
            Fm = vl.to_numpy(F).astype(np.float64).reshape(Kp, Cc, order="F")                 # column m = filter m
            F1m = vl.to_numpy(F1).astype(np.float64).reshape(Cc, int(F1.shape[3]), order="F")
            scn, shn = vl.to_numpy(sc).astype(np.float64).ravel(), vl.to_numpy(sh).astype(np.float64).ravel()
            bn_ = vl.to_numpy(bias).astype(np.float64).ravel() if bias is not None else np.zeros(Cc)
            f10 = (Fm * scn[None, :]) @ F1m                                                   # [C/4, C/16]
            b10 = F1m.T @ (scn * bn_ + shn)
            if b1 is not None:
                b10 = b10 + vl.to_numpy(b1).astype(np.float64).ravel()
            self._reduce = (key,
                            vl.from_numpy(np.asfortranarray(f10.astype(np.float32).reshape(1, 1, Kp, -1, order="F")), F.device),
                            vl.from_numpy(b10.astype(np.float32).reshape(-1, 1), F.device))
        return self._reduce[1], self._reduce[2]

    def forward(self, net):
        r, blk, se = self.rec, self.rec.block, self.se
        u = net.vars[r.inputs[0]].value
        prm = self._params(net)
        bias = prm[1] if blk.hasBias else None
        sc, sh = self._fold(net)
        ubar = vl.vl_nnpool(u, [int(u.shape[0]), int(u.shape[1])], method="avg")
        f1 = [net.params[p].value for p in se["fc1"].params]
        f2 = [net.params[p].value for p in se["fc2"].params]
        b1 = f1[1] if se["fc1"].block.hasBias else None
        if net.foldSEReduce:
            # the squeeze's projection and the reduction layer are both affine and nothing sits between them:
            #   fc1(scale .* (F' ubar + b) + shift) = (F1' diag(scale) F') ubar + F1' (scale .* b + shift) + b1
            # -- ONE skinny product C/4 -> C/16 per block instead of C/4 -> C -> C/16 (the wider one was a 19 us launch at
            # 256 faces, 7-9 us at 32-128), the folded filter built once per parameter set
            f10, b10 = self._fold_reduce(net, prm[0], bias, sc, sh, f1[0], b1)
            h = vl.vl_nnconv(ubar, f10, b10, relu=True)
        else:
            z = vl.vl_nnconv(ubar, prm[0], bias, scale=sc, shift=sh)
            h = vl.vl_nnconv(z, f1[0], b1, relu=True)
        a = vl.vl_nnconv(h, f2[0], f2[1] if se["fc2"].block.hasBias else None, sigmoid=True)
        y = vl.vl_nnconv(u, prm[0], bias, stride=blk.stride, pad=blk.pad, dilate=blk.dilate, scale=sc, shift=sh,
                         gate=a, residual=net.vars[se["shortcut"]].value, relu=se["relu"] is not None)
        net.vars[self.out_name].value = y


def _match_se_tail(net, recs, consumers, order, conv_rec, bn_rec, keep_values=False):
    """the SE tail behind a 1 x 1 / stride-1 projection + bnorm: {GlobalPooling('avg') -> Conv -> ReLU -> Conv -> Sigmoid}
    and Axpy(gate, x, shortcut) [-> ReLU] as the only consumers of x, nothing precious in between (`keep_values`: the
    caller still materialises every variable of the gate path -- training plans -- so precious ones there are fine)"""
    blk = conv_rec.block
    if blk.size[0] != 1 or blk.size[1] != 1 or tuple(vl._pair(blk.stride, "STRIDE")) != (1, 1) or any(vl._pad4(blk.pad)):
        return None
    x = bn_rec.outputs[0]
    cs = consumers.get(x, [])
    if len(cs) != 2 or net.vars[x].precious:
        return None
    gp = [c for c in cs if isinstance(c.block, GlobalPooling) and c.block.method == "avg"]
    ax = [c for c in cs if isinstance(c.block, Axpy)]
    if len(gp) != 1 or len(ax) != 1:
        return None
    gp, ax = gp[0], ax[0]

    def only(var, cls):
        c = consumers.get(var, [])
        return c[0] if len(c) == 1 and isinstance(c[0].block, cls) and (keep_values or not net.vars[var].precious) else None
    fc1 = only(gp.outputs[0], Conv)
    r1 = only(fc1.outputs[0], ReLU) if fc1 else None
    fc2 = only(r1.outputs[0], Conv) if r1 else None
    sg = only(fc2.outputs[0], Sigmoid) if fc2 else None
    if sg is None or r1.block.leak != 0.0 or only(sg.outputs[0], Axpy) is not ax:
        return None
    for fc in (fc1, fc2):
        b = fc.block
        if b.size[0] != 1 or b.size[1] != 1 or tuple(vl._pair(b.stride, "STRIDE")) != (1, 1) or any(vl._pad4(b.pad)):
            return None
    if len(ax.inputs) != 3 or ax.inputs[0] != sg.outputs[0] or ax.inputs[1] != x:
        return None
    shortcut = ax.inputs[2]
    prod = [q for q in recs if shortcut in q.outputs]
    if prod and not all(order[id(q)] < order[id(conv_rec)] for q in prod):
        return None
    out, relu = ax.outputs[0], None
    rl = only(out, ReLU)
    if rl is not None and rl.block.leak == 0.0:
        relu, out = rl, rl.outputs[0]
    return {"gp": gp, "fc1": fc1, "relu1": r1, "fc2": fc2, "sig": sg, "axpy": ax, "relu": relu, "out": out,
            "shortcut": shortcut}


def _link_bias_conv(bn_step, steps, r, consumers, training):
    """training plans: a biased Conv whose only consumer is this bnorm gets its dzdb from the bnorm backward (sum of
    dx) instead of a pass of its own over dzdy"""
    if not training:
        return
    prod = [q for q in steps if type(q) is _Step and r.inputs[0] in q.rec.outputs]
    if prod and isinstance(prod[0].rec.block, Conv) and prod[0].rec.block.hasBias and \
            len(consumers.get(r.inputs[0], [])) == 1:
        bn_step.bias_conv = prod[0]
        prod[0].bias_from = bn_step


def build_plan(net, training):
    recs = net.layers
    if not net.fuse:
        return [_Step(r) for r in recs]
    consumers = {}
    for r in recs:
        for v in r.inputs:
            consumers.setdefault(v, []).append(r)
    produced_before = {}
    order = {id(r): i for i, r in enumerate(recs)}

    def sole_consumer(var, cls):
        cs = consumers.get(var, [])
        if len(cs) == 1 and isinstance(cs[0].block, cls) and not net.vars[var].precious:
            return cs[0]
        return None

    steps, skip = [], set()
    fold_ok = (not training) and net.mode == "test"
    for r in recs:
        if id(r) in skip:
            continue
        if fold_ok and isinstance(r.block, Conv):
            bn = sole_consumer(r.outputs[0], BatchNorm)
            se = _match_se_tail(net, recs, consumers, order, r, bn) if (bn is not None and net.fuseSE) else None
            if se is not None:
                steps.append(_SEFoldStep(r, bn, se))
                skip.update(id(q) for q in (bn, se["gp"], se["fc1"], se["relu1"], se["fc2"], se["sig"], se["axpy"],
                                            se["relu"]) if q is not None)
                continue
            if bn is not None:
                out = bn.outputs[0]
                sm = sole_consumer(out, Sum)
                # the shortcut operand must already be computed when the conv runs
                if sm is not None and len(sm.inputs) == 2:
                    other = [v for v in sm.inputs if v != out][0]
                    prod = [q for q in recs if other in q.outputs]
                    ready = (not prod) or all(order[id(q)] < order[id(r)] for q in prod)
                    if ready:
                        out = sm.outputs[0]
                    else:
                        sm = None
                else:
                    sm = None
                rl = sole_consumer(out, ReLU)
                if rl is not None and rl.block.leak == 0.0:
                    out = rl.outputs[0]
                else:
                    rl = None
                steps.append(_ConvFoldStep(r, bn, sm, rl, out))
                skip.update(id(q) for q in (bn, sm, rl) if q is not None)
                continue
            act = sole_consumer(r.outputs[0], (ReLU, Sigmoid))
            if act is not None and (isinstance(act.block, Sigmoid) or act.block.leak == 0.0):
                steps.append(_ConvActStep(r, act))
                skip.add(id(act))
                continue
        if isinstance(r.block, (Sum, Axpy)) and (not isinstance(r.block, Sum) or len(r.inputs) == 2):
            rl = sole_consumer(r.outputs[0], ReLU)
            if rl is not None and rl.block.leak == 0.0:
                steps.append(_AddReluStep(r, rl))
                skip.add(id(rl))
                continue
        if isinstance(r.block, BatchNorm):
            rl = sole_consumer(r.outputs[0], ReLU)
            if rl is not None and rl.block.leak == 0.0:
                pl = sole_consumer(rl.outputs[0], Pooling)
                if pl is not None and not isinstance(pl.block, GlobalPooling) and \
                        _BnReluPoolStep.eligible(pl.block):
                    bias_conv = None
                    prod = [q for q in steps if type(q) is _Step and r.inputs[0] in q.rec.outputs]
                    if training and prod and isinstance(prod[0].rec.block, Conv) and prod[0].rec.block.hasBias \
                            and len(consumers.get(r.inputs[0], [])) == 1:
                        bias_conv = prod[0]
                    steps.append(_BnReluPoolStep(r, rl, pl, bias_conv))
                    skip.update((id(rl), id(pl)))
                    continue
                st_ = _BnReluStep(r, rl)
                _link_bias_conv(st_, steps, r, consumers, training)
                steps.append(st_)
                skip.add(id(rl))
                continue
            st_ = _Step(r)
            _link_bias_conv(st_, steps, r, consumers, training)
            steps.append(st_)
            continue
        steps.append(_Step(r))
    # a fused step may now sit before the producer of one of its operands is scheduled; the
    # `ready` test above guarantees producers precede the conv, so plain order is still valid.
    del produced_before
    if training:
        # Conv -> BatchNorm (train mode): the convolution's epilogue computes the batch moments (vl_nnconv
        # moments_out), the bnorm step -- plain, +relu or +relu+pool -- takes them instead of re-reading its input
        conv_of = {st.rec.outputs[0]: st for st in steps if type(st) is _Step and isinstance(st.rec.block, Conv)}
        for st in steps:
            r = st.rec
            if isinstance(r.block, BatchNorm) and type(st) in (_Step, _BnReluStep, _BnReluPoolStep):
                cs = conv_of.get(r.inputs[0])
                if cs is not None and len(consumers.get(r.inputs[0], [])) == 1:
                    cs.moments_for = r
                    if type(st) is _BnReluPoolStep:
                        st.producer_conv = cs
                        cs.stem_into = st
        # SE blocks: the plain bnorm behind the projection, its squeeze and its excite (+ relu) share one fused backward
        by_rec = {id(st.rec): st for st in steps}
        for i, st in enumerate(steps):
            r = st.rec
            if type(st) is not _Step or not isinstance(r.block, BatchNorm):
                continue
            cs = conv_of.get(r.inputs[0])
            if cs is None or len(consumers.get(r.inputs[0], [])) != 1:
                continue
            se = _match_se_tail(net, recs, consumers, order, cs.rec, r, keep_values=True)
            if se is None or se["relu"] is None:
                continue
            ax_step, gp_step = by_rec.get(id(se["axpy"])), by_rec.get(id(se["gp"]))
            if type(ax_step) is not _AddReluStep or type(gp_step) is not _Step:
                continue
            new_st = _SEBnTrainStep(r, ax_step, gp_step)
            new_st.bias_conv, new_st.bias_from, new_st.moments_for = st.bias_conv, st.bias_from, st.moments_for
            for q in steps:
                if q.bias_from is st:
                    q.bias_from = new_st
            steps[i] = new_st
    return steps
