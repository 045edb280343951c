"""emoVoxZoo / ferPlusZoo mirrors: the student and teacher graphs of the reference.

The reference downloads the networks as .mat files (emoVoxCeleb/emoVoxZoo.m:36-40,95-97;
teacher/ferPlusZoo.m:93-101); those files are not available offline, so the architectures are
restated from the published models and the weights are seeded synthetic
ones.  The one structural pin the reference holds -- the pool6 width table at
emoVoxZoo.m:258-259 -- is reproduced exactly by `vggvox()` (tests/test_pins.py).

Net surgery follows the reference line by line where it exists:
    prepareFromDagNN        emoVoxZoo.m:187-253   strip Loss/SoftMax, last FC -> numOutputs
    configureForRegression  emoVoxZoo.m:105-183   loss + classerror + ErrorStats layers
    updatePooling           emoVoxZoo.m:256-269   pool6 <- [1 p1] from the bucket table
"""
import numpy as np
import torch

from . import dagnn, vl

EMOTIONS = ["neutral", "happiness", "surprise", "sadness", "anger", "disgust", "fear", "contempt"]

# emoVoxZoo.m:258-259 (also external/compute_audio_feats.m:45-46)
BUCKETS_POOL = [2, 5, 8, 11, 14, 17, 20, 23, 27, 30]
BUCKETS_WIDTH = list(range(100, 1001, 100))


# ---------------------------------------------------------------------------------------------
# raw architectures ("what the .mat files contain")
# ---------------------------------------------------------------------------------------------
def vggvox(num_classes=1251, width_mult=1.0):
    """VGGVox (VGG-M, BN variant), input 512 x W x 1 -- SURVEY Appendix B.1.
    `width_mult` < 1 shrinks the channel counts for CPU-sized tests (geometry unchanged)."""
    def c(n):
        return max(4, int(round(n * width_mult)))

    net = dagnn.DagNN()
    spec = [  # name, FH, FW, Cin, Cout, stride, pad
        ("1", 7, 7, 1, c(96), 2, 1), ("2", 5, 5, c(96), c(256), 2, 1), ("3", 3, 3, c(256), c(384), 1, 1),
        ("4", 3, 3, c(384), c(256), 1, 1), ("5", 3, 3, c(256), c(256), 1, 1),
    ]
    x = "input"
    for name, fh, fw, ci, co, s, p in spec:
        net.addLayer("conv" + name, dagnn.Conv([fh, fw, ci, co], True, (s, s), (p, p, p, p)), x,
                     "x_conv" + name, ["conv%sf" % name, "conv%sb" % name])
        net.addLayer("bn" + name, dagnn.BatchNorm(co), "x_conv" + name, "x_bn" + name,
                     ["bn%sm" % name, "bn%sb" % name, "bn%sx" % name])
        net.addLayer("relu" + name, dagnn.ReLU(), "x_bn" + name, "x_relu" + name)
        x = "x_relu" + name
        if name in ("1", "2"):
            net.addLayer("mpool" + name, dagnn.Pooling([3, 3], (2, 2), (0, 0, 0, 0), "max"), x,
                         "x_mpool" + name)
            x = "x_mpool" + name
        if name == "5":
            net.addLayer("mpool5", dagnn.Pooling([5, 3], (3, 2), (0, 0, 0, 0), "max"), x, "x_mpool5")
            x = "x_mpool5"
    net.addLayer("fc6", dagnn.Conv([9, 1, c(256), c(4096)], True), x, "x_fc6", ["fc6f", "fc6b"])
    net.addLayer("bn6", dagnn.BatchNorm(c(4096)), "x_fc6", "x_bn6", ["bn6m", "bn6b", "bn6x"])
    net.addLayer("relu6", dagnn.ReLU(), "x_bn6", "x_relu6")
    net.addLayer("pool6", dagnn.Pooling([1, 8], (1, 1), (0, 0, 0, 0), "avg"), "x_relu6", "x_pool6")
    net.addLayer("fc7", dagnn.Conv([1, 1, c(4096), c(1024)], True), "x_pool6", "x_fc7", ["fc7f", "fc7b"])
    net.addLayer("bn7", dagnn.BatchNorm(c(1024)), "x_fc7", "x_bn7", ["bn7m", "bn7b", "bn7x"])
    net.addLayer("relu7", dagnn.ReLU(), "x_bn7", "x_relu7")
    net.addLayer("fc8", dagnn.Conv([1, 1, c(1024), num_classes], True), "x_relu7", "x_fc8",
                 ["fc8f", "fc8b"])
    net.addLayer("softmax", dagnn.SoftMax(), "x_fc8", "prob")
    net.meta = {"normalization": {"imageSize": [512, 300, 1]},
                "audio": {"fs": 16000, "Tw": 25, "Ts": 10, "nfft": 1024}}
    return net


def resnet50(se=False, num_classes=8, width_mult=1.0, blocks=(3, 4, 6, 3)):
    """(SE-)ResNet-50, Caffe style (stride on the first 1x1 of a stage), input 224 x 224 x 3 --
    SURVEY Appendix B.2 / B.3.  Variable / layer names follow the Caffe imports."""
    def c(n):
        return max(4, int(round(n * width_mult)))

    net = dagnn.DagNN()
    net.addLayer("conv1", dagnn.Conv([7, 7, 3, c(64)], True, (2, 2), (3, 3, 3, 3)), "data", "conv1",
                 ["conv1_filter", "conv1_bias"])
    net.addLayer("bn_conv1", dagnn.BatchNorm(c(64), 1e-5), "conv1", "conv1_bn",
                 ["bn_conv1_mult", "bn_conv1_bias", "bn_conv1_moments"])
    net.addLayer("conv1_relu", dagnn.ReLU(), "conv1_bn", "conv1x")
    net.addLayer("pool1", dagnn.Pooling([3, 3], (2, 2), (0, 1, 0, 1), "max"), "conv1x", "pool1")
    x, cin = "pool1", c(64)
    for si, nb in enumerate(blocks):
        mid, cout = c(64 * 2 ** si), c(256 * 2 ** si)
        for bi in range(nb):
            tag = "res%d%s" % (si + 2, "abcdefgh"[bi])
            stride = 2 if (bi == 0 and si > 0) else 1
            sc = x
            if bi == 0:
                net.addLayer(tag + "_branch1", dagnn.Conv([1, 1, cin, cout], False, (stride, stride)), x,
                             tag + "_branch1", [tag + "_branch1_filter"])
                net.addLayer("bn" + tag[3:] + "_branch1", dagnn.BatchNorm(cout, 1e-5), tag + "_branch1",
                             tag + "_branch1_bn", [tag + "_b1_mult", tag + "_b1_bias", tag + "_b1_moments"])
                sc = tag + "_branch1_bn"
            y = x
            for li, (fh, ci, co, s, p) in enumerate([(1, cin, mid, stride, 0), (3, mid, mid, 1, 1),
                                                      (1, mid, cout, 1, 0)]):
                nm = tag + "_branch2" + "abc"[li]
                net.addLayer(nm, dagnn.Conv([fh, fh, ci, co], False, (s, s), (p, p, p, p)), y, nm,
                             [nm + "_filter"])
                net.addLayer("bn" + nm[3:], dagnn.BatchNorm(co, 1e-5), nm, nm + "_bn",
                             [nm + "_mult", nm + "_bias", nm + "_moments"])
                y = nm + "_bn"
                if li < 2:
                    net.addLayer(nm + "_relu", dagnn.ReLU(), y, nm + "x")
                    y = nm + "x"
            if se:
                r = max(4, cout // 16)
                net.addLayer(tag + "_global_pool", dagnn.GlobalPooling("avg"), y, tag + "_gp")
                net.addLayer(tag + "_fc1", dagnn.Conv([1, 1, cout, r], True), tag + "_gp", tag + "_fc1",
                             [tag + "_fc1_filter", tag + "_fc1_bias"])
                net.addLayer(tag + "_fc1_relu", dagnn.ReLU(), tag + "_fc1", tag + "_fc1x")
                net.addLayer(tag + "_fc2", dagnn.Conv([1, 1, r, cout], True), tag + "_fc1x", tag + "_fc2",
                             [tag + "_fc2_filter", tag + "_fc2_bias"])
                net.addLayer(tag + "_prob", dagnn.Sigmoid(), tag + "_fc2", tag + "_prob")
                net.addLayer(tag, dagnn.Axpy(), [tag + "_prob", y, sc], tag)
            else:
                net.addLayer(tag, dagnn.Sum(), [sc, y], tag)
            net.addLayer(tag + "_relu", dagnn.ReLU(), tag, tag + "x")
            x, cin = tag + "x", cout
    net.addLayer("pool5", dagnn.Pooling([7, 7], (1, 1), (0, 0, 0, 0), "avg"), x, "pool5")
    net.addLayer("classifier", dagnn.Conv([1, 1, cin, num_classes], True), "pool5", "prediction",
                 ["classifier_filter", "classifier_bias"])
    net.addLayer("loss", dagnn.Loss("softmaxlog"), ["prediction", "label"], "objective")
    net.addLayer("top1error", dagnn.Loss("classerror"), ["prediction", "label"], "top1error")
    net.meta = {"normalization": {"imageSize": [224, 224, 3], "averageImage": [131.0912, 103.8827, 91.4953]},
                "classes": {"name": list(EMOTIONS), "description": list(EMOTIONS)}}
    return net


def synthetic_pretrained(net, seed):
    """Seeded stand-in for downloaded weights: He-normal filters, zero biases,
    BN g = 1, b = 0, stored moments mean ~ N(0, .1), sigma ~ U(.5, 1.5)."""
    net.initParams(seed)
    rng = np.random.default_rng(seed + 1)
    for l in net.layers:
        if isinstance(l.block, dagnn.BatchNorm):
            C = l.block.numChannels
            mom = np.zeros((C, 2), np.float32, order="F")
            mom[:, 0] = rng.standard_normal(C) * 0.1
            mom[:, 1] = rng.uniform(0.5, 1.5, C)
            net.params[l.params[2]].value = mom
    return net


def calibrate_moments(net, inputs):
    """Give a synthetic teacher realistic stored moments: one train-mode forward over `inputs`
    and copy each BatchNorm's batch moments into its `moments` parameter (what training with
    trainMethod 'average' converges to).  Runs on the GPU through the HIP kernels."""
    old_mode, old_fuse = net.mode, net.fuse
    net.mode, net.fuse = "normal", False
    try:
        if not isinstance(inputs, dict):
            inputs = {inputs[i]: inputs[i + 1] for i in range(0, len(inputs), 2)}
        for v in net.vars.values():
            v.value = None
        for k, t in inputs.items():
            if k in net.vars:
                net.vars[k].value = t
        for l in net.layers:
            if isinstance(l.block, dagnn.LossBase):
                continue
            ins = [net.vars[v].value for v in l.inputs]
            prm = [net.params[p].value for p in l.params]
            outs = l.block.forward(ins, prm)
            if isinstance(l.block, dagnn.BatchNorm):
                net.params[l.params[2]].value.copy_(l.block.moments)
                outs = l.block.forward(ins, prm)
            for v, t in zip(l.outputs, outs):
                net.vars[v].value = t
        for v in net.vars.values():
            v.value = None
    finally:
        net.mode, net.fuse = old_mode, old_fuse
    return net


# ---------------------------------------------------------------------------------------------
# ferPlusZoo (teacher)  -- teacher/ferPlusZoo.m:93-114,127-133
# ---------------------------------------------------------------------------------------------
def ferPlusZoo(modelName, seed=100, width_mult=1.0, blocks=(3, 4, 6, 3)):
    """dag = ferPlusZoo(modelName): pretrained branch only (the non-pretrained branch of the
    reference is unreachable as shipped -- SURVEY Appendix C)."""
    if modelName == "resnet50-ferplus":
        net = resnet50(False, 8, width_mult, blocks)
    elif modelName == "senet50-ferplus":
        net = resnet50(True, 8, width_mult, blocks)
    else:
        raise ValueError("%s is not a recognised FER+ teacher" % modelName)  # ferPlusZoo.m:87
    synthetic_pretrained(net, seed)
    # ferPlusZoo.m:127-133: make sure the input variable is called 'data'
    for old in ("input", "x0"):
        if old in net.vars:
            net.renameVar(old, "data")
    net.meta["modelName"] = modelName
    return net


def strip_losses(net):
    """fetch_emovoxceleb_imdb.m:101-106: remove every dagnn.Loss layer before inference."""
    names = [l.name for l in net.layers if isinstance(l.block, dagnn.LossBase)]
    net.removeLayer(names)
    return net


class FrozenTeacher:
    """The teacher loop of fetch_emovoxceleb_imdb.m:98-136 -- `dag.eval({in, data})` in test mode,
    logits = value of the last variable (:130) -- with the batch cut into `lanes` sample slices that
    run concurrently on `lanes` HIP streams (one DagNN replica per lane, parameters shared).
    A ResNet forward is a chain of ~55 dependent launches; with a single stream every launch pays its
    ramp-up and tail alone, two lanes fill each other's gaps (measured on MI355X: -7 % time at batch
    128, no gain at 32).  Samples are independent in test mode, so the logits are those of the
    single-lane evaluation up to the summation order of the tile configuration chosen per shape."""

    def __init__(self, net, lanes=2, output=None):
        if net.mode != "test":
            raise ValueError("FrozenTeacher needs a test-mode network (fetch_emovoxceleb_imdb.m:107)")
        self.output = output or net.getOutputs()[-1]
        net.vars[self.output].precious = True
        self.nets = [net] + [net.replica() for _ in range(max(1, int(lanes)) - 1)]
        self.streams = [torch.cuda.Stream(device=net.device) for _ in self.nets] if len(self.nets) > 1 else []

    def logits(self, faces, input_name="data"):
        """faces: 224 x 224 x 3 x N mat -> 1 x 1 x numClasses x N mat."""
        N = int(faces.shape[3])
        L = len(self.nets)
        if L == 1 or N < 2 * L:
            self.nets[0].eval([input_name, faces])
            return self.nets[0].vars[self.output].value
        cur = torch.cuda.current_stream()
        bounds = [N * i // L for i in range(L + 1)]
        for i, (net, st) in enumerate(zip(self.nets, self.streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                net.eval([input_name, faces[..., bounds[i]:bounds[i + 1]]])
        for st in self.streams:
            cur.wait_stream(st)
        outs = [net.vars[self.output].value for net in self.nets]
        C = int(outs[0].shape[2])
        out = vl.mat_empty((1, 1, C, N), device=faces.device)
        for i, o in enumerate(outs):
            out[..., bounds[i]:bounds[i + 1]].copy_(o)
        return out


# ---------------------------------------------------------------------------------------------
# emoVoxZoo (student)
# ---------------------------------------------------------------------------------------------
def prepareFromDagNN(net, numOutputs, seed=0):
    """emoVoxZoo.m:187-253: drop Loss / SoftMax layers, resize the last FC to numOutputs with
    1e-4 * randn filters (rng(0), :217-220), name the output 'prediction', input 'data'."""
    drop = [l.name for l in net.layers
            if isinstance(l.block, (dagnn.LossBase, dagnn.SoftMax))]
    net.removeLayer(drop)
    convs = [l for l in net.layers if isinstance(l.block, dagnn.Conv)]
    last = convs[-1]
    FH, FW, FC, _ = last.block.size
    last.block.size = (FH, FW, FC, numOutputs)
    rng = np.random.default_rng(seed)
    net.params[last.params[0]].value = np.asfortranarray(
        (1e-4 * rng.standard_normal((FH, FW, FC, numOutputs))).astype(np.float32))
    net.params[last.params[1]].value = np.zeros((numOutputs, 1), np.float32)
    net.renameVar(last.outputs[0], "prediction")
    inputs = net.getInputs()
    assert len(inputs) == 1, "expected a single-input network"  # emoVoxZoo.m:35
    if inputs[0] != "data":
        net.renameVar(inputs[0], "data")
    return net


def insert_dropout(net, prev, nxt, rate):
    """emoVoxZoo.m:272-277: a dagnn.DropOut named <prev>_drop between layer `prev` and its consumer `nxt`."""
    prec = net.layers[net.getLayerIndex(prev)]
    out = "%s_drop" % prev
    net.insertLayerAfter(prev, out, dagnn.DropOut(rate=rate, seed=net.getLayerIndex(prev) + 1), prec.outputs, out)
    nrec = net.layers[net.getLayerIndex(nxt)]
    net.setLayerInputs(nxt, [out if v == prec.outputs[0] else v for v in nrec.inputs])
    return net


def configureForRegression(net, lossType, numOutputs, dropout=0):
    """emoVoxZoo.m:105-177 (the dropout layers of :116-135 first, then the loss heads)."""
    if dropout and dropout > 0 and not any(isinstance(l.block, dagnn.DropOut) for l in net.layers):
        convs = [l for l in net.layers if isinstance(l.block, dagnn.Conv)]
        for sel in convs[-3:-1]:                      # convLayers(end-2:end-1): "reduce aggression" (:120)
            nxt = [l.name for l in net.layers if sel.outputs[0] in l.inputs]
            assert nxt, "target layer was not found"  # :132
            insert_dropout(net, sel.name, nxt[0], float(dropout))
    if lossType == "softmaxlog":
        layer, inputs = dagnn.Loss("softmaxlog"), ["prediction", "maxLabel"]
    elif lossType == "hot-cross-ent":
        # emoVoxZoo.m:152 -- the temperature is hard-coded to 2 (opts.temperature only names
        # the experiment directory, run_distillation.m:85,102-104)
        layer, inputs = dagnn.SoftmaxCELoss(temperature=2, logitTargets=True), ["prediction", "logitTarget"]
    elif lossType == "euclidean":
        layer, inputs = dagnn.EuclideanLoss(), ["prediction", "logitTarget", "instanceWeights"]
        # emoVoxZoo.m:141-145: "scale down a lot to prevent exploding gradients" -- the filters of the
        # last layer are divided by 10
        p = net.params[net.layers[-1].params[0]]
        p.value = p.value / 10
    elif lossType == "huber":
        layer, inputs = dagnn.HuberLoss(sigma=1), ["prediction", "logitTarget", "instanceWeights"]
    else:
        raise ValueError("unrecognised regression loss: %s" % lossType)
    net.addLayer("loss", layer, inputs, "objective")
    net.addLayer("classerror", dagnn.VerboseLoss("classerror"), ["prediction", "maxLabel"], "classerror")
    net.addLayer("classAccs", dagnn.ErrorStats(numOutputs), ["prediction", "maxLabel"], "classAccs")
    net.meta.setdefault("classes", {})
    net.meta["classes"]["name"] = list(EMOTIONS)
    net.meta["classes"]["description"] = list(EMOTIONS)
    return net


def updatePooling(net, numSeconds):
    """emoVoxZoo.m:256-269: pool6.poolSize = [1 p1] for clips of `numSeconds` seconds."""
    width = int(round(100 * numSeconds))
    if width not in BUCKETS_WIDTH:
        raise ValueError("no pooling bucket for width %d" % width)
    p1 = BUCKETS_POOL[BUCKETS_WIDTH.index(width)]
    pools = [l for l in net.layers if l.name == "pool6"]
    assert len(pools) == 1, "expected a single pool6 layer"  # emoVoxZoo.m:268
    pools[0].block.poolSize = [1, p1]
    return net


def emoVoxZoo(modelName="emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=4,
              numOutputs=8, seed=200, width_mult=1.0, dropout=False):
    """dag = emoVoxZoo(name, 'scratch', 1, 'lossType', 'hot-cross-ent', 'numSeconds', 4,
    'numOutputs', 8, 'dropout', rate) -- emoVoxZoo.m:1-62 (call site run_distillation.m:125-129)."""
    if modelName not in ("emovoxceleb-student", "vggvox-ver", "vggvox-ident"):
        raise ValueError("%s is not a recognised student model" % modelName)
    net = vggvox(1251, width_mult)
    net.initParams(seed)               # stands in for the downloaded weights
    if scratch:
        prepareFromDagNN(net, numOutputs)
        net.initParams(seed)           # emoVoxZoo.m:54: re-randomise everything
        configureForRegression(net, lossType, numOutputs, dropout)
    else:
        prepareFromDagNN(net, numOutputs)
    updatePooling(net, numSeconds)
    net.meta["modelName"] = modelName
    return net
