"""cnn_train_dag step mirror: forward + backward + ParameterServer + accumulateGradients.

Reference call: cnn_train_dag(net, imdb, getBatchFn, 'learningRate', logspace(-4,-5,300),
'batchSize', 64, 'gpus', ..., 'parameterServer', struct('method','tmove'), ...)
(emoVoxCeleb/run_distillation.m:170-182).  The loop itself stays on the host; per minibatch it is
    net.eval(inputs, {'objective', 1})          -> HIP kernels
    parserv.push / sync / pull                  -> one RCCL sum-all-reduce of the flat der buffer
    accumulateGradients (momentum 0.9, wd 5e-4) -> fused HIP SGD / moving-average kernels
Data parallelism follows MatConvNet: each worker takes a shard of the minibatch, BN statistics
are per worker, gradients are summed and divided by the GLOBAL batch size.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, vl


class TrainOpts:
    def __init__(self, learningRate=None, momentum=0.9, weightDecay=5e-4, batchSize=64,
                 derOutputs=("objective", 1)):
        # run_distillation.m:76,87: 300 mini-epochs, logspace(-4, -5, 300)
        self.learningRate = np.logspace(-4, -5, 300) if learningRate is None else np.atleast_1d(learningRate)
        self.momentum = momentum
        self.weightDecay = weightDecay
        self.batchSize = batchSize
        self.derOutputs = list(derOutputs)


class ParameterServer:
    """ParameterServer.{start, push, pull, sync} collapsed to a sum-all-reduce.

    backend 'rccl-capi' : libxmodal_hip's own communicator (xm_comm_init + xm_allreduce_sum_f32),
                          unique id distributed through torch.distributed's store;
    backend 'torch'     : torch.distributed.all_reduce (nccl == RCCL on ROCm, gloo on CPU tests)."""

    def __init__(self, backend="torch"):
        self.backend = backend
        self.world = 1
        self.rank = 0
        self._started = False

    def start(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(), dist.get_rank()
        if self.world > 1 and self.backend == "rccl-capi":
            L = _lib.load()
            buf = (C.c_char * 128)()
            if self.rank == 0:
                _lib.check(L.xm_comm_unique_id(buf))
            t = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.broadcast(t, 0)
            raw = bytes(t.cpu().tolist())
            _lib.check(L.xm_comm_init(C.c_char_p(raw), self.rank, self.world))
        self._started = True

    force = False  # debugging: run the collective even with a single worker
    overlap = True  # 'torch' backend: bucketed exchange overlapped with the backward pass (GradBuckets)

    def allreduce_(self, flat):
        if self.world == 1 and not self.force:
            return
        if self.backend == "rccl-capi":
            _lib.check(_lib.load().xm_allreduce_sum_f32(C.c_void_p(flat.data_ptr()), flat.numel(),
                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        else:
            import torch.distributed as dist
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)

    def stop(self):
        if self.backend == "rccl-capi" and self.world > 1:
            _lib.check(_lib.load().xm_comm_destroy())


class GradBuckets:
    """Overlapped gradient exchange: the filters of the last FC layers are 82 % of the
    student's gradient bytes (fc6 37.7 MB + fc7 16.8 MB of 66.6 MB) and their derivatives are the FIRST
    to be ready in the backward pass.  They form one contiguous range of the flat buffer (the tail of the
    filter segment), which is all-reduced asynchronously as soon as the earliest of those layers has
    enqueued its wgrad -- the rest of the backward pass (conv5 ... conv1, ~4 ms) hides it.  The
    remainder (conv filters, biases, BN parameters and moments) goes out after the backward pass.
    Same sums as one all-reduce over the whole buffer: every element is reduced exactly once."""

    def __init__(self, net, early_layers=("fc6", "fc7", "fc8")):
        flat = net._flat
        total = int(flat.der.numel())
        recs = [net.getLayer(n) for n in early_layers]
        recs = [r for r in recs if r is not None]
        self.early = None
        self.trigger = None
        if recs:
            ps = [net.params[r.params[0]] for r in recs]
            a = min(p._flat_off for p in ps)
            b = max(p._flat_off + (int(p.value.numel()) + 3) // 4 * 4 for p in ps)
            inside = {id(q) for q in net.params.values() if a <= getattr(q, "_flat_off", -1) < b}
            if inside == {id(p) for p in ps}:          # nothing else lives inside the range
                self.early = (a, b)
                order = {l.name: i for i, l in enumerate(net.layers)}
                self.trigger = min((r.name for r in recs), key=lambda n: order[n])   # its backward runs last
        self.rest = [(0, total)] if self.early is None else [(0, self.early[0]), (self.early[1], total)]
        self.rest = [(a, b) for a, b in self.rest if b > a]
        self.handles = []
        self.flat = flat

    def ranges(self):
        return ([self.early] if self.early else []) + self.rest

    def begin(self):
        self.handles = []

    def on_layer(self, name):
        if name == self.trigger:
            import torch.distributed as dist
            a, b = self.early
            self.handles.append(dist.all_reduce(self.flat.der[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        import torch.distributed as dist
        if self.early is not None and not self.handles:       # trigger layer absent from this pass
            self.rest_all = [self.early] + self.rest
        else:
            self.rest_all = self.rest
        for a, b in self.rest_all:
            dist.all_reduce(self.flat.der[a:b], op=dist.ReduceOp.SUM)
        for h in self.handles:
            h.wait()
        self.handles = []


def shard_batch(batch, rank, world):
    """cnn_train_dag: batch(labindex:numlabs:end) -- interleaved shards."""
    return list(batch)[rank::world]


def accumulate_gradients(net, opts, lr, global_batch, nworkers=1):
    """accumulateGradients of cnn_train_dag (solver = []): per flat segment one fused launch."""
    flat = net._flat
    for (method, lr_mult, wd_mult), a, b in flat.segments:
        if b == a:
            continue
        if method == "average":
            vl.average_update(flat.val[a:b], flat.der[a:b], lr_mult, nworkers)
        else:
            vl.sgd_update(flat.val[a:b], flat.mom[a:b], flat.der[a:b], lr * lr_mult, opts.momentum,
                          opts.weightDecay * wd_mult, global_batch)


def train_step(net, inputs, opts, epoch=0, parserv=None, global_batch=None, input_events=None):
    """One minibatch of cnn_train_dag's processEpoch in training mode."""
    if net._flat is None:
        net.pack_params()
    net.mode = "normal"
    world = parserv.world if parserv is not None else 1
    exchange = parserv is not None and (world > 1 or parserv.force)
    buckets = None
    if exchange and parserv.backend == "torch" and parserv.overlap:
        buckets = net.__dict__.get("_grad_buckets")
        if buckets is None or buckets.flat is not net._flat:
            buckets = net.__dict__["_grad_buckets"] = GradBuckets(net)
        buckets.begin()
        net.gradHook = buckets.on_layer
    else:
        net.gradHook = None
    net.eval(inputs, opts.derOutputs, input_events=input_events)
    if buckets is not None:
        buckets.finish()
    elif exchange:
        parserv.allreduce_(net._flat.der)
    lr = float(opts.learningRate[min(epoch, len(opts.learningRate) - 1)])
    accumulate_gradients(net, opts, lr, global_batch or opts.batchSize, world)


def extractStats(stats, net):
    """stats = extractStats(stats, net) -- run_distillation.m:186-207: `.average` of every dagnn.Loss
    block under its output name; for dagnn.ErrorStats the per-class accuracies under the class names,
    their mean as `meanAcc`, and the class population as `<name>Pop`."""
    from . import dagnn
    stats = dict(stats or {})
    for l in net.layers:
        b = l.block
        if not isinstance(b, dagnn.LossBase):
            continue
        if isinstance(b, dagnn.ErrorStats):
            metrics, dist = b.average, b.classDist
            pop = dist / dist.sum() if dist.sum() > 0 else dist
            names = (net.meta.get("classes", {}).get("name") or
                     ["class%d" % (i + 1) for i in range(len(metrics))])
            stats["meanAcc"] = float(np.mean(metrics))
            for name, m in zip(names, metrics):
                stats[name] = float(m)
            for name, q in zip(names, pop):
                stats["%sPop" % name] = float(q)
        else:
            if b.ignoreAverage:
                continue
            stats[l.outputs[0]] = b.average
    return stats


def _reset_losses(net):
    from . import dagnn
    for l in net.layers:
        if isinstance(l.block, dagnn.LossBase):
            l.block.reset()


def process_epoch(net, imdb, getBatch, subset, opts, epoch, mode, parserv=None, extractStatsFn=extractStats):
    """processEpoch of cnn_train_dag [EXT]: one pass over `subset` in minibatches of opts.batchSize;
    each worker evaluates the interleaved shard batch(labindex:numlabs:end)."""
    import time
    world = parserv.world if parserv is not None else 1
    rank = parserv.rank if parserv is not None else 0
    _reset_losses(net)
    subset = list(subset)
    t0 = time.perf_counter()
    num = 0
    inflight = []
    for t in range(0, len(subset), opts.batchSize):
        batch = subset[t:t + opts.batchSize]
        shard = shard_batch(batch, rank, world)
        if not shard:
            continue
        inputs = getBatch(imdb, shard)
        if mode == "train":
            train_step(net, inputs, opts, epoch, parserv, len(batch))
        else:
            net.mode = "test"
            net.eval(inputs)
        num += len(batch)
        # bounded run-ahead (two minibatches): a full HIP queue stalls launches for milliseconds
        ev = torch.cuda.Event()
        ev.record()
        inflight.append(ev)
        if len(inflight) > 2:
            inflight.pop(0).synchronize()
    stats = extractStatsFn({}, net)   # the only host synchronisation of the pass
    stats["num"] = num
    stats["time"] = time.perf_counter() - t0
    return stats


def cnn_train_dag(net, imdb, getBatch, learningRate=None, batchSize=64, numEpochs=300, train=None, val=None,
                  cont=True, expDir=None, epochSize=float("inf"), parameterServer=None, extractStatsFn=extractStats,
                  momentum=0.9, weightDecay=5e-4, derOutputs=("objective", 1), randomSeed=0, verbose=False):
    """[net, info] = cnn_train_dag(net, imdb, getBatch, 'learningRate', ..., 'batchSize', ..., 'numEpochs', ...,
    'train', ..., 'val', ..., 'continue', ..., 'expDir', ..., 'epochSize', ..., 'parameterServer', ...,
    'extractStatsFn', ...) -- the MatConvNet driver [EXT] as run_distillation.m:170-182 calls it.

    Per epoch: shuffle `train` with the epoch's seed, keep the first `epochSize` samples (the reference's
    "mini-epochs"), one training pass, one validation pass in test mode, then a checkpoint
    `net-epoch-<n>.pt` in `expDir` (flat parameters + momentum + info) from which `cont` resumes.
    `gpus` is implicit: one process per GPU (torchrun), the process group gives the worker count."""
    import os
    opts = TrainOpts(learningRate=learningRate, momentum=momentum, weightDecay=weightDecay, batchSize=batchSize,
                     derOutputs=derOutputs)
    if len(opts.learningRate) < numEpochs:   # MatConvNet indexes min(epoch, numel(learningRate))
        pass
    parserv = parameterServer if isinstance(parameterServer, ParameterServer) else ParameterServer("torch")
    parserv.start()
    if net._flat is None:
        net.pack_params()
    train = list(train if train is not None else [])
    val = list(val if val is not None else [])
    info = {"train": [], "val": []}
    start = 0
    path = (lambda e: os.path.join(expDir, "net-epoch-%d.pt" % e)) if expDir else None
    if expDir and parserv.rank == 0:
        os.makedirs(expDir, exist_ok=True)
    if cont and expDir:
        done = [e for e in range(1, numEpochs + 1) if os.path.exists(path(e))]
        if done:
            start = max(done)
            ck = torch.load(path(start), map_location=net.device, weights_only=False)
            net._flat.val.copy_(ck["val"])
            net._flat.mom.copy_(ck["mom"])
            info = ck["info"]
    for epoch in range(start, numEpochs):
        rng = np.random.default_rng(epoch + 1 + randomSeed)       # rng(epoch + opts.randomSeed)
        order = [train[i] for i in rng.permutation(len(train))]
        if epochSize < len(order):
            order = order[:int(epochSize)]
        info["train"].append(process_epoch(net, imdb, getBatch, order, opts, epoch, "train", parserv, extractStatsFn))
        info["val"].append(process_epoch(net, imdb, getBatch, val, opts, epoch, "val", parserv, extractStatsFn))
        if verbose and parserv.rank == 0:
            print("epoch %d: train %s | val %s" % (epoch + 1, info["train"][-1], info["val"][-1]), flush=True)
        if path and parserv.rank == 0:
            torch.save({"val": net._flat.val, "mom": net._flat.mom, "info": info, "epoch": epoch + 1}, path(epoch + 1))
    parserv.stop()
    return net, info
