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

    def allreduce_(self, flat):
        if self.world == 1:
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
    net.eval(inputs, opts.derOutputs, input_events=input_events)
    world = parserv.world if parserv is not None else 1
    if parserv is not None and world > 1:
        parserv.allreduce_(net._flat.der)
    lr = float(opts.learningRate[min(epoch, len(opts.learningRate) - 1)])
    accumulate_gradients(net, opts, lr, global_batch or opts.batchSize, world)


def extractStats(net, num_samples):
    """run_distillation.m:186-207: average of every dagnn.Loss output (per sample)."""
    from . import dagnn
    stats = {}
    for l in net.layers:
        if isinstance(l.block, dagnn.LossBase) and getattr(l.block, "lastValue", None) is not None:
            stats[l.outputs[0]] = float(vl.to_numpy(l.block.lastValue).ravel()[0]) / max(num_samples, 1)
    return stats
