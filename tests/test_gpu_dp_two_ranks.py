"""GPU, world_size 2 on ONE MI355X: two processes share cuda:0 and exchange over gloo (RCCL refuses two ranks on one
device), so that the PRODUCT's data-parallel step runs with the real HIP operators on more than one rank:
train.process_epoch -> train_step -> dagnn.eval (fused plans, side stream) -> GradBuckets -> ParameterServer ->
accumulate_gradients.  Same scenario and same oracle-side reference as tests/test_dp_train_step_gloo.py (interleaved
shards, per-worker bnorm statistics, ragged minibatch 3 + 2, a tail minibatch with an empty shard, merged validation
statistics); what changes is that every operator is the HIP kernel behind the C ABI.
Reference call: run_distillation.m:170-182 (numGpus > 1, parameterServer.method = 'tmove')."""
import os

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests import test_dp_train_step_gloo as T

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcncrossmodalemotions_amd import train, vl, zoo
    data, lgo, lab = T._dataset()

    def getBatch(imdb, idx):
        idx = [int(i) for i in idx]
        return ["data", vl.from_numpy(data[..., idx]), "logitTarget", vl.from_numpy(lgo[..., idx]),
                "maxLabel", vl.from_numpy(lab[..., idx])]

    net = zoo.emoVoxZoo(numSeconds=T.W / 100.0, width_mult=T.WIDTH, seed=3)    # fused plans, device resident
    net.pack_params()
    # side stream for the filter derivatives, made to LAG (a sleep in front of every filter transposition it runs
    # during the forward pass): a bucket pushed from the main stream (conv1's hook) covers derivatives enqueued on
    # the side stream and must wait for them
    net.wgradStream = torch.cuda.Stream()
    real_prep = vl.conv_prepare_backward

    def slow_prep(*a, **k):
        torch.cuda._sleep(4_000_000)
        return real_prep(*a, **k)
    vl.conv_prepare_backward = slow_prep
    ps = train.ParameterServer("torch")
    ps.start()
    assert ps.world == world and ps.comm_count() == world
    # one bucket per filter segment, triggered by its EARLIEST layer: conv1 (no input derivative -> its hook runs
    # on the main stream) pushes the derivatives of conv2 ... fc8, which were enqueued on the lagging side stream
    # (the many-small-buckets plan is covered by the CPU test)
    net._grad_buckets = train.GradBuckets(net, target_bytes=1 << 30)
    assert any(t == "conv1" and b - a > 4096 for a, b, t in net._grad_buckets.buckets)
    opts = train.TrainOpts(learningRate=[1e-2], batchSize=T.BATCH)
    stats = train.process_epoch(net, None, getBatch, list(range(T.N_DATA)), opts, 0, "train", ps)
    vstats = train.process_epoch(net, None, getBatch, list(range(T.N_DATA)), opts, 0, "val", ps)
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, "val_%d.npy" % rank), net._flat.val.cpu().numpy())
    np.save(os.path.join(outdir, "stats_%d.npy" % rank),
            np.array([stats["objective"], stats["num"], vstats["objective"], vstats["classerror"], vstats["num"]]))
    offs = {k: (int(p._flat_off), int(p.value.numel())) for k, p in net.params.items()}
    np.save(os.path.join(outdir, "offs_%d.npy" % rank), np.array([(k,) + v for k, v in offs.items()], dtype=object),
            allow_pickle=True)
    ps.stop()
    dist.destroy_process_group()


def test_product_train_step_two_ranks_on_one_gpu(gpu, tmp_path):
    world, port = 2, T._free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    v0, v1 = np.load(tmp_path / "val_0.npy"), np.load(tmp_path / "val_1.npy")
    assert np.array_equal(v0, v1), "workers diverged"
    s0, s1 = np.load(tmp_path / "stats_0.npy"), np.load(tmp_path / "stats_1.npy")
    assert s0[1] == T.N_DATA and s0[4] == T.N_DATA
    assert np.allclose(s0, s1, rtol=1e-6, atol=1e-7), (s0, s1)
    _, P, vobj, verr = T._reference()
    for k, off, n in np.load(tmp_path / "offs_0.npy", allow_pickle=True):
        got = v0[int(off):int(off) + int(n)]
        ref = np.asarray(P[k], np.float32).ravel(order="F")
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= 1e-4 * scale, (k, float(np.abs(got - ref).max()))
    assert abs(s0[2] - vobj) <= 1e-4 * max(1.0, abs(vobj)), (s0[2], vobj)
    assert abs(s0[3] - verr) <= 1e-6, (s0[3], verr)
