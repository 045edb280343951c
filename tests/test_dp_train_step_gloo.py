"""CPU, world_size 2 over gloo: the PRODUCT's own data-parallel training code -- train.process_epoch ->
train_step -> dagnn.eval -> GradBuckets -> ParameterServer -> accumulate_gradients -- run end to end on a tiny
VGGVox-shaped student, with tests/cpu_standin.py standing in for the HIP operators (oracle, fp64 accumulate).

What is pinned (reference call run_distillation.m:170-182):
  * interleaved shards batch(labindex:numlabs:end), per-worker bnorm statistics (no sync-BN);
  * gradients summed over workers, update divided by the GLOBAL batch size;
  * bnorm moments averaged with weights shard-size / global-batch (a ragged last minibatch: 3 + 2 samples);
  * a tail minibatch smaller than the worker count (1 sample, rank 1's shard is empty) does not hang: the idle
    worker joins every collective with a zero contribution;
  * every element of the flat derivative buffer is pushed exactly once per step (bucket plan with many buckets);
  * validation statistics are merged over the workers before extractStats reads them;
  * all workers end with identical parameters, equal to an oracle-side reference of the same arithmetic."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from oracle import oracle as O
from oracle import oracle_net

N_DATA, BATCH, W, WIDTH = 6, 5, 100, 1.0 / 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dataset():
    rng = np.random.default_rng(7)
    data = O.spec_rownorm(O.F(np.abs(rng.standard_normal((512, W, 1, N_DATA)))))
    lgo = O.F(rng.standard_normal((1, 1, 8, N_DATA)) * 3)
    lab = O.F(lgo.reshape(8, N_DATA).argmax(0).reshape(1, 1, 1, N_DATA) + 1)
    return data, lgo, lab


def _make_net():
    from mcncrossmodalemotions_amd import zoo
    net = zoo.emoVoxZoo(numSeconds=W / 100.0, width_mult=WIDTH, seed=3)
    net.fuse = False          # the stand-in implements the plain operators only
    return net


def _worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import cpu_standin
    cpu_standin.install()
    from mcncrossmodalemotions_amd import train, vl
    data, lgo, lab = _dataset()

    def getBatch(imdb, idx):
        idx = [int(i) for i in idx]
        return ["data", vl.from_numpy(data[..., idx]), "logitTarget", vl.from_numpy(lgo[..., idx]),
                "maxLabel", vl.from_numpy(lab[..., idx])]

    net = _make_net()
    net.pack_params()
    ps = train.ParameterServer("torch")
    ps.start()
    assert ps.world == world and ps.rank == rank and ps.comm_count() == world
    bk = net._grad_buckets = train.GradBuckets(net, target_bytes=2048)     # many buckets
    bk.log = []
    assert len(bk.buckets) >= 3
    opts = train.TrainOpts(learningRate=[1e-2], batchSize=BATCH)
    stats = train.process_epoch(net, None, getBatch, list(range(N_DATA)), opts, 0, "train", ps)
    # pushes of the FIRST step (the second step runs bucket-less on the idle rank): a partition of the buffer
    total = int(net._flat.der.numel())
    first = sorted(bk.log[:len(bk.ranges())])
    assert first[0][0] == 0 and first[-1][1] == total and all(a[1] == b[0] for a, b in zip(first, first[1:])), first
    vstats = train.process_epoch(net, None, getBatch, list(range(N_DATA)), opts, 0, "val", ps)
    np.save(os.path.join(outdir, "val_%d.npy" % rank), net._flat.val.numpy())
    np.save(os.path.join(outdir, "stats_%d.npy" % rank),
            np.array([stats["objective"], stats["num"], vstats["objective"], vstats["classerror"], vstats["num"]]))
    ps.stop()
    dist.destroy_process_group()


def _reference():
    """the same two minibatches by hand: per-shard oracle passes, summed derivatives, weighted moments"""
    net = _make_net()
    data, lgo, lab = _dataset()
    P = oracle_net.host_params(net)
    mom = {k: np.zeros_like(v) for k, v in P.items()}
    for batch in ([0, 1, 2, 3, 4], [5]):
        B = len(batch)
        G, Msum = None, {}
        for r in range(2):
            idx = batch[r::2]
            if not idx:
                continue
            ins = {"data": data[..., idx], "logitTarget": lgo[..., idx], "maxLabel": lab[..., idx]}
            V = oracle_net.forward(net, ins, P, mode="normal")
            _, DP = oracle_net.backward(net, V, {"objective": np.float32(1)}, P, mode="normal")
            for k, d in DP.items():
                if net.params[k].trainMethod == "average":
                    Msum[k] = Msum.get(k, 0) + np.asarray(d, np.float64) * len(idx)
            G = DP if G is None else {k: G[k] + DP[k] for k in DP}
        for k, p in net.params.items():
            if p.trainMethod == "average":
                P[k] = ((1 - p.learningRate) * P[k] + p.learningRate * (Msum[k] / B)).astype(np.float32)
            else:
                P[k], mom[k] = O.sgd_update(P[k], mom[k], np.reshape(G[k], P[k].shape, order="F"),
                                            1e-2 * p.learningRate, 0.9, 5e-4 * p.weightDecay, B)
    # validation pass over all 6 samples in test mode with the updated parameters
    net.mode = "test"
    V = oracle_net.forward(net, {"data": data, "logitTarget": lgo, "maxLabel": lab}, P, mode="test")
    return net, P, float(V["objective"]) / N_DATA, float(V["classerror"]) / N_DATA


def test_product_train_step_world2(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    v0, v1 = np.load(tmp_path / "val_0.npy"), np.load(tmp_path / "val_1.npy")
    assert np.array_equal(v0, v1), "workers diverged"
    s0, s1 = np.load(tmp_path / "stats_0.npy"), np.load(tmp_path / "stats_1.npy")
    assert s0[1] == N_DATA and s0[4] == N_DATA
    assert np.allclose(s0, s1, rtol=1e-6, atol=1e-7), (s0, s1)      # merged statistics: the same on every worker
    net, P, vobj, verr = _reference()
    # flat layout of the product: rebuild it on a scratch net to map names -> offsets (CPU stand-in for move())
    from tests import cpu_standin
    undo = cpu_standin.install()
    try:
        scratch = _make_net()
        scratch.pack_params()
        for k, p in scratch.params.items():
            n = int(p.value.numel())
            got = v0[p._flat_off:p._flat_off + n]
            ref = np.asarray(P[k], np.float32).ravel(order="F")
            scale = max(1.0, float(np.abs(ref).max()))
            assert np.abs(got - ref).max() <= 2e-5 * scale, (k, float(np.abs(got - ref).max()))
    finally:
        undo()
    assert abs(s0[2] - vobj) <= 1e-5 * max(1.0, abs(vobj)), (s0[2], vobj)
    assert abs(s0[3] - verr) <= 1e-6, (s0[3], verr)
