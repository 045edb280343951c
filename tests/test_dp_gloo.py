"""CPU, world_size 2 over gloo: the data-parallel contract of cnn_train_dag + ParameterServer
 -- interleaved shards, per-worker gradients summed by the all-reduce, update
divided by the GLOBAL batch size -- exercised through the product's own ParameterServer /
shard_batch code, with the oracle standing in for the per-worker compute."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net_grads(x, lgo, f1, b1, f2, b2):
    """conv3x3 -> relu -> global avg pool -> fc (1x1 conv) -> distillation loss; returns grads."""
    y1 = O.vl_nnconv(x, f1, b1, pad=1, acc64=True)
    r1 = O.vl_nnrelu(y1)
    p1 = O.vl_nnpool(r1, r1.shape[:2], method="avg")
    pred = O.vl_nnconv(p1, f2, b2, acc64=True)
    dpred = O.vl_nnsoftmaxceloss(pred, lgo, np.ones(1, np.float32), temperature=2, logit_targets=True)
    dp1, df2, db2 = O.vl_nnconv(p1, f2, b2, dpred, acc64=True)
    dr1 = O.vl_nnpool(r1, r1.shape[:2], dp1, method="avg")
    dy1 = O.vl_nnrelu(y1, dr1)
    _, df1, db1 = O.vl_nnconv(x, f1, b1, dy1, pad=1, acc64=True, no_der_data=True)
    return np.concatenate([df1.ravel(order="F"), db1.ravel(), df2.ravel(order="F"), db2.ravel()])


def _data():
    rng = np.random.default_rng(0)
    N = 6
    x = O.F(rng.standard_normal((8, 7, 3, N)))
    lgo = O.F(rng.standard_normal((1, 1, 8, N)) * 2)
    f1, b1 = O.F(rng.standard_normal((3, 3, 3, 5)) * 0.3), O.F(rng.standard_normal(5) * 0.1)
    f2, b2 = O.F(rng.standard_normal((1, 1, 5, 8)) * 0.3), O.F(np.zeros(8))
    return x, lgo, f1, b1, f2, b2


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcncrossmodalemotions_amd import train
    x, lgo, f1, b1, f2, b2 = _data()
    idx = train.shard_batch(range(x.shape[3]), rank, world)  # batch(labindex:numlabs:end)
    g = _net_grads(x[..., idx], lgo[..., idx], f1, b1, f2, b2)
    flat = torch.from_numpy(g.astype(np.float32))
    ps = train.ParameterServer("torch")
    ps.start()
    assert ps.world == world and ps.rank == rank
    ps.allreduce_(flat)  # push / sync / pull
    ps.stop()
    if rank == 0:
        np.save(out, flat.numpy())
    dist.destroy_process_group()


def test_sharded_gradient_sum_equals_full_batch(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "g.npy")
    mp.start_processes(_worker, args=(world, port, out), nprocs=world, join=True, start_method="spawn")
    summed = np.load(out)
    full = _net_grads(*_data())
    # loss is a SUM over samples and there is no BN here, so the shard sums equal the full gradient
    assert np.abs(summed - full).max() <= 1e-5 * max(1.0, np.abs(full).max())


def test_shard_batch_is_interleaved():
    from mcncrossmodalemotions_amd import train
    assert train.shard_batch(range(8), 0, 2) == [0, 2, 4, 6] and train.shard_batch(range(8), 1, 2) == [1, 3, 5, 7]
    assert sorted(sum((train.shard_batch(range(10), r, 4) for r in range(4)), [])) == list(range(10))


def _agree_worker(rank, world, port, out, fail="init@1"):
    """ParameterServer.start_agreed over 2 gloo workers: the communicator's id travels through the store, and when ONE
    worker cannot start the library's communicator every worker ends up on torch.distributed (none is left waiting in a
    collective the others never enter)."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mcncrossmodalemotions_amd import _lib, train
    L = _lib.load()
    ids = []
    real_id, real_init = L.xm_comm_unique_id, L.xm_comm_init

    class FakeLib:
        """the library with a communicator that comes up on rank 0 only (no GPU / RCCL in this test)"""
        def __getattr__(self, name):
            return getattr(L, name)

        def xm_comm_unique_id(self, buf):
            for i in range(128):
                buf[i] = bytes([65 + (i % 26)])
            return 0

        def xm_comm_init(self, raw, r, w):
            ids.append(bytes(raw.value if hasattr(raw, "value") else raw)[:16])
            return 0 if r == 0 else 3        # XM_EHIP on rank 1

        def xm_comm_destroy(self):
            return 0

        def xm_debug_comm_force_single(self, on):
            return 0

        def xm_last_error(self):
            return b"simulated communicator failure"
    fake = FakeLib()
    _lib._lib = fake                               # what _lib.load() hands out from now on
    real_load = _lib.load
    if fail.startswith("load@"):
        # the library cannot even be loaded on ONE worker: it fails before it has touched any store key
        bad = int(fail[5:])

        def load():
            if rank == bad:
                raise OSError("simulated: libxmodal_hip.so cannot be loaded")
            return fake
        _lib.load = load

        def blocking_init(raw, r, w):
            # like ncclCommInitRank: returns only when EVERY rank has called it (round-5 advisor: with the immediate fake
            # the test could not see healthy workers stuck in the bootstrap of a communicator the failed worker never joins)
            from torch.distributed.distributed_c10d import _get_default_store
            import datetime
            st = _get_default_store()
            st.set("fake_init/%d" % r, b"1")
            st.wait(["fake_init/%d" % q for q in range(w)], datetime.timedelta(seconds=20))
            ids.append(b"x")
            return 0
        fake.xm_comm_init = blocking_init
    train.ParameterServer.store_timeout_s = 30.0   # a lost key fails the test instead of hanging it
    try:
        ps = train.ParameterServer.start_agreed("rccl-capi")
        backend = ps.backend
        flat = torch.full((4,), float(rank + 1))
        ps.allreduce_(flat)
        ps.stop()
    finally:
        _lib._lib = L
        _lib.load = real_load
    np.save(out + ".%d.npy" % rank, np.array([1.0 if backend == "torch" else 0.0, float(flat[0]), float(len(ids))]))
    dist.destroy_process_group()
    assert real_id is not None and real_init is not None


def test_start_agreed_falls_back_on_every_worker(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "agree")
    mp.start_processes(_agree_worker, args=(world, port, out), nprocs=world, join=True, start_method="spawn")
    for r in range(world):
        backend_is_torch, summed, inits = np.load(out + ".%d.npy" % r)
        assert backend_is_torch == 1.0, "worker %d stayed on the failed backend" % r
        assert summed == 3.0                      # 1 + 2 through the fallback's all-reduce
        assert inits == 1.0                       # every worker tried the library's communicator exactly once


@pytest.mark.parametrize("bad", [0, 1])
def test_start_agreed_when_one_worker_cannot_load_the_library(tmp_path, bad):
    """round-4 advisor: a worker that fails BEFORE the id exchange (no library) must still meet the others in the
    agreement round -- its own round counter -- and worker 0 must tell the others that no id is coming."""
    import time
    world, port = 2, _free_port()
    out = str(tmp_path / "agree_load")
    t0 = time.time()
    mp.start_processes(_agree_worker, args=(world, port, out, "load@%d" % bad), nprocs=world, join=True,
                       start_method="spawn")
    assert time.time() - t0 < 25.0, "a worker sat in a store timeout"
    for r in range(world):
        backend_is_torch, summed, inits = np.load(out + ".%d.npy" % r)
        assert backend_is_torch == 1.0 and summed == 3.0
