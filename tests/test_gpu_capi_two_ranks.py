"""GPU, world_size 2 on TWO MI355X: the library's own communicator (xm_comm_init + xm_parserv_push / xm_parserv_sync --
the ParameterServer a MATLAB spmd host binds, run_distillation.m:88,181) carries the gradient exchange of the PRODUCT's
data-parallel step, one rank per device over RCCL / xGMI.  Same scenario and oracle-side reference as
tests/test_dp_train_step_gloo.py (interleaved shards, per-worker bnorm statistics, ragged minibatch 3 + 2, a tail
minibatch with an empty shard).  Skipped on a one-GPU box (RCCL refuses two ranks on one device).

ONE RCCL communicator per rank: the torch process group of these tests is the control plane only
(rendezvous store, barriers) and runs on gloo; the library's communicator is the only RCCL communicator of a process.

On a one-GPU box `test_capi_live_communicator_per_process_one_gpu` runs instead: two processes share the device, EACH
with a live one-rank communicator of the library (xm_debug_comm_force_single) that carries every bucket of the
overlapped exchange -- xm_parserv_push on the producer stream, the collective on the communicator's own stream, the
event hand-off back in xm_parserv_sync -- and the cross-process sum of each pushed range then goes over gloo.  Same
reference, same tolerance: what it cannot show is RCCL moving bytes between two devices."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import test_dp_train_step_gloo as T

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, outdir, one_gpu=False):
    import ctypes as C
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0 if one_gpu else rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only: no RCCL communicator
    from mcncrossmodalemotions_amd import _lib, train, vl, zoo
    if not one_gpu:
        ps = train.ParameterServer.start_agreed("rccl-capi")          # communicator first (xmodal.h "CALL ORDER")
        assert ps.backend == "rccl-capi" and ps.world == world and ps.comm_count() == world and ps.rccl_count() == world
    else:
        class LivePerProcess(train.ParameterServer):
            """the library's communicator with ONE rank in every process (two ranks cannot share a device) + the sum
            over the processes through gloo behind it"""
            def start(self):
                L = _lib.load()
                self.world, self.rank = dist.get_world_size(), dist.get_rank()
                buf = (C.c_char * 128)()
                _lib.check(L.xm_debug_comm_force_single(1))
                _lib.check(L.xm_comm_unique_id(buf))
                _lib.check(L.xm_comm_init(C.c_char_p(bytes(buf)), 0, 1))
                self._started, self._parts = True, []

            def push(self, part):
                if part.numel():
                    super().push(part)                    # xm_parserv_push: one-rank all-reduce on the comm stream
                    self._parts.append(part)

            def sync(self):
                super().sync()                            # xm_parserv_sync: the current stream waits for the pushes
                for part in self._parts:
                    dist.all_reduce(part)                 # gloo, through the host (synchronises the current stream)
                self._parts = []
        ps = LivePerProcess("rccl-capi")
        ps.start()
        assert ps.comm_count() == 1 and ps.rccl_count() == 1
    data, lgo, lab = T._dataset()

    def getBatch(imdb, idx):
        idx = [int(i) for i in idx]
        return ["data", vl.from_numpy(data[..., idx]), "logitTarget", vl.from_numpy(lgo[..., idx]),
                "maxLabel", vl.from_numpy(lab[..., idx])]

    net = zoo.emoVoxZoo(numSeconds=T.W / 100.0, width_mult=T.WIDTH, seed=3)
    net.pack_params()
    net.wgradStream = torch.cuda.Stream()
    net._grad_buckets = train.GradBuckets(net, target_bytes=2048)      # many buckets, pushed along the backward pass
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
    _lib.load().xm_debug_comm_force_single(0)
    dist.destroy_process_group()


def _check(tmp_path):
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


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one RCCL rank per device)")
def test_capi_parameter_server_two_ranks(gpu, tmp_path):
    world, port = 2, T._free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    _check(tmp_path)


def test_capi_live_communicator_per_process_one_gpu(gpu, tmp_path):
    world, port = 2, T._free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True, start_method="spawn")
    _check(tmp_path)
