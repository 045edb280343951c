"""bench.py --gpus N is the command the driver runs: it must produce an N-rank line by itself, or refuse.

cnn_train_dag starts numel(opts.gpus) workers (run_distillation.m:71,77,88,179-181); here one process per GPU under
torch.distributed.run, which `bench.py --gpus N` execs itself into when no launcher is around it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "XM_DEBUG_DIST")}
    env.update(kw)
    return env


def test_refuses_to_run_fewer_ranks_than_asked_for():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU node would really launch the ranks")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(), timeout=300)
    assert r.returncode != 0
    assert b"refusing to run fewer ranks" in r.stderr and not r.stdout.strip()


def test_launcher_world_must_match_the_flag():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and b"must agree" in r.stderr


@pytest.mark.gpu
def test_gpus_flag_starts_the_ranks_itself(gpu):
    """`python bench.py --gpus 2` with no launcher: two ranks (XM_DEBUG_DIST=gloo0: both on the one GPU of the test
    box, exchange over gloo -- a functional run of the whole N > 1 path; the throughput means nothing) and ONE json line
    that says so: n_gpus 2, world 2, dp2, global batch = 2 shards; rccl_ranks is null because no RCCL communicator
    carried that exchange (it is reported from ncclCommCount only)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline", "--no-roofline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=_env(XM_DEBUG_DIST="gloo0"), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world"] == 2 and d["rccl_ranks"] is None and d["steps"] == 3
    assert d["control_group"] == "gloo"
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]
    assert d["value"] > 0 and d["scaling"] == "weak"


@pytest.mark.gpu
def test_gpus_8_functional_run_on_one_gpu(gpu):
    """The driver's 8-GPU command line, `python bench.py --gpus 8`, as far as a one-GPU box can take it (round-5 review,
    item 7): eight ranks under torch.distributed.run (XM_DEBUG_DIST=gloo0: all on cuda:0, exchange over gloo), the collective
    settle / step-count agreement of eight workers, eight interleaved shards of BASELINE config 4 (8 x 32 = 256 pairs), the
    bucketed exchange with eight participants, max-over-ranks timing, ONE json line.  What it cannot show is RCCL moving bytes
    over xGMI.  Small per-rank batch so that eight processes share one device comfortably; the throughput means nothing."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--per-gpu-batch", "4", "--no-cpu-baseline", "--no-roofline", "--north-star", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(XM_DEBUG_DIST="gloo0"), timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["world"] == 8 and d["rccl_ranks"] is None and d["steps"] == 2
    assert d["control_group"] == "gloo" and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 8 * d["config"]["per_gpu_batch"] == 32
    assert d["value"] > 0


def test_imdb_windows_are_time2idx_ranges():
    """bench.py --imdb-windows: the ragged ranges follow getBatchEmoVoxCeleb.m:137-158 -- contiguous 1-based inclusive row
    ranges, a 3 s window of a track with a logit row every 6th frame at 25 fps spans 13-14 rows (time2idx, :210-214) unless
    the track ends first, never an empty window, the teacher's frame count is the sum of the windows."""
    sys.path.insert(0, ROOT)
    import bench
    w = bench.imdb_windows(256, 300, 0)
    cnt = [b - a + 1 for a, b in zip(w["first"], w["last"])]
    assert w["first"][0] == 1 and all(a == b + 1 for a, b in zip(w["first"][1:], w["last"][:-1]))
    assert min(cnt) >= 1 and max(cnt) <= 14 and w["frames"] == sum(cnt) == w["last"][-1]
    assert sum(c in (13, 14) for c in cnt) > 200 and (w["min"], w["max"]) == (min(cnt), max(cnt))
    assert bench.imdb_windows(256, 300, 0) == w and bench.imdb_windows(256, 300, 1) != w


@pytest.mark.gpu
def test_imdb_windows_line_runs(gpu):
    """SURVEY 8f row 1 on the bench line with ragged windows: the teacher runs on exactly the windows' frames, the per-pair
    max goes through xm_aggregate_logits, the student steps on it; the line says what it ran."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--imdb-windows", "1", "--teacher", "senet50", "--steps",
                        "2", "--warmup", "1", "--per-gpu-batch", "8", "--no-cpu-baseline", "--no-roofline", "--north-star", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert "imdb windows: " in d["config"]["workload"] and d["config"]["per_gpu_batch"] == 8 and d["value"] > 0
