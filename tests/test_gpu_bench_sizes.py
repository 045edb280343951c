"""GPU parity at the sizes bench.py actually TIMES (round-4 review, item 3): the tile / split-K / hybrid / eight-wave /
patch-kernel selections a size triggers run in a whole net only at that size, so the numeric check has to be there too.

    (a) config-5 joint step at its 64-pair shard   SE-ResNet-50 train_step (softmaxlog head) + student train_step on the
                                                    teacher's logits, side stream on (ferplus_baselines.m:140-141,
                                                    run_distillation.m:170-182) -- tests/test_gpu_composed_steps.py has it at 16
    (b) north_star's 256-pair step                  SE-ResNet-50 logits on 256 faces (128 x the two fixture faces: every
                                                    logit known), then the student step on 256 spectrograms -- the only
                                                    place the eight-wave 128 x 128 configuration, the >= 1024-tile rule and
                                                    the large-launch patch kernels fire inside a net
    (c) the reference's REAL default shape          numSeconds = 4 (run_distillation.m:74): 512 x 400 spectrograms,
                                                    pool6 = 1 x 11 (emoVoxZoo.m:258-259), batch 64 (:75)

The oracle (oracle/graphs.py, fp64 accumulate + its fp32 path for the floor) runs on the GPU box's host cores: about a
minute per case.  Method and tolerances are those of tests/test_gpu_nets_full.py (1e-4 forward; decisions identified;
every sampled derivative within 1e-4 of the largest entry + 4 x the reference arithmetic's own fp32 deviation)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import graphs as G
from oracle import oracle as O
from test_gpu_composed_steps import check_updated, updated_reference
from test_gpu_nets_full import GateRecorder, check_derivatives, check_gates, close, inject

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def Z():
    return np.load(os.path.join(HERE, "golden", "nets_full.npz"))


@pytest.fixture(scope="module")
def M():
    spec = importlib.util.spec_from_file_location("make_golden_nets", os.path.join(HERE, "golden", "make_golden_nets.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _host_memory_gb():
    """memory the oracle may use on this host: MemAvailable, capped by the cgroup's limit (v1 / v2) minus its usage"""
    avail = 1e30
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = float(line.split()[1]) * 1024
    except OSError:
        pass
    for lim, use in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            l_, u_ = open(lim).read().strip(), open(use).read().strip()
            if l_ != "max":
                avail = min(avail, float(l_) - float(u_))
        except (OSError, ValueError):
            pass
    return avail / 2 ** 30


def _student_step_against_oracle(Z, monkeypatch, N, W, data, targets, what, side_stream=True, events=None):
    """one train_step of the full-width student on `data` with teacher logits `targets` (device mats) -> every check"""
    import torch
    from mcncrossmodalemotions_amd import train, vl, zoo
    tl, ml, lgo = targets
    student = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=W / 100.0, numOutputs=8)
    gs = G.vggvox_student(W)
    Ps = G.perturb_bn(G.make_params(gs, 200), gs, 201)
    inject(student, Ps)
    student.pack_params()
    if side_stream:
        student.wgradStream = torch.cuda.Stream()
    student.vars["prediction"].precious = True
    opts = train.TrainOpts(batchSize=N)
    lr = float(opts.learningRate[0])
    rec = GateRecorder(student, monkeypatch)
    train.train_step(student, ["data", vl.from_numpy(data), "logitTarget", tl, "maxLabel", ml], opts, 0, None, N,
                     input_events=events)
    torch.cuda.synchronize()
    lab = O.F(lgo.reshape(8, N).argmax(0).reshape(1, 1, 1, N) + 1)
    assert np.array_equal(vl.to_numpy(ml).ravel(), lab.ravel()), "maxLabel of the teacher's logits"
    ins = {"data": data, "logitTarget": lgo, "maxLabel": lab}
    V = G.forward(gs, ins, Ps, mode="normal", acc64=True)
    close(vl.to_numpy(student.vars["prediction"].value), V["prediction"], 1e-4, what + " prediction")
    close(vl.to_numpy(student.vars["objective"].value).ravel()[0], V["objective"], 1e-4, what + " objective")
    close(vl.to_numpy(student.vars["classerror"].value).ravel()[0], V["classerror"], 0, what + " classerror")
    gates = rec.gates(gs)
    total, flips = check_gates(gs, V, gates)
    _, DPm = G.backward(gs, V, {"objective": np.float32(1)}, Ps, mode="normal", acc64=True, gates=gates)
    del V
    V32 = G.forward(gs, ins, Ps, mode="normal", acc64=False)
    _, DP32 = G.backward(gs, V32, {"objective": np.float32(1)}, Ps, mode="normal", acc64=False, gates=gates)
    del V32
    worst = check_derivatives(Z, "none", student, DPm, DP32=DP32)
    check_updated(student, updated_reference(student, Ps, DPm, opts, lr, N), Ps, what, DPm, DP32, lr, N)
    print("%s: %d of %d decisions differ from the oracle's (all at the boundary); worst derivative error / allowance = %.3f"
          % (what, flips, total, worst))


def test_joint_step_at_its_shard_of_64_pairs(gpu, Z, monkeypatch):
    """BASELINE config 5's per-GPU shard: 64 pairs, both branches backward, one ParameterServer, side stream."""
    import torch
    from mcncrossmodalemotions_amd import train, vl, zoo
    N, W = 64, 300
    teacher = zoo.ferPlusZoo("senet50-ferplus")
    teacher.removeLayer("top1error")                      # bench.py joint
    gt = [l for l in G.resnet50_teacher(se=True, heads=True) if l.name != "top1error"]
    Pt = G.perturb_bn(G.make_params(gt, 300), gt, 301)
    inject(teacher, Pt)
    teacher.pack_params()
    side = torch.cuda.Stream()
    teacher.wgradStream = side
    parserv = train.ParameterServer("torch")              # one worker: inactive, as in bench.py at N = 1
    parserv.start()
    opts = train.TrainOpts(batchSize=N)
    lr = float(opts.learningRate[0])
    faces = G.face_batch(N, 5)
    flab = np.asfortranarray(np.random.default_rng(105).integers(1, 9, (1, 1, 1, N)).astype(np.float32))
    teacher.vars["prediction"].precious = True
    rec_t = GateRecorder(teacher, monkeypatch)
    train.train_step(teacher, ["data", vl.from_numpy(faces), "label", vl.from_numpy(flab)], opts, 0, parserv, N)
    torch.cuda.synchronize()
    gates_t = rec_t.gates(gt)
    tl = teacher.vars["prediction"].value
    ml = vl.max_label(tl)
    Vt = G.forward(gt, {"data": faces, "label": flab}, Pt, mode="normal", acc64=True)
    close(vl.to_numpy(tl), Vt["prediction"], 1e-4, "teacher prediction")
    close(vl.to_numpy(teacher.vars["objective"].value).ravel()[0], Vt["objective"], 1e-5, "teacher objective")
    total, flips = check_gates(gt, Vt, gates_t)
    _, DPt = G.backward(gt, Vt, {"objective": np.float32(1)}, Pt, mode="normal", acc64=True, gates=gates_t)
    lgo = np.asfortranarray(Vt["prediction"].astype(np.float32))
    del Vt
    Vt32 = G.forward(gt, {"data": faces, "label": flab}, Pt, mode="normal", acc64=False)
    _, DPt32 = G.backward(gt, Vt32, {"objective": np.float32(1)}, Pt, mode="normal", acc64=False, gates=gates_t)
    del Vt32
    worst_t = check_derivatives(Z, "none", teacher, DPt, DP32=DPt32)
    check_updated(teacher, updated_reference(teacher, Pt, DPt, opts, lr, N), Pt, "teacher", DPt, DPt32, lr, N)
    print("joint step, teacher branch at %d faces: %d of %d decisions differ (all at the boundary); worst derivative "
          "error / allowance = %.3f" % (N, flips, total, worst_t))
    data, _, _ = G.spectrogram_batch(N, W, 7)
    _student_step_against_oracle(Z, monkeypatch, N, W, data, (tl, ml, lgo), "joint step, student branch at 64")


def test_north_star_step_at_256_pairs(gpu, Z, M, monkeypatch):
    """north_star's configuration on one GPU: SE-ResNet-50 teacher forward on 256 faces (on its own stream, as bench.py
    runs it) -> logits -> student train_step on 256 spectrograms 512 x 300."""
    import torch
    from mcncrossmodalemotions_amd import vl, zoo
    from test_gpu_nets_full import build_teacher
    N, W = 256, 300
    mem = _host_memory_gb()
    if mem < 96:
        # (round-5 review: no silent skip) the oracle's pass over 256 spectrograms keeps ~50 GB of activations.  A smaller
        # host still runs the step at 128 pairs -- the eight-wave configuration, the >= 1024-tile rule and conv_dgrad_s2_kernel
        # fire there as well (conv3: 1530 tiles, conv2's dgrad: 4736 blocks >= 6 rounds); only the >= 4096-column rule of the
        # 3 x 3 patch filter derivative needs the full 256 -- and says so.
        import warnings
        N = 128
        warnings.warn("test_north_star_step_at_256_pairs runs at 128 pairs: host has %.0f GB for the oracle (needs 96)" % mem)
        assert mem >= 40, "host memory %.0f GB: not even the 128-pair oracle pass fits -- the north_star step is UNTESTED here" % mem
    teacher = build_teacher(Z, M, "se50", True, 300)
    teacher.move("gpu")
    teacher.mode = "test"
    teacher.vars["prediction"].precious = True
    faces = np.asfortranarray(np.tile(G.face_batch(M.TEACHER_N, 3), (1, 1, 1, N // M.TEACHER_N)))
    tstream = torch.cuda.Stream()
    with torch.cuda.stream(tstream):
        teacher.eval(["data", vl.from_numpy(faces)])
        tl = teacher.vars["prediction"].value
        ml = vl.max_label(tl)
        ev = torch.cuda.Event()
        ev.record(tstream)
    torch.cuda.synchronize()
    del faces
    ref_logits = np.asfortranarray(np.tile(Z["se50_logits"], (1, 1, 1, N // M.TEACHER_N)))
    close(vl.to_numpy(tl), ref_logits, 1e-4, "SE-ResNet-50 logits at 256 faces")
    teacher = None
    data, _, _ = G.spectrogram_batch(N, W, 256)
    _student_step_against_oracle(Z, monkeypatch, N, W, data, (tl, ml, ref_logits.astype(np.float32)),
                                 "north_star step, student at 256", events={"logitTarget": ev, "maxLabel": ev})


def test_student_step_at_the_reference_default_width_400(gpu, Z, monkeypatch):
    """run_distillation.m:74-75: numSeconds = 4 -> 512 x 400 spectrograms, pool6 = 1 x 11, batch 64."""
    from mcncrossmodalemotions_amd import vl, zoo
    N, W = 64, 400
    net = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=4)
    assert list(net.getLayer("pool6").block.poolSize) == [1, 11]          # emoVoxZoo.m:258-259
    data, lgo, lab = G.spectrogram_batch(N, W, 400)
    tl = vl.from_numpy(lgo)
    ml = vl.max_label(tl)
    _student_step_against_oracle(Z, monkeypatch, N, W, data, (tl, ml, np.asfortranarray(lgo.astype(np.float32))),
                                 "student step at W = 400, batch 64")
