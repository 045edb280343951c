"""GPU parity of the two COMPOSED steps bench.py times, at the sizes it times them (round-3 review: they had no numeric
assertion):

    config 5  joint step   SE-ResNet-50 train_step (softmaxlog head, ferPlusZoo.m:240-249, ferplus_baselines.m:140-141)
                            + student train_step on the teacher's logits, one ParameterServer, wgrad side stream on --
                            exactly bench.py's `joint` branch (run_distillation.m:170-182 for the student half)
    config 4  32-pair shard frozen ResNet-50 forward on 32 faces -> logits -> student train_step on 32 spectrograms
                            (the shapes N = 32 selects: other tile configurations / split-K / hybrid than N = 4 or 64)

The oracle (fp64 accumulate, oracle/graphs.py) runs on the GPU box's host cores; derivatives are the identification
test of tests/test_gpu_nets_full.py (decisions of the HIP pass compared with the oracle's, oracle backward re-run with
them injected, every sampled entry within 1e-4 * max|ref| + 4 x the fp32 floor of the reference's own arithmetic), and
the parameters AFTER the update are compared with oracle.sgd_update / average_update applied to the oracle's
derivatives."""
import os

import numpy as np
import pytest

from oracle import graphs as G
from oracle import oracle as O
from test_gpu_nets_full import GateRecorder, check_derivatives, check_gates, close, inject

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def Z():
    return np.load(os.path.join(HERE, "golden", "nets_full.npz"))


def updated_reference(net, P, DP, opts, lr, batch):
    """accumulateGradients (cnn_train_dag, solver = []) on the oracle side: momentum starts at zero"""
    out = {}
    for name, p in net.params.items():
        d = np.asarray(DP[name], np.float32).reshape(P[name].shape, order="F")
        if p.trainMethod == "average":
            out[name] = O.average_update(P[name], d, float(p.learningRate), 1.0)
        else:
            out[name], _ = O.sgd_update(P[name], np.zeros_like(P[name]), d, lr * float(p.learningRate), opts.momentum,
                                        opts.weightDecay * float(p.weightDecay), batch)
    return out


def check_updated(net, ref, P, what, DP=None, DP32=None, lr=0.0, batch=1.0):
    """parameters after the step: the CHANGE of every parameter within 1e-3 of the largest change of that tensor (the
    derivative tolerance carried through the update) + the derivative's fp32 floor carried through the update (4 x
    what the reference's own fp32 arithmetic deviates by, as in check_derivatives) + two ulps of the parameter"""
    from mcncrossmodalemotions_amd import vl
    for name in net.params:
        floor = 0.0
        if DP32 is not None and net.params[name].trainMethod != "average":
            dev = float(np.abs(np.asarray(DP32[name], np.float64) - np.asarray(DP[name], np.float64)).max())
            floor = 4.0 * dev * lr * float(net.params[name].learningRate) / batch
        got = vl.to_numpy(net.params[name].value).astype(np.float64)
        want = np.asarray(ref[name], np.float64).reshape(got.shape, order="F")
        old = np.asarray(P[name], np.float64).reshape(got.shape, order="F")
        step = float(np.abs(want - old).max())
        ulp = float(np.abs(want).max()) * 1.2e-7
        err = float(np.abs(got - want).max())
        assert err <= 1e-3 * step + floor + 2 * ulp, "%s %s: updated value off by %.3e (largest change %.3e)" % (what, name, err, step)


def test_joint_step_as_bench_runs_it(gpu, Z, monkeypatch):
    """BASELINE config 5's step at 16 pairs: both branches backward, shared ParameterServer, side stream."""
    import torch
    from mcncrossmodalemotions_amd import train, vl, zoo
    N, W = 16, 300
    teacher = zoo.ferPlusZoo("senet50-ferplus")
    teacher.removeLayer("top1error")                      # bench.py joint
    gt = [l for l in G.resnet50_teacher(se=True, heads=True) if l.name != "top1error"]
    Pt = G.perturb_bn(G.make_params(gt, 300), gt, 301)
    inject(teacher, Pt)
    teacher.pack_params()
    student = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=W / 100.0,
                            numOutputs=8)
    gs = G.vggvox_student(W)
    Ps = G.perturb_bn(G.make_params(gs, 200), gs, 201)
    inject(student, Ps)
    student.pack_params()
    side = torch.cuda.Stream()
    teacher.wgradStream = student.wgradStream = side
    parserv = train.ParameterServer("torch")              # one worker: inactive, as in bench.py at N = 1
    parserv.start()
    opts = train.TrainOpts(batchSize=N)
    lr = float(opts.learningRate[0])

    faces = G.face_batch(N, 5)
    flab = np.asfortranarray(np.random.default_rng(105).integers(1, 9, (1, 1, 1, N)).astype(np.float32))
    data, _, _ = G.spectrogram_batch(N, W, 7)

    teacher.vars["prediction"].precious = True
    student.vars["prediction"].precious = True
    rec_t = GateRecorder(teacher, monkeypatch)
    train.train_step(teacher, ["data", vl.from_numpy(faces), "label", vl.from_numpy(flab)], opts, 0, parserv, N)
    gates_t = rec_t.gates(gt)
    tl = teacher.vars["prediction"].value
    ml = vl.max_label(tl)
    rec_s = GateRecorder(student, monkeypatch)
    train.train_step(student, ["data", vl.from_numpy(data), "logitTarget", tl, "maxLabel", ml], opts, 0, parserv, N)
    torch.cuda.synchronize()
    gates_s = rec_s.gates(gs)

    # ---- teacher branch against the oracle ------------------------------------------------------------------
    Vt = G.forward(gt, {"data": faces, "label": flab}, Pt, mode="normal", acc64=True)
    close(vl.to_numpy(tl), Vt["prediction"], 1e-4, "teacher prediction")
    close(vl.to_numpy(teacher.vars["objective"].value).ravel()[0], Vt["objective"], 1e-5, "teacher objective")
    total, flips = check_gates(gt, Vt, gates_t)
    _, DPt = G.backward(gt, Vt, {"objective": np.float32(1)}, Pt, mode="normal", acc64=True, gates=gates_t)
    Vt32 = G.forward(gt, {"data": faces, "label": flab}, Pt, mode="normal", acc64=False)
    _, DPt32 = G.backward(gt, Vt32, {"objective": np.float32(1)}, Pt, mode="normal", acc64=False, gates=gates_t)
    worst_t = check_derivatives(Z, "none", teacher, DPt, DP32=DPt32)
    check_updated(teacher, updated_reference(teacher, Pt, DPt, opts, lr, N), Pt, "teacher", DPt, DPt32, lr, N)
    print("joint step, teacher branch at %d faces: %d of %d decisions differ (all at the boundary); worst derivative "
          "error / allowance = %.3f" % (N, flips, total, worst_t))

    # ---- student branch: targets = the ORACLE teacher's logits (the composition is what is under test) ---------
    lgo = np.asfortranarray(Vt["prediction"].astype(np.float32))
    lab = O.F(lgo.reshape(8, N).argmax(0).reshape(1, 1, 1, N) + 1)
    assert np.array_equal(vl.to_numpy(ml).ravel(), lab.ravel()), "maxLabel of the HIP teacher's logits"
    ins = {"data": data, "logitTarget": lgo, "maxLabel": lab}
    Vs = G.forward(gs, ins, Ps, mode="normal", acc64=True)
    close(vl.to_numpy(student.vars["prediction"].value), Vs["prediction"], 1e-4, "student prediction")
    close(vl.to_numpy(student.vars["objective"].value).ravel()[0], Vs["objective"], 1e-4, "student objective")
    total, flips = check_gates(gs, Vs, gates_s)
    _, DPs = G.backward(gs, Vs, {"objective": np.float32(1)}, Ps, mode="normal", acc64=True, gates=gates_s)
    Vs32 = G.forward(gs, ins, Ps, mode="normal", acc64=False)
    _, DPs32 = G.backward(gs, Vs32, {"objective": np.float32(1)}, Ps, mode="normal", acc64=False, gates=gates_s)
    worst_s = check_derivatives(Z, "none", student, DPs, DP32=DPs32)
    check_updated(student, updated_reference(student, Ps, DPs, opts, lr, N), Ps, "student", DPs, DPs32, lr, N)
    print("joint step, student branch at %d spectrograms: %d of %d decisions differ; worst derivative error / "
          "allowance = %.3f" % (N, flips, total, worst_s))


def test_distillation_step_at_the_bench_shard(gpu, Z, monkeypatch):
    """BASELINE config 4's per-GPU shard, numerically: ResNet-50 logits on 32 faces (16 x the two fixture faces: every
    logit known), then the student step on 32 spectrograms with those logits as targets -- prediction, objective,
    decisions, every parameter derivative and the updated parameters against the oracle at N = 32."""
    import torch
    from mcncrossmodalemotions_amd import train, vl, zoo
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_nets", os.path.join(HERE, "golden", "make_golden_nets.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    N, W = 32, 300
    teacher = zoo.ferPlusZoo("resnet50-ferplus")
    zoo.strip_losses(teacher)
    g, Pt = M.teacher_params(False, 100)
    for l in g:
        if l.type == "bnorm":
            Pt[l.params[2]] = Z["r50_mom_%s" % l.params[2]]
    inject(teacher, Pt)
    teacher.move("gpu")
    teacher.mode = "test"
    teacher.vars["prediction"].precious = True
    faces = np.asfortranarray(np.tile(G.face_batch(M.TEACHER_N, 1), (1, 1, 1, N // M.TEACHER_N)))
    tstream = torch.cuda.Stream()
    with torch.cuda.stream(tstream):                       # bench.py: the teacher on its own stream
        teacher.eval(["data", vl.from_numpy(faces)])
        tl = teacher.vars["prediction"].value
        ml = vl.max_label(tl)
        ev = torch.cuda.Event()
        ev.record(tstream)
    ref_logits = np.asfortranarray(np.tile(Z["r50_logits"], (1, 1, 1, N // M.TEACHER_N)))

    student = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=W / 100.0,
                            numOutputs=8)
    gs = G.vggvox_student(W)
    Ps = G.perturb_bn(G.make_params(gs, 200), gs, 201)
    inject(student, Ps)
    student.pack_params()
    student.wgradStream = torch.cuda.Stream()
    student.vars["prediction"].precious = True
    data, _, _ = G.spectrogram_batch(N, W, 32)
    opts = train.TrainOpts(batchSize=N)
    rec = GateRecorder(student, monkeypatch)
    train.train_step(student, ["data", vl.from_numpy(data), "logitTarget", tl, "maxLabel", ml], opts, 0, None, N,
                     input_events={"logitTarget": ev, "maxLabel": ev})
    torch.cuda.synchronize()
    close(vl.to_numpy(tl), ref_logits, 1e-4, "ResNet-50 logits at 32 faces")

    lgo = ref_logits.astype(np.float32)
    lab = O.F(lgo.reshape(8, N).argmax(0).reshape(1, 1, 1, N) + 1)
    ins = {"data": data, "logitTarget": lgo, "maxLabel": lab}
    V = G.forward(gs, ins, Ps, mode="normal", acc64=True)
    close(vl.to_numpy(student.vars["prediction"].value), V["prediction"], 1e-4, "prediction")
    close(vl.to_numpy(student.vars["objective"].value).ravel()[0], V["objective"], 1e-4, "objective")
    close(vl.to_numpy(student.vars["classerror"].value).ravel()[0], V["classerror"], 0, "classerror")
    gates = rec.gates(gs)
    total, flips = check_gates(gs, V, gates)
    _, DPm = G.backward(gs, V, {"objective": np.float32(1)}, Ps, mode="normal", acc64=True, gates=gates)
    V32 = G.forward(gs, ins, Ps, mode="normal", acc64=False)
    _, DP32 = G.backward(gs, V32, {"objective": np.float32(1)}, Ps, mode="normal", acc64=False, gates=gates)
    worst = check_derivatives(Z, "none", student, DPm, DP32=DP32)
    check_updated(student, updated_reference(student, Ps, DPm, opts, float(opts.learningRate[0]), N), Ps, "student", DPm, DP32,
                  float(opts.learningRate[0]), N)
    print("distillation step at the 32-pair shard: %d of %d decisions differ (all at the boundary); worst derivative "
          "error / allowance = %.3f" % (flips, total, worst))
