"""GPU parity at the reference's REAL architectures and sizes against tests/golden/nets_full.npz (oracle fp64
accumulate over the oracle's own layer tables; generator: tests/golden/make_golden_nets.py).

    full 16-block ResNet-50 / SE-ResNet-50, 224x224x3, test mode          fetch_emovoxceleb_imdb.m:98-131
    full-width VGGVox student step, 512x300, train mode, every derivative   run_distillation.m:125-131,170-182
    SE-ResNet-50 fwd + bwd with its softmaxlog head (config-5 teacher)      ferplus_baselines.m:140
    teacher validation pass with loss + classerror attached, batch 32       benchmark_ferplus_models.m:46-54,
                                                                             ferplus_baselines.m:120-141
Inputs / parameters are regenerated from seeds by oracle.graphs (numpy only); nothing here runs the oracle's
operators except the two scalar loss heads of the validation-pass test.
Tolerances: logits / predictions 1e-4 * max(1, max|ref|) (north_star).  Parameter derivatives: a network with ReLUs
and max pooling is piecewise linear -- a pre-activation within round-off of zero (or two window entries within
round-off of each other) takes one branch in the fp64 oracle and the other in ANY fp32 arithmetic, and each such
flip moves every derivative upstream of it by one element's worth.  The tests therefore IDENTIFY the flips instead
of tolerating them (check_gates / masked_reference): the discrete decisions of the HIP forward pass (ReLU masks from
the layer outputs, the fused bnorm+relu+pool steps' routing tables) are compared with the oracle's own -- every
difference must sit at an oracle value within the forward tolerance of the decision boundary -- and the oracle's
backward pass is then re-run with the HIP path's decisions injected (oracle.graphs.backward(gates=...)).  Against
that reference EVERY sampled derivative entry must be within 1e-4 of the largest entry (north_star's own figure; 5e-4
until round 4 -- the measured worst case is 0.27 of this allowance) plus 4 x the deviation of the
reference's own fp32 CPU arithmetic from the fp64 values (make_golden_nets.fp32_deviation: cancellation-dominated
sums such as conv1's filter derivative, or the exactly-zero bias derivatives in front of a train-mode BatchNorm, sit
below any relative bound in any fp32 summation order).  No percentile, no outlier clause."""
import importlib.util
import os

import numpy as np
import torch
import pytest

from oracle import graphs as G
from oracle import oracle as O

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


def close(a, b, tol, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, "%s: max err %.3e > %.1e * %.3g" % (what, err, tol, scale)


def inject(net, P):
    """product net <- oracle-generated parameters, by name (host arrays; move()/pack_params() uploads them)"""
    assert set(P) == set(net.params), set(P) ^ set(net.params)
    for k, v in P.items():
        assert tuple(net.params[k].value.shape) == tuple(v.shape), k
        net.params[k].value = np.asfortranarray(v)


def check_summary(Z, prefix, name, got, tol=1e-4):
    """sampled entries and L2 norm of a forward tensor against the fixture: EVERY sample within tol * max|ref|"""
    flat = np.asarray(got, np.float32).ravel(order="F")
    ref_n, ref_s = float(Z["%s_%s_norm" % (prefix, name)]), Z["%s_%s_samp" % (prefix, name)]
    idx = np.unique(np.linspace(0, flat.size - 1, min(flat.size, 256)).astype(np.int64))
    e = np.abs(flat[idx].astype(np.float64) - ref_s)
    scale = max(float(np.abs(ref_s).max()), 1.0)
    assert float(e.max()) <= tol * scale, "%s %s: max err %.3e > %.1e * %.3g" % (prefix, name, e.max(), tol, scale)
    n = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
    assert abs(n - ref_n) <= 2 * tol * max(ref_n, 1e-30), "%s %s: norm %.6g vs %.6g" % (prefix, name, n, ref_n)


class GateRecorder:
    """Collects the discrete decisions of the HIP forward pass of a training plan WITHOUT changing the plan:
    plain / fused-with-bnorm / fused-with-sum ReLUs leave their output variable (open gate <=> output > 0), which is
    marked precious; a fused bnorm + relu + max-pool step never materialises the rectified tensor, so its routing
    table and pooled output are taken from the operator call itself (vl.bnorm_relu_pool is wrapped)."""

    def __init__(self, net, monkeypatch):
        from mcncrossmodalemotions_amd import dagnn, vl
        self.net, self.vl = net, vl
        self.calls = []
        real = vl.bnorm_relu_pool

        def wrapped(*a, **k):
            out = real(*a, **k)
            self.calls.append((out[0], out[1]))
            return out
        monkeypatch.setattr(vl, "bnorm_relu_pool", wrapped)
        real2 = vl.conv_bnorm_relu_pool

        def wrapped2(*a, **k):
            # the first layer's conv -> bnorm -> relu -> pool in one kernel (round 6): its table marks windows whose maximum
            # did not pass the ReLU with 255; all entries of such a window are zero after the ReLU, so the first-maximum
            # rule gives code 0 there -- recorded in that form
            out = real2(*a, **k)
            if out is not None:
                self.calls.append((out[0], torch.where(out[1] == 255, torch.zeros_like(out[1]), out[1])))
            return out
        monkeypatch.setattr(vl, "conv_bnorm_relu_pool", wrapped2)
        plan = net._plan(True)
        self.pooled = [st for st in plan if isinstance(st, dagnn._BnReluPoolStep)]
        swallowed = {st.relu_rec.name for st in self.pooled}
        self.relus = [l for l in net.layers if isinstance(l.block, dagnn.ReLU) and l.name not in swallowed]
        for l in self.relus:
            net.vars[l.outputs[0]].precious = True
        assert net._plan(True) is not None and [type(x) for x in net._plan(True)] == [type(x) for x in plan], \
            "marking the ReLU outputs precious must not change the fused plan"

    def gates(self, graph):
        """{layer name: gate} in the oracle graph's terms (oracle.graphs.backward)"""
        vl = self.vl
        assert len(self.calls) == len(self.pooled)
        out = {}
        for l in self.relus:
            out[l.name] = vl.to_numpy(self.net.vars[l.outputs[0]].value) > 0
        for st, (y, am) in zip(self.pooled, self.calls):
            yp = vl.to_numpy(y)
            out[st.pool_rec.name] = (am.cpu().numpy().reshape(yp.shape, order="F"), yp > 0)
            out[st.relu_rec.name] = True       # its gate is the pooled output's sign, applied at the pooling layer
        return out


def check_gates(graph, V, gates, tolf=1e-4):
    """Every discrete decision of the HIP pass that differs from the oracle's own must sit at the decision boundary:
    the oracle value that decides it within the forward tolerance (tolf * max(1, max|tensor|)) of zero (ReLU) / of the
    window maximum (routing).  Returns (#decisions, #flips)."""
    total = flips = 0
    by_name = {l.name: l for l in graph}
    for name, g in gates.items():
        l = by_name[name]
        if l.type == "relu" and g is not True:
            x = V[l.inputs[0]]
            scale = tolf * max(1.0, float(np.abs(x).max()))
            diff = g != (x > 0)
            total += x.size
            flips += int(diff.sum())
            assert float(np.abs(x[diff]).max(initial=0.0)) <= scale, \
                "%s: a flipped ReLU gate has oracle pre-activation %.3e (> %.1e)" % (name, np.abs(x[diff]).max(), scale)
        elif l.type == "pool":
            code, flows = g
            x, y = V[l.inputs[0]], V[l.outputs[0]]
            a = l.attrs
            scale = tolf * max(1.0, float(np.abs(x).max()))
            ref = G.pool_argmax_codes(x, a["poolSize"], a["stride"], a["pad"])
            xf = x.ravel(order="F")
            chosen = xf[G.pool_positions(code, x.shape, a["poolSize"], a["stride"], a["pad"])]
            total += 2 * y.size
            d1, d2 = code != ref, flows != (y > 0)
            flips += int(d1.sum()) + int(d2.sum())
            assert float(np.abs(chosen - y)[d1].max(initial=0.0)) <= scale, \
                "%s: a re-routed window picked an entry %.3e below the oracle's maximum" % (name, np.abs(chosen - y)[d1].max())
            assert float(np.abs(y[d2]).max(initial=0.0)) <= scale, "%s: flipped gate at pooled value %.3e" % (name, np.abs(y[d2]).max())
    return total, flips


def check_derivatives(Z, prefix, net, DPm, tol=1e-4, DP32=None):
    """HIP parameter derivatives against the mask-matched oracle pass: EVERY one of the fixture's sample positions
    within tol * max|ref| + 4 * dev32 (dev32: how far the reference's own fp32 CPU arithmetic is from the fp64 values
    on those positions -- from the fixture, or from `DP32`, the same mask-matched pass in the oracle's fp32 path), and
    the whole tensor within twice that floor."""
    from mcncrossmodalemotions_amd import vl
    worst = 0.0
    fails = []
    for name in net.params:
        got = vl.to_numpy(net.params[name].der).astype(np.float64).ravel(order="F")
        ref = np.asarray(DPm[name], np.float64).ravel(order="F")
        key = "%s_%s_dev32" % (prefix, name)
        if DP32 is not None:
            sidx = np.unique(np.linspace(0, ref.size - 1, min(ref.size, 256)).astype(np.int64))
            floor = 4.0 * float(np.abs(np.asarray(DP32[name], np.float64).ravel(order="F") - ref)[sidx].max())
        else:
            floor = 4.0 * float(Z[key]) if key in Z.files else 0.0
        scale = max(float(np.abs(ref).max()), 1e-30)
        idx = np.unique(np.linspace(0, ref.size - 1, min(ref.size, 256)).astype(np.int64))
        e_s, e_all = float(np.abs(got - ref)[idx].max()), float(np.abs(got - ref).max())
        worst = max(worst, e_s / (tol * scale + floor), e_all / (tol * scale + 2 * floor))
        if e_s > tol * scale + floor:
            fails.append("%s %s: max sampled err %.3e > %.3e (max|ref| %.3g, floor %.1e)" % (prefix, name, e_s, tol * scale + floor, scale, floor))
        elif e_all > tol * scale + 2 * floor:
            fails.append("%s %s: max err %.3e > %.3e over the whole tensor" % (prefix, name, e_all, tol * scale + 2 * floor))
    assert not fails, "\n".join(fails)
    return worst


def build_teacher(Z, M, tag, se, seed, heads=False):
    from mcncrossmodalemotions_amd import zoo
    net = zoo.ferPlusZoo("senet50-ferplus" if se else "resnet50-ferplus")
    if not heads:
        zoo.strip_losses(net)
    g, P = M.teacher_params(se, seed)
    for l in g:
        if l.type == "bnorm":
            P[l.params[2]] = Z["%s_mom_%s" % (tag, l.params[2])]
    inject(net, P)
    return net


@pytest.mark.parametrize("tag,se,seed,in_seed", [("r50", False, 100, 1), ("se50", True, 300, 3)])
@pytest.mark.parametrize("fuse", [True, False])
def test_full_teacher_logits(gpu, Z, M, tag, se, seed, in_seed, fuse):
    """all 53 (+32 SE) conv layers at full width on 224x224 faces: logits and stage outputs vs the fixture."""
    from mcncrossmodalemotions_amd import vl
    net = build_teacher(Z, M, tag, se, seed)
    net.move("gpu")
    net.mode = "test"
    net.fuse = fuse
    keep = ("pool1", "res2cx", "res3dx", "res4fx", "res5cx", "pool5", "prediction")
    for v in keep:
        net.vars[v].precious = True
    x = G.face_batch(M.TEACHER_N, in_seed)
    net.eval(["data", vl.from_numpy(x)])
    for v in keep[:-1]:
        check_summary(Z, tag + "_var", v, vl.to_numpy(net.vars[v].value), tol=1e-4)
    close(vl.to_numpy(net.vars["prediction"].value), Z[tag + "_logits"], 1e-4, tag + " logits")


def test_full_teacher_lanes_and_batch(gpu, Z, M):
    """zoo.FrozenTeacher at full size: a batch of 6 faces (the 2 fixture faces x3) cut over 2 stream lanes
    reproduces the fixture logits for every copy (samples are independent in test mode)."""
    from mcncrossmodalemotions_amd import vl, zoo
    net = build_teacher(Z, M, "se50", True, 300)
    net.move("gpu")
    net.mode = "test"
    x = np.asfortranarray(np.tile(G.face_batch(M.TEACHER_N, 3), (1, 1, 1, 3)))
    got = vl.to_numpy(zoo.FrozenTeacher(net, lanes=2).logits(vl.from_numpy(x)))
    close(got, np.tile(Z["se50_logits"], (1, 1, 1, 3)), 1e-4, "lanes")


def test_full_teacher_at_imdb_batch_size(gpu, Z, M):
    """the imdb-building loop runs the teacher at batch 128 (fetch_emovoxceleb_imdb.m:63,122-136; BASELINE config 3):
    test-mode samples are independent, so 128 faces = the 2 fixture faces x 64 must give the fixture logits for every
    copy -- whatever tile configuration / split-K the batch-128 shapes select."""
    from mcncrossmodalemotions_amd import vl
    for tag, se, seed, in_seed in (("se50", True, 300, 3), ("r50", False, 100, 1)):
        net = build_teacher(Z, M, tag, se, seed)
        net.move("gpu")
        net.mode = "test"
        x = np.asfortranarray(np.tile(G.face_batch(M.TEACHER_N, in_seed), (1, 1, 1, 64)))
        net.vars["prediction"].precious = True
        net.eval(["data", vl.from_numpy(x)])
        got = vl.to_numpy(net.vars["prediction"].value)
        assert got.shape == (1, 1, 8, 128)
        close(got, np.tile(Z[tag + "_logits"], (1, 1, 1, 64)), 1e-4, tag + " logits at batch 128")


@pytest.mark.parametrize("side_stream", [False, True])
def test_full_student_step(gpu, Z, M, side_stream, monkeypatch):
    """full-width VGGVox-BN, 4 spectrograms 512x300, train mode: prediction, loss, classerror against the fixture;
    the HIP pass's ReLU / routing decisions against the oracle's (flips only at the decision boundary); every
    parameter derivative (incl. the batch moments handed to trainMethod 'average') against the oracle's backward pass
    with those decisions injected -- all sampled entries, no percentile."""
    import torch
    from mcncrossmodalemotions_amd import vl, zoo
    net = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=M.STUDENT_W / 100.0)
    g, P = M.student_params()
    inject(net, P)
    net.pack_params()
    if side_stream:
        net.wgradStream = torch.cuda.Stream()
    data, lgo, lab = G.spectrogram_batch(M.STUDENT_N, M.STUDENT_W, M.STUDENT_IN_SEED)
    net.vars["prediction"].precious = True
    net.mode = "normal"
    rec = GateRecorder(net, monkeypatch)
    net.eval(["data", vl.from_numpy(data), "logitTarget", vl.from_numpy(lgo), "maxLabel", vl.from_numpy(lab)],
             ["objective", 1])
    torch.cuda.synchronize()
    close(vl.to_numpy(net.vars["prediction"].value), Z["stu_prediction"], 1e-4, "prediction")
    close(vl.to_numpy(net.vars["objective"].value).ravel()[0], Z["stu_objective"], 1e-5, "objective")
    close(vl.to_numpy(net.vars["classerror"].value).ravel()[0], Z["stu_classerror"], 0, "classerror")
    # the oracle's own pass (fp64 accumulate) -- reproduces the fixture -- then its backward with the HIP decisions
    ins = {"data": data, "logitTarget": lgo, "maxLabel": lab}
    V = G.forward(g, ins, P, mode="normal", acc64=True)
    close(V["prediction"], Z["stu_prediction"], 1e-6, "oracle pass == fixture")
    gates = rec.gates(g)
    total, flips = check_gates(g, V, gates)
    _, DPm = G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=True, gates=gates)
    worst = check_derivatives(Z, "stu_der", net, DPm)
    print("student step: %d of %d discrete decisions differ from the oracle's (all at the boundary); worst "
          "derivative error / allowance = %.3f" % (flips, total, worst))


def test_full_student_step_at_baseline_batch(gpu, Z, M, monkeypatch):
    """the student's REAL training batch: 64 spectrograms 512x300 (run_distillation.m:75; BASELINE config 2), train
    mode -- batch-norm statistics over 64 samples (conv1: 2.4 M values per channel), side stream on.  The oracle runs
    here (fp64 accumulate, ~15 s on the GPU box's host cores): forward values, the discrete decisions, and every
    parameter derivative against the mask-matched pass; the fp32 floor comes from the same pass in the oracle's fp32
    path (there is no fixture at this size)."""
    import torch
    from mcncrossmodalemotions_amd import vl, zoo
    N = 64
    net = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=M.STUDENT_W / 100.0)
    g, P = M.student_params()
    inject(net, P)
    net.pack_params()
    net.wgradStream = torch.cuda.Stream()
    data, lgo, lab = G.spectrogram_batch(N, M.STUDENT_W, 64)
    net.vars["prediction"].precious = True
    net.mode = "normal"
    rec = GateRecorder(net, monkeypatch)
    net.eval(["data", vl.from_numpy(data), "logitTarget", vl.from_numpy(lgo), "maxLabel", vl.from_numpy(lab)],
             ["objective", 1])
    torch.cuda.synchronize()
    ins = {"data": data, "logitTarget": lgo, "maxLabel": lab}
    V = G.forward(g, ins, P, mode="normal", acc64=True)
    close(vl.to_numpy(net.vars["prediction"].value), V["prediction"], 1e-4, "prediction")
    close(vl.to_numpy(net.vars["objective"].value).ravel()[0], V["objective"], 1e-5, "objective")
    close(vl.to_numpy(net.vars["classerror"].value).ravel()[0], V["classerror"], 0, "classerror")
    gates = rec.gates(g)
    total, flips = check_gates(g, V, gates)
    _, DPm = G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=True, gates=gates)
    V32 = G.forward(g, ins, P, mode="normal", acc64=False)
    _, DP32 = G.backward(g, V32, {"objective": np.float32(1)}, P, mode="normal", acc64=False, gates=gates)
    worst = check_derivatives(Z, "none", net, DPm, DP32=DP32)
    print("student step at batch 64: %d of %d discrete decisions differ from the oracle's (all at the boundary); "
          "worst derivative error / allowance = %.3f" % (flips, total, worst))


def check_slopes(net, g, ins, P, names, change=3e-3, tol=1e-2):
    """for every tensor of `names`: the oracle's fp64-accumulate forward objective, stepped by +- e along the HIP derivative
    of that tensor (e such that the predicted change is 2 * `change`), changes by what the derivative predicts"""
    from mcncrossmodalemotions_amd import vl

    def objective(Q):
        return float(G.forward(g, ins, Q, mode="normal", acc64=True)["objective"])

    close(vl.to_numpy(net.vars["objective"].value).ravel()[0], objective(P), 1e-5, "objective")
    worst = 0.0
    for name in names:
        gp = vl.to_numpy(net.params[name].der).astype(np.float64).reshape(P[name].shape, order="F")
        nrm = float(np.sqrt((gp ** 2).sum()))
        assert nrm > 0, name
        step = change / nrm * (gp / nrm)
        Pp, Pm = dict(P), dict(P)
        Pp[name] = (P[name].astype(np.float64) + step).astype(np.float32)
        Pm[name] = (P[name].astype(np.float64) - step).astype(np.float32)
        predicted = float(((Pp[name].astype(np.float64) - Pm[name].astype(np.float64)) * gp).sum())   # the step as rounded to fp32
        measured = objective(Pp) - objective(Pm)
        worst = max(worst, abs(measured / predicted - 1.0))
        assert abs(measured / predicted - 1.0) < tol, "%s: objective changed by %.6e, the HIP derivative predicts %.6e" % (
            name, measured, predicted)
    return worst


def test_student_derivatives_are_the_slope_of_the_oracle_forward(gpu, M):
    """An INDEPENDENT check of the backward pass (round-5 review, weak 1 (i)): nothing the HIP pass decided -- no ReLU gate,
    no routing table -- is handed to the oracle here.  For a parameter tensor p with HIP derivative g_p, the oracle's
    fp64-accumulate FORWARD objective L must change by <g_p, d> along d = g_p / |g_p|:
        L(p + e d) - L(p - e d) = 2 e |g_p|            (central difference; train-mode bnorm statistics included)
    with e chosen so that the predicted change is 6e-3 (L ~ 9: well above the forward's fp32 storage noise, small enough
    that the kinks a step crosses do not matter -- calibrated with the oracle's own backward: ratio 0.997 ... 1.0002 on
    these tensors).  One tensor per kernel family of the student's backward: the Gram route of conv1 / bn1, the stride-2
    and 3 x 3 patch filter derivatives, the generic filter derivative, the skinny FC layers, a late bnorm.  (No
    convolution bias except fc8's: a train-mode bnorm follows every other one, the objective does not depend on it and its
    derivative is rounding noise around zero.)"""
    import torch
    from mcncrossmodalemotions_amd import vl, zoo
    N = 4
    net = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=M.STUDENT_W / 100.0)
    g, P = M.student_params()
    inject(net, P)
    net.pack_params()
    net.wgradStream = torch.cuda.Stream()
    data, lgo, lab = G.spectrogram_batch(N, M.STUDENT_W, 164)
    net.mode = "normal"
    net.eval(["data", vl.from_numpy(data), "logitTarget", vl.from_numpy(lgo), "maxLabel", vl.from_numpy(lab)],
             ["objective", 1])
    torch.cuda.synchronize()
    ins = {"data": data, "logitTarget": lgo, "maxLabel": lab}
    worst = check_slopes(net, g, ins, P, ("conv1f", "bn1m", "bn1b", "conv2f", "conv3f", "conv5f", "bn5m", "fc6f", "fc7f", "fc8f",
                                          "fc8b"))
    print("student derivatives as slopes of the oracle's forward objective: worst |ratio - 1| = %.2e" % worst)


def test_teacher_derivatives_are_the_slope_of_the_oracle_forward(gpu, M):
    """the same independent check for config 5's teacher branch (SE-ResNet-50, softmaxlog head, train mode, the two fixture
    faces): the 7 x 7 / 2 stem, 3 x 3 and 1 x 1 layers of every stage incl. a strided projection, the SE expansion layers,
    bnorm multipliers / biases, the classifier.  (Not the SE reduction layers: 2 samples x C / 16 rectified units put one
    kink inside most steps -- the oracle's own backward reads 0.98 ... 1.02 there.)"""
    import torch
    from mcncrossmodalemotions_amd import vl, zoo
    net = zoo.ferPlusZoo("senet50-ferplus")
    g = G.resnet50_teacher(se=True, heads=True)
    P = G.perturb_bn(G.make_params(g, 300), g, 301)
    inject(net, P)
    net.pack_params()
    net.wgradStream = torch.cuda.Stream()
    net.mode = "normal"
    x, lab = G.face_batch(M.JOINT_N, M.JOINT_IN_SEED), M.joint_labels()
    net.eval(["data", vl.from_numpy(x), "label", vl.from_numpy(lab)], ["objective", 1])
    torch.cuda.synchronize()
    worst = check_slopes(net, g, {"data": x, "label": lab}, P,
                         ("conv1_filter", "res2a_branch2b_filter", "res3a_branch1_filter", "res4b_branch2b_filter",
                          "res5c_branch2c_filter", "res3b_fc2_filter", "res4c_fc2_filter", "res2a_branch2a_mult",
                          "res5c_branch2c_bias", "classifier_filter", "classifier_bias"))
    print("teacher derivatives as slopes of the oracle's forward objective: worst |ratio - 1| = %.2e" % worst)


def test_full_joint_teacher_backward(gpu, Z, M, monkeypatch):
    """SE-ResNet-50 with the softmaxlog head in train mode, fwd + bwd at full width / depth (config 5's
    teacher branch): logits, loss; decisions and every parameter derivative as in test_full_student_step."""
    import torch
    from mcncrossmodalemotions_amd import vl, zoo
    net = zoo.ferPlusZoo("senet50-ferplus")
    g = G.resnet50_teacher(se=True, heads=True)
    P = G.perturb_bn(G.make_params(g, 300), g, 301)
    inject(net, P)
    net.pack_params()
    net.wgradStream = torch.cuda.Stream()
    net.vars["prediction"].precious = True
    net.mode = "normal"
    x, lab = G.face_batch(M.JOINT_N, M.JOINT_IN_SEED), M.joint_labels()
    rec = GateRecorder(net, monkeypatch)
    net.eval(["data", vl.from_numpy(x), "label", vl.from_numpy(lab)], ["objective", 1])
    torch.cuda.synchronize()
    close(vl.to_numpy(net.vars["prediction"].value), Z["jnt_prediction"], 1e-4, "prediction")
    close(vl.to_numpy(net.vars["objective"].value).ravel()[0], Z["jnt_objective"], 1e-5, "objective")
    V = G.forward(g, {"data": x, "label": lab}, P, mode="normal", acc64=True)
    close(V["prediction"], Z["jnt_prediction"], 1e-6, "oracle pass == fixture")
    gates = rec.gates(g)
    total, flips = check_gates(g, V, gates)
    _, DPm = G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=True, gates=gates)
    worst = check_derivatives(Z, "jnt_der", net, DPm)
    print("joint teacher: %d of %d discrete decisions differ from the oracle's (all at the boundary); worst "
          "derivative error / allowance = %.3f" % (flips, total, worst))


def test_teacher_validation_pass_with_heads(gpu, Z, M):
    """the FER+ evaluation path -- cnn_train_dag's val pass over the teacher with its loss and
    classerror layers attached, batch 32, test mode (benchmark_ferplus_models.m:46-54).  32 faces = the 2 fixture
    faces x16, so every logit is known; the per-sample averages of both heads are checked against the oracle's
    vl_nnloss on the fixture logits."""
    from mcncrossmodalemotions_amd import vl, train
    net = build_teacher(Z, M, "r50", False, 100, heads=True)
    net.move("gpu")
    faces = np.asfortranarray(np.tile(G.face_batch(M.TEACHER_N, 1), (1, 1, 1, 16)))
    labels = np.asfortranarray((np.arange(32) % 8 + 1).reshape(1, 1, 1, 32).astype(np.float32))

    def getBatch(imdb, batch):
        idx = [int(i) for i in batch]
        return ["data", vl.from_numpy(faces[..., idx]), "label", vl.from_numpy(labels[..., idx])]

    opts = train.TrainOpts(batchSize=32)
    stats = train.process_epoch(net, None, getBatch, list(range(32)), opts, 0, "val")
    logits = np.asfortranarray(np.tile(Z["r50_logits"], (1, 1, 1, 16)))
    ref_obj = float(O.vl_nnloss(logits, labels, loss="softmaxlog")) / 32
    ref_err = float(O.vl_nnloss(logits, labels, loss="classerror")) / 32
    assert stats["num"] == 32
    assert abs(stats["objective"] - ref_obj) <= 1e-4 * max(1.0, abs(ref_obj)), (stats["objective"], ref_obj)
    assert abs(stats["top1error"] - ref_err) < 1e-6, (stats["top1error"], ref_err)


def test_se_ops_full_size_properties(gpu):
    """SE path at BASELINE config-3 size (56 x 56 x 256 x 128): squeeze = per-(c, n) plane means (vs float64 on a
    strided subset of planes), excite + residual + ReLU fused vs unfused (1 ulp: fma), and <dz, scale_axpy(x,a)> adjointness of
    scale_backward."""
    import torch
    from mcncrossmodalemotions_amd import vl
    H = W = 56
    C, N = 256, 128
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    x = torch.randn((N, C, W, H), generator=g, device="cuda").permute(3, 2, 1, 0)
    r = torch.randn((N, C, W, H), generator=g, device="cuda").permute(3, 2, 1, 0)
    a = torch.rand((N, C, 1, 1), generator=g, device="cuda").permute(3, 2, 1, 0)
    y = vl.vl_nnpool(x, [H, W], method="avg")
    ref = x.double().sum(dim=(0, 1)) / (H * W)
    assert float((y.reshape(C, N).double() - ref).abs().max()) <= 1e-6
    fused = vl.scale_axpy(x, a, r, relu=True)
    plain = vl.vl_nnrelu(vl.sum2(vl.scale_axpy(x, a), r))
    # the fused kernel contracts a .* x + r into one fma (one rounding), the unfused pair rounds twice
    assert float((fused - plain).abs().max()) <= 1e-6
    dz = torch.randn((N, C, W, H), generator=g, device="cuda").permute(3, 2, 1, 0)
    dx, da = vl.scale_backward(x, a, dz)
    lhs = float((dz.double() * vl.scale_axpy(x, a).double()).sum())
    # y = a .* x is bilinear: <dz, y> = <dx, x> = <da, a>
    assert abs(float((dx.double() * x.double()).sum()) - lhs) <= 1e-5 * abs(lhs) + 1e-3
    assert abs(float((da.double() * a.double()).sum()) - lhs) <= 1e-5 * abs(lhs) + 1e-3
