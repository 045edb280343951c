"""GPU parity at the reference's REAL architectures and sizes against tests/golden/nets_full.npz (oracle fp64
accumulate over the oracle's own layer tables; generator: tests/golden/make_golden_nets.py).

    full 16-block ResNet-50 / SE-ResNet-50, 224x224x3, test mode          fetch_emovoxceleb_imdb.m:98-131
    full-width VGGVox student step, 512x300, train mode, every derivative   run_distillation.m:125-131,170-182
    SE-ResNet-50 fwd + bwd with its softmaxlog head (config-5 teacher)      ferplus_baselines.m:140
    teacher validation pass with loss + classerror attached, batch 32       benchmark_ferplus_models.m:46-54,
                                                                             ferplus_baselines.m:120-141
Inputs / parameters are regenerated from seeds by oracle.graphs (numpy only); nothing here runs the oracle's
operators except the two scalar loss heads of the validation-pass test.
Tolerances: logits / predictions 1e-4 * max(1, max|ref|) (north_star); parameter derivatives: see check_summary
(5e-4 of the largest sampled entry plus 4 x the deviation of the reference's own fp32 CPU arithmetic from the fp64
values -- cancellation-dominated sums such as conv1's filter derivative or the exactly-zero bias derivatives in
front of a train-mode BatchNorm sit below 5e-4 in ANY fp32 summation order -- on the 98th percentile of the
samples, with a separate cap for ReLU-gate flips)."""
import importlib.util
import os

import numpy as np
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


def check_summary(Z, prefix, name, got, tol=5e-4, fails=None):
    """Sampled entries of a tensor against the fixture.

    bulk   98th percentile of |err| <= tol * max|ref| + 4 * dev32  (dev32 = deviation of the reference's own fp32
           CPU arithmetic from the fp64 values on the same samples, recorded by make_golden_nets.fp32_deviation)
    flips  max |err| <= 40 * tol * max|ref| + 4 * dev32: a pre-activation within fp32 round-off of zero opens its
           ReLU gate in one arithmetic and not in the other (fp64 oracle vs any fp32 path, the CPU one included);
           every such flip moves the few derivative entries it feeds by one element's worth.  Switching every
           bnorm / bias reduction of the HIP path to fp64 accumulation left these outliers unchanged to four
           digits, which is how they were told apart from round-off.
    norm   L2 norm within 2 * tol (+ the same floor)."""
    flat = np.asarray(got, np.float32).ravel(order="F")
    ref_n, ref_s = float(Z["%s_%s_norm" % (prefix, name)]), Z["%s_%s_samp" % (prefix, name)]
    key = "%s_%s_dev32" % (prefix, name)
    floor = 4.0 * float(Z[key]) if key in Z.files else 0.0
    idx = np.unique(np.linspace(0, flat.size - 1, min(flat.size, 256)).astype(np.int64))
    e = np.abs(flat[idx].astype(np.float64) - ref_s)
    scale = max(float(np.abs(ref_s).max()), 1e-30)
    bulk, worst = float(np.percentile(e, 98)), float(e.max())
    allowed = tol * scale + floor
    n = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
    nallowed = 2 * tol * max(ref_n, 1e-30) + floor * np.sqrt(flat.size)
    msg = None
    if bulk > allowed:
        msg = "%s %s: 98th-percentile err %.3e > allowed %.3e (max|ref| %.3g, floor %.1e)" % (
            prefix, name, bulk, allowed, scale, floor)
    elif worst > 40 * tol * scale + floor:
        msg = "%s %s: max err %.3e > %.3e (max|ref| %.3g)" % (prefix, name, worst, 40 * tol * scale + floor, scale)
    elif abs(n - ref_n) > nallowed:
        msg = "%s %s: norm %.6g vs %.6g (allowed %.2e)" % (prefix, name, n, ref_n, nallowed)
    if msg and fails is not None:
        fails.append(msg)
    elif msg:
        raise AssertionError(msg)
    return bulk / allowed


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


@pytest.mark.parametrize("side_stream", [False, True])
def test_full_student_step(gpu, Z, M, side_stream):
    """full-width VGGVox-BN, 4 spectrograms 512x300, train mode: prediction, loss, classerror and every
    parameter derivative (incl. the batch moments handed to trainMethod 'average')."""
    import torch
    from mcncrossmodalemotions_amd import vl, zoo
    net = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=M.STUDENT_W / 100.0)
    _, P = M.student_params()
    inject(net, P)
    net.pack_params()
    if side_stream:
        net.wgradStream = torch.cuda.Stream()
    data, lgo, lab = G.spectrogram_batch(M.STUDENT_N, M.STUDENT_W, M.STUDENT_IN_SEED)
    net.vars["prediction"].precious = True
    net.mode = "normal"
    net.eval(["data", vl.from_numpy(data), "logitTarget", vl.from_numpy(lgo), "maxLabel", vl.from_numpy(lab)],
             ["objective", 1])
    torch.cuda.synchronize()
    close(vl.to_numpy(net.vars["prediction"].value), Z["stu_prediction"], 1e-4, "prediction")
    close(vl.to_numpy(net.vars["objective"].value).ravel()[0], Z["stu_objective"], 1e-5, "objective")
    close(vl.to_numpy(net.vars["classerror"].value).ravel()[0], Z["stu_classerror"], 0, "classerror")
    fails, worst = [], 0.0
    for name in net.params:
        worst = max(worst, check_summary(Z, "stu_der", name, vl.to_numpy(net.params[name].der), fails=fails))
    print("student step: worst derivative error / allowance = %.3f" % worst)
    assert not fails, "\n".join(fails)


def test_full_joint_teacher_backward(gpu, Z, M):
    """SE-ResNet-50 with the softmaxlog head in train mode, fwd + bwd at full width / depth (config 5's
    teacher branch): logits, loss, every parameter derivative."""
    import torch
    from mcncrossmodalemotions_amd import vl, zoo
    net = zoo.ferPlusZoo("senet50-ferplus")
    g = G.resnet50_teacher(se=True, heads=True)
    inject(net, G.perturb_bn(G.make_params(g, 300), g, 301))
    net.pack_params()
    net.wgradStream = torch.cuda.Stream()
    net.vars["prediction"].precious = True
    net.mode = "normal"
    x = G.face_batch(M.JOINT_N, M.JOINT_IN_SEED)
    net.eval(["data", vl.from_numpy(x), "label", vl.from_numpy(M.joint_labels())], ["objective", 1])
    torch.cuda.synchronize()
    close(vl.to_numpy(net.vars["prediction"].value), Z["jnt_prediction"], 1e-4, "prediction")
    close(vl.to_numpy(net.vars["objective"].value).ravel()[0], Z["jnt_objective"], 1e-5, "objective")
    fails, worst = [], 0.0
    for name in net.params:
        worst = max(worst, check_summary(Z, "jnt_der", name, vl.to_numpy(net.params[name].der), fails=fails))
    print("joint teacher: worst derivative error / allowance = %.3f" % worst)
    assert not fails, "\n".join(fails)


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
