"""The product's zoo.py must build exactly the graphs of the oracle's own layer tables (oracle/graphs.py, restated
from SURVEY Appendix B without importing the product), and those tables must reproduce the known answers the
reference / SURVEY hold: the pool6 bucket table (emoVoxZoo.m:258-259), the MAC totals of SURVEY 8d, the
parameter counts of Appendix B.  CPU only."""
import numpy as np
import pytest

from oracle import graphs as G


def _pad4(p):
    p = list(p) if not np.isscalar(p) else [p] * 4
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    return tuple(int(v) for v in p)


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[-1]))


def _describe(rec):
    """product layer record -> (type, attrs) in the vocabulary of oracle/graphs.py"""
    from mcncrossmodalemotions_amd import dagnn
    b = rec.block
    if isinstance(b, dagnn.Conv):
        return "conv", dict(size=tuple(int(s) for s in b.size), hasBias=bool(b.hasBias), stride=_pair(b.stride),
                            pad=_pad4(b.pad))
    if isinstance(b, dagnn.BatchNorm):
        return "bnorm", dict(numChannels=b.numChannels, epsilon=b.epsilon)
    if isinstance(b, dagnn.ReLU):
        assert b.leak == 0
        return "relu", {}
    if isinstance(b, dagnn.Sigmoid):
        return "sigmoid", {}
    if isinstance(b, dagnn.DropOut):
        return "dropout", dict(rate=float(b.rate))
    if isinstance(b, dagnn.GlobalPooling):
        return "gpool", dict(method=b.method)
    if isinstance(b, dagnn.Pooling):
        return "pool", dict(poolSize=[int(v) for v in b.poolSize], stride=_pair(b.stride), pad=_pad4(b.pad),
                            method=b.method)
    if isinstance(b, dagnn.Sum):
        return "sum", {}
    if isinstance(b, dagnn.Axpy):
        return "axpy", {}
    if isinstance(b, dagnn.SoftmaxCELoss):
        return "softmaxceloss", dict(temperature=b.temperature, logitTargets=b.logitTargets)
    if isinstance(b, dagnn.ErrorStats):
        return "errorstats", dict(numClasses=b.numClasses)
    if isinstance(b, dagnn.Loss):
        return "loss", dict(loss=b.loss)
    raise AssertionError("layer type without a table entry: %r" % type(b))


def assert_same_graph(net, table):
    assert [l.name for l in net.layers] == [l.name for l in table]
    for rec, ref in zip(net.layers, table):
        typ, attrs = _describe(rec)
        assert typ == ref.type, (rec.name, typ, ref.type)
        assert list(rec.inputs) == list(ref.inputs), (rec.name, rec.inputs, ref.inputs)
        assert list(rec.outputs) == list(ref.outputs), (rec.name, rec.outputs, ref.outputs)
        assert list(rec.params) == list(ref.params), (rec.name, rec.params, ref.params)
        want = dict(ref.attrs)
        if "pad" in want:
            want["pad"] = _pad4(want["pad"])
        if "stride" in want:
            want["stride"] = _pair(want["stride"])
        assert attrs == want, (rec.name, attrs, want)


@pytest.mark.parametrize("width", [100, 300, 400, 1000])
def test_student_graph_equals_oracle_table(width):
    from mcncrossmodalemotions_amd import zoo
    net = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=width / 100.0)
    assert_same_graph(net, G.vggvox_student(width))
    for k, v in G.make_params(G.vggvox_student(width), 0).items():
        assert tuple(net.params[k].value.shape) == tuple(v.shape), k


def test_student_graph_with_dropout_equals_oracle_table():
    """emoVoxZoo('dropout', r): configureForRegression puts a dagnn.DropOut behind the outputs of the third-last and
    second-last convolution -- fc6 and fc7 -- and rewires their consumers (emoVoxZoo.m:116-135,272-277)."""
    from mcncrossmodalemotions_amd import zoo
    net = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=3, dropout=0.5)
    assert_same_graph(net, G.vggvox_student(300, dropout=0.5))
    names = [l.name for l in net.layers]
    assert names.index("fc6_drop") == names.index("fc6") + 1 and names.index("fc7_drop") == names.index("fc7") + 1
    assert net.layers[names.index("bn6")].inputs == ["fc6_drop"] and net.layers[names.index("bn7")].inputs == ["fc7_drop"]
    # the default (opts.dropout = false, emoVoxZoo.m:18) has none
    assert not any(l.type == "dropout" for l in G.vggvox_student(300))


@pytest.mark.parametrize("se", [False, True])
@pytest.mark.parametrize("heads", [False, True])
def test_teacher_graph_equals_oracle_table(se, heads):
    from mcncrossmodalemotions_amd import zoo
    net = zoo.ferPlusZoo("senet50-ferplus" if se else "resnet50-ferplus")
    if not heads:
        zoo.strip_losses(net)      # fetch_emovoxceleb_imdb.m:101-106
    table = G.resnet50_teacher(se=se, heads=heads)
    assert_same_graph(net, table)
    P = G.make_params(table, 0)
    assert set(P) == set(net.params)
    for k, v in P.items():
        assert tuple(net.params[k].value.shape) == tuple(v.shape), k


def test_table_known_answers():
    # MACs per sample and conv/FC parameter counts
    st = G.vggvox_student(300)
    assert G.macs(st, (512, 300, 1)) == 2831114496                    # 2.8311 GMAC -> 5.662 GFLOP fwd
    assert round(G.macs(G.vggvox_student(400), (512, 400, 1)) / 1e9, 4) == 3.8010   # W = 400
    r50, se50 = G.resnet50_teacher(False), G.resnet50_teacher(True)
    assert G.macs(r50, (224, 224, 3)) == 3855941632                   # 3.8559 GMAC
    assert G.macs(se50, (224, 224, 3)) == 3858456576                  # 3.8585 GMAC

    def nparams(g):
        return sum(int(np.prod(l.attrs["size"])) for l in g if l.type == "conv")
    assert nparams(st) == 4704 + 614400 + 2 * 884736 + 589824 + 9437184 + 4194304 + 8192   # B.1 column: 16.62 M
    assert abs(nparams(r50) - 23.47e6) < 0.01e6
    assert abs(nparams(se50) - 25.99e6) < 0.02e6
    # emoVoxZoo.m:258-259: pool6 consumes the whole fc6 output width for every bucket
    for W, p1 in G.POOL6_WIDTH.items():
        S = G.shapes(G.vggvox_student(W), (512, W, 1))
        assert S["x_fc6"] == (1, p1, 4096), (W, S["x_fc6"])
        assert S["x_pool6"] == (1, 1, 4096)
    # teacher: 224 -> 112 -> 56 (Caffe ceil pooling via pad [0 1 0 1]) -> 28 -> 14 -> 7 -> 1
    S = G.shapes(r50, (224, 224, 3))
    assert S["conv1"] == (112, 112, 64) and S["pool1"] == (56, 56, 64)
    assert S["res2cx"] == (56, 56, 256) and S["res3dx"] == (28, 28, 512)
    assert S["res4fx"] == (14, 14, 1024) and S["res5cx"] == (7, 7, 2048) and S["prediction"] == (1, 1, 8)


def test_golden_nets_reproducible_on_cpu():
    """the committed fixture is what the oracle computes today (ResNet-50 logits, 2 faces, ~5 s)."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "nets_full.npz")
    Z = np.load(path)
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_nets",
                                                  os.path.join(os.path.dirname(path), "make_golden_nets.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    g, P = M.teacher_params(False, 100)
    for l in g:
        if l.type == "bnorm":
            P[l.params[2]] = Z["r50_mom_" + l.params[2]]
    V = G.forward(g, {"data": G.face_batch(M.TEACHER_N, 1)}, P, mode="test", acc64=True, keep=("prediction",))
    np.testing.assert_allclose(V["prediction"], Z["r50_logits"], rtol=0, atol=1e-6 * max(1, np.abs(Z["r50_logits"]).max()))
