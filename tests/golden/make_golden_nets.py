#!/usr/bin/env python
"""Generates tests/golden/nets_full.npz: whole-network results at the reference's REAL architectures and sizes.

    teacher   full 16-block ResNet-50 and SE-ResNet-50 on 224x224x3, test mode, N = 2  -> 8 logits per face
              (fetch_emovoxceleb_imdb.m:98-131; BASELINE configs 1, 3, 4)
    student   full-width VGGVox-BN, 512x300 spectrograms, train mode, N = 4             -> prediction, objective,
              classerror and, per parameter, the derivative's L2 norm + 256 strided samples
              (run_distillation.m:125-131,170-182; BASELINE configs 2, 4)
    joint     SE-ResNet-50 with its softmaxlog head, train mode fwd + bwd, N = 2         -> same per-parameter
              summary (BASELINE config 5's teacher branch; ferplus_baselines.m:140)

Everything is computed by the oracle's fp64-accumulate operators over the oracle's OWN layer tables
(oracle/graphs.py -- nothing here imports the product package).  Inputs and parameters are NOT stored: they are
regenerated from seeds by oracle.graphs (numpy Generator streams); the teachers' calibrated BN moments are stored
because they come out of an oracle forward pass.  The reference itself cannot produce vectors (MATLAB +
un-vendored MatConvNet), so parity with the true binaries stays "unpinned" (oracle/xm_oracle.c header).

    python tests/golden/make_golden_nets.py       # ~1.5 minutes on 8 cores
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import graphs as G  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# (fixture prefix, se?, parameter seed, input seed) -- SURVEY 8d seeds
TEACHERS = (("r50", False, 100, 1), ("se50", True, 300, 3))
TEACHER_N = 2
STUDENT_N, STUDENT_W, STUDENT_SEED, STUDENT_IN_SEED = 4, 300, 200, 2
JOINT_N, JOINT_IN_SEED = 2, 5
NSAMP = 256


def sample_idx(n):
    return np.unique(np.linspace(0, n - 1, min(n, NSAMP)).astype(np.int64))


def summarize(d):
    flat = np.asarray(d, np.float32).ravel(order="F")
    return np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum())), flat[sample_idx(flat.size)]


def fp32_deviation(out, tag, g, inputs, P):
    """How far the reference's OWN arithmetic type is from exact: the same pass through the oracle's fp32 path
    (MatConvNet's CPU algorithm shape: im2row + SGEMM, fp32 accumulation) against the fp64-accumulate values stored
    above, per parameter derivative, on the stored samples.  Several derivatives are sums with massive cancellation
    (the filter derivative of conv1 sums 150 k products whose BN-centred factors sum to zero; a conv bias in front
    of a train-mode BatchNorm has an exactly-zero derivative), so "5e-4 of the largest entry" is below fp32
    round-off for them whatever the summation order; the GPU tests allow 5e-4 * max|ref| + 4 * dev32."""
    V = G.forward(g, inputs, P, mode="normal", acc64=False)
    _, DP = G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=False)
    out[tag + "_prediction_dev32"] = np.float32(np.abs(V["prediction"] - out[tag + "_prediction"]).max())
    for k, d in DP.items():
        _, s = summarize(d)
        out["%s_der_%s_dev32" % (tag, k)] = np.float32(np.abs(s.astype(np.float64) - out["%s_der_%s_samp" % (tag, k)]).max())


def teacher_params(se, seed):
    g = G.resnet50_teacher(se=se)
    P = G.perturb_bn(G.make_params(g, seed), g, seed + 1)
    return g, P


def student_params():
    g = G.vggvox_student(STUDENT_W)
    return g, G.perturb_bn(G.make_params(g, STUDENT_SEED), g, STUDENT_SEED + 1)


def joint_labels():
    return np.asfortranarray(np.random.default_rng(JOINT_IN_SEED + 100).integers(1, 9, (1, 1, 1, JOINT_N))
                             .astype(np.float32))


def main():
    out = {}
    for tag, se, seed, in_seed in TEACHERS:
        t0 = time.time()
        g, P = teacher_params(se, seed)
        x = G.face_batch(TEACHER_N, in_seed)
        G.calibrate_moments(g, P, x)
        keep = ("pool1", "res2cx", "res3dx", "res4fx", "res5cx", "pool5", "prediction")
        V = G.forward(g, {"data": x}, P, mode="test", acc64=True, keep=keep)
        for l in g:
            if l.type == "bnorm":
                out["%s_mom_%s" % (tag, l.params[2])] = P[l.params[2]]
        out[tag + "_logits"] = V["prediction"]
        for v in keep[:-1]:
            n, s = summarize(V[v])
            out["%s_var_%s_norm" % (tag, v)] = n
            out["%s_var_%s_samp" % (tag, v)] = s
        print("%s: logits %s  (%.1f s)" % (tag, np.round(V["prediction"].ravel()[:8], 3), time.time() - t0), flush=True)

    t0 = time.time()
    g, P = student_params()
    data, lgo, lab = G.spectrogram_batch(STUDENT_N, STUDENT_W, STUDENT_IN_SEED)
    V = G.forward(g, {"data": data, "logitTarget": lgo, "maxLabel": lab}, P, mode="normal", acc64=True)
    _, DP = G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=True)
    out["stu_prediction"] = V["prediction"]
    out["stu_objective"] = np.float32(V["objective"])
    out["stu_classerror"] = np.float32(V["classerror"])
    for k, d in DP.items():
        out["stu_der_%s_norm" % k], out["stu_der_%s_samp" % k] = summarize(d)
    fp32_deviation(out, "stu", g, {"data": data, "logitTarget": lgo, "maxLabel": lab}, P)
    print("student: objective %.6f (%.1f s)" % (V["objective"], time.time() - t0), flush=True)

    t0 = time.time()
    g = G.resnet50_teacher(se=True, heads=True)
    P = G.perturb_bn(G.make_params(g, 300), g, 301)
    x = G.face_batch(JOINT_N, JOINT_IN_SEED)
    lab = joint_labels()
    V = G.forward(g, {"data": x, "label": lab}, P, mode="normal", acc64=True)
    _, DP = G.backward(g, V, {"objective": np.float32(1)}, P, mode="normal", acc64=True)
    out["jnt_prediction"] = V["prediction"]
    out["jnt_objective"] = np.float32(V["objective"])
    for k, d in DP.items():
        out["jnt_der_%s_norm" % k], out["jnt_der_%s_samp" % k] = summarize(d)
    fp32_deviation(out, "jnt", g, {"data": x, "label": lab}, P)
    print("joint: objective %.6f (%.1f s)" % (V["objective"], time.time() - t0), flush=True)

    path = os.path.join(HERE, "nets_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
