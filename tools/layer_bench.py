#!/usr/bin/env python
"""Per-layer convolution timing of a whole network of the hot path (forward, tuned tile config).
usage: python tools/layer_bench.py [--net resnet50|senet50|vggvox] [--n 32] [--reps 20] [--bwd]
Prints one row per dagnn.Conv layer: geometry, time, TFLOP/s, share of the summed conv time."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import dagnn, vl, zoo  # noqa: E402


def timeit(fn, reps, tag=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    prof = os.environ.get("LB_PROF") and tag
    if prof:
        import ctypes as C
        from mcncrossmodalemotions_amd import _lib
        L = _lib.load()
        L.xm_prof_enable(1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    if prof:
        L.xm_prof_enable(0)
        cap = 64
        keys = (C.c_int * cap)(); ms = (C.c_double * cap)(); fl = (C.c_double * cap)(); cnt = (C.c_longlong * cap)()
        n = L.xm_prof_collect(cap, keys, ms, fl, cnt)
        out = []
        for i in range(n):
            buf = C.create_string_buffer(128); L.xm_prof_kernel_name(keys[i], buf, 128)
            out.append("%s x%d avg %.3f ms" % (buf.value.decode(), cnt[i], ms[i] / cnt[i]))
        print("[prof] %s: wall %.3f ms/call | %s | ws %d MB" % (tag, s.elapsed_time(e) / reps, "; ".join(out),
                                                          L.xm_workspace_bytes() >> 20), file=sys.stderr)
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="resnet50")
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--bwd", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if os.environ.get("LB_RESERVE"):
        from mcncrossmodalemotions_amd import _lib
        _lib.check(_lib.load().xm_workspace_reserve(int(os.environ["LB_RESERVE"]) << 20))
    rng = np.random.default_rng(0)
    if args.net == "vggvox":
        net = zoo.emoVoxZoo("emovoxceleb-student", numSeconds=3, seed=200)
        x = vl.from_numpy(rng.standard_normal((512, 300, 1, args.n)).astype(np.float32))
        net = zoo.strip_losses(net)
        for nm in [l.name for l in net.layers if isinstance(l.block, (dagnn.ErrorStats,))]:
            net.removeLayer(nm)
    else:
        net = zoo.ferPlusZoo(args.net + "-ferplus", seed=100)
        net = zoo.strip_losses(net)
        x = vl.from_numpy(rng.standard_normal((224, 224, 3, args.n)).astype(np.float32))
    net.move("gpu")
    net.mode = "test"
    net.fuse = False
    for v in net.vars.values():
        v.precious = True
    net.eval(["data", x])
    rows = []
    for rec in net.layers:
        if not isinstance(rec.block, dagnn.Conv):
            continue
        xin = net.vars[rec.inputs[0]].value
        if xin is None:
            continue
        blk = rec.block
        f = net.params[rec.params[0]].value
        b = net.params[rec.params[1]].value if (blk.hasBias and not os.environ.get("LB_NOBIAS")) else None
        H, W, C, N = (int(s) for s in xin.shape)
        FH, FW, FC, K = blk.size
        y = vl.vl_nnconv(xin, f, b, stride=blk.stride, pad=blk.pad)
        Ho, Wo = int(y.shape[0]), int(y.shape[1])
        fl = 2.0 * Ho * Wo * N * K * FH * FW * FC
        ms = timeit(lambda: vl.vl_nnconv(xin, f, b, stride=blk.stride, pad=blk.pad), args.reps)
        row = [rec.name, "%dx%dx%d" % (H, W, C), "%dx%d/%d" % (FH, FW, blk.stride[0]), K, fl, ms]
        if args.bwd:
            dz = torch.randn_like(y)
            row.append(timeit(lambda: vl.vl_nnconv(xin, f, None, dz, stride=blk.stride, pad=blk.pad,
                                                   no_der_filters=True), args.reps))
            row.append(timeit(lambda: vl.vl_nnconv(xin, f, None, dz, stride=blk.stride, pad=blk.pad,
                                                   no_der_data=True), args.reps, rec.name + " wgrad"))
        rows.append(row)
    tot = sum(r[5] for r in rows)
    print("%-22s %-14s %-7s %5s %8s %8s %7s %6s" % ("layer", "in", "filt", "K", "GFLOP", "us", "TF", "%"))
    for r in rows:
        extra = ""
        if args.bwd:
            extra = "  dgrad %7.1f us %6.1f TF  wgrad %7.1f us %6.1f TF" % (
                r[6] * 1e3, r[4] / r[6] / 1e9, r[7] * 1e3, r[4] / r[7] / 1e9)
        print("%-22s %-14s %-7s %5d %8.2f %8.1f %7.1f %6.1f%s" % (
            r[0], r[1], r[2], r[3], r[4] / 1e9, r[5] * 1e3, r[4] / r[5] / 1e9, 100 * r[5] / tot, extra))
    fl = sum(r[4] for r in rows)
    print("TOTAL conv fwd: %.3f ms, %.1f GFLOP, %.1f TFLOP/s" % (tot, fl / 1e9, fl / tot / 1e9))
    if args.bwd:
        td, tw = sum(r[6] for r in rows), sum(r[7] for r in rows)
        print("TOTAL dgrad %.3f ms (%.1f TF)   wgrad %.3f ms (%.1f TF)" % (td, fl / td / 1e9, tw, fl / tw / 1e9))


if __name__ == "__main__":
    main()
