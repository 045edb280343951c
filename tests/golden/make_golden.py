#!/usr/bin/env python
"""Generates tests/golden/*.npz.

The reference (MATLAB + un-vendored MatConvNet) cannot run in the build container and ships no
test vectors, so these fixtures come from the repo's own CPU restatement (oracle/, fp64-accumulate
variant) AFTER it has been cross-checked against an independent second opinion (torch CPU ops,
tests/test_oracle.py).  They pin the oracle against silent drift and give the GPU tests a
reference that does not depend on the oracle being rebuilt identically.  Parity with the true
MatConvNet binaries stays formally "unpinned" (see oracle/xm_oracle.c).

    python tests/golden/make_golden.py        # rewrites the fixtures (seeded, deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20260928)
    out = {}
    # conv: student conv2-like geometry (5x5 / 2, pad 1), small channels
    x = O.F(rng.standard_normal((21, 18, 6, 2)))
    f = O.F(rng.standard_normal((5, 5, 6, 9)) * 0.2)
    b = O.F(rng.standard_normal(9))
    y = O.vl_nnconv(x, f, b, stride=2, pad=1, acc64=True)
    dz = O.F(rng.standard_normal(y.shape))
    dx, df, db = O.vl_nnconv(x, f, b, dz, stride=2, pad=1, acc64=True)
    out.update(conv_x=x, conv_f=f, conv_b=b, conv_y=y, conv_dzdy=dz, conv_dx=dx, conv_df=df, conv_db=db)
    # bnorm train + backward
    xb = O.F(rng.standard_normal((7, 5, 4, 3)) * 2 + 1)
    g, bb = O.F(rng.uniform(0.5, 1.5, 4)), O.F(rng.standard_normal(4))
    yb, mom = O.vl_nnbnorm(xb, g, bb, acc64=True)
    dzb = O.F(rng.standard_normal(xb.shape))
    dxb, dg, dbb, _ = O.vl_nnbnorm(xb, g, bb, dzb, acc64=True)
    out.update(bn_x=xb, bn_g=g, bn_b=bb, bn_y=yb, bn_moments=mom, bn_dzdy=dzb, bn_dx=dxb, bn_dg=dg, bn_db=dbb)
    # max pool with ties (post-ReLU zeros) -- pins the first-maximum routing rule
    xp = np.maximum(O.F(rng.standard_normal((9, 8, 3, 2))), 0)
    yp = O.vl_nnpool(xp, [3, 3], stride=2, method="max")
    dzp = O.F(rng.standard_normal(yp.shape))
    dxp = O.vl_nnpool(xp, [3, 3], dzp, stride=2, method="max")
    out.update(pool_x=xp, pool_y=yp, pool_dzdy=dzp, pool_dx=dxp)
    # distillation loss, T = 2, logit targets (emoVoxZoo.m:152)
    xl, pl = O.F(rng.standard_normal((1, 1, 8, 5)) * 3), O.F(rng.standard_normal((1, 1, 8, 5)) * 3)
    out.update(loss_x=xl, loss_p=pl,
               loss_y=np.float32(O.vl_nnsoftmaxceloss(xl, pl, temperature=2, logit_targets=True)),
               loss_dx=O.vl_nnsoftmaxceloss(xl, pl, np.ones(1, np.float32), temperature=2, logit_targets=True))
    # spectrogram row normalisation (getBatchEmoVoxCeleb.m:164-169)
    sp = O.F(np.abs(rng.standard_normal((16, 12, 1, 2))) * 4)
    out.update(spec=sp, spec_norm=O.spec_rownorm(sp))
    # ---- second fixture file (added later; separate file so that ops_small.npz stays byte-stable)
    ext = {}
    rng2 = np.random.default_rng(20260929)
    xr, tr = O.F(rng2.standard_normal((1, 1, 8, 6)) * 2), O.F(rng2.standard_normal((1, 1, 8, 6)) * 2)
    wr = O.F(rng2.uniform(0.5, 2, (1, 1, 1, 6)))
    one = np.ones(1, np.float32)
    ext.update(reg_x=xr, reg_t=tr, reg_w=wr,
               euclid_y=np.float32(O.vl_nnregloss(xr, tr, kind="euclidean", instance_weights=wr)),
               euclid_dx=O.vl_nnregloss(xr, tr, one, kind="euclidean", instance_weights=wr),
               huber_y=np.float32(O.vl_nnregloss(xr, tr, kind="huber", sigma=1.0, instance_weights=wr)),
               huber_dx=O.vl_nnregloss(xr, tr, one, kind="huber", sigma=1.0, instance_weights=wr))
    xs, ds = O.F(rng2.standard_normal((2, 3, 8, 2)) * 2), O.F(rng2.standard_normal((2, 3, 8, 2)))
    ext.update(sm_x=xs, sm_dzdy=ds, sm_y=O.vl_nnsoftmaxt(xs, 1.0), sm_dx=O.vl_nnsoftmaxt_backward(xs, ds, 1.0))
    z = O.F(rng2.standard_normal((400 + 160 * 9, 2)) * 0.1)                # 10 analysis frames per clip
    ext.update(wav=z, wav_spec=O.run_spec(z))
    face = O.F(rng2.integers(0, 256, (96, 80, 3, 2)))
    ext.update(face_src=face, face_out=O.crop_resize_face(face, (131.0912, 103.8827, 91.4953), (56, 56)))
    fl = O.F(rng2.standard_normal((23, 8)) * 3)
    ext.update(agg_logits=fl, agg_first=np.array([1, 4, 9], np.int32), agg_last=np.array([3, 9, 23], np.int32))
    for k, (a, b) in enumerate(zip(ext["agg_first"], ext["agg_last"])):
        ext["agg_max_%d" % k] = O.aggregate_logits(fl, int(a), int(b), "max")
        ext["agg_mean_%d" % k] = O.aggregate_logits(fl, int(a), int(b), "mean")
    np.savez_compressed(os.path.join(HERE, "ops_extra.npz"), **ext)
    print("wrote", os.path.join(HERE, "ops_extra.npz"), sum(v.nbytes for v in ext.values()), "bytes raw")
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **out)
    print("wrote", os.path.join(HERE, "ops_small.npz"), sum(v.nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    main()
