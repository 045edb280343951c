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
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **out)
    print("wrote", os.path.join(HERE, "ops_small.npz"), sum(v.nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    main()
