#!/usr/bin/env python
"""Sanity: the student overfits a fixed synthetic batch of teacher logits (loss per sample must fall)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mcncrossmodalemotions_amd import vl, zoo, train
dev = torch.device("cuda", 0)
nb = 16
net = zoo.emoVoxZoo(numSeconds=1, seed=3, width_mult=0.25); net.pack_params()
g = torch.Generator(device=dev); g.manual_seed(0)
spec = vl.spec_rownorm(torch.randn((nb, 1, 100, 512), generator=g, device=dev).abs_().permute(3, 2, 1, 0))
rng = np.random.default_rng(0)
lgo = vl.from_numpy((rng.standard_normal((1, 1, 8, nb)) * 3).astype(np.float32), dev); lab = vl.max_label(lgo)
opts = train.TrainOpts(batchSize=nb, learningRate=[1e-2])
loss_rec = net.getLayer("loss").block
for it in range(300):
    train.train_step(net, ["data", spec, "logitTarget", lgo, "maxLabel", lab], opts, 0, None, nb)
    if it % 50 == 0 or it == 299:
        print("step %3d  objective/sample %.4f  classerror %.3f" % (
            it, float(vl.to_numpy(loss_rec.lastValue).ravel()[0]) / nb,
            float(vl.to_numpy(net.getLayer("classerror").block.lastValue).ravel()[0]) / nb), flush=True)
