#!/usr/bin/env python
"""device spectrogram front-end (runSpec as a strided convolution + magnitude + row norm)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcncrossmodalemotions_amd import batch as xbatch, vl
dev = torch.device("cuda", 0)
for N, W in ((32, 300), (64, 300), (64, 400)):
    L = int(xbatch.aud_samples(W))
    z = (torch.randn((N, L), device=dev) * 0.1).t()
    f = lambda: vl.spec_rownorm(xbatch.runSpec(z))
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("runSpec + rownorm: %3d clips x %d frames: %.3f ms  (%.0f clips/s)" % (N, W, dt * 1e3, N / dt))
