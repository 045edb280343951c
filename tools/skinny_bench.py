#!/usr/bin/env python
"""time the skinny 1x1 layers (SE gates, classifier, fc8): python tools/skinny_bench.py   (XM_NO_SKINNY=1 for the MFMA path, XM_NO_SKINNY4=1 for one row per block)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl
dev = torch.device("cuda:0")
cases = [(1, 1, 256, 32, 16, "relu"), (1, 1, 16, 32, 256, "sigmoid"), (1, 1, 2048, 32, 128, "relu"),
         (1, 1, 128, 32, 2048, "sigmoid"), (1, 1, 2048, 64, 128, "relu"), (1, 1, 128, 64, 2048, "sigmoid"),
         (1, 1, 2048, 32, 8, None), (1, 8, 1024, 32, 8, None),
         # four rows per block (XM_NO_SKINNY4=1: one row per block / the MFMA path beyond 128 pixels)
         (1, 1, 4096, 32, 1024, None), (1, 1, 2048, 256, 128, "relu"), (1, 1, 128, 256, 2048, "sigmoid"),
         (1, 1, 256, 256, 16, "relu"), (1, 1, 16, 256, 256, "sigmoid")]
for H, W, C, N, K, act in cases:
    x = torch.randn(N, C, W, H, device=dev).permute(3, 2, 1, 0)
    f = torch.randn(K, C, 1, 1, device=dev).permute(3, 2, 1, 0)
    b = torch.randn(1, K, device=dev).permute(1, 0)
    fn = lambda: vl.vl_nnconv(x, f, b, relu=act == "relu", sigmoid=act == "sigmoid")
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200):
        fn()
    e.record()
    torch.cuda.synchronize()
    print("%dx%dx%dx%d -> %d %-8s %.2f us" % (H, W, C, N, K, act, s.elapsed_time(e) * 5))
