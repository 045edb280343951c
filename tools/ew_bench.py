#!/usr/bin/env python
"""Timing of the student's non-convolution operators at their real sizes (batch 32, 3 s spectrograms):
fused bnorm+relu+pool forward / backward and bnorm(+relu) forward / backward.
usage: python tools/ew_bench.py [--n 32] [--reps 20]
Prints ms and the effective HBM rate over the ALGORITHMIC bytes (each tensor once)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    N = a.n
    dev = torch.device("cuda:0")
    pooled = [("bn1+pool1", 254, 148, 96, (3, 3), (2, 2)), ("bn2+pool2", 62, 36, 256, (3, 3), (2, 2)),
              ("bn5+pool5", 30, 17, 256, (5, 3), (3, 2))]
    plain = [("bn3", 30, 17, 384), ("bn4", 30, 17, 256), ("bn6", 1, 8, 4096)]
    big = torch.randn(462 * 250000, device=dev)
    dst = torch.empty_like(big)
    t = timeit(lambda: dst.copy_(big), a.reps)
    print("copy 462 MB (read + write)      %.3f ms  %6.0f GB/s" % (t, 2 * big.numel() * 4 / t / 1e6))
    t = timeit(lambda: big.sum(), a.reps)
    print("sum 462 MB (read)               %.3f ms  %6.0f GB/s" % (t, big.numel() * 4 / t / 1e6))
    del big, dst
    for name, H, W, Cc, pool, stride in pooled:
        x = torch.randn(N, Cc, W, H, device=dev).permute(3, 2, 1, 0)
        g = vl.from_numpy((1 + 0.1 * torch.randn(Cc, 1)).numpy())
        b = vl.from_numpy((0.1 * torch.randn(Cc, 1)).numpy())
        y, am, mo = vl.bnorm_relu_pool(x, g, b, pool, stride)
        dz = torch.randn(N, Cc, y.shape[1], y.shape[0], device=dev).permute(3, 2, 1, 0)
        nx, ny = x.numel() * 4, y.numel() * 4
        t = timeit(lambda: vl.bnorm_relu_pool(x, g, b, pool, stride), a.reps)
        bf = 2 * nx + ny + ny // 4
        print("%-10s fwd (stats + pool)   %.3f ms  %6.0f GB/s" % (name, t, bf / t / 1e6))
        t2 = timeit(lambda: vl.bnorm_relu_pool(x, g, b, pool, stride, moments=mo), a.reps)
        print("%-10s fwd (pool only)      %.3f ms  %6.0f GB/s" % (name, t2, (nx + ny + ny // 4) / t2 / 1e6))
        dxs = vl.mat_empty(Cc, 1)
        t = timeit(lambda: vl.bnorm_relu_pool_backward(x, g, b, mo, am, dz, pool, stride, dxsum_out=dxs, y_pool=y), a.reps)
        bb = 3 * nx + 2 * (ny + ny // 4)
        print("%-10s bwd (partial+apply)  %.3f ms  %6.0f GB/s" % (name, t, bb / t / 1e6))
        t = timeit(lambda: vl.bnorm_relu_pool_backward(x, g, b, mo, am, dz, pool, stride, need_dx=False, y_pool=y), a.reps)
        print("%-10s bwd (partial only)   %.3f ms  %6.0f GB/s" % (name, t, (nx + ny + ny // 4) / t / 1e6))
    for name, H, W, Cc in plain:
        x = torch.randn(N, Cc, W, H, device=dev).permute(3, 2, 1, 0)
        g = vl.from_numpy((1 + 0.1 * torch.randn(Cc, 1)).numpy())
        b = vl.from_numpy((0.1 * torch.randn(Cc, 1)).numpy())
        dz = torch.randn(N, Cc, W, H, device=dev).permute(3, 2, 1, 0)
        nx = x.numel() * 4
        y, mo = vl.vl_nnbnorm(x, g, b, relu=True)
        t = timeit(lambda: vl.vl_nnbnorm(x, g, b, relu=True), a.reps)
        print("%-10s fwd                  %.3f ms  %6.0f GB/s" % (name, t, 3 * nx / t / 1e6))
        t = timeit(lambda: vl.vl_nnbnorm(x, g, b, dzdy=dz, relu=True, y=y, moments=mo, batch_moments=True), a.reps)
        print("%-10s bwd                  %.3f ms  %6.0f GB/s" % (name, t, 5 * nx / t / 1e6))


if __name__ == "__main__":
    main()
