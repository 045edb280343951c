#!/usr/bin/env python
"""The student's conv1 (512 x 300 x 1 spectrograms, 7 x 7 / stride 2, 96 filters) through conv_stem_kernel and through
the implicit-GEMM kernel it replaces, with and without the fused batch moments.  usage: python tools/stem_bench.py [N=32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcncrossmodalemotions_amd import vl, _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = _lib.load()


def t(fn, reps=30):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for W in (300,):
    x = torch.randn((N, 1, W, 512), device="cuda").permute(3, 2, 1, 0)
    f = (torch.randn((96, 1, 7, 7), device="cuda") * 0.05).permute(3, 2, 1, 0)
    b = vl.mat_empty(96, 1, device=x.device); b.fill_(0.1)
    mo = vl.mat_empty(96, 2, device=x.device)
    dz = torch.randn((N, 96, (W + 2 - 7) // 2 + 1, 254), device="cuda").permute(3, 2, 1, 0)
    out_bytes = 96 * 254 * ((W + 2 - 7) // 2 + 1) * N * 4
    for force in (0, 1, 0, 1):
        L.xm_debug_force_conv_stem(force)
        t0 = t(lambda: vl.vl_nnconv(x, f, b, stride=2, pad=1))
        t1 = t(lambda: vl.vl_nnconv(x, f, b, stride=2, pad=1, moments_out=mo))
        t2 = t(lambda: vl.vl_nnconv(x, f, b, dz, stride=2, pad=1, no_der_data=True, no_der_biases=True))
        print("conv1 N=%d W=%d %-14s conv %.1f us (%.2f TB/s of output)   conv + moments %.1f us   wgrad %.1f us" %
              (N, W, "stem kernels" if force else "implicit GEMM", t0, out_bytes / t0 / 1e6, t1, t2))
    L.xm_debug_force_conv_stem(-1)

if os.environ.get("XM_LIB_PATH", "").endswith("_cyc.so"):       # library built with -DXM_DEBUG_CYCLES: phase clocks
    import ctypes as C, numpy as np
    L.xm_debug_force_conv_stem(1)
    buf = (C.c_ulonglong * (512 * 4))()
    L.xm_debug_conv_cycles(1, buf, 0)
    for mom in (None, mo):
        vl.vl_nnconv(x, f, b, stride=2, pad=1, moments_out=mom)
        torch.cuda.synchronize()
        L.xm_debug_conv_cycles(1, buf, 512)
        d = np.array(buf[:], dtype=np.float64).reshape(512, 4)
        print("moments" if mom is not None else "plain  ", "shader clocks per block: geometry %.0f  MFMA phase + load issue %.0f  epilogue + wait for the loads %.0f  "
              "patch write %.0f  (wave 0, mean over 512 blocks)" % tuple(d.mean(0)))
    L.xm_debug_conv_cycles(0, buf, 0)
