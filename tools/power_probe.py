#!/usr/bin/env python
"""Samples rocm-smi (socket power, sclk, temperature) while a sustained kernel loop runs.
usage: python tools/power_probe.py conv <case> [cfg]   |   python tools/power_probe.py mfma
Answers one question: is a long fp32-MFMA kernel cycle-bound or clock(power)-bound on this box?"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sampler(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True,
                                 text=True, timeout=5).stdout
            rows.append((time.time(), out))
        except Exception as e:  # noqa
            rows.append((time.time(), "ERR %r" % e))
        time.sleep(0.25)


def main():
    what = sys.argv[1]
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, rows))
    if what == "mfma":
        th.start()
        subprocess.run([os.path.join(os.path.dirname(__file__), "_bin", "mfma_peak"), "sustained"])
    else:
        import numpy as np
        import torch
        from mcncrossmodalemotions_amd import vl, _lib
        from tools.conv_bench import CASES
        nm = sys.argv[2]
        H, W, C, FH, FW, K, s, p = CASES[nm][:8]
        N = CASES[nm][8] if len(CASES[nm]) > 8 else 32
        x = vl.from_numpy(np.random.default_rng(0).standard_normal((H, W, C, N)).astype(np.float32))
        f = vl.from_numpy(np.random.default_rng(1).standard_normal((FH, FW, C, K)).astype(np.float32))
        if len(sys.argv) > 3:
            _lib.load().xm_debug_force_conv_cfg(int(sys.argv[3]))
        if os.environ.get("ZERO"):
            x.zero_()
        y = vl.vl_nnconv(x, f, None, stride=s, pad=p)
        torch.cuda.synchronize()
        th.start()
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        while time.time() - t0 < 3.0:
            e0.record()
            for _ in range(50):
                vl.vl_nnconv(x, f, None, stride=s, pad=p)
            e1.record()
            torch.cuda.synchronize()
            n += 1
            if n % 10 == 1:
                ms = e0.elapsed_time(e1) / 50
                print("t=%.2fs  %.3f ms  %.1f TFLOP/s" % (time.time() - t0, ms, 2.0 * y.numel() * FH * FW * C / ms / 1e9), flush=True)
    stop.set()
    th.join()
    for t, o in rows[::2]:
        lines = [l for l in o.strip().splitlines() if l and not l.startswith("WARNING")]
        print("--", " | ".join(lines[-2:])[:400])


if __name__ == "__main__":
    main()
