"""Every kernel-path selector of the library chooses between two complete implementations of the same operator
(include/xmodal_prof.h): with a selector set, the same seeded passes -- a frozen SE-ResNet forward, a student training step, a
trainable SE-ResNet training step -- must give the results of the default paths.  Outputs (logits, predictions): 2e-4 of
the tensor's largest magnitude (fp32 summation order differs between the two arms).  Parameter derivatives: 0.1 -- another
summation order flips the ReLU / max-pool decisions that sit exactly at the boundary (about one of the 0.8 M outputs of
the narrow student's bn3 at 8 spectrograms), and ONE flipped element moves the derivative of that bnorm's input by 11 % of its
largest magnitude at that element, the bnorm's bias derivative by 0.8 % and the filter derivative below it by 2 % (measured
with the halo kernels forced on / off: every forward value agrees to 1e-4, the derivatives differ from the flipped element
downwards).  tests/test_gpu_nets_full.py identifies such flips against the oracle and checks everything else at 1e-4; here the
point is that a selector's arm is a working, equivalent path inside whole passes (a wrong kernel is off by O(1)), and the
operator-level tests (tests/test_gpu_ops.py) force the arms one by one and compare them with the oracle at rounding level.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_switch_worker.py")

LIBRARY = ["XM_NO_HYBRID", "XM_NO_HALO", "XM_NO_SKINNY", "XM_NO_SKINNY4", "XM_NO_STEM", "XM_NO_STEM_WGRAD", "XM_NO_DMA",
           "XM_NO_FUSED_STATS", "XM_DGRAD_MERGE", "XM_NO_FAST_TRANSPOSE", "XM_NO_POOL_LDS", "XM_NO_POOL_PATCH",
           "XM_NO_POOL_POOLED", "XM_NO_W8", "XM_NO_WGRAD_PATCH", "XM_NO_WGRAD_PATCH_S2", "XM_NO_DGRAD_S2", "XM_NO_STEM3"]
EXECUTOR = ["XM_NO_FUSED_BIASDER", "XM_NO_FUSED_STEM_BWD", "XM_NO_STEM_GRAM", "XM_NO_STEM_FWD", "XM_NO_FORK_SUMS", "XM_NO_FUSED_SE", "XM_NO_SE_FOLD_FC", "XM_NO_FUSED_SE_BWD", "XM_NO_PREPARE",
            "XM_WGRAD_AFTER_DGRAD"]


def _run(tmp_path, tag, env_extra):
    out = str(tmp_path / (tag + ".npz"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("XM_NO_") and k not in ("XM_DGRAD_MERGE", "XM_WGRAD_AFTER_DGRAD")}
    env["XM_TUNE_FILE"] = ""          # find mode: the shipped table may name a configuration the selector removes
    env.update(env_extra)
    r = subprocess.run([sys.executable, WORKER, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, "%s: %s" % (tag, r.stderr.decode()[-2000:])
    return dict(np.load(out))


def test_selector_lists_match_the_library():
    """the lists above are the selectors the sources read (a new one must be added here)"""
    import re
    src = open(os.path.join(ROOT, "mcncrossmodalemotions_amd", "csrc", "context.cpp")).read()
    names = set(re.findall(r'"(XM_[A-Z0-9_]+)"', src[src.index("kEnv[kPathCount]"):src.index("static bool on[")]))
    assert names == set(LIBRARY), names ^ set(LIBRARY)
    dag = open(os.path.join(ROOT, "mcncrossmodalemotions_amd", "dagnn.py")).read()
    ex = set(re.findall(r'os\.environ\.get\("(XM_(?:NO_|WGRAD_AFTER)[A-Z0-9_]+)"\)', dag)) - {"XM_NO_FUSED_STATS"}
    assert ex == set(EXECUTOR), ex ^ set(EXECUTOR)


@pytest.mark.gpu
def test_every_selector_gives_the_default_results(gpu, tmp_path):
    base = _run(tmp_path, "default", {})
    assert any(k.startswith("der_") for k in base) and any(k.startswith("jder_") for k in base)
    worst = {}
    group = {}
    for k, ref in base.items():
        group[k.split("_")[0]] = max(group.get(k.split("_")[0], 1e-6), float(np.abs(ref).max()))
    for name in LIBRARY + EXECUTOR:
        got = _run(tmp_path, name, {name: "1"})
        assert got.keys() == base.keys(), name
        w = 0.0
        for k, ref in base.items():
            # (the bias derivatives in front of a training-mode bnorm are exactly zero in exact arithmetic: what is left is
            # rounding noise of the pass, measured against the group's largest derivative instead of against itself)
            scale = max(float(np.abs(ref).max()), 1e-3 * group[k.split("_")[0]])
            err = float(np.abs(got[k] - ref).max()) / scale
            tol = 0.1 if k.startswith(("der_", "jder_")) else 2e-4
            assert err < tol, "%s changes %s by %.2e of its magnitude" % (name, k, err)
            w = max(w, err)
        worst[name] = w
    print("largest relative difference per selector:", {k: "%.1e" % v for k, v in worst.items()})
