"""CPU: the product path has no CPU fallback and never touches the oracle.
  * bench.py on a host without a GPU exits with a clear message instead of timing anything;
  * no module of the product package imports `oracle` (the oracle is test infrastructure: only tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it);
  * a vl operator called on host tensors is rejected (no silent host computation)."""
import ast
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mcncrossmodalemotions_amd")


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)
    assert '"metric"' not in r.stdout          # nothing that looks like a result line


def test_product_package_never_imports_the_oracle():
    offenders = []
    for name in sorted(os.listdir(PKG)):
        if not name.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(PKG, name)).read(), name)
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or ""]
            if any(m == "oracle" or m.startswith("oracle.") for m in mods):
                offenders.append((name, node.lineno))
    assert not offenders, offenders
    # the library's sources include nothing from oracle/, and the built library does not link the oracle
    import re
    for name in sorted(os.listdir(os.path.join(PKG, "csrc"))):
        if name.endswith((".hip", ".cpp", ".h")):
            txt = open(os.path.join(PKG, "csrc", name)).read()
            assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', txt), name
    from mcncrossmodalemotions_amd import _lib, build
    build.build()
    dyn = subprocess.run(["readelf", "-d", _lib.SO_PATH], capture_output=True, text=True).stdout
    assert "NEEDED" in dyn and "oracle" not in dyn, dyn


def test_operators_reject_host_tensors():
    import torch
    from mcncrossmodalemotions_amd import vl
    x = torch.zeros(4, 4, 2, 1).permute(3, 2, 1, 0).contiguous().permute(3, 2, 1, 0)
    with pytest.raises(Exception):
        vl.vl_nnrelu(x)
