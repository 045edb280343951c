"""CPU: the C-ABI library builds, loads and exports every symbol include/xmodal.h declares (no
compute calls -- there is no GPU here), and argument validation works without touching a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(xm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mcncrossmodalemotions_amd import _lib, build
    build.build()
    lib = C.CDLL(_lib.SO_PATH)
    names = _declared("xmodal.h") + _declared("xmodal_prof.h")
    assert len(names) > 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes table covers the whole drop-in header
    untyped = [n for n in _declared("xmodal.h") if n not in _lib.SIGNATURES]
    assert not untyped, untyped


def test_error_reporting_without_gpu():
    from mcncrossmodalemotions_amd import _lib
    L = _lib.load()
    assert L.xm_version() >= 101
    # an exchange without a communicator is an error (a worker that skipped xm_comm_init must not train on
    # derivatives that were silently not exchanged); after a single-worker init it is the documented no-op
    assert L.xm_parserv_sync(None) == 1 and b"no communicator" in L.xm_last_error()
    assert L.xm_allreduce_sum_f32(None, 4, None) == 1
    assert L.xm_comm_init(None, 0, 1) == 0
    assert L.xm_parserv_sync(None) == 0 and L.xm_parserv_push(None, 0, None) == 0
    assert L.xm_comm_destroy() == 0
    assert L.xm_parserv_push(None, 4, None) == 1
    assert L.xm_out_size(512, 1, 1, 7, 1, 2) == 254
    # execution hint: explicit, process-wide, unknown bits rejected
    assert L.xm_get_exec_hint() == 0 and L.xm_set_exec_hint(1) == 0 and L.xm_get_exec_hint() == 1
    assert L.xm_set_exec_hint(6) == 1 and b"unknown flag" in L.xm_last_error() and L.xm_get_exec_hint() == 1
    assert L.xm_set_exec_hint(0) == 0
    # invalid arguments are rejected before any device work, with a MATLAB-style message
    rc = L.xm_nnconv_forward(None, 8, 8, 4, 2, None, 3, 3, 3, 5, None, None, 1, 1, 0, 0, 0, 0, 1, 1, None)
    assert rc == 1 and b"does not divide" in L.xm_last_error()
    rc = L.xm_nnconv_forward(None, 4, 4, 1, 1, None, 9, 9, 1, 1, None, None, 1, 1, 0, 0, 0, 0, 1, 1, None)
    assert rc == 1 and b"larger than padded input" in L.xm_last_error()
    rc = L.xm_nnpool_forward(None, 8, 8, 1, 1, 3, 3, 2, 2, 3, 0, 0, 0, 0, None, None)
    assert rc == 1 and b"pad must be smaller" in L.xm_last_error()
    rc = L.xm_nnsoftmaxceloss(None, None, 100, 4, C.c_float(2.0), 1, None, None, None, None)
    assert rc == 5  # XM_ENOTSUP: more than 64 classes
    with pytest.raises(_lib.XmError):
        _lib.check(rc)


def test_product_path_has_no_cpu_fallback():
    """operators refuse CPU tensors instead of silently computing elsewhere."""
    import torch
    from mcncrossmodalemotions_amd import vl
    with pytest.raises(RuntimeError, match="no CPU path"):
        vl.vl_nnrelu(torch.zeros(2, 2, 1, 1).permute(3, 2, 1, 0))
    # and nothing under the package imports the oracle
    pkg = os.path.join(ROOT, "mcncrossmodalemotions_amd")
    import ast
    for fn in os.listdir(pkg):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or ""]
            assert not any(m.split(".")[0] == "oracle" for m in mods), (fn, mods)


def test_shipped_tuning_table_matches_the_library():
    """mcncrossmodalemotions_amd/tune_gfx950.txt must carry the revision / configuration count of the library it
    ships with (a kernel change bumps XM_TUNE_REV; a stale table would be ignored silently and every process
    would re-tune): the library's own loader accepts it, and it covers the default bench step."""
    from mcncrossmodalemotions_amd import _lib
    L = _lib.load()
    path = os.path.join(ROOT, "mcncrossmodalemotions_amd", "tune_gfx950.txt")
    assert os.path.exists(path)
    first = L.xm_tune_load(path.encode())
    tot, new = C.c_int(), C.c_int()
    assert L.xm_tune_entries(C.byref(tot), C.byref(new)) == 0
    # either this call or an earlier lookup in this process loaded it; nothing in it was measured here
    assert tot.value >= 300 and new.value == 0 and first in (0, tot.value), (first, tot.value, new.value)
    rows = [l.split() for l in open(path).read().splitlines()[1:]]
    # student conv2 forward at 32 samples: kind 0, M 256, NP 62*36*32, Rp 2400
    assert any(r[:4] == ["0", "256", str(62 * 36 * 32), "2400"] for r in rows)
