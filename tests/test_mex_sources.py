"""The MEX gateways under mex/ cannot be built here (no MATLAB).  This test keeps them honest at the level that is
possible: every gateway must be valid C++ against the C ABI header as it is today (g++ -fsyntax-only with a
declaration-only stand-in for mex.h), and every ABI function a gateway calls must be exported by the built library."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mex_gateways_are_valid_cxx_against_the_abi():
    srcs = sorted(glob.glob(os.path.join(ROOT, "mex", "*.cpp")))
    names = {os.path.basename(s)[:-4] for s in srcs}
    # the three compiled MatConvNet operators + the five M-file operators of the hot path + the device helper
    assert {"vl_nnconv", "vl_nnpool", "vl_nnbnorm", "vl_nnrelu", "vl_nnsigmoid", "vl_nnsoftmaxt", "vl_nnsoftmaxceloss",
            "vl_nnloss", "xm_device"} <= names
    for s in srcs:
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=return-type",
                            "-I" + os.path.join(ROOT, "tests", "mex_stub"), "-I" + os.path.join(ROOT, "include"),
                            "-I" + os.path.join(ROOT, "mex"), s], capture_output=True, text=True)
        assert r.returncode == 0, "%s\n%s" % (s, r.stderr)


def test_mex_gateways_call_only_exported_symbols():
    from mcncrossmodalemotions_amd import _lib
    used = set()
    for s in glob.glob(os.path.join(ROOT, "mex", "*")):
        if os.path.isfile(s):
            used |= set(re.findall(r"\b(xm_[a-z0-9_]+)\s*\(", open(s).read()))
    used -= {"xm_check", "xm_intvec", "xm_streq", "xm_ignored_option", "xm_device", "xm_mex", "xm_mex_startup"}
    missing = sorted(u for u in used if u not in _lib.SIGNATURES)
    assert not missing, missing
