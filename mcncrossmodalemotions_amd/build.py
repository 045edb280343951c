"""Builds libxmodal_hip.so in-tree for gfx950 with hipcc (no JIT cache, no cmake)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# experiments: XM_BUILD_TAG=v1 XM_DEFINES="-DXM_VARIANT=1" python -m mcncrossmodalemotions_amd.build  builds
# libxmodal_hip_v1.so next to the product library; XM_LIB_PATH=<that file> makes _lib.py load it (A/B runs in one
# gpurun call).  The product build uses neither.
TAG = os.environ.get("XM_BUILD_TAG", "")
OUT = os.path.join(HERE, "libxmodal_hip%s.so" % ("_" + TAG if TAG else ""))
SOURCES = ["context.cpp", "conv.hip", "norm_pool.hip", "misc.hip", "comm.cpp"]
HEADERS = ["xm_common.h", "conv_kernels.h", "stem_pool_kernels.h", os.path.join("..", "..", "include", "xmodal.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if os.environ.get("XM_DEBUG_CYCLES"):   # per-block clock trace in the conv kernels (tools/conv_bench.py --cycles)
    FLAGS.append("-DXM_DEBUG_CYCLES")
FLAGS += os.environ.get("XM_DEFINES", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "_obj" + ("_" + TAG if TAG else ""))
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + ["-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs +
            ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
