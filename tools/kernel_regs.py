#!/usr/bin/env python
"""VGPR / AGPR / SGPR / spill / LDS / occupancy of every kernel of a csrc/*.hip file (compiles it with
-Rpass-analysis=kernel-resource-usage; nothing is written).  usage: python tools/kernel_regs.py conv.hip [filter] [-D...]"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(HERE, "mcncrossmodalemotions_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-D") else ""
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src, "-o",
       "/dev/null", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"] + defs
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE).stderr.decode()
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: .*?\s{2,}([A-Za-z][A-Za-z \[\]/]+): (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for n, r in rows.items():
    try:
        dem = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n]).decode().strip()
    except Exception:
        dem = n
    dem = re.sub(r"\(.*", "", dem).replace("void xm::", "")
    if flt not in dem:
        continue
    print("%-58s vgpr %3s agpr %3s sgpr %3s spill v%s s%s scratch %s occ %s lds %s" % (
        dem[:58], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("VGPR Spill"), r.get("SGPR Spill"),
        r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
