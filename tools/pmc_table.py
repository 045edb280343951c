#!/usr/bin/env python
"""Merge rocprofv3 --pmc passes (separate directories) into one per-kernel table of per-launch
averages + a JSON of HBM traffic per launch.
usage: pmc_table.py <sq_dir> <fetch_dir> <write_dir> <out.txt> <out.json> [commit]
FETCH_SIZE / WRITE_SIZE are KiB as reported by rocprofv3; per MI355X_MICROARCH.md ("HBM") the gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x, so the table carries fetch x 2 as well and the
JSON traffic figure uses the doubled value (upper estimate for the dword gathers)."""
import collections, csv, glob, json, re, sys

def load(d):
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void |\(.*$", "", r["Kernel_Name"])
            rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(cnt[k]) for c, v in rows[k].items()} | {"_n": len(cnt[k])} for k in rows}

def durations(d):
    """average kernel duration (us) from the kernel trace written next to a counter pass"""
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void |\(.*$", "", r["Kernel_Name"])
            tot[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
            cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


sq, fe, wr = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
dur = durations(sys.argv[2])
names = sorted(sq, key=lambda k: -sq[k].get("SQ_BUSY_CU_CYCLES", 0) * sq[k]["_n"])
out = ["# rocprofv3 --pmc, per-launch averages per kernel (three separate passes: SQ | FETCH_SIZE | WRITE_SIZE)",
       "# mfma% = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); wait% of SQ_WAVE_CYCLES (quad-cycles)",
       "# HBM_GB/s = (2 x FETCH_SIZE + WRITE_SIZE) / kernel duration in the FETCH pass (kernels run alone there)",
       "%-46s %7s %7s %9s %10s %11s %11s %10s %9s %9s" % ("kernel", "launch", "mfma%", "wait_any%", "wait_inst%",
                                                         "fetch_MB", "fetch_x2_MB", "write_MB", "avg_us", "HBM_GB/s")]
traffic = {}
for k in names:
    s = sq[k]
    busy = s.get("SQ_BUSY_CU_CYCLES", 0.0)
    wave = s.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    mf = 100.0 * s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy) if busy else 0.0
    f = fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 / 1e6
    w = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
    us = dur.get(k, 0.0)
    gbs = (2 * f + w) * 1e6 / (us * 1e-6) / 1e9 if us > 0 else 0.0
    out.append("%-46s %7d %7.1f %9.1f %10.1f %11.1f %11.1f %10.1f %9.1f %9.0f" % (
        k[:46], s["_n"], mf, 100 * s.get("SQ_WAIT_ANY", 0) / wave, 100 * s.get("SQ_WAIT_INST_ANY", 0) / wave, f, 2 * f, w,
        us, gbs))
    traffic[k] = {"fetch_bytes_x2": 2 * f * 1e6, "write_bytes": w * 1e6, "launches": s["_n"]}
open(sys.argv[4], "w").write("\n".join(out) + "\n")
# the figures belong to the build they were measured on: bench.py reports this commit next to roofline.traffic
traffic["_meta"] = {"commit": sys.argv[6] if len(sys.argv) > 6 else None}
json.dump(traffic, open(sys.argv[5], "w"), indent=1)
