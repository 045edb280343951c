#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel launch.  usage: pmc_avg.py <dir-with-*counter_collection.csv> [regex]"""
import csv, glob, re, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void |\(.*$", "", r["Kernel_Name"])
        if pat and not pat.search(k):
            continue
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k in sorted(rows):
    n = len(cnt[k])
    print("%s  (launches %d)" % (k, n))
    for c in sorted(rows[k]):
        print("    %-34s %16.0f" % (c, rows[k][c] / n))
