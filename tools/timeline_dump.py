#!/usr/bin/env python
"""Dump the tail of a rocprofv3 --kernel-trace results.db as CSV (start_ns, end_ns, queue, stream, kernel name) for
off-box analysis (tools/timeline_stats.py).  usage: python tools/timeline_dump.py <results.db> <out.csv> [tail_ms=200]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
out = sys.argv[2]
tail_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 200.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print("tables:", tabs, file=sys.stderr)
for t in tabs:
    if "kernel" in t.lower():
        print(t, [c[1] for c in db.execute("pragma table_info('%s')" % t)], file=sys.stderr)
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t.lower()][0]
cols = [c[1] for c in db.execute("pragma table_info('%s')" % view)]
q = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else "0")
s = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else "0")
n = "name" if "name" in cols else "kernel_name"
rows = db.execute("select start, end, %s, %s, %s from '%s' order by start" % (q, s, n, view)).fetchall()
t1 = max(r[1] for r in rows)
lo = t1 - tail_ms * 1e6
with open(out, "w") as f:
    for a, b, qq, ss, nm in rows:
        if b >= lo:
            f.write("%d,%d,%s,%s,%s\n" % (a - int(lo), b - int(lo), qq, ss, str(nm).replace(",", ";")[:90]))
print("wrote", out, file=sys.stderr)
