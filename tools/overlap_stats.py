#!/usr/bin/env python
"""How busy is the GPU in the overlapped step?  From a rocprofv3 --kernel-trace results.db: over the last `frac` of
the trace, the share of wall time with >= 1 kernel resident, the time-weighted number of resident kernels, and the
sum of kernel durations per wall second (how much kernel time the overlap packs into one second).
usage: python tools/overlap_stats.py <results.db> [frac=0.5]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if "kernel_dispatch" in t.lower() or t.lower() == "kernels"]
rows = None
for t in cand:
    cols = [c[1] for c in db.execute("pragma table_info('%s')" % t)]
    if "start" in cols and "end" in cols:
        rows = db.execute("select start, end from '%s'" % t).fetchall()
        break
if rows is None:
    print("no start/end table; tables:", tabs)
    sys.exit(1)
rows = sorted((int(a), int(b)) for a, b in rows if b > a)
t0, t1 = rows[0][0], max(b for _, b in rows)
lo = t1 - (t1 - t0) * frac
ev = []
for a, b in rows:
    if b <= lo:
        continue
    ev.append((max(a, lo), 1))
    ev.append((b, -1))
ev.sort()
busy = conc = 0.0
cur, last = 0, lo
hist = {}
for t, d in ev:
    if cur > 0:
        busy += t - last
    conc += cur * (t - last)
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d
    last = t
wall = t1 - lo
print("window %.1f ms: >= 1 kernel resident %.1f %% of the time; resident kernels (time-weighted) %.2f; "
      "kernel-time per wall-time %.3f" % (wall / 1e6, 100 * busy / wall, conc / max(busy, 1), conc / wall))
print("share of time by number of resident kernels:", {k: round(100 * v / wall, 1) for k, v in sorted(hist.items())})
