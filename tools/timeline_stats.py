#!/usr/bin/env python
"""Per-stream view of an overlapped step from tools/timeline_dump.py's CSV: busy time, idle gaps, kernel time by name.
usage: python tools/timeline_stats.py tl.csv [step_ms]"""
import sys
from collections import defaultdict

rows = []
for line in open(sys.argv[1]):
    a, b, q, s, nm = line.rstrip("\n").split(",", 4)
    rows.append((int(a), int(b), q, s, nm))
t0 = min(r[0] for r in rows)
t1 = max(r[1] for r in rows)
wall = (t1 - t0) / 1e6
print("window %.2f ms, %d dispatches" % (wall, len(rows)))
by = defaultdict(list)
for r in rows:
    by[r[3]].append(r)
for s, rs in sorted(by.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1])):
    rs.sort()
    busy = sum(r[1] - r[0] for r in rs) / 1e6
    gaps = [(rs[i + 1][0] - rs[i][1]) / 1e3 for i in range(len(rs) - 1)]
    small = [g for g in gaps if 0 <= g < 30]
    print("stream %s: %d kernels, busy %.2f ms (%.1f %% of window), gaps<30us: n=%d sum %.2f ms median %.1f us; "
          "gaps>=30us: n=%d sum %.2f ms" % (s, len(rs), busy, 100 * busy / wall, len(small), sum(small) / 1e3,
                                            sorted(small)[len(small) // 2] if small else 0,
                                            len([g for g in gaps if g >= 30]), sum(g for g in gaps if g >= 30) / 1e3))
    agg = defaultdict(lambda: [0, 0.0])
    for r in rs:
        agg[r[4]][0] += 1
        agg[r[4]][1] += (r[1] - r[0]) / 1e6
    for nm, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print("     %-70s x%-4d %.3f ms  avg %.1f us" % (nm[:70], c, ms, ms / c * 1e3))
