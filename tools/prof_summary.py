#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats results.db (sqlite) as a text table.
usage: python tools/prof_summary.py <results.db> [steps]   (steps: launches-per-step divisor)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
print("%-100s %8s %12s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
for name, calls, total, avg, pct in rows:
    print("%-100s %8d %12.3f %10.2f %6.2f" % (name[:100], calls, total / 1e3, avg, pct))
print("TOTAL kernel time: %.3f ms  (%.3f ms per step over %g steps)" % (tot / 1e3, tot / 1e3 / steps, steps))
