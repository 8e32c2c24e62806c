#!/usr/bin/env python3
"""Print per-kernel stats and the timeline of the last steps from a rocprofv3 rocpd (sqlite) results file."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 14"):
    print(f"{r[0][:70]:70s} calls={r[1]:5d} avg_us={r[3] / 1e3:9.2f} pct={r[4]:.1f}")
rows = list(db.execute("select name,start,end from kernels order by start"))
sel = rows[-2 * n_last:-n_last]
base = sel[0][1]
prev_end = base
for n, s, e in sel:
    print(f"{(s - base) / 1e3:9.1f} gap={(s - prev_end) / 1e3:6.1f} dur={(e - s) / 1e3:7.1f} {n[:60]}")
    prev_end = e
