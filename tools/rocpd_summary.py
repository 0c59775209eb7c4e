#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into the per-kernel summary
committed under profiles/ (rocprofv3 in this ROCm 7.2 image writes .db, not CSV, by default)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
grand = sum(r[2] for r in rows) or 1
print(f"# source: {sys.argv[1]} (rocprofv3 --kernel-trace --stats); durations in ns")
print(f"{'kernel':<110} {'calls':>6} {'total_ns':>14} {'avg_ns':>12} {'min_ns':>10} {'max_ns':>10} {'pct':>7}")
for name, calls, total, avg, mn, mx in rows:
    print(f"{name[:110]:<110} {calls:>6} {int(total):>14} {avg:>12.1f} {mn:>10} {mx:>10} {100.0 * total / grand:>7.2f}")
regs = db.execute("select distinct name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x from kernels").fetchall()
print("\n# per-kernel resources (from the dispatch records)")
for r in sorted(regs):
    print(f"{r[0][:90]:<90} vgpr={r[1]} agpr={r[2]} sgpr={r[3]} lds={r[4]} scratch={r[5]} wg={r[6]} grid={r[7]}")
