#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
rows = db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall() if "kernel_name" in cols else []
if not rows:
    print("columns:", cols)
    sys.exit(0)
acc = defaultdict(lambda: defaultdict(list))
for k, c, v, d in rows:
    acc[k][c].append(v)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if flt and flt not in k:
        continue
    print(k[:100])
    for c in sorted(acc[k]):
        vals = acc[k][c]
        print(f"    {c:<34} avg {sum(vals) / len(vals):>16.1f}   n={len(vals)}")
