#!/usr/bin/env python3
"""Per-frame timeline of a rocprofv3 rocpd database (--kernel-trace): for the last timed frames, every dispatch with its start
relative to the frame's first kernel, its duration, the queue / stream it ran on and the gap to the previous dispatch on the same
queue - where the time between kernels goes.   Usage: python tools/timeline.py results.db [frames]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
start = "start" if "start" in cols else "start_timestamp"
end = "end" if "end" in cols else "end_timestamp"
queue = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute(f"select name, {start}, {end}, {queue} from kernels order by {start}").fetchall()
# frames begin with k_prepass
firsts = [i for i, r in enumerate(rows) if "k_prepass" in r[0]]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for f in firsts[-want - 1:-1]:
    nxt = firsts[firsts.index(f) + 1]
    t0 = rows[f][1]
    print(f"--- frame starting at dispatch {f}: {(rows[nxt][1] - t0) / 1000.0:.1f} us to the next frame's first kernel")
    last_end = {}
    busy = 0
    for name, s, e, q in rows[f:nxt]:
        gap = (s - last_end[q]) / 1000.0 if q in last_end else 0.0
        last_end[q] = e
        short = name.replace("void hkd::", "").replace("hkd::", "").split("(")[0][:44]
        print(f"  q{q} +{(s - t0) / 1000.0:8.1f} us  {(e - s) / 1000.0:7.1f} us  gap {gap:6.1f}  {short}")
