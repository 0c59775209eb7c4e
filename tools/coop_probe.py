#!/usr/bin/env python3
"""Round 5: dependent divergent record fetches, every lane its own 128-B record (8 lane-loads) against the same records fetched
cooperatively (8 consecutive lanes load the 8 pieces of one lane's record: 8 lines per wave-level load instead of 64) - hk_measure_gather
modes 128 / 129, by footprint and occupancy.  Prints one JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bevy_hikari_amd as hk  # noqa: E402

e = hk.Engine(device=0)
out = {}
for fp_name, fp in (("16KiB_L1", 16 << 10), ("1MiB_L2", 1 << 20), ("64MiB_infinity_cache", 64 << 20), ("640MiB_hbm", 640 << 20)):
    for waves in (2, 5, 8):
        row = {}
        for mode, name in ((64, "own_64B_4_loads"), (128, "own_128B_8_loads"), (129, "cooperative_128B")):
            gl, gb = e.measure_gather(fp, mode, waves, 256)
            loads = 8 if mode >= 128 else mode // 16
            row[name] = {"g_lane_records_s": round(gl / loads * 64, 2), "gbytes_s": round(gb, 1)}
        out[f"{fp_name}_{waves}_waves_per_simd"] = row
print(json.dumps(out, indent=1))
