#!/usr/bin/env python3
"""Wall clock per frame of one band rendered alone (product mode): python tools/band_wall.py --config 2 --band 3 --bands 8 [--balanced]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from band_anatomy import Runner

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--band", type=int, nargs="+", default=[3])
ap.add_argument("--bands", type=int, default=8)
ap.add_argument("--balanced", action="store_true")
ap.add_argument("--frames", type=int, default=None)
a = ap.parse_args()
K = a.frames or (48 if a.config in (2, 5) else 8)
r = Runner(a.config, 0)
r.frames(12)
full = r.wall(K)
bounds = r.balanced_bounds(a.bands) if a.balanced else None
out = {"config": a.config, "full_ms": round(full, 4), "bounds": bounds, "band_ms": {}}
for b in a.band:
    r.to_band(b, a.bands, bounds)
    out["band_ms"][b] = round(min(r.wall(K) for _ in range(3)), 4)
print(json.dumps(out))
