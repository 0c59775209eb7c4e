#!/usr/bin/env python3
"""A/B in one process: the main stream joins the side stream at the end of every frame (1: rounds 1-5) or not at all (0: round 6) -
full frames of configs 2 / 3 / 4 / 5 and bands of the 8-way split of configs 2 / 4, interleaved; the last frame's image must be the same."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from band_anatomy import Runner
from bevy_hikari_amd import _ffi as F
out = {}
for config in (2, 3, 4, 5):
    K = 48 if config == 2 else 8
    r = Runner(config, 0)
    r.frames(12)
    bounds = r.balanced_bounds(8) if config == 4 else None
    res = {}
    for what in (("full", 0, 3, 7) if config in (2, 4) else ("full",)):
        if what == "full":
            r.e.set_band(0, 1); r.e.set_band_bounds(None); r.frames(4)
        else:
            r.to_band(what, 8, bounds)
        t = {0: [], 1: []}
        for rep in range(4):
            for mode in (1, 0):
                r.e.set_debug_option(F.DEBUG_OPT_SIDE_JOIN, mode)
                r.frames(4)
                t[mode].append(r.wall(K))
        res[str(what)] = {"join_every_frame_ms": round(min(t[1]), 4), "no_join_ms": round(min(t[0]), 4), "all": {k: [round(x, 4) for x in v] for k, v in t.items()}}
    out[str(config)] = res
print(json.dumps(out, indent=1))
