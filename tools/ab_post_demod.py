#!/usr/bin/env python3
"""A/B in one process: demodulation on the post stream (1) or on the main stream (0) - full frame and bands of configs 2 / 4, interleaved."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from band_anatomy import Runner
from bevy_hikari_amd import _ffi as F
out = {}
for config, balanced in ((2, False), (4, True)):
    K = 48 if config == 2 else 8
    r = Runner(config, 0)
    r.frames(12)
    bounds = r.balanced_bounds(8) if balanced else None
    res = {}
    for what in ("full", 0, 3, 7):
        if what == "full":
            r.e.set_band(0, 1); r.e.set_band_bounds(None); r.frames(4)
        else:
            r.to_band(what, 8, bounds)
        t = {0: [], 1: []}
        for rep in range(4):
            for mode in (0, 1):
                r.e.set_debug_option(F.DEBUG_OPT_POST_DEMODULATION, mode)
                r.frames(4)
                t[mode].append(r.wall(K))
        res[str(what)] = {"demod_on_main_ms": round(min(t[0]), 4), "demod_on_post_ms": round(min(t[1]), 4), "all": {k: [round(x, 4) for x in v] for k, v in t.items()}}
    out[str(config)] = res
print(json.dumps(out, indent=1))
