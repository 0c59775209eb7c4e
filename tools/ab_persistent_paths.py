#!/usr/bin/env python3
"""A/B in one process: the queue-based indirect pass as ONE launch for every bounce (HK_DEBUG_OPT_PERSISTENT_PATHS = 1) against one
trace + one shade launch per bounce (0) - configs 3 / 4: the full frame (product streams), the indirect pass alone (HIP events of a
timed pass), and bands of an 8-way split (where every band pays every stage's end), interleaved.

    python tools/ab_persistent_paths.py [--configs 3 4] > profiles/r06_persistent_paths_ab.json"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from band_anatomy import Runner
from bevy_hikari_amd import _ffi as F

ap = argparse.ArgumentParser()
ap.add_argument("--configs", type=int, nargs="+", default=[3, 4])
ap.add_argument("--bands", type=int, nargs="*", default=[0, 3, 7])
ap.add_argument("--reps", type=int, default=4)
args = ap.parse_args()
out = {}
for config in args.configs:
    K = 12 if config == 3 else 8
    r = Runner(config, 0)
    r.frames(8)
    bounds = r.balanced_bounds(8)
    res = {"workload": r.description, "bounds_8": bounds}
    for what in ["full"] + list(args.bands):
        if what == "full":
            r.e.set_band(0, 1); r.e.set_band_bounds(None); r.frames(4)
        else:
            r.to_band(what, 8, bounds)
        t = {0: [], 1: []}
        p = {0: [], 1: []}
        for rep in range(args.reps):
            for mode in (0, 1):
                r.e.set_debug_option(F.DEBUG_OPT_PERSISTENT_PATHS, mode)
                r.frames(3)
                t[mode].append(r.wall(K))
                if rep < 2:
                    p[mode].append(r.passes(4).get("indirect_lit_ambient"))
        res[str(what)] = {"staged_ms": round(min(t[0]), 4), "one_launch_ms": round(min(t[1]), 4), "one_launch_over_staged": round(min(t[1]) / min(t[0]), 4),
                          "indirect_pass_alone_ms": {"staged": min(p[0]), "one_launch": min(p[1])},
                          "all_ms": {("one_launch" if k else "staged"): [round(x, 4) for x in v] for k, v in t.items()}}
    # every dispatch alone on the GPU (one stream): what the launch itself costs, apart from what it leaves to the other streams
    del r
    r = Runner(config, F.CTX_SINGLE_STREAM)
    r.frames(6)
    alone = {}
    for what in ["full"] + list(args.bands):
        if what == "full":
            r.e.set_band(0, 1); r.e.set_band_bounds(None); r.frames(3)
        else:
            r.to_band(what, 8, bounds)
        p = {0: [], 1: []}
        for rep in range(3):
            for mode in (0, 1):
                r.e.set_debug_option(F.DEBUG_OPT_PERSISTENT_PATHS, mode)
                r.frames(2)
                p[mode].append(r.passes(4).get("indirect_lit_ambient"))
        alone[str(what)] = {"staged": min(p[0]), "one_launch": min(p[1]), "one_launch_over_staged": round(min(p[1]) / min(p[0]), 4)}
    res["indirect_pass_on_one_stream_ms"] = alone
    out[str(config)] = res
    del r
print(json.dumps(out, indent=1))
