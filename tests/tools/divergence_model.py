#!/usr/bin/env python3
"""How much of k_indirect's wave time is lost to trip-count divergence, and what would re-grouping buy?

The CPU oracle records, for every pixel, the traversal work of each ray of the indirect pass in call
order (inner-node visits, triangle tests, instance entries per traverse_top / stand-alone
traverse_bottom call; the GPU performs the same visits per lane, minus the folded leaf navigators).
A wave executes a traversal until its slowest lane is done, so the model charges each 8x8 tile
max-over-lanes of the per-lane work and compares it with
  * two pixels per lane, traced back to back in one loop (tile 8x16): max over lanes of the SUM,
  * four pixels per lane (tile 16x16),
  * the bound of perfect re-grouping: total work / 64.
Usage: python tests/tools/divergence_model.py [width height bounces frames]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bevy_hikari_amd as hk
from oracle_lib import oracle_plugin

# VALU instructions per event on the GPU (ISA of k_indirect<true,false,true>): node step, triangle test, instance entry
C_NODE, C_TRI, C_ENTRY = 35.0, 60.0, 100.0

if __name__ == "__main__":
    w, h, bounces, frames = (int(a) for a in (sys.argv[1:5] + ["960", "544", "2", "12"][len(sys.argv) - 1:]))
    p = oracle_plugin()
    p.set_scene(hk.load_cornell())
    s = hk.HikariSettings(indirect_bounces=bounces, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = hk.cornell_camera(w, h)
    slots = 3 * bounces
    rec = np.zeros((h, w, slots, 3), dtype=np.uint16)
    fn = p.engine.api.dll.orc_debug_record_steps
    fn.argtypes, fn.restype = [C.c_void_p, C.c_void_p, C.c_uint32], C.c_int
    for n in range(1, frames + 1):
        if n == frames:
            assert fn(p.engine.ctx, rec.ctypes.data, slots) == 0
        p.render(cam, s, frame_number=n)
    fn(p.engine.ctx, None, 0)
    work = rec[..., 0] * C_NODE + rec[..., 1] * C_TRI + rec[..., 2] * C_ENTRY      # [h][w][slots]
    active = rec.reshape(h, w, -1).any(axis=2)
    names = ["closest", "emitter-blas", "shadow"]
    out = {"size": [w, h], "bounces": bounces, "geometry_pixels": float(active.mean())}
    for k in range(slots):
        r = rec[..., k, :][active]
        out[f"b{k // 3}:{names[k % 3]}"] = {"traced": float((r.sum(axis=1) > 0).mean()), "nodes_mean": float(r[:, 0].mean()),
                                           "nodes_p99": float(np.percentile(r[:, 0], 99)), "tris_mean": float(r[:, 1].mean()),
                                           "entries_mean": float(r[:, 2].mean())}

    def tiles(a, th, tw):      # [h][w][slots] -> [ntiles][th*tw][slots]
        hh, ww = a.shape[0] // th * th, a.shape[1] // tw * tw
        a = a[:hh, :ww]
        return a.reshape(hh // th, th, ww // tw, tw, -1).transpose(0, 2, 1, 3, 4).reshape(-1, th * tw, a.shape[2])

    total = float(work.sum())
    base = tiles(work, 8, 8)                                                       # one pixel per lane
    cost1 = float(base.max(axis=1).sum())
    t2 = tiles(work, 16, 8).reshape(-1, 2, 64, slots).sum(axis=1)                  # lanes pair (x, y) with (x, y + 8)
    cost2 = float(t2.max(axis=1).sum())
    t4 = tiles(work, 16, 16)                                                       # 2x2 tiles of 8x8 -> lane sums its 4 pixels
    t4 = t4.reshape(-1, 2, 8, 2, 8, slots).transpose(0, 2, 4, 1, 3, 5).reshape(-1, 64, 4, slots).sum(axis=2)
    cost4 = float(t4.max(axis=1).sum())
    out["wave_cost_model"] = {"lane_utilisation_now": total / (64.0 * cost1), "two_pixels_per_lane_speedup": cost1 / cost2,
                              "four_pixels_per_lane_speedup": cost1 / cost4, "perfect_regrouping_speedup": cost1 / (total / 64.0)}
    print(json.dumps(out, indent=1))
