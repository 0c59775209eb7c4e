#!/usr/bin/env python3
"""Bulk / tail split of the trace stages of the queue-based indirect pass: with HK_DEBUG_OPT_WF_TIMELINE the trace
kernel's instrumented twin records per stage when the ray queue ran dry, when the last persistent wave left and how long the rays'
walks were.  Prints one JSON object per config.    python tools/wf_timeline.py [--no-wide-walk] [3 4]
(default: the wide kernel k_wf_trace_wide, whose "node steps" are 128-B records of two tree levels and whose long walks are those of
>= 128 records; --no-wide-walk: the skip-link kernel k_wf_trace, long walks >= 256 node steps)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bevy_hikari_amd as hk  # noqa: E402
from bench import workload  # noqa: E402


def main():
    no_wide = "--no-wide-walk" in sys.argv
    configs = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [3, 4]
    persistent = 0 if "--staged" in sys.argv else (1 if "--one-launch" in sys.argv else -1)   # HK_DEBUG_OPT_PERSISTENT_PATHS (-1: the library's rule)
    out = {}
    for cfg in configs:
        scene, camera, settings, lights, description = workload(hk, cfg, None, None, None)
        e = hk.Engine(device=0, flags=256 if no_wide else 0)
        e.set_debug_option(2, 1)   # HK_DEBUG_OPT_WF_TIMELINE
        e.set_debug_option(8, persistent)
        e.upload_noise(); e.upload_scene(scene); e.resize(camera.width, camera.height, 1.0)
        view, pview, sc = camera.view_uniform(), camera.previous_view_uniform(), settings.to_c()
        for n in range(1, 9):
            e.frame_render(hk.frame_uniform(settings, n), view, pview, lights, sc)
        e.wait()
        raw = np.zeros(64 * 32, dtype=np.uint64)
        e.api.call("debug_read_wf_timeline", e.ctx, raw.ctypes.data_as(C.POINTER(C.c_uint64)), raw.size)
        raw = raw.reshape(64, 32)
        tick_us = 1e3 / float(raw[63, 31])   # wall_clock64 rate in kHz -> microseconds per tick
        stages = []
        inv = np.uint64(0xFFFFFFFFFFFFFFFF)
        for s in range(settings.indirect_bounces + 1):
            r = raw[s]
            if r[4] == 0:
                continue
            t0, tdry, tend = int(inv - r[0]), int(inv - r[1]), int(r[2])
            total, bulk = (tend - t0) * tick_us, (tdry - t0) * tick_us
            stages.append({"stage": s, "launch_us": round(total, 1), "queue_dry_after_us": round(bulk, 1), "tail_us": round(total - bulk, 1),
                           "tail_fraction": round((total - bulk) / total, 3), "mean_wave_residency": round(int(r[3]) / int(r[4]) / (tend - t0), 3),
                           "rays": int(r[7]), "mean_node_steps": round(int(r[6]) / max(1, int(r[7])), 1), "max_node_steps": int(r[5]),
                           "rays_by_log2_steps": [int(x) for x in r[8:24]],
                           "long_walks": {"rays": int(r[26]), "mean_us_per_node_step": round(int(r[24]) * tick_us / max(1, int(r[25])), 3),
                                                       "slowest": {"us": round((int(r[27]) >> 32) * tick_us, 1), "node_steps": int(r[27]) & 0xFFFFFFFF},
                                                       "handed_out_last": {"us_after_wave_start": round((int(r[28]) >> 32) * tick_us, 1), "node_steps": int(r[28]) & 0xFFFFFFFF}}})
        out[str(cfg)] = {"workload": description, "trace_kernel": "k_wf_trace_wide" if e.wide_walk() else "k_wf_trace", "wall_clock_khz": int(raw[63, 31]), "trace_stages": stages}
        del e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
