#!/usr/bin/env python3
"""Wide GPU-only sweep: the wavefront schedule of indirect_lit_ambient against the fused kernel on seeded scenes BEYOND the LDS
copy (global-memory traversal, where the wavefront schedule is the default), random settings / sizes / camera, three frames each,
every buffer bit for bit.  Two pairs per seed: product defaults (direction-threaded BVHs) and HK_CTX_EXACT_TRAVERSAL.
Prints one JSON line.   Usage: python tests/tools/schedule_sweep.py 0 100"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["HIKARI_HIP_DEFAULT_CTX_FLAGS"] = "0"
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.scenes import synthetic_large
from cases import diff_buffers, snapshot

first, last = int(sys.argv[1]), int(sys.argv[2])
bad, t0, frames, schedules = {}, time.time(), 0, set()
for seed in range(first, last):
    rng = np.random.default_rng(9100 + seed)
    scene, sun = synthetic_large(int(rng.integers(1, 1 << 30)), int(rng.integers(2, 7)), int(rng.integers(6, 16)), int(rng.integers(12, 28)), int(rng.integers(30, 400)), 12,
                                 int(rng.integers(0, 5)), float(rng.uniform(6.0, 14.0)))
    s = hk.HikariSettings(indirect_bounces=int(rng.integers(2, 6)), emissive_spatial_reuse=bool(rng.random() < 0.5), indirect_spatial_reuse=bool(rng.random() < 0.7),
                          denoise=bool(rng.random() < 0.6), temporal_reuse=bool(rng.random() < 0.85), max_indirect_luminance=float(rng.choice([0.5, 10.0])),
                          upscale=hk.Upscale.SmaaTu4x(float(rng.choice([1.0, 1.0, 1.5, 2.0]))))
    w, h = int(rng.integers(97, 420)), int(rng.integers(65, 300))
    eye = tuple(np.array([14.0, 9.0, 17.0]) * rng.uniform(0.5, 1.2) + rng.normal(0, 1.0, 3))
    cam = hk.Camera(hk.look_at_transform(eye, (0.0, 0.6, 0.0)), w, h)
    lights = hk.lights_uniform(directional=sun)
    first_frame = int(rng.integers(1, 9))
    for base in (0, F.CTX_EXACT_TRAVERSAL):
        a, b = hk.HikariPlugin(device=0, flags=base | F.CTX_WAVEFRONT), hk.HikariPlugin(device=0, flags=base | F.CTX_FUSED_INDIRECT)
        for p in (a, b):
            p.set_scene(scene)
        for n in range(first_frame, first_frame + 3):
            for p in (a, b):
                p.render(cam, s, lights=lights, frame_number=n)
            frames += 1
            d = diff_buffers(snapshot(a), snapshot(b))
            if d and seed not in bad:
                bad[seed] = {"flags": base, "frame": n, **{k: v[:80] for k, v in d.items()}}
        schedules.add(a.engine.indirect_schedule() + "/" + b.engine.indirect_schedule())
print(json.dumps({"seeds": [first, last], "frames_compared": frames, "schedules": sorted(schedules), "mismatching_seeds": len(bad), "first": dict(list(bad.items())[:3]),
                  "seconds": round(time.time() - t0, 1)}))
