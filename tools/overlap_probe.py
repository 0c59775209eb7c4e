#!/usr/bin/env python3
"""The default frame path (direct-light dispatches on a second stream) against HK_CTX_SINGLE_STREAM:
wall time per frame for the whole image and for one band of an N-way split, plus a bit-exactness check."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F

W, H = 1920, 1080
s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
sc = s.to_c()
cam = hk.cornell_camera(W, H)
view, pview, lights = cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform()


def run(flags, band=None, nb=1, frames=64, reps=3):
    e = hk.Engine(device=0, flags=flags)
    e.upload_noise(); e.upload_scene(hk.load_cornell()); e.resize(W, H, 1.0)
    for n in range(1, 17):
        e.frame_render(hk.frame_uniform(s, n), view, pview, lights, sc)
    if band is not None:
        e.set_band(band, nb)
    e.wait()
    best = 1e9
    n = 17
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(frames):
            e.frame_render(hk.frame_uniform(s, n), view, pview, lights, sc)
            n += 1
        e.wait()
        best = min(best, (time.perf_counter() - t0) / frames * 1e3)
    return best, e.read(F.BUF_TONE_MAPPED)


if __name__ == "__main__":
    for label, kw in (("full frame", {}), ("band 3 of 8", dict(band=3, nb=8)), ("band 1 of 4", dict(band=1, nb=4)), ("band 0 of 2", dict(band=0, nb=2))):
        a, ia = run(F.CTX_SINGLE_STREAM, **kw)
        b, ib = run(0, **kw)
        print(f"{label:12s}: single stream {a:.4f} ms/frame, overlapped {b:.4f} ms/frame ({(a / b - 1) * 100:+.1f} %), identical: {bool((ia == ib).all())}")
