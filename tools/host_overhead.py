#!/usr/bin/env python3
"""Host-side enqueue cost per frame (what bounds strong scaling once a band's GPU time gets small)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bevy_hikari_amd as hk
from bevy_hikari_amd.distributed import BandRenderer

W, H = 1920, 1080
s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0); sc = s.to_c()
cam = hk.cornell_camera(W, H); view, pview, lights = cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform()
e = hk.Engine(device=0); e.upload_noise(); e.upload_scene(hk.load_cornell()); e.resize(W, H, 1.0)
for band_rows, label in ((None, "full frame"), (8, "1/8 band (set_band 3 of 8)")):
    if band_rows: e.set_band(3, 8)
    r = BandRenderer(e, 0, 1)
    if band_rows: e.set_band(3, 8)
    for mode in ("frame_render", "stages (BandRenderer.render)"):
        for n in range(1, 9):
            e.frame_render(hk.frame_uniform(s, n), view, pview, lights, sc)
        e.wait()
        t0 = time.perf_counter()
        N = 100
        for n in range(9, 9 + N):
            f = hk.frame_uniform(s, n)
            if mode == "frame_render": e.frame_render(f, view, pview, lights, sc)
            else: r.render(f, view, pview, lights, s, W, H)
        t1 = time.perf_counter(); e.wait(); t2 = time.perf_counter()
        print(f"{label:28s} {mode:30s} enqueue {1e3*(t1-t0)/N:.3f} ms/frame, total {1e3*(t2-t0)/N:.3f} ms/frame")
