#!/usr/bin/env python3
"""Per-pass GPU time of ONE band of an N-band split, on one GPU (no exchanges): what each rank of a
multi-GPU run spends in kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F

W, H = 1920, 1080
s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0); sc = s.to_c()
cam = hk.cornell_camera(W, H); view, pview, lights = cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for band in range(nb):
    e = hk.Engine(device=0); e.upload_noise(); e.upload_scene(hk.load_cornell()); e.resize(W, H, 1.0)
    for n in range(1, 17): e.frame_render(hk.frame_uniform(s, n), view, pview, lights, sc)   # converge full-frame state first
    e.set_band(band, nb); e.wait(); e.reset_stats(); e.set_timing_mask(0xFFFF)
    N = 24
    for n in range(17, 17 + N): e.frame_render(hk.frame_uniform(s, n), view, pview, lights, sc)
    st = e.stats()
    p = {F.PASS_NAMES[i][:10]: round(st.pass_ms_total[i] / N, 3) for i in range(F.PASS_COUNT) if st.pass_launches[i]}
    print(f"band {band}/{nb}: sum {sum(p.values()):.3f} ms  {p}")
