#!/usr/bin/env python3
"""Time the BASELINE configs that are not bench lines (3: Sponza-class, 4: city-class 4K on one
GPU, 5: Cornell 4K 8 bounces) - frame ms, rays/frame, Mray/s, per-pass ms."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large


def run(name, scene, cam, settings, lights, warm=8, steps=16, antialias=False):
    res = {}
    for flags in (0, F.CTX_COUNT_RAYS):
        p = hk.HikariPlugin(device=0, flags=flags)
        p.set_scene(scene)
        for n in range(1, warm + 1):
            p.render(cam, settings, lights=lights, frame_number=n, antialias=antialias)
        p.engine.wait()
        p.engine.reset_stats()
        if flags == 0:
            p.engine.set_timing_mask(0xFFFF)
        t0 = time.perf_counter()
        for n in range(warm + 1, warm + steps + 1):
            p.render(cam, settings, lights=lights, frame_number=n, antialias=antialias)
        p.engine.wait()
        dt = time.perf_counter() - t0
        st = p.engine.stats()
        if flags == 0:
            res["ms_per_frame"] = round(dt / steps * 1e3, 3)
            res["pass_ms"] = {F.PASS_NAMES[i]: round(st.pass_ms_total[i] / steps, 4) for i in range(F.PASS_COUNT) if st.pass_launches[i]}
        else:
            res["rays_per_frame"] = (st.rays_primary + st.rays_tlas + st.rays_blas) / steps
    res["mray_per_s"] = round(res["rays_per_frame"] / res["ms_per_frame"] / 1e3, 1)
    print(json.dumps({"config": name, **res}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4", "5"]
    U = hk.Upscale.SMAA_TU_1_0
    if "3" in which:
        scene, sun = synthetic_large()
        run("3: sponza-class 1920x1080, 3 bounces, denoise", scene, synthetic_camera(1920, 1080, extent=9.0), hk.HikariSettings(indirect_bounces=3, upscale=U),
            hk.lights_uniform(directional=sun))
    if "4" in which:
        scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
        run("4: city-class 3840x2160, 2 bounces, ONE GPU", scene, synthetic_camera(3840, 2160, extent=30.0), hk.HikariSettings(indirect_bounces=2, upscale=U),
            hk.lights_uniform(directional=dict(sun, illuminance=10000.0)))
    if "aa" in which:   # the reference's full post-process tail on BASELINE config 2: SMAA Tu4x to 3840x2160 + TAA there
        run("2+aa: cornell 1920x1080 traced, 2 bounces, SMAA Tu4x -> 3840x2160 + TAA", hk.load_cornell(), hk.cornell_camera(1920, 1080),
            hk.HikariSettings(indirect_bounces=2, upscale=U), hk.lights_uniform(), warm=16, steps=32, antialias=True)
        run("default settings at a 1920x1080 window: 960x540 traced, SMAA Tu4x -> 1920x1080 + TAA", hk.load_cornell(), hk.cornell_camera(1920, 1080),
            hk.HikariSettings(indirect_bounces=2), hk.lights_uniform(), warm=16, steps=32, antialias=True)
    if "helmet" in which:   # the reference's textured glTF asset: 94 722 triangles in 6 BLAS, 10 textures
        from bevy_hikari_amd.scenes import flight_helmet_scene

        scene, sun, camera = flight_helmet_scene()
        run("FlightHelmet 1920x1080, 2 bounces, textured", scene, camera(1920, 1080), hk.HikariSettings(indirect_bounces=2, upscale=U),
            hk.lights_uniform(directional=sun))
    if "5" in which:
        run("5: cornell 3840x2160, 8 bounces, emissive+indirect spatial, denoise off", hk.load_cornell(), hk.cornell_camera(3840, 2160),
            hk.HikariSettings(indirect_bounces=8, emissive_spatial_reuse=True, denoise=False, upscale=U), hk.lights_uniform())
