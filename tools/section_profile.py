#!/usr/bin/env python3
"""Where does k_indirect spend its wave time?  Builds (``--build``, no GPU needed) a variant of the library with
-DHK_PROFILE_SECTIONS into build_ab/ and, on the GPU box, renders the bench workload with it and prints the share of
wave cycles per section (scalar clock, charged once per wave when any lane is inside the section)."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWICE = "--walk-twice" in sys.argv   # ablation instead of the clock: every BVH walk of k_indirect runs twice
VARIANT = os.path.join(ROOT, "build_ab", "libhikari_hip_walk_twice.so" if TWICE else "libhikari_hip_sections.so")
NAMES = ["prologue+idle", "sample+ray setup", "closest-hit traversal", "hit_info+surface", "light candidate", "shadow ray setup",
         "shadow traversal", "radiance+shading+throughput", "ReSTIR temporal", "stores"]

if "--build" in sys.argv:   # (through tools/build_lib.py, like every variant: the same sources, flags and build stamp as the shipped library)
    os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_lib.py"), "-o", VARIANT, "--objdir", os.path.join(ROOT, "build", "obj_" + os.path.basename(VARIANT)[:-3]),
                    "-DHK_ABLATE_WALK_TWICE" if TWICE else "-DHK_PROFILE_SECTIONS"], check=True)
    print("built", VARIANT)
    sys.exit(0)

if TWICE:   # time k_indirect alone with the shipped library and with the variant (bench.py --passes does the timing)
    out = {}
    for name, lib in (("shipped", None), ("walks_twice", VARIANT)):
        env = dict(os.environ)
        if lib:
            env["HIKARI_HIP_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--passes", "--no-cpu-baseline"], env=env, capture_output=True, text=True, check=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out[name] = {"k_indirect_alone_ms": d["pass_ms"]["indirect_lit_ambient"], "frame_ms": d["ms_per_step"]}
    a, b = out["shipped"]["k_indirect_alone_ms"], out["walks_twice"]["k_indirect_alone_ms"]
    out["walk_share_of_k_indirect"] = round((b - a) / a, 4)
    print(json.dumps(out, indent=1))
    sys.exit(0)

os.environ["HIKARI_HIP_LIB"] = VARIANT
sys.path.insert(0, ROOT)
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F

w, h, bounces = 1920, 1080, 2
p = hk.HikariPlugin(device=0, flags=F.CTX_SINGLE_STREAM)
p.set_scene(hk.load_cornell())
s = hk.HikariSettings(indirect_bounces=bounces, upscale=hk.Upscale.SMAA_TU_1_0)
cam = hk.cornell_camera(w, h)
read = p.engine.api.dll.hk_debug_read_sections
read.argtypes, read.restype = [C.POINTER(C.c_ulonglong), C.c_int], C.c_int
out = (C.c_ulonglong * 24)()
for n in range(1, 9):
    p.render(cam, s, frame_number=n)
p.engine.wait()
assert read(out, 1) == 0
for n in range(9, 29):
    p.render(cam, s, frame_number=n)
p.engine.wait()
assert read(out, 0) == 0
tot = float(sum(out[:10]))
print(json.dumps({"workload": f"cornell {w}x{h} b{bounces}", "frames": 20,
                  "share": {NAMES[i]: round(out[i] / tot, 4) for i in range(len(NAMES))},
                  # every walk of the frame (all ray kernels): what a wave pays per loop iteration
                  "walk": {"wave_iterations_per_frame": out[16] / 20, "with_a_triangle_test": round(out[17] / out[16], 4),
                           "with_an_instance_entry": round(out[18] / out[16], 4), "with_a_blas_exit": round(out[19] / out[16], 4),
                           "active_lanes_per_iteration": round(out[20] / out[16], 2)},
                  # k_spatial_reuse<false>'s sixteen taps per pixel: where they end (per frame; emissive pass off in this config)
                  "spatial_reuse_taps": {"taps_per_frame": out[10] / 20, "outside_the_image": round(out[11] / max(out[10], 1), 4),
                                         "rejected_by_depth_ratio": round(out[12] / max(out[10], 1), 4), "empty_or_normal_miss": round(out[13] / max(out[10], 1), 4),
                                         "facing_away": round(out[14] / max(out[10], 1), 4), "occluded_by_the_depth_march": round(out[15] / max(out[10], 1), 4),
                                         "merged": round(1.0 - sum(out[11:16]) / max(out[10], 1), 4)},
                  # the one-level walk counts differently (hk_device.hpp traverse_flat): slots 0 / 1 / 4..7
                  "one_level_walk": {"traversal": list(p.engine.traversal_mode()), "rays_per_frame": out[22] / 20, "wave_iterations_per_frame": out[16] / 20,
                                     "iterations_with_the_test_block": round(out[17] / max(out[16], 1), 4),
                                     "lanes_in_the_loop_per_iteration": round(out[23] / max(out[16], 1), 2),
                                     "lanes_walking_a_node_per_iteration": round(out[20] / max(out[16], 1), 2),
                                     "lanes_testing_per_test_block": round(out[21] / max(out[17], 1), 2),
                                     "node_steps_per_ray": round(out[20] / max(out[22], 1), 2), "triangle_tests_per_ray": round(out[21] / max(out[22], 1), 2)}}, indent=1))
