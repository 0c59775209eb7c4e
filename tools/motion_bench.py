#!/usr/bin/env python3
"""Frames under MOTION, and the price of determinism.

Every other timed frame of bench.py is a static-camera steady state.  Here:
  * config 2 (Cornell 1920x1080, 2 bounces) with the camera ORBITING its target - what examples/cornell.rs's OrbitCameraController does
    with the mouse held: a fixed angle per frame around (0, 1, 0) at the example's radius;
  * config 3 (Sponza-class stand-in) with the same orbit AND instances moving on their own, poses pushed through
    hk_refit_scene_instances every frame (device refit of both trees).
Each in three modes: HK_CTX_RACING_SCATTER - the reference's own write-write race on previous_spatial_reservoir_buffer
(light.wgsl:1092-1095,1199-1202,1456-1459), whichever store lands last stays: the product default through round 5 -, the product
default since round 6 - the race resolved by highest invocation index (the oracle's rule) in the LIGHT form, for the channels whose
buffer has a reader -, and HK_CTX_DETERMINISTIC_SCATTER, the verification mode (all three channels, one set of planes, no frame
pipelining).  Reported: ms per frame of each, what the default costs over the racing mode, the relative L2 of the racing frames against
the default's per frame (GPU against GPU, full size) and - `oracle_frames` > 0 - of all three against the CPU oracle on the first
frames of the same sequence.

    python tools/motion_bench.py [--configs 2 3] [--oracle-frames 6] > profiles/r06_motion.json        (bench.py --motion prints the same)
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ORBIT_RAD_PER_FRAME = 0.002   # 0.11 degrees per frame: a 360-degree mouse orbit in 3 s at 1 000 frames per second


def orbit_camera(hk, eye, target, width, height, n, radians_per_frame=ORBIT_RAD_PER_FRAME):
    """The camera at frame n: the eye turned about the vertical axis through the target by n x radians_per_frame, looking at the target
    (the example's OrbitCameraController)."""
    a = n * radians_per_frame
    d = np.asarray(eye, dtype=np.float64) - np.asarray(target, dtype=np.float64)
    rot = np.array([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
    return hk.Camera(hk.look_at_transform(tuple(np.asarray(target) + rot @ d), tuple(target)), width, height)


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else 0.0


def run(hk, F, config, device=0, frames=48, blocks=3, warmup=24, compare_frames=12, oracle_frames=0, oracle_engine=None, orbit=ORBIT_RAD_PER_FRAME):
    import bench

    scene, camera, settings, lights, description = bench.workload(hk, config, None, None, None)
    W, H = camera.width, camera.height
    sc = settings.to_c()
    moving_instances = config == 3
    movers = rest = None
    if moving_instances:   # a twentieth of the instances drift and spin, new poses every frame
        rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in scene.instances], dtype=np.float32)
        movers = np.random.default_rng(3).choice(len(rest), size=max(1, len(rest) // 20), replace=False)

    def pose(engine, n):
        if not moving_instances:
            return
        b = scene.builder
        setter = b.api.raw("scene_builder_set_instance_transform")
        for k, i in enumerate(movers):
            m = rest[i].reshape(4, 4).T.astype(np.float64)
            ang = 0.01 * n * (1 + k % 3)
            c, s_ = math.cos(ang), math.sin(ang)
            rot = np.array([[c, 0, s_, 0], [0, 1, 0, 0], [-s_, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)
            shift = np.eye(4)
            shift[0, 3] = 0.004 * n * (1 if k % 2 == 0 else -1)
            t = (shift @ m @ rot).T.astype(np.float32).reshape(-1)
            setter(b.h, int(i), t.ctypes.data_as(C.POINTER(F.f32)))
        engine.refit_instances(b)

    # (bench.workload's cameras: examples/cornell.rs:49-50, scenes.synthetic_camera)
    eye, target = ((0.0, 1.0, 4.0), (0.0, 1.0, 0.0)) if config in (2, 5) else ((1.6 * 9.0, 1.1 * 9.0, 2.0 * 9.0), (0.0, 0.6, 0.0))

    def frame(engine, n):
        pose(engine, n)
        cam, prev = orbit_camera(hk, eye, target, W, H, n, orbit), orbit_camera(hk, eye, target, W, H, n - 1, orbit)
        engine.frame_render(hk.frame_uniform(settings, n), cam.view_uniform(), cam.previous_view_uniform(prev), lights, sc)

    def make(flags):
        e = hk.Engine(device=device, flags=flags)
        e.upload_noise()
        e.upload_scene(scene)
        e.resize(W, H, 1.0)
        return e

    out = {"workload": description.replace("static camera", "camera orbiting its target at %.3f rad per frame" % orbit) +
           (", %d of %d instances moving every frame through hk_refit_scene_instances" % (len(movers), len(rest)) if moving_instances else ""),
           "frames_per_block": frames, "blocks": blocks, "warmup_frames": warmup}
    engines = {}
    for mode, flags in (("racing", F.CTX_RACING_SCATTER), ("default", 0), ("verification_mode", F.CTX_DETERMINISTIC_SCATTER)):
        e = make(flags)
        n = 0
        for _ in range(warmup):
            n += 1
            frame(e, n)
        e.wait()
        ms = []
        for _ in range(blocks):
            t0 = time.perf_counter()
            for _ in range(frames):
                n += 1
                frame(e, n)
            e.wait()
            ms.append((time.perf_counter() - t0) / frames * 1e3)
        out[mode] = {"ms_per_frame": round(float(np.median(ms)), 4), "blocks_ms_per_frame": [round(x, 4) for x in ms]}
        engines[mode] = e
    out["default_costs_over_racing"] = round(out["default"]["ms_per_frame"] / out["racing"]["ms_per_frame"] - 1.0, 4)
    out["verification_mode_costs_over_racing"] = round(out["verification_mode"]["ms_per_frame"] / out["racing"]["ms_per_frame"] - 1.0, 4)
    # the racing frames against the default's, frame by frame over a fresh sequence (both from zeroed reservoirs)
    del engines
    a, b = make(F.CTX_RACING_SCATTER), make(0)
    o = None
    if oracle_frames > 0 and oracle_engine is not None:
        o = oracle_engine()
        o.upload_noise()
        o.upload_scene(scene)
        o.resize(W, H, 1.0)
    dev, dev_o_racing, dev_o_det = [], [], []
    for n in range(1, compare_frames + 1):
        frame(a, n)
        frame(b, n)
        ta, tb = a.read_f16(F.BUF_TONE_MAPPED), b.read_f16(F.BUF_TONE_MAPPED)
        dev.append(rel_l2(ta, tb))
        if o is not None and n <= oracle_frames:
            frame(o, n)
            to = o.read_f16(F.BUF_TONE_MAPPED)
            dev_o_racing.append(rel_l2(ta, to))
            dev_o_det.append(rel_l2(tb, to))
    out["racing_vs_default_rel_l2_per_frame"] = [float("%.3e" % x) for x in dev]
    out["racing_vs_default_rel_l2_max"] = float("%.3e" % max(dev))
    if o is not None:
        out["vs_oracle_rel_l2_per_frame"] = {"racing": [float("%.3e" % x) for x in dev_o_racing], "default": [float("%.3e" % x) for x in dev_o_det]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="+", default=[2, 3])
    ap.add_argument("--oracle-frames", type=int, default=6)
    ap.add_argument("--frames", type=int, default=48)
    args = ap.parse_args()
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import oracle_engine   # (the checker: CPU restatement of the reference, compared against - never timed here)

    res = {"orbit_radians_per_frame": ORBIT_RAD_PER_FRAME, "configs": {}}
    for cfg in args.configs:
        if cfg == 2:   # the same orbit ten times faster (1.1 degrees per frame: a flick of the mouse) - where the reference's race starts to show
            res["configs"]["2_fast_orbit"] = run(hk, F, 2, frames=args.frames, oracle_frames=args.oracle_frames, oracle_engine=oracle_engine, orbit=10.0 * ORBIT_RAD_PER_FRAME)
        res["configs"][str(cfg)] = run(hk, F, cfg, frames=args.frames if cfg == 2 else max(8, args.frames // 4), oracle_frames=args.oracle_frames if cfg == 2 else 0,   # (the oracle has no device refit: config 3's deterministic mode is held to it by tests/test_device_refit.py)
                                       oracle_engine=oracle_engine)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
