#!/usr/bin/env python3
"""Instance motion: host rebuild + upload (hk_upload_scene_instances: the reference's prepare_instances, instance.rs:286-437)
against the device refit (hk_refit_scene_instances) at 2 000 and 20 000 instances, 10 % of them moving per frame.
Reports host time per update, GPU time of the update's kernels (HIP events on the context's stream) and the frame time of
an animated sequence enqueued back to back.   Usage: python tools/refit_probe.py [n_instances ...]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large


def probe(n_instances):
    # few small unique meshes: the instance count is what is being varied
    scene, sun = synthetic_large(0x5EED0004, 20, 16, 32, n_instances, 50, 8, 40.0)
    cam = synthetic_camera(1280, 720, extent=30.0)
    s = hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)
    lights = hk.lights_uniform(directional=sun)
    rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in scene.instances], dtype=np.float32)
    movers = np.random.default_rng(1).choice(len(rest), size=max(1, len(rest) // 10), replace=False)
    b = scene.builder
    set_t = b.api.raw("scene_builder_set_instance_transform")
    out = {"instances": len(rest), "triangles": len(scene.primitives), "moved_per_update": int(len(movers))}

    def move(n):
        for i in movers:
            m = rest[i].copy()
            m[12] += 0.01 * n
            set_t(b.h, int(i), m.ctypes.data_as(C.POINTER(F.f32)))

    for mode in ("host", "device"):
        p = hk.HikariPlugin(device=0)
        p.set_scene(scene)
        p.render(cam, s, lights=lights, frame_number=1)
        p.engine.wait()
        stream = torch.cuda.ExternalStream(p.engine.stream())
        t_host, t_gpu = [], []
        for n in range(2, 14):
            move(n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p.engine.wait()
            e0.record(stream)
            t0 = time.perf_counter()
            if mode == "host":
                b.api.call("scene_builder_finish", b.h)
                p.engine.api.call("upload_scene_instances", p.engine.ctx, b.h)
                p.engine.frame_begin(hk.frame_uniform(s, n), cam.view_uniform(), cam.previous_view_uniform(), lights)
                p.engine.api.call("indirect_schedule", p.engine.ctx, C.byref(C.c_uint32()))  # lays the scene out + enqueues the upload, no dispatch
            else:
                p.engine.refit_instances(b)
            t1 = time.perf_counter()
            e1.record(stream)
            p.engine.wait()
            t_host.append(t1 - t0)
            t_gpu.append(e0.elapsed_time(e1))

        def animated(frames, first, animate):
            t0 = time.perf_counter()
            for n in range(first, first + frames):
                if animate:
                    move(n)
                    if mode == "host":
                        b.api.call("scene_builder_finish", b.h)
                        p.engine.api.call("upload_scene_instances", p.engine.ctx, b.h)
                    else:
                        p.engine.refit_instances(b)
                p.render(cam, s, lights=lights, frame_number=n)
            p.engine.wait()
            return (time.perf_counter() - t0) / frames * 1e3

        animated(4, 100, True)
        out[mode] = {"host_ms_per_update": round(float(np.median(t_host)) * 1e3, 3), "stream_ms_per_update": round(float(np.median(t_gpu)), 3),
                     "frame_ms_static": round(animated(20, 200, False), 3), "frame_ms_animated": round(animated(20, 300, True), 3)}
        st = p.engine.stats()
        out[mode]["stats"] = {"instance_builds": int(st.scene_instance_builds), "async_uploads": int(st.scene_async_instance_uploads), "device_refits": int(st.scene_device_refits)}
        if mode == "device":   # the LBVH rebuild of both trees on the device, alone on the stream
            for name, tree in (("lbvh", F.TREE_LBVH), ("sah", F.TREE_SAH)):
                t_gpu = []
                for _ in range(8):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    p.engine.wait()
                    e0.record(stream)
                    p.engine.rebuild_trees(tree)
                    e1.record(stream)
                    p.engine.wait()
                    t_gpu.append(e0.elapsed_time(e1))
                out[mode][f"rebuild_trees_{name}_stream_ms"] = round(float(np.median(t_gpu)), 3)
                out[mode][f"frame_ms_static_after_{name}_rebuild"] = round(animated(20, 400 if name == "lbvh" else 500, False), 3)
        if mode == "device":   # leave the builder finished for the next scene
            b.api.call("scene_builder_finish", b.h)
        del p
    # instance SET edits (round 3): every update adds one instance and removes another.  "host" = the reference's path
    # (hk_scene_builder_finish: both SAH tree builds on the host + upload); "device" = hk_update_scene_instances (records laid out
    # on the host, both trees built on the device: HK_TREE_SAH = the same trees).
    extra = rest[int(movers[0])].copy()
    for mode in ("host", "device"):
        p = hk.HikariPlugin(device=0)
        p.set_scene(scene)
        p.render(cam, s, lights=lights, frame_number=1)
        p.engine.wait()
        stream = torch.cuda.ExternalStream(p.engine.stream())
        t_host, t_gpu = [], []
        for n in range(2, 12):
            m = extra.copy()
            m[12] += 0.37 * n
            b.add_instance(0, 1, m)
            b.remove_instance(int(movers[n]))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p.engine.wait()
            e0.record(stream)
            t0 = time.perf_counter()
            if mode == "host":
                b.api.call("scene_builder_finish", b.h)
                p.engine.api.call("upload_scene_instances", p.engine.ctx, b.h)
                p.engine.frame_begin(hk.frame_uniform(s, n), cam.view_uniform(), cam.previous_view_uniform(), lights)
                p.engine.api.call("indirect_schedule", p.engine.ctx, C.byref(C.c_uint32()))
            else:
                p.engine.update_instances_on_device(b, F.TREE_SAH)
            t1 = time.perf_counter()
            e1.record(stream)
            p.engine.wait()
            t_host.append(t1 - t0)
            t_gpu.append(e0.elapsed_time(e1))
        out["edit_" + mode] = {"host_ms_per_update": round(float(np.median(t_host)) * 1e3, 3), "stream_ms_per_update": round(float(np.median(t_gpu)), 3)}
        del p
    b.api.call("scene_builder_finish", b.h)
    return out


def emitter_probe():
    """The emitter half of the refit: an emissive sphere of ~1 900 / ~3 700 triangles moving (examples/scene.rs:231-235)."""
    from bevy_hikari_amd.scenes import synthetic_scene

    res = {}
    for rings, segs in ((24, 40), (40, 48)):
        scene, sun = synthetic_scene(n_boxes=12, n_spheres=2, n_emitters=2, sphere_rings=rings, sphere_segs=segs, n_emissive_spheres=1)
        p = hk.HikariPlugin(device=0)
        p.set_scene(scene)
        cam, s = synthetic_camera(640, 360), hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)
        p.render(cam, s, lights=hk.lights_uniform(directional=sun), frame_number=1)
        p.engine.wait()
        stream = torch.cuda.ExternalStream(p.engine.stream())
        idx = len(scene.instances) - 1
        rest = np.ctypeslib.as_array(scene.instances[idx].model).copy()
        t_gpu = []
        for n in range(2, 14):
            m = rest.copy()
            m[12] += 0.02 * n
            scene.builder.set_instance_transform(idx, m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p.engine.wait()
            e0.record(stream)
            p.engine.refit_instances(scene.builder)
            e1.record(stream)
            p.engine.wait()
            t_gpu.append(e0.elapsed_time(e1))
        res[f"{int(scene.emissives[-1].alias_table[1])}_triangles"] = {"refit_stream_ms": round(float(np.median(t_gpu)), 4)}
    return {"emissive_sphere_refit": res}


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [2000, 20000]
    for n in sizes:
        print(json.dumps(probe(n)), flush=True)
    print(json.dumps(emitter_probe()), flush=True)
