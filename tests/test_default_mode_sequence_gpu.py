"""The TIMED kernels of the large configs held to the bar where they are timed, and over a sequence.

The product default for scenes beyond the LDS copy (flags 0: direction-threaded trees, the queue-based indirect pass, the wide walk -
what bench.py --config 3 / 4 times) visits candidates in another order than the reference; its bar is the north star's 1e-3
relative L2 (since round 5 its hits are the reference's, and what is measured is orders of magnitude below the bar).  ReSTIR feeds its reservoirs back through temporal reuse: a deviation that is fine after two frames
says nothing about frame 32.  Here:

  * configs 3 and 4 at their FULL sizes, flags 0 against HK_CTX_EXACT_TRAVERSAL (itself bit-exact against the oracle: test_parity_gpu.py)
    over 32 frames - relative L2 and differing-pixel fraction PER FRAME, every frame under the bar, the curve kept under gpurun_out/
    (copied to profiles/ by the round's script);
  * config 4 at 3840x2160, flags 0, against the ORACLE itself on three row ranges (the oracle renders only those rows and the aprons
    their passes read: orc_frame_stage_rows).

Reference semantics compared: light.wgsl:442-486 (traverse_top), 1263-1498 (indirect_lit_ambient)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import ALL_BUFFERS, _as_float, product_default_traversal
from conftest import ROOT

pytestmark = pytest.mark.gpu

# what a frame's consumers see of the light / denoise path: per-channel radiance before and after the denoiser, the variance the
# denoiser is steered by, the tone-mapped image - and the G-buffer's primary hits (where another closest hit would show first)
SEQUENCE_BUFFERS = {b: n for b, n in ALL_BUFFERS.items()
                    if n in ("tone_mapped", "render0", "render1", "render2", "denoise_render0", "denoise_render1", "denoise_render2", "variance2")}


def _deviation(x, y):
    ne = (x.view(np.uint8).reshape(x.shape[0], x.shape[1], -1) != y.view(np.uint8).reshape(y.shape[0], y.shape[1], -1)).any(axis=2)
    if not ne.any():
        return 0.0, 0.0
    fx, fy = np.nan_to_num(_as_float(x), posinf=0.0, neginf=0.0), np.nan_to_num(_as_float(y), posinf=0.0, neginf=0.0)
    return float(np.linalg.norm(fx - fy) / max(float(np.linalg.norm(fy)), 1e-30)), float(ne.mean())


def _report(name, data):
    print(name, json.dumps(data)[:2000])
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, name + ".json"), "w") as f:
            json.dump(data, f, indent=1)


def _sequence(name, scene, cam, s, lights, n_frames, flags_fast=0):
    exact = hk.HikariPlugin(device=0, flags=F.CTX_EXACT_TRAVERSAL)
    with product_default_traversal():
        fast = hk.HikariPlugin(device=0, flags=flags_fast)
    for p in (exact, fast):
        p.set_scene(scene)
    curve = []
    for n in range(1, n_frames + 1):
        for p in (exact, fast):
            p.render(cam, s, lights=lights, frame_number=n)
        per = {}
        for b, bn in SEQUENCE_BUFFERS.items():
            per[bn] = _deviation(fast.engine.read(b), exact.engine.read(b))
        ia, ib = fast.engine.read(F.BUF_INSTANCE_MATERIAL), exact.engine.read(F.BUF_INSTANCE_MATERIAL)
        worst = max(per, key=lambda k: per[k][0])
        curve.append({"frame": n, "output_rel_l2": per["tone_mapped"][0], "output_pixels_differing": per["tone_mapped"][1],
                      "worst_buffer": worst, "worst_rel_l2": per[worst][0], "worst_pixels_differing": max(v[1] for v in per.values()),
                      "primary_hit_instance_differs": float((ia[..., 0] != ib[..., 0]).mean()),
                      "per_buffer_rel_l2": {k: v[0] for k, v in per.items()}})
    assert exact.engine.traversal_mode()[0] == "reference" and fast.engine.traversal_mode()[0] == "threaded"
    assert fast.engine.indirect_schedule() == "wavefront" and fast.engine.wide_walk() and fast.engine.stats().wide_stack_lost == 0
    data = {"case": name, "frames": n_frames, "bar": 1e-3, "held_to": 2e-4, "fast": "flags 0 (threaded trees + queue-based indirect pass + wide walk: what bench.py times)",
            "against": "HK_CTX_EXACT_TRAVERSAL (bit-exact vs the oracle)", "max_worst_rel_l2": max(c["worst_rel_l2"] for c in curve),
            "max_output_rel_l2": max(c["output_rel_l2"] for c in curve), "curve": curve}
    _report(f"default_mode_sequence_{name}", data)
    # The north star's bar is 1e-3.  Since round 5 the default mode's hits ARE the reference's (ties by the leaves' ranks, kept occluders
    # in the reference's order): what is left is a pixel or two per 8 M and frame (6e-5 at most over these sequences,
    # profiles/r05_default_mode_sequence_*.json) - the test holds the curve to 2e-4, so that a return of either order dependence
    # (8.6e-4 after two frames, 1.6e-2 after thirty: round 4's state) fails it long before the bar is in sight.
    over = [(c["frame"], c["worst_buffer"], c["worst_rel_l2"]) for c in curve if not c["worst_rel_l2"] <= 2e-4]
    assert not over, over
    return data


def test_config2_default_mode_every_buffer_of_24_frames_1080p():
    """BASELINE config 2 (the configuration the metric is quoted on) at 1920x1080 in the product default - the one-level walk from the
    LDS copy, frame pipelining - against HK_CTX_EXACT_TRAVERSAL (bit-exact against the oracle): EVERY buffer, the ten reservoir
    buffers included, byte for byte, every fourth frame of 24.  (Round 3-4 excepted the reservoir records: the one-level walk kept
    whichever occluder it met first.  Since round 5 a ray whose occluder is kept takes the reference's own walk -
    hk_device.hpp traverse_top<true> - and nothing is excepted.)"""
    from cases import diff_buffers, snapshot

    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    scene, cam = hk.load_cornell(), hk.cornell_camera(1920, 1080)
    exact = hk.HikariPlugin(device=0, flags=F.CTX_EXACT_TRAVERSAL)
    with product_default_traversal():
        fast = hk.HikariPlugin(device=0)
    for p in (exact, fast):
        p.set_scene(scene)
    for n in range(1, 25):
        for p in (exact, fast):
            p.render(cam, s, frame_number=n)
        if n % 4 == 0:
            bad = diff_buffers(snapshot(fast), snapshot(exact))
            assert bad == {}, (n, bad)
    assert fast.engine.traversal_mode()[0] == "one-level" and exact.engine.traversal_mode()[0] == "reference"


def test_config3_default_mode_32_frames_1080p():
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large()
    _sequence("config3_1080p", scene, synthetic_camera(1920, 1080, extent=9.0), hk.HikariSettings(indirect_bounces=3, upscale=hk.Upscale.SMAA_TU_1_0),
              hk.lights_uniform(directional=sun), 32)


def test_config4_default_mode_32_frames_4k():
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
    _sequence("config4_4k", scene, synthetic_camera(3840, 2160, extent=30.0), hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0),
              hk.lights_uniform(directional=dict(sun, illuminance=10000.0)), 32)


def test_config4_default_mode_full_4k_row_ranges_vs_oracle():
    """Config 4 at 3840x2160 in the mode it is timed in (flags 0) against the ORACLE on three row ranges: relative L2 of every
    rendered buffer over the rows of the three ranges <= 1e-3 (measured: 0 differing bytes in all 16 buffers over both frames,
    profiles/r05_default_mode_config4_4k_row_ranges_vs_oracle.json)."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large
    from oracle_lib import oracle_api, oracle_engine

    scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    sc = s.to_c()
    W, H = 3840, 2160
    cam = synthetic_camera(W, H, extent=30.0)
    lights = hk.lights_uniform(directional=dict(sun, illuminance=10000.0))
    view, pview = cam.view_uniform(), cam.previous_view_uniform()
    with product_default_traversal():
        gpu = hk.Engine(device=0)
    cpu = oracle_engine()
    for e in (gpu, cpu):
        e.upload_noise(); e.upload_scene(scene); e.resize(W, H, 1.0)
    stage_rows = oracle_api().dll.orc_frame_stage_rows
    ranges = [(0, 24), (1068, 1092), (2136, 2160)]
    SP, DEN = 21, 16                     # spatial-reuse and denoiser aprons (rows), as in hk_band_plan_for
    clamp = lambda v: min(max(v, 0), H)
    worst = {}
    for n in (1, 2):
        f = hk.frame_uniform(s, n)
        gpu.frame_render(f, view, pview, lights, sc)
        cpu.frame_begin(f, view, pview, lights)
        extra = (SP + DEN) if n == 1 else 0   # frame 1 also produces what frame 2 reads of it (same pixel: static camera)
        for r0, r1 in ranges:
            for stage, apron in ((F.STAGE_TEMPORAL, SP + DEN), (F.STAGE_SPATIAL, DEN), (F.STAGE_POST_PROCESS, 0)):
                rc = stage_rows(cpu.ctx, stage, C.byref(sc), 0, clamp(r0 - apron - extra), clamp(r1 + apron + extra))
                assert rc == 0, cpu.api.last_error()
        gpu.wait()
        for b, name in ALL_BUFFERS.items():
            if name.startswith(("previous_", "reservoir", "internal")) or name in ("upscale_output", "taa_output", "upscale_sharpened"):
                continue
            a, o = gpu.read(b), cpu.read(b)
            x = np.concatenate([a[r0:r1] for r0, r1 in ranges])
            y = np.concatenate([o[r0:r1] for r0, r1 in ranges])
            dev = _deviation(x, y)
            worst[name] = max(worst.get(name, (0.0, 0.0)), dev)
            assert dev[0] <= 1e-3, f"frame {n}: {name} rows of the three ranges: relative L2 {dev[0]:.3e} vs the oracle at 4K (product default mode)"
    assert gpu.traversal_mode()[0] == "threaded" and gpu.indirect_schedule() == "wavefront" and gpu.wide_walk() and gpu.stats().wide_stack_lost == 0
    _report("default_mode_config4_4k_row_ranges_vs_oracle", {"ranges": ranges, "frames": 2, "worst_relative_l2": max(v[0] for v in worst.values()),
                                                             "worst_fraction_of_pixels_differing": max(v[1] for v in worst.values()),
                                                             "per_buffer": {k: list(v) for k, v in worst.items()}})
