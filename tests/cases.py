"""Shared parity cases: (scene, camera, settings, lights, frame numbers).  Used by the golden
generator, the oracle tests and the GPU parity tests."""
import numpy as np

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.scenes import synthetic_camera, synthetic_scene

ALL_BUFFERS = {F.BUF_POSITION: "position", F.BUF_NORMAL: "normal", F.BUF_DEPTH_GRADIENT: "depth_gradient",
               F.BUF_INSTANCE_MATERIAL: "instance_material", F.BUF_VELOCITY_UV: "velocity_uv", F.BUF_ALBEDO: "albedo",
               F.BUF_DENOISE_INTERNAL_VARIANCE: "internal_variance", F.BUF_TONE_MAPPED: "tone_mapped"}
for _i in range(3):
    ALL_BUFFERS[F.BUF_VARIANCE0 + _i] = f"variance{_i}"
    ALL_BUFFERS[F.BUF_RENDER0 + _i] = f"render{_i}"
    ALL_BUFFERS[F.BUF_DENOISE_RENDER0 + _i] = f"denoise_render{_i}"
for _i in range(10):
    ALL_BUFFERS[F.BUF_RESERVOIR0 + _i] = f"reservoir{_i}"
for _i in range(4):
    ALL_BUFFERS[F.BUF_DENOISE_INTERNAL0 + _i] = f"internal{_i}"
ALL_BUFFERS.update({F.BUF_PREVIOUS_POSITION: "previous_position", F.BUF_PREVIOUS_VELOCITY_UV: "previous_velocity_uv",
                    F.BUF_PREVIOUS_TONE_MAPPED: "previous_tone_mapped", F.BUF_UPSCALE_OUTPUT: "upscale_output",
                    F.BUF_TAA_OUTPUT: "taa_output", F.BUF_PREVIOUS_TAA_OUTPUT: "previous_taa_output",
                    F.BUF_UPSCALE_SHARPENED: "upscale_sharpened"})


import contextlib
import json
import os


def oracle():
    """The CPU oracle behind the plugin interface (test infrastructure: tests/oracle_lib.py)."""
    from oracle_lib import oracle_plugin

    return oracle_plugin()


def report(name, data):
    """Printed, and kept under gpurun_out/ when the suite runs on the GPU box (copied to profiles/ by the round's scripts)."""
    from conftest import ROOT

    print(name, data)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, name + ".json"), "w") as f:
            json.dump(data, f, indent=1)


@contextlib.contextmanager
def product_default_traversal():
    """Engines created inside use the product's default flags (direction-threaded BVHs for scenes beyond the LDS copy) instead
    of the suite-wide HK_CTX_EXACT_TRAVERSAL (tests/conftest.py)."""
    old = F.DEFAULT_CTX_FLAGS
    F.DEFAULT_CTX_FLAGS = 0
    try:
        yield
    finally:
        F.DEFAULT_CTX_FLAGS = old


class Case:
    def __init__(self, name, scene, camera, settings, lights=None, frames=(1, 2, 3, 4), antialias=False):
        self.name, self.scene, self.camera, self.settings, self.frames = name, scene, camera, settings, list(frames)
        self.lights = lights or hk.lights_uniform()
        self.antialias = antialias   # also run the SMAA Tu4x / TAA dispatches of PostProcessNode::run


_CACHE = {}


def cornell_scene():
    if "cornell" not in _CACHE:
        _CACHE["cornell"] = hk.load_cornell()
    return _CACHE["cornell"]


def yard_scene():
    if "yard" not in _CACHE:
        _CACHE["yard"] = synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=3, sphere_rings=6, sphere_segs=8)
    return _CACHE["yard"]


def yard_textured_scene():
    if "yard_tex" not in _CACHE:
        _CACHE["yard_tex"] = synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=3, sphere_rings=6, sphere_segs=8, textured=True)
    return _CACHE["yard_tex"]


def make_case(name):
    S, U = hk.HikariSettings, hk.Upscale
    if name == "cornell_b2":       # BASELINE config 2 at a test size
        return Case(name, cornell_scene(), hk.cornell_camera(96, 64), S(indirect_bounces=2, upscale=U.SMAA_TU_1_0), frames=range(1, 7))
    if name == "cornell_b1":       # BASELINE config 1 (1 bounce, defaults otherwise) at a test size, odd size: guards the grid rounding
        return Case(name, cornell_scene(), hk.cornell_camera(100, 76), S(upscale=U.SMAA_TU_1_0), frames=range(1, 6))
    if name == "cornell_upscale2":  # default Upscale (ratio 2.0, SMAA jitter) + emissive spatial reuse
        return Case(name, cornell_scene(), hk.cornell_camera(128, 96), S(emissive_spatial_reuse=True), frames=range(1, 6))
    if name == "cornell_ratio15_fsr":
        return Case(name, cornell_scene(), hk.cornell_camera(90, 66), S(indirect_bounces=3, upscale=U.Fsr1(1.5, 0.2), taa=hk.Taa.NONE), frames=range(3, 7))
    if name == "cornell_b0_nodenoise":
        return Case(name, cornell_scene(), hk.cornell_camera(64, 64), S(indirect_bounces=0, denoise=False, upscale=U.SMAA_TU_1_0), frames=range(1, 4))
    if name == "cornell_notemporal":
        return Case(name, cornell_scene(), hk.cornell_camera(64, 48), S(temporal_reuse=False, indirect_spatial_reuse=False, upscale=U.SMAA_TU_1_0,
                                                                       max_reservoir_lifetime=1.0), frames=range(1, 4))
    if name == "cornell_b8":       # BASELINE config 5 at a test size: 8 bounces (path compaction kernel), both spatial passes, no denoise
        return Case(name, cornell_scene(), hk.cornell_camera(112, 80), S(indirect_bounces=8, emissive_spatial_reuse=True, denoise=False,
                                                                       upscale=U.SMAA_TU_1_0), frames=range(1, 5))
    if name == "yard_sun":          # several emitters (light-BVH pick, alias tables), sun cone, strips, scaled/rotated instances
        scene, sun = yard_scene()
        return Case(name, scene, synthetic_camera(96, 72), S(indirect_bounces=2, upscale=U.SMAA_TU_1_0, emissive_spatial_reuse=True),
                    lights=hk.lights_uniform(directional=sun), frames=range(1, 8))
    if name == "yard_no_emitters":  # empty emissive list / light BVH: every light pick falls back to the sun cone
        if "yard_dark" not in _CACHE:
            _CACHE["yard_dark"] = synthetic_scene(n_boxes=10, n_spheres=3, n_emitters=0, sphere_rings=5, sphere_segs=6)
        scene, sun = _CACHE["yard_dark"]
        return Case(name, scene, synthetic_camera(72, 40), S(indirect_bounces=2, upscale=U.SMAA_TU_1_0), lights=hk.lights_uniform(directional=sun),
                    frames=range(1, 5))
    if name == "background_only":   # camera looks away from the box: every pixel takes the depth < eps paths
        cam = hk.Camera(hk.look_at_transform((0.0, 1.0, 4.0), (0.0, 1.0, 9.0)), 40, 24)
        return Case(name, cornell_scene(), cam, S(indirect_bounces=2, upscale=U.SMAA_TU_1_0), frames=range(1, 4))
    if name == "tiny_3x5":          # smaller than one 8x8 tile, odd in both axes
        return Case(name, cornell_scene(), hk.cornell_camera(3, 5), S(indirect_bounces=2, upscale=U.SMAA_TU_1_0, emissive_spatial_reuse=True), frames=range(1, 5))
    if name == "yard_textured":     # the textured pipelines (light.wgsl:749-793): base colour / metallic / occlusion / emissive textures
        scene, sun = yard_textured_scene()
        return Case(name, scene, synthetic_camera(96, 72), S(indirect_bounces=2, upscale=U.SMAA_TU_1_0),
                    lights=hk.lights_uniform(directional=sun), frames=range(1, 6))
    if name == "cornell_aa_default":   # the reference's default settings end to end: ratio 2, SMAA Tu4x to the window size, TAA
        return Case(name, cornell_scene(), hk.cornell_camera(120, 88), S(indirect_bounces=2), frames=range(1, 7), antialias=True)
    if name == "yard_aa_smaa2x":       # SMAA Tu4x at ratio 1 (2x the window) + TAA on the textured yard with a sun
        scene, sun = yard_textured_scene()
        return Case(name, scene, synthetic_camera(80, 56), S(indirect_bounces=1, upscale=U.SMAA_TU_1_0), lights=hk.lights_uniform(directional=sun),
                    frames=range(1, 6), antialias=True)
    if name == "cornell_aa_fsr":       # Upscale::Fsr1 end to end: TAA at the scaled size, EASU to the window, RCAS (post_process.rs:1260-1308)
        return Case(name, cornell_scene(), hk.cornell_camera(120, 88), S(indirect_bounces=2, upscale=U.Fsr1(1.5, 0.2)), frames=range(1, 6), antialias=True)
    if name == "yard_aa_fsr_notaa":    # FSR1 straight from the tone-mapped image (Taa::None), ratio 2, full sharpness, odd window size
        scene, sun = yard_textured_scene()
        return Case(name, scene, synthetic_camera(83, 57), S(indirect_bounces=1, upscale=U.Fsr1(2.0, 0.0), taa=hk.Taa.NONE),
                    lights=hk.lights_uniform(directional=sun), frames=range(1, 4), antialias=True)
    if name == "yard_ortho":           # OrthographicProjection: the projection[3].w == 1 branches (light.wgsl:714-727,1040), parallel primary rays
        scene, sun = yard_scene()
        cam = hk.Camera(hk.look_at_transform((6.0, 7.0, 8.0), (0.0, 0.5, 0.0)), 88, 64, ortho_height=9.0)
        return Case(name, scene, cam, S(indirect_bounces=2, upscale=U.SMAA_TU_1_0), lights=hk.lights_uniform(directional=sun), frames=range(1, 5))
    if name == "flight_helmet":        # the reference's textured glTF asset: 94 722 triangles, 10 textures
        if "helmet" not in _CACHE:
            from bevy_hikari_amd.scenes import flight_helmet_scene

            _CACHE["helmet"] = flight_helmet_scene()
        scene, sun, camera = _CACHE["helmet"]
        return Case(name, scene, camera(96, 96), S(indirect_bounces=2, upscale=U.SMAA_TU_1_0), lights=hk.lights_uniform(directional=sun),
                    frames=range(1, 5))
    raise KeyError(name)


CASE_NAMES = ["cornell_b2", "cornell_b1", "cornell_upscale2", "cornell_ratio15_fsr", "cornell_b0_nodenoise", "cornell_notemporal", "cornell_b8", "yard_sun", "yard_textured", "yard_no_emitters", "background_only", "tiny_3x5", "cornell_aa_default", "yard_aa_smaa2x", "flight_helmet", "yard_ortho", "cornell_aa_fsr", "yard_aa_fsr_notaa"]


def run_case(plugin, case, on_frame=None):
    plugin.set_scene(case.scene)
    for n in case.frames:
        plugin.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        if on_frame:
            on_frame(n)
    plugin.engine.wait()


def snapshot(plugin):
    return {name: plugin.engine.read(b) for b, name in ALL_BUFFERS.items()}


def diff_buffers(a, b):
    """Names of buffers whose bytes differ, with the first differing pixel."""
    bad = {}
    for name in a:
        x, y = a[name], b[name]
        ne = (x.view(np.uint8).reshape(x.shape[0], x.shape[1], -1) != y.view(np.uint8).reshape(y.shape[0], y.shape[1], -1)).any(axis=2)
        if ne.any():
            ys, xs = np.nonzero(ne)
            bad[name] = f"{int(ne.sum())} px, first (x={xs[0]}, y={ys[0]}): {x[ys[0], xs[0]]} vs {y[ys[0], xs[0]]}"
    return bad


def product_default_plugin(flags=0):
    """A plugin in the mode bench.py times: no suite-wide HK_CTX_EXACT_TRAVERSAL, no ray counters - one-level walk from LDS for
    scenes under one transform, direction-threaded trees + the queue-based indirect pass beyond LDS."""
    with product_default_traversal():
        return hk.HikariPlugin(device=0, flags=flags)


def _as_float(x):
    if x.dtype == np.uint16:   # rgba16f
        return x.view(np.float16).astype(np.float32)
    if x.dtype == np.uint32:   # rgba8snorm normals
        b = x.view(np.int8).astype(np.float32)
        return np.maximum(b / 127.0, -1.0)
    return x.astype(np.float32)


def rendered_deviation(a, b):
    """{name: (relative L2, fraction of pixels with any differing byte)} over every RENDERED buffer of two snapshots - everything a
    frame's consumers see: the G-buffer, albedo, render / variance, the denoiser's planes, the tone-mapped image and the
    anti-aliasing tail.  (The reservoir records are compared where the claim is bit equality - test_default_mode_sequence_gpu.py
    holds config 2's default mode to the exact walk in EVERY buffer; until round 5 an any-hit walk in another order could keep another
    occluder in an occluded sample's sample_position - since then the rays whose occluder is kept walk the reference's order)"""
    out = {}
    for name in a:
        if name.startswith("reservoir"):
            continue
        x, y = a[name], b[name]
        ne = (x.view(np.uint8).reshape(x.shape[0], x.shape[1], -1) != y.view(np.uint8).reshape(y.shape[0], y.shape[1], -1)).any(axis=2)
        if not ne.any():
            out[name] = (0.0, 0.0)
            continue
        fx, fy = np.nan_to_num(_as_float(x), posinf=0.0, neginf=0.0), np.nan_to_num(_as_float(y), posinf=0.0, neginf=0.0)
        out[name] = (float(np.linalg.norm(fx - fy) / max(float(np.linalg.norm(fy)), 1e-30)), float(ne.mean()))
    return out


def assert_rendered_within(a, b, what, tol=1e-3):
    """The north star's bar for a mode that may visit candidates in another order than the reference: relative L2 <= 1e-3 per
    buffer.  Returns (worst relative L2, worst differing-pixel fraction) for the report."""
    dev = rendered_deviation(a, b)
    bad = {k: v for k, v in dev.items() if not v[0] <= tol}
    assert not bad, f"{what}: {bad}"
    return max(v[0] for v in dev.values()), max(v[1] for v in dev.values())


GBUFFER = ("position", "normal", "depth_gradient", "instance_material", "velocity_uv", "albedo")   # names, as diff_buffers reports them
GBUFFER_IDS = (F.BUF_POSITION, F.BUF_NORMAL, F.BUF_DEPTH_GRADIENT, F.BUF_INSTANCE_MATERIAL, F.BUF_VELOCITY_UV)


def run_case_with_host_gbuffer(source, plugin, case):
    """A host that keeps its raster prepass: per frame hk_frame_begin, then the five G-buffer
    planes written with hk_write_buffer, then hk_frame_render(HK_FRAME_EXTERNAL_GBUFFER).  `source` renders the same
    frames normally and supplies the planes."""
    source.set_scene(case.scene)
    plugin.set_scene(case.scene)
    s = case.settings
    for n in case.frames:
        source.render(case.camera, s, lights=case.lights, frame_number=n, antialias=case.antialias)
        size = (case.camera.width, case.camera.height, s.upscale.ratio())
        if plugin._size != size:
            plugin.engine.resize(*size)
            plugin._size = size
        frame, view, pview = hk.frame_uniform(s, n), case.camera.view_uniform(), case.camera.previous_view_uniform()
        plugin.engine.frame_begin(frame, view, pview, case.lights)        # selects the planes frame n writes
        for b in GBUFFER_IDS:
            plugin.engine.write(b, source.engine.read(b))
        plugin.engine.frame_render(frame, view, pview, case.lights, s.to_c(), F.FRAME_EXTERNAL_GBUFFER | (F.FRAME_ANTIALIAS if case.antialias else 0))
    plugin.engine.wait()


def random_case(seed):
    """Seeded point of the HikariSettings space (lib.rs:402-433) x scene x odd image size x AA tail, for the sweeps."""
    rng = np.random.default_rng(1000 + seed)
    ratio = float(rng.choice([1.0, 1.25, 1.5, 2.0]))
    upscale = hk.Upscale.SmaaTu4x(ratio) if rng.random() < 0.5 else hk.Upscale.Fsr1(ratio, 0.2)
    s = hk.HikariSettings(
        direct_validate_interval=int(rng.integers(1, 5)), emissive_validate_interval=int(rng.integers(1, 7)),
        max_temporal_reuse_count=int(rng.choice([1, 8, 50])), max_spatial_reuse_count=int(rng.choice([4, 100, 800])),
        max_reservoir_lifetime=float(rng.choice([0.5, 3.0, 100.0])), solar_angle=float(rng.choice([0.0, 0.046, 0.2])),
        indirect_bounces=int(rng.integers(0, 4)), max_indirect_luminance=float(rng.choice([0.5, 10.0])),
        temporal_reuse=bool(rng.random() < 0.8), emissive_spatial_reuse=bool(rng.random() < 0.5),
        indirect_spatial_reuse=bool(rng.random() < 0.7), denoise=bool(rng.random() < 0.7),
        taa=hk.Taa.Jasmine if rng.random() < 0.6 else hk.Taa.NONE, upscale=upscale)
    w, h = int(rng.integers(33, 120)), int(rng.integers(25, 90))
    if rng.random() < 0.5:
        scene, cam, lights = cornell_scene(), hk.cornell_camera(w, h), hk.lights_uniform()
    else:
        scene, sun = synthetic_scene(n_boxes=8, n_spheres=3, n_emitters=2, sphere_rings=5, sphere_segs=6, textured=bool(rng.random() < 0.5))
        cam, lights = synthetic_camera(w, h), hk.lights_uniform(directional=sun)
    antialias = bool(rng.random() < 0.6)
    first = int(rng.integers(1, 7))
    return Case(f"random{seed}", scene, cam, s, lights=lights, frames=range(first, first + 3), antialias=antialias)


def motion_case(seed):
    """Seeded moving-camera + moving-instances sequence on a synthetic yard (40 % of them above the 32 KB LDS limit, i.e.
    on the two-slot asynchronous instance upload), random settings and anti-aliasing tail.  Returns a dict; run_motion_case
    drives a plugin through it."""
    from bevy_hikari_amd.scenes import synthetic_scene

    rng = np.random.default_rng(7000 + seed)
    big = bool(rng.random() < 0.4)
    scene, sun = synthetic_scene(n_boxes=int(rng.integers(6, 26)), n_spheres=int(rng.integers(1, 6)), n_emitters=int(rng.integers(1, 4)),
                                 sphere_rings=12 if big else 5, sphere_segs=16 if big else 6, seed=int(rng.integers(1, 1 << 30)))
    n_inst = len(scene.instances)
    movers = tuple(int(i) for i in rng.choice(n_inst, size=min(n_inst, int(rng.integers(1, 8))), replace=False))
    w, h = int(rng.integers(48, 160)), int(rng.integers(40, 110))
    ratio = float(rng.choice([1.0, 1.5, 2.0]))
    s = hk.HikariSettings(indirect_bounces=int(rng.integers(0, 3)), upscale=hk.Upscale.SmaaTu4x(ratio) if rng.random() < 0.5 else hk.Upscale.Fsr1(ratio, 0.3),
                          taa=hk.Taa.Jasmine if rng.random() < 0.7 else hk.Taa.NONE, emissive_spatial_reuse=bool(rng.random() < 0.5))
    return dict(scene=scene, lights=hk.lights_uniform(directional=sun), movers=movers, size=(w, h), settings=s, antialias=bool(rng.random() < 0.7),
                eye=np.array([6.4, 4.4, 8.0]) + rng.normal(0, 0.5, 3), drift=rng.normal(0, 0.08, 3), frames=5)


def run_motion_case(plugins, case, on_frame):
    """Both plugins through the same animated sequence; on_frame(n) after every frame."""
    from bevy_hikari_amd.scenes import animate

    for p in plugins:
        p.set_scene(case["scene"])
    cur = case["scene"]
    for n in range(1, case["frames"] + 1):
        if n > 1:
            cur = animate(cur, n - 1, movers=case["movers"])
            for p in plugins:
                p.update_instances(cur)
        cam = hk.Camera(hk.look_at_transform(tuple(case["eye"] + case["drift"] * n), (0.0, 0.6, 0.0)), *case["size"])
        for p in plugins:
            p.render(cam, case["settings"], lights=case["lights"], frame_number=n, antialias=case["antialias"])
        on_frame(n)
