"""Parity on the large scenes (BASELINE configs 3, 4, 5 and their relatives) at FULL size: the exact walk bit for bit against the
oracle, the product default (threaded trees, queue-based indirect pass, wide walk) within the north star's 1e-3 - measured: the
reference's own hits since round 5 (test_default_mode_sequence_gpu.py has the 32-frame curves).  Split from test_parity_gpu.py."""
import ctypes as C
import os

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import ALL_BUFFERS, assert_rendered_within, diff_buffers, make_case, oracle, product_default_plugin, report, run_case, snapshot
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_sponza_class_vs_oracle():
    """BASELINE config 3 stand-in (seeded synthetic, ~256 k unique triangles, 409 instances, 50
    materials, 8 emitters, sun 100 000 lux, 3 bounces + denoise): too big for LDS staging, so this
    is the global-memory traversal path; compared with the oracle at a reduced resolution."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large()
    s = hk.HikariSettings(indirect_bounces=3, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = synthetic_camera(320, 180, extent=9.0)
    lights = hk.lights_uniform(directional=sun)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    for n in (1, 2, 3):
        for p in (gpu, cpu):
            p.render(cam, s, lights=lights, frame_number=n)
    bad = diff_buffers(snapshot(gpu), snapshot(cpu))
    assert bad == {}, bad
    sg, sc = gpu.engine.stats(), cpu.engine.stats()
    assert (sg.rays_tlas, sg.rays_blas) == (sc.rays_tlas, sc.rays_blas) and sg.rays_tlas > 320 * 180 * 3
    out = gpu.output(s)
    assert np.isfinite(out).all() and out[..., :3].max() > 0.05


def test_config3_full_1080p_vs_oracle():
    """BASELINE config 3 stand-in at the FULL 1920x1080 (3 bounces, denoise, sun + 8 emitters): every buffer of two frames bit for
    bit against the oracle - the global-memory (non-LDS) traversal path at the size the config is quoted on."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large()
    s = hk.HikariSettings(indirect_bounces=3, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = synthetic_camera(1920, 1080, extent=9.0)
    lights = hk.lights_uniform(directional=sun)
    gpu, cpu, dflt = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS), oracle(), product_default_plugin()
    for p in (gpu, cpu, dflt):
        p.set_scene(scene)
    worst = (0.0, 0.0)
    for n in (1, 2):
        for p in (gpu, cpu, dflt):
            p.render(cam, s, lights=lights, frame_number=n)
        want = snapshot(cpu)
        bad = diff_buffers(snapshot(gpu), want)
        assert bad == {}, (n, bad)
        # what bench.py times for this config - direction-threaded trees, the queue-based indirect pass - against the oracle directly
        worst = max(worst, assert_rendered_within(snapshot(dflt), want, f"config 3 at 1920x1080 frame {n}, product default mode"))
    assert dflt.engine.traversal_mode()[0] == "threaded" and dflt.engine.indirect_schedule() == "wavefront" and dflt.engine.wide_walk()
    assert dflt.engine.stats().wide_stack_lost == 0
    report("default_mode_config3_1080p_vs_oracle", {"worst_relative_l2": worst[0], "worst_fraction_of_pixels_differing": worst[1], "frames": 2})
    sg, sc = gpu.engine.stats(), cpu.engine.stats()
    assert (sg.rays_tlas, sg.rays_blas) == (sc.rays_tlas, sc.rays_blas) and sg.rays_tlas > 1920 * 1080 * 2


def test_config4_city_class_vs_oracle():
    """BASELINE config 4 stand-in (1.5 M unique triangles, 2002 instances, sun 10 000 lux, 2 bounces) compared with the ORACLE:
    640x360, every buffer of two frames bit for bit (the 4K run of the same scene below checks size-independent properties)."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
    sun = dict(sun, illuminance=10000.0)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = synthetic_camera(640, 360, extent=30.0)
    lights = hk.lights_uniform(directional=sun)
    gpu, cpu, dflt = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS), oracle(), product_default_plugin()
    for p in (gpu, cpu, dflt):
        p.set_scene(scene)
    worst = (0.0, 0.0)
    for n in (1, 2):
        for p in (gpu, cpu, dflt):
            p.render(cam, s, lights=lights, frame_number=n)
        want = snapshot(cpu)
        bad = diff_buffers(snapshot(gpu), want)
        assert bad == {}, (n, bad)
        worst = max(worst, assert_rendered_within(snapshot(dflt), want, f"config 4 (city class) frame {n}, product default mode"))
    assert dflt.engine.traversal_mode()[0] == "threaded" and dflt.engine.indirect_schedule() == "wavefront" and dflt.engine.wide_walk()
    assert dflt.engine.stats().wide_stack_lost == 0
    report("default_mode_config4_city_class_vs_oracle", {"worst_relative_l2": worst[0], "worst_fraction_of_pixels_differing": worst[1], "frames": 2})
    sg, sc = gpu.engine.stats(), cpu.engine.stats()
    assert (sg.rays_primary, sg.rays_tlas, sg.rays_blas) == (sc.rays_primary, sc.rays_tlas, sc.rays_blas)
    out = gpu.output(s)
    assert np.isfinite(out).all() and out[..., :3].max() > 0.05

def test_wide_walk_against_the_threaded_walk_and_the_oracle():
    """Scenes beyond LDS, product default: the closest-hit walks read the wide records (HK_TRAVERSAL_WIDE; hk_wide.hpp).  The same
    frames with HK_CTX_NO_WIDE_WALK (the threaded skip-link walk everywhere) and on the oracle: both within the north star's 1e-3
    of the oracle in every rendered buffer, and the two G-buffers - primary rays, where a different closest hit would show first -
    agree in all but exact ties.  (Instance motion - the records are derived again after a device refit - is
    test_device_refit.py::test_refit_with_direction_threaded_orderings_stays_within_tolerance, which runs in this mode.)"""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large
    from cases import product_default_traversal

    scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = synthetic_camera(640, 360, extent=30.0)
    lights = hk.lights_uniform(directional=dict(sun, illuminance=10000.0))
    cpu = oracle()
    with product_default_traversal():
        wide, again, threaded, exact = (hk.HikariPlugin(device=0), hk.HikariPlugin(device=0), hk.HikariPlugin(device=0, flags=F.CTX_NO_WIDE_WALK),
                                        hk.HikariPlugin(device=0, flags=F.CTX_EXACT_TRAVERSAL))
    for p in (cpu, wide, again, threaded, exact):
        p.set_scene(scene)

    def frames(numbers):
        for n in numbers:
            for p in (cpu, wide, again, threaded):
                p.render(cam, s, lights=lights, frame_number=n)
        want = snapshot(cpu)
        # the same frames on a second context: every byte equal - which lanes of a dry wave helped which walk (k_wf_trace_wide's work
        # sharing) depends on timing, the result must not (the order-independent tie rule of wide_triangle)
        assert diff_buffers(snapshot(again), snapshot(wide)) == {}
        a = assert_rendered_within(snapshot(wide), want, f"wide walk, frame {numbers[-1]}")
        b = assert_rendered_within(snapshot(threaded), want, f"threaded walk, frame {numbers[-1]}")
        ia, ib = wide.engine.read(F.BUF_INSTANCE_MATERIAL), threaded.engine.read(F.BUF_INSTANCE_MATERIAL)
        assert float((ia[..., 0] != ib[..., 0]).mean()) <= 1e-5
        return a, b

    first = frames((1, 2))
    assert wide.engine.wide_walk() and not threaded.engine.wide_walk()
    assert wide.engine.stats().wide_stack_lost == 0  # (no pending subtree was dropped: HkStats)
    assert wide.engine.traversal_mode() == threaded.engine.traversal_mode() == ("threaded", 8)
    exact.render(cam, s, lights=lights, frame_number=1)
    assert exact.engine.traversal_mode()[0] == "reference" and not exact.engine.wide_walk()
    report("wide_walk_vs_threaded_vs_oracle", {"wide_vs_oracle": first[0], "threaded_vs_oracle": first[1]})

def test_wide_walk_inside_few_large_meshes():
    """The other shape of a long walk: few instances of two 100 k-triangle meshes - the walks are long INSIDE a mesh tree (17 levels),
    so what a dry wave of the trace stage hands to its idle lanes are mesh-tree entries (hk_wide.hpp: blas_base, tombstones), and the
    stacks are at their deepest.  Product default against the oracle (1e-3), two contexts byte for byte, no dropped stack entry."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large
    from cases import product_default_traversal

    scene, sun = synthetic_large(0x5EED0007, 2, 160, 320, 6, 8, 2, 3.0)
    s = hk.HikariSettings(indirect_bounces=3, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = synthetic_camera(320, 180, extent=3.0)
    lights = hk.lights_uniform(directional=sun)
    cpu = oracle()
    with product_default_traversal():
        wide, again = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0)
    for p in (cpu, wide, again):
        p.set_scene(scene)
    worst = (0.0, 0.0)
    for n in (1, 2, 3):
        for p in (cpu, wide, again):
            p.render(cam, s, lights=lights, frame_number=n)
        worst = max(worst, assert_rendered_within(snapshot(wide), snapshot(cpu), f"two large meshes, frame {n}, product default mode"))
        assert diff_buffers(snapshot(again), snapshot(wide)) == {}
    assert wide.engine.wide_walk() and wide.engine.indirect_schedule() == "wavefront" and wide.engine.stats().wide_stack_lost == 0
    report("wide_walk_few_large_meshes_vs_oracle", {"worst_relative_l2": worst[0], "worst_fraction_of_pixels_differing": worst[1], "frames": 3})


def test_config5_full_4k_8_bounces_vs_oracle():
    """BASELINE config 5 at its FULL size (Cornell 3840x2160, 8 bounces, emissive + indirect spatial reuse, denoise off): two
    frames, every buffer bit for bit against the oracle."""
    s = hk.HikariSettings(indirect_bounces=8, emissive_spatial_reuse=True, denoise=False, upscale=hk.Upscale.SMAA_TU_1_0)
    scene, cam = hk.load_cornell(), hk.cornell_camera(3840, 2160)
    gpu, cpu, dflt = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS), oracle(), product_default_plugin()
    for p in (gpu, cpu, dflt):
        p.set_scene(scene)
    for n in (1, 2):
        for p in (gpu, cpu, dflt):
            p.render(cam, s, frame_number=n)
    want = snapshot(cpu)
    bad = diff_buffers(snapshot(gpu), want)
    assert bad == {}, bad
    rel, frac = assert_rendered_within(snapshot(dflt), want, "config 5 at 3840x2160 x 8 bounces, product default mode")
    assert dflt.engine.traversal_mode()[0] == "one-level"
    # a 4K launch is many rounds of workgroups: both spatial passes of both frames took the WINDOWED form of k_spatial_reuse (kernels.hip)
    assert gpu.engine.spatial_windowed_launches() == 4 and dflt.engine.spatial_windowed_launches() == 4
    report("default_mode_config5_4k_vs_oracle", {"worst_relative_l2": rel, "worst_fraction_of_pixels_differing": frac, "frames": 2})
    sg, sc = gpu.engine.stats(), cpu.engine.stats()
    assert (sg.rays_primary, sg.rays_tlas, sg.rays_blas) == (sc.rays_primary, sc.rays_tlas, sc.rays_blas)


def _threaded_vs_exact(name, scene, cam, s, lights, frames, tol_pixels):
    """The product default for scenes beyond the LDS copy (flags 0, NO ray counters - HK_CTX_COUNT_RAYS would switch the queue-based
    schedule off, context.hip use_wavefront: the kernels bench.py times are the ones that run here: direction-threaded trees, the
    wavefront schedule of the indirect pass, the wide walk) against HK_CTX_EXACT_TRAVERSAL (the reference's single order, bit-exact vs
    the oracle in the tests above) on the same frames: the north star's 1e-3 relative L2 on the output, and the fraction of pixels
    whose primary hit (instance id) or any G-buffer byte differs - exact ties between two candidates are the only thing the order
    can change."""
    from cases import product_default_traversal

    exact = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS)
    with product_default_traversal():
        fast = hk.HikariPlugin(device=0)
    for p in (exact, fast):
        p.set_scene(scene)
    for n in frames:
        for p in (exact, fast):
            p.render(cam, s, lights=lights, frame_number=n)
    a, b = fast.output(s), exact.output(s)
    rel = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    ia, ib = fast.engine.read(F.BUF_INSTANCE_MATERIAL), exact.engine.read(F.BUF_INSTANCE_MATERIAL)
    pa, pb = fast.engine.read(F.BUF_POSITION), exact.engine.read(F.BUF_POSITION)
    hit_diff = float((ia[..., 0] != ib[..., 0]).mean())
    pos_diff = float((pa.view(np.uint32) != pb.view(np.uint32)).any(axis=2).mean())
    se = exact.engine.stats()
    report = {"case": name, "traversal": list(fast.engine.traversal_mode()), "schedule": fast.engine.indirect_schedule(), "wide_walk": bool(fast.engine.wide_walk()),
              "rel_l2": rel, "primary_hit_instance_differs": hit_diff, "gbuffer_position_differs": pos_diff, "rays_exact": int(se.rays_tlas + se.rays_blas)}
    assert fast.engine.traversal_mode()[0] == "threaded" and exact.engine.traversal_mode()[0] == "reference"
    assert fast.engine.indirect_schedule() == "wavefront" and fast.engine.wide_walk() and fast.engine.stats().wide_stack_lost == 0
    print("threaded vs exact traversal:", report)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json

        with open(os.path.join(out_dir, f"threaded_traversal_{name}.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert rel <= 1e-3 and hit_diff <= tol_pixels and pos_diff <= 10 * tol_pixels, report
    return report


def test_threaded_traversal_config3_within_tolerance():
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large()
    _threaded_vs_exact("config3_1080p", scene, synthetic_camera(1920, 1080, extent=9.0), hk.HikariSettings(indirect_bounces=3, upscale=hk.Upscale.SMAA_TU_1_0),
                       hk.lights_uniform(directional=sun), (1, 2, 3, 4), 1e-5)


def test_threaded_traversal_config4_within_tolerance():
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
    _threaded_vs_exact("config4_1080p", scene, synthetic_camera(1920, 1080, extent=30.0), hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0),
                       hk.lights_uniform(directional=dict(sun, illuminance=10000.0)), (1, 2, 3), 1e-5)


def test_threaded_traversal_config4_full_4k_within_tolerance():
    """BASELINE config 4 at the size it is benchmarked at (3840x2160) in the mode it is benchmarked in - the product default:
    threaded orderings + wavefront schedule - against HK_CTX_EXACT_TRAVERSAL on the same frames."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
    _threaded_vs_exact("config4_4k", scene, synthetic_camera(3840, 2160, extent=30.0), hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0),
                       hk.lights_uniform(directional=dict(sun, illuminance=10000.0)), (1, 2, 3), 1e-5)


def test_config4_full_4k_row_ranges_vs_oracle():
    """Config 4 at its full 3840x2160, exact traversal, against the ORACLE on three row ranges of the frame (top edge, middle,
    bottom edge): the oracle renders only those rows plus the aprons their passes read (orc_frame_stage_rows) - frame 1 with the
    aprons frame 2's history needs, then frame 2 - and every buffer's rows must equal the GPU's full-frame rows bit for bit."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large
    from oracle_lib import oracle_api, oracle_engine

    scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    sc = s.to_c()
    W, H = 3840, 2160
    cam = synthetic_camera(W, H, extent=30.0)
    lights = hk.lights_uniform(directional=dict(sun, illuminance=10000.0))
    view, pview = cam.view_uniform(), cam.previous_view_uniform()
    gpu = hk.Engine(device=0)           # (conftest: HK_CTX_EXACT_TRAVERSAL)
    cpu = oracle_engine()
    for e in (gpu, cpu):
        e.upload_noise(); e.upload_scene(scene); e.resize(W, H, 1.0)
    stage_rows = oracle_api().dll.orc_frame_stage_rows
    ranges = [(0, 24), (1068, 1092), (2136, 2160)]
    SP, DEN = 21, 16                     # spatial-reuse and denoiser aprons (rows), as in hk_band_plan_for
    clamp = lambda v: min(max(v, 0), H)
    checked = 0
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
        cur, prev = n % 2, 1 - n % 2
        for b, name in ALL_BUFFERS.items():
            if name.startswith("previous_") or name in ("upscale_output", "taa_output", "upscale_sharpened"):
                continue
            if name.startswith("reservoir") and (int(name[9:]) % 2) != prev:
                continue                  # (the buffers this frame wrote: the ping-pong half temporal / spatial store into)
            if name.startswith("internal"):
                continue                  # a-trous scratch: holds the last channel's intermediate levels with their shrinking aprons
            a, o = gpu.read(b), cpu.read(b)
            for r0, r1 in ranges:
                x, y = a[r0:r1], o[r0:r1]
                assert (x.view(np.uint8) == y.view(np.uint8)).all(), f"frame {n}: {name} rows [{r0},{r1}) differ from the oracle at 4K"
                checked += 1
    assert checked >= 2 * 3 * 20


def test_config3_default_mode_under_instance_motion_with_refit_1080p():
    """Config 3 at 1920x1080 in the PRODUCT DEFAULT (threaded orderings + wavefront), instances moving every frame through the
    device refit: against HK_CTX_EXACT_TRAVERSAL fed the same poses - 1e-3 on the output, G-buffer hits equal but for ties.
    Both contexts resolve the scatter race the same way (HK_CTX_DETERMINISTIC_SCATTER), so what is compared is the traversal."""
    from cases import product_default_traversal
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large
    from test_device_refit import pose

    s = hk.HikariSettings(indirect_bounces=3, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = synthetic_camera(1920, 1080, extent=9.0)
    view, pview = cam.view_uniform(), cam.previous_view_uniform()
    engines, scenes = [], []
    for default_mode in (False, True):
        scene, sun = synthetic_large()
        scenes.append(scene)
        if default_mode:
            with product_default_traversal():
                e = hk.Engine(device=0, flags=F.CTX_DETERMINISTIC_SCATTER)
        else:
            e = hk.Engine(device=0, flags=F.CTX_DETERMINISTIC_SCATTER | F.DEFAULT_CTX_FLAGS)
        e.upload_noise(); e.upload_scene(scene); e.resize(1920, 1080, 1.0)
        engines.append(e)
    lights = hk.lights_uniform(directional=sun)
    rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in scenes[0].instances], dtype=np.float32)
    movers = list(range(3, len(rest), 9))
    for n in range(1, 6):
        if n > 1:
            for e, scene in zip(engines, scenes):
                for k, i in enumerate(movers):
                    scene.builder.set_instance_transform(i, pose(rest[i], n - 1, k))
                assert e.refit_instances(scene.builder) == len(movers)
        for e in engines:
            e.frame_render(hk.frame_uniform(s, n), view, pview, lights, s.to_c())
    exact, fast = engines
    assert fast.indirect_schedule() == "wavefront" and fast.stats().scene_device_refits == 4
    a = np.stack([fast.read_f16(F.BUF_DENOISE_RENDER0 + i) for i in range(3)]).astype(np.float64)
    b = np.stack([exact.read_f16(F.BUF_DENOISE_RENDER0 + i) for i in range(3)]).astype(np.float64)
    rel = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    ia, ib = fast.read(F.BUF_INSTANCE_MATERIAL), exact.read(F.BUF_INSTANCE_MATERIAL)
    hit_diff = float((ia[..., 0] != ib[..., 0]).mean())
    report = {"case": "config3_1080p_motion_refit", "rel_l2": rel, "primary_hit_instance_differs": hit_diff, "movers": len(movers), "frames": 5}
    print(report)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json

        with open(os.path.join(out_dir, "threaded_traversal_config3_1080p_motion_refit.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert rel <= 1e-3 and hit_diff <= 1e-5, report


def test_threaded_traversal_flight_helmet_vs_oracle():
    """... and against the ORACLE itself on the reference's textured asset (deep BLASes): default flags, 1e-3."""
    from cases import product_default_traversal

    case = make_case("flight_helmet")
    with product_default_traversal():
        gpu = hk.HikariPlugin(device=0)
    cpu = oracle()
    for p in (gpu, cpu):
        run_case(p, case)
    a, b = gpu.output(case.settings), cpu.output(case.settings)
    rel = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert rel <= 1e-3, rel


def test_cornell_8k_row_ranges_vs_oracle():
    """The largest frame a 16:9 display asks for, 7680x4320 (33 M pixels, 21 GB of reservoir buffers - sized for 288 GB of HBM):
    two frames of Cornell, 2 bounces, exact traversal, against the oracle on three row ranges (top edge, the middle of the box,
    bottom edge) with the aprons their passes read, bit for bit; plus whole-frame properties."""
    import psutil
    from oracle_lib import oracle_api, oracle_engine

    if psutil.virtual_memory().available < 96 * 2 ** 30:   # the ORACLE's 8K context is ~30 GB of host memory
        pytest.skip("not enough host memory for the oracle's 8K buffers")
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    sc = s.to_c()
    W, H = 7680, 4320
    cam = hk.cornell_camera(W, H)
    lights = hk.lights_uniform()
    view, pview = cam.view_uniform(), cam.previous_view_uniform()
    scene = hk.load_cornell()
    gpu, cpu = hk.Engine(device=0), oracle_engine()
    for e in (gpu, cpu):
        e.upload_noise(); e.upload_scene(scene); e.resize(W, H, 1.0)
    stage_rows = oracle_api().dll.orc_frame_stage_rows
    ranges = [(0, 8), (2156, 2164), (4312, 4320)]
    SP, DEN = 21, 16
    clamp = lambda v: min(max(v, 0), H)
    checked = 0
    for n in (1, 2):
        f = hk.frame_uniform(s, n)
        gpu.frame_render(f, view, pview, lights, sc)
        cpu.frame_begin(f, view, pview, lights)
        extra = (SP + DEN) if n == 1 else 0
        for r0, r1 in ranges:
            for stage, apron in ((F.STAGE_TEMPORAL, SP + DEN), (F.STAGE_SPATIAL, DEN), (F.STAGE_POST_PROCESS, 0)):
                rc = stage_rows(cpu.ctx, stage, C.byref(sc), 0, clamp(r0 - apron - extra), clamp(r1 + apron + extra))
                assert rc == 0, cpu.api.last_error()
        gpu.wait()
        prev = 1 - n % 2
        for b, name in ALL_BUFFERS.items():
            if name.startswith("previous_") or name in ("upscale_output", "taa_output", "upscale_sharpened") or name.startswith("internal"):
                continue
            if name.startswith("reservoir") and (int(name[9:]) % 2) != prev:
                continue
            a, o = gpu.read(b), cpu.read(b)
            for r0, r1 in ranges:
                assert (a[r0:r1].view(np.uint8) == o[r0:r1].view(np.uint8)).all(), f"frame {n}: {name} rows [{r0},{r1}) differ from the oracle at 8K"
                checked += 1
    assert checked >= 2 * 3 * 20
    tone = gpu.read_f16(F.BUF_TONE_MAPPED)
    assert tone.shape[:2] == (H, W) and np.isfinite(tone).all() and tone[H // 2].max() > 0.0 and (tone[0] == tone[0, 0]).all()


def test_city_class_4k_properties():
    """BASELINE config 4 stand-in at its full size on one GPU (seeded synthetic, ~1.5 M unique
    triangles, 2002 instances, 3840x2160, 2 bounces): determinism, dispatch row-range independence
    (what the 8-band split relies on), finite output, sane ray counts."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
    sun = dict(sun, illuminance=10000.0)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = synthetic_camera(3840, 2160, extent=30.0)
    lights = hk.lights_uniform(directional=sun)
    runs = []
    for rep in range(2):
        p = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS if rep == 0 else 0)
        p.set_scene(scene)
        for n in (1, 2):
            p.render(cam, s, lights=lights, frame_number=n)
        runs.append(p)
    a = snapshot(runs[0])
    assert diff_buffers(a, snapshot(runs[1])) == {}
    out = runs[0].output(s)
    assert np.isfinite(out).all() and out[..., :3].max() > 0.05
    st = runs[0].engine.stats()
    px = 3840 * 2160 * 2
    assert st.rays_primary == px and px < st.rays_tlas <= px * 7 and st.rays_blas <= px * 4
    e = runs[1].engine
    for b0, b1 in ((0, 270), (270, 1000), (1000, 2160)):
        e.pass_run(F.PASS_INDIRECT, 0, b0, b1)
    for b0, b1 in ((0, 1111), (1111, 2160)):
        e.pass_run(F.PASS_INDIRECT_SPATIAL_REUSE, 0, b0, b1)
    assert diff_buffers(snapshot(runs[1]), a) == {}
