"""Parity of the HIP path (through the C ABI) with the CPU oracle and the committed golden
fixtures.  Bar: BIT-EXACT on every screen-space buffer for static scenes (the numeric contract
makes f32 arithmetic reproducible); <= 1e-3 relative L2 (BASELINE.json north_star)
where the reference itself races (moving camera)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import ALL_BUFFERS, CASE_NAMES, assert_rendered_within, diff_buffers, make_case, product_default_plugin, run_case, snapshot
from conftest import ROOT

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")


def oracle():
    from oracle_lib import oracle_plugin

    return oracle_plugin()


def report(name, data):
    """Printed, and kept under gpurun_out/ when the suite runs on the GPU box (copied to profiles/ by the round's scripts)."""
    print(name, data)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json

        with open(os.path.join(out_dir, name + ".json"), "w") as f:
            json.dump(data, f, indent=1)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_bit_exact_vs_oracle_every_frame(name):
    """Every buffer of every frame, bit for bit, in the reference's walk (the suite's HK_CTX_EXACT_TRAVERSAL) - and, beside it, the
    SAME frames from a context in the product default (what bench.py times: flags 0, no counters) held to the oracle directly:
    every rendered buffer within the north star's 1e-3."""
    case = make_case(name)
    gpu, cpu, dflt = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS), oracle(), product_default_plugin()
    for p in (gpu, cpu, dflt):
        p.set_scene(case.scene)
    for n in case.frames:
        for p in (gpu, cpu, dflt):
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        want = snapshot(cpu)
        bad = diff_buffers(snapshot(gpu), want)
        assert bad == {}, f"{name} frame {n}: {bad}"
        assert_rendered_within(snapshot(dflt), want, f"{name} frame {n}, product default mode")
    sg, sc = gpu.engine.stats(), cpu.engine.stats()
    assert (sg.rays_primary, sg.rays_tlas, sg.rays_blas) == (sc.rays_primary, sc.rays_tlas, sc.rays_blas)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_matches_golden_fixture(name):
    """Same check without the oracle in the loop: committed fixtures (tests/tools/make_golden.py)."""
    case = make_case(name)
    gpu = hk.HikariPlugin(device=0)
    run_case(gpu, case)
    snap = snapshot(gpu)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    for key in g.files:
        if key.startswith("sha256_"):
            got = np.frombuffer(hashlib.sha256(snap[key[7:]].tobytes()).digest(), dtype=np.uint8)
            assert (got == g[key]).all(), f"{name}: {key[7:]}"
    a = gpu.output(case.settings)
    base = "denoise_render" if case.settings.denoise else "render"
    if base == "denoise_render":
        b = np.stack([g[f"denoise_render{i}"].view(np.float16).astype(np.float32) for i in range(3)])
        assert np.linalg.norm(a - b) <= 1e-3 * np.linalg.norm(b)
        # the product default mode (what bench.py times) against the same committed frames, no oracle in the loop
        dflt = product_default_plugin()
        run_case(dflt, case)
        a = dflt.output(case.settings)
        assert np.linalg.norm(a - b) <= 1e-3 * np.linalg.norm(b), f"{name}: product default mode vs the golden fixture"


def test_nodes_path_equals_frame_render():
    case = make_case("cornell_upscale2")
    a, b = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0)
    for p, by_nodes in ((a, False), (b, True)):
        p.set_scene(case.scene)
        for n in case.frames:
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n, by_nodes=by_nodes, antialias=case.antialias)
    assert diff_buffers(snapshot(a), snapshot(b)) == {}


def test_full_size_1080p_vs_oracle():
    """BASELINE config 2 at its real size: three frames, every buffer, bit for bit."""
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    scene, cam = hk.load_cornell(), hk.cornell_camera(1920, 1080)
    gpu, cpu, dflt = hk.HikariPlugin(device=0), oracle(), product_default_plugin()
    for p in (gpu, cpu, dflt):
        p.set_scene(scene)
    assert dflt.engine.traversal_mode()[0] == "one-level"
    for n in (1, 2, 3):
        for p in (gpu, cpu, dflt):
            p.render(cam, s, frame_number=n)
    want = snapshot(cpu)
    bad = diff_buffers(snapshot(gpu), want)
    assert bad == {}, bad
    # the kernels bench.py TIMES - one-level walk from LDS, no counters, frame pipelining - against the oracle directly, at the size
    # the metric is quoted on (until now only through a chain of replays)
    rel, frac = assert_rendered_within(snapshot(dflt), want, "config 2 at 1920x1080, product default mode")
    report("default_mode_config2_1080p_vs_oracle", {"worst_relative_l2": rel, "worst_fraction_of_pixels_differing": frac, "frames": 3})


def test_full_size_properties_4k_8_bounces():
    """BASELINE config 5 (Cornell 4K, 8 bounces, emissive + indirect spatial reuse, denoise off):
    size-independent properties - run-to-run determinism, row-range independence of a dispatch,
    finite output, ray count within the analytic bound of SURVEY 8d."""
    s = hk.HikariSettings(indirect_bounces=8, emissive_spatial_reuse=True, denoise=False, upscale=hk.Upscale.SMAA_TU_1_0)
    scene, cam = hk.load_cornell(), hk.cornell_camera(3840, 2160)
    runs = []
    for rep in range(2):
        p = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS)
        p.set_scene(scene)
        for n in (1, 2, 3):
            p.render(cam, s, frame_number=n)
        runs.append(p)
    a, b = snapshot(runs[0]), snapshot(runs[1])
    assert diff_buffers(a, b) == {}
    out = runs[0].output(s)
    assert np.isfinite(out).all() and out[..., :3].max() > 0.1
    st = runs[0].engine.stats()
    px = 3840 * 2160 * 3
    assert st.rays_primary == px
    assert st.rays_tlas <= px * (2 + 2 * 8 + 1) and st.rays_blas <= px * (1 + 8 + 1) and st.rays_tlas > px
    # re-run the last frame's indirect dispatch in two halves on run 1: identical reservoirs / render
    e = runs[1].engine
    e.pass_run(F.PASS_INDIRECT, 0, 0, 1000)
    e.pass_run(F.PASS_INDIRECT, 0, 1000, 2160)
    e.pass_run(F.PASS_INDIRECT_SPATIAL_REUSE, 0, 0, 777)
    e.pass_run(F.PASS_INDIRECT_SPATIAL_REUSE, 0, 777, 2160)
    assert diff_buffers(snapshot(runs[1]), a) == {}


def test_moving_camera_within_tolerance():
    """Camera motion makes the reference's scatter-store race observable; the oracle
    resolves it by thread index, the GPU by arrival.  The image must stay within the north-star
    tolerance and almost all reservoirs must still agree bit for bit."""
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    scene = hk.load_cornell()
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    rels = []
    for n in range(1, 9):
        eye = (0.02 * n, 1.0 + 0.01 * n, 4.0 - 0.03 * n)
        cam = hk.Camera(hk.look_at_transform(eye, (0.0, 1.0, 0.0)), 128, 96)
        for p in (gpu, cpu):
            p.render(cam, s, frame_number=n)
        a, b = gpu.output(s), cpu.output(s)
        rels.append(float(np.linalg.norm(a - b) / np.linalg.norm(b)))
    assert max(rels) <= 1e-3, rels
    bad = diff_buffers(snapshot(gpu), snapshot(cpu))
    assert not any(k in bad for k in ("position", "normal", "velocity_uv", "albedo")), bad


def test_error_paths():
    eng = hk.Engine(device=0)
    with pytest.raises(hk.HikariError) as e:
        eng.pass_run(F.PASS_INDIRECT)
    assert e.value.code == F.HK_E_NOT_READY      # reference: node silently returns Ok(()) (light.rs:606-617)
    # a material that references a texture which was never uploaded is rejected when the scene is finalised
    bad = hk.Engine(device=0)
    bad.upload_noise()
    scene = hk.load_cornell()
    scene.materials[2].base_color_texture = 3
    bad.upload_scene(scene)
    bad.resize(32, 32, 1.0)
    cam = hk.cornell_camera(32, 32)
    bad.frame_begin(hk.frame_uniform(hk.HikariSettings(), 1), cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform())
    with pytest.raises(hk.HikariError) as e:
        bad.pass_run(F.PASS_PREPASS)
    assert e.value.code == F.HK_E_INVALID and "texture" in str(e.value)
    with pytest.raises(hk.HikariError) as e:
        eng.api.call("upload_noise", eng.ctx, None, 0)
    assert e.value.code == F.HK_E_INVALID
    eng.resize(64, 64, 1.0)
    with pytest.raises(hk.HikariError):
        eng.read(F.BUF_COUNT + 3)
    with pytest.raises(hk.HikariError) as e:
        ctx = C.c_void_p()
        eng.api.call("create", 99, 0, C.byref(ctx))
    assert e.value.code == F.HK_E_NO_DEVICE
    # sizes beyond the 32-bit pixel index are refused, and a refused resize leaves no half-allocated screen behind
    with pytest.raises(hk.HikariError) as e:
        eng.resize(20000, 16, 1.0)
    assert e.value.code == F.HK_E_INVALID
    assert eng.buffer_info(F.BUF_TONE_MAPPED)[:2] == (64, 64)          # the earlier resources are untouched by a rejected call
    eng.resize(48, 32, 1.5)
    assert eng.buffer_info(F.BUF_TONE_MAPPED)[:2] == (32, 22)


def test_band_renderer_single_gpu_views():
    """Zero-copy torch views of the library's device buffers (the halo-exchange transport)."""
    import torch

    from bevy_hikari_amd.distributed import BandRenderer

    case = make_case("cornell_b2")
    p = hk.HikariPlugin(device=0)
    run_case(p, case)
    r = BandRenderer(p.engine, 0, 1)
    t = r._view(F.BUF_TONE_MAPPED)
    host = p.engine.read(F.BUF_TONE_MAPPED)
    assert t.is_cuda and t.numel() == host.nbytes
    assert (t.cpu().numpy() == host.view(np.uint8).reshape(-1)).all()


def _gpu_band_worker(rank, world, port, case_name, out_dir):
    import sys

    import torch.distributed as dist

    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from rendezvous import init_gloo

    init_gloo(rank, world, port)   # (`port`: a file:// rendezvous token, tests/rendezvous.py)
    import torch

    import bevy_hikari_amd as hk
    from bevy_hikari_amd.distributed import BandRenderer
    from cases import ALL_BUFFERS, make_case, random_case

    case = random_case(int(case_name[6:])) if case_name.startswith("random") else make_case(case_name)
    s = case.settings
    e = hk.Engine(device=0)
    e.upload_noise()
    e.upload_scene(case.scene)
    w, h = case.camera.width, case.camera.height
    e.resize(w, h, s.upscale.ratio())
    r = BandRenderer(e, rank, world, transport="host")  # several ranks on ONE device: halos through host memory over gloo
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    for n in case.frames:
        r.render(hk.frame_uniform(s, n), view, pview, case.lights, s, w, h, antialias=case.antialias)
    e.wait()
    _, rh, _ = e.buffer_info(F.BUF_TONE_MAPPED)
    b0, b1 = r.band(rh)
    prev = 1 - case.frames[-1] % 2
    want = [F.BUF_TONE_MAPPED, F.BUF_DENOISE_RENDER0, F.BUF_DENOISE_RENDER0 + 1, F.BUF_DENOISE_RENDER0 + 2, F.BUF_RENDER0 + 2, F.BUF_VARIANCE0 + 2,
            F.BUF_RESERVOIR0 + prev + 6, F.BUF_RESERVOIR0 + prev + 8]
    out = {ALL_BUFFERS[b]: e.read(b)[b0:b1] for b in want if e.buffer_info(b)[1] == rh}   # (reservoirs are indexed with the scaled width inside full-size storage)
    if case.antialias:  # output rows: two per render row with SMAA Tu4x
        fsr = s.upscale.kind == F.UPSCALE_FSR1
        for b in (F.BUF_UPSCALE_OUTPUT, F.BUF_TAA_OUTPUT) + ((F.BUF_UPSCALE_SHARPENED,) if fsr else ()):
            _, bh, _ = e.buffer_info(b)
            if fsr and b != F.BUF_TAA_OUTPUT:   # FSR1: a band owns its share of the window rows (HK_STAGE_UPSCALE)
                y0, y1 = r.band(bh)
            else:
                scale = 2 if bh > rh else 1
                y0, y1 = min(bh, scale * b0), (bh if b1 == rh else min(bh, scale * b1))
            out[ALL_BUFFERS[b]] = e.read(b)[y0:y1]
            out[ALL_BUFFERS[b] + "_rows"] = np.array([y0, y1])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), b0=b0, b1=b1, **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case_name", [(2, "cornell_b2"), (3, "yard_sun"), (3, "cornell_aa_default"), (2, "random7"), (4, "random12"), (3, "random21"), (2, "cornell_aa_fsr"),
                                              (3, "random33")])
def test_gpu_bands_equal_single_gpu(tmp_path, world, case_name):
    """The band-sharded GPU path (hk_set_band + hk_frame_stage + halo exchange) on ONE GPU: every
    rank renders its band on device 0, halos travel over gloo (staged through host memory because
    RCCL refuses two ranks on one device); the union must equal the single-context frame."""
    import socket

    import torch.multiprocessing as mp

    from rendezvous import new_rendezvous

    mp.spawn(_gpu_band_worker, args=(world, new_rendezvous(), case_name, str(tmp_path)), nprocs=world, join=True)
    from cases import random_case

    case = random_case(int(case_name[6:])) if case_name.startswith("random") else make_case(case_name)
    ref = hk.HikariPlugin(device=0)
    run_case(ref, case)
    full = snapshot(ref)
    for rank in range(world):
        d = np.load(tmp_path / f"rank{rank}.npz")
        b0, b1 = int(d["b0"]), int(d["b1"])
        for key in d.files:
            if key in ("b0", "b1") or key.endswith("_rows"):
                continue
            y0, y1 = (int(v) for v in d[key + "_rows"]) if key + "_rows" in d.files else (b0, b1)
            assert (d[key].view(np.uint8) == full[key][y0:y1].view(np.uint8)).all(), f"rank {rank} [{y0},{y1}) differs in {key}"


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


@pytest.mark.parametrize("size", [(1, 1), (1, 37), (41, 1), (2, 2), (9, 8), (8, 9)])
def test_degenerate_image_sizes_vs_oracle(size):
    """Images of one pixel, one row, one column, and just over / under one 8x8 tile: every buffer against the oracle, with the
    spatial passes, the denoiser and the SMAA Tu4x + TAA tail on (their footprints all reach past such an image)."""
    from oracle_lib import oracle_plugin

    w, h = size
    s = hk.HikariSettings(indirect_bounces=2, emissive_spatial_reuse=True, upscale=hk.Upscale.SmaaTu4x(1.5), taa=hk.Taa.Jasmine)
    cam = hk.cornell_camera(w, h)
    gpu, cpu = hk.HikariPlugin(device=0), oracle_plugin()
    for p in (gpu, cpu):
        p.set_scene(hk.load_cornell())
        for n in range(1, 5):
            p.render(cam, s, frame_number=n, antialias=True)
    d = diff_buffers(snapshot(gpu), snapshot(cpu))
    assert not d, f"{w}x{h}: {d}"


def test_single_triangle_scene_vs_oracle():
    """The smallest scene the builders accept: one instance of a one-triangle mesh (a tree of a single leaf at both levels), lit by
    the sun only."""
    from bevy_hikari_amd.plugin import SceneBuilder, standard_material
    from oracle_lib import oracle_plugin

    b = SceneBuilder()
    pos = np.array([[-1.0, 0.0, -1.0], [1.0, 0.0, -1.0], [0.0, 0.0, 1.5]], dtype=np.float32)
    nrm = np.tile(np.array([[0.0, 1.0, 0.0]], dtype=np.float32), (3, 1))
    uv = np.array([[0.0, 0.0], [1.0, 0.0], [0.5, 1.0]], dtype=np.float32)
    mesh = b.add_mesh(pos, nrm, uv, np.array([0, 1, 2], dtype=np.uint32))
    mat = b.add_material(standard_material((0.8, 0.6, 0.4, 1.0), (0, 0, 0), 0.7, 0.0, 0.5))
    b.add_instance(mesh, mat, np.eye(4, dtype=np.float32))
    scene = b.finish()
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = hk.Camera(hk.look_at_transform((0.0, 3.0, 3.0), (0.0, 0.0, 0.0)), 64, 48)
    lights = hk.lights_uniform(directional=dict(color=(1.0, 1.0, 1.0), illuminance=50000.0, direction_to_light=(0.2, 0.9, 0.3)))
    gpu, cpu = hk.HikariPlugin(device=0), oracle_plugin()
    for p in (gpu, cpu):
        p.set_scene(scene)
        for n in range(1, 4):
            p.render(cam, s, lights=lights, frame_number=n)
    d = diff_buffers(snapshot(gpu), snapshot(cpu))
    assert not d, d
    assert np.isfinite(gpu.output(s)).all() and gpu.output(s).max() > 0.0


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


GBUFFER = ("position", "normal", "depth_gradient", "instance_material", "velocity_uv", "albedo")


def test_dynamic_instances_vs_oracle():
    """Moving instances (prepare_instances re-runs, instance.rs:352-437; PreviousMeshUniform feeds the
    velocity output, prepass.wgsl:50,96).  The G-buffer has no races: bit-exact.  Reprojection across a
    moving object triggers the reference's scatter-store race like camera motion does: image <= 1e-3.
    The library must rewrite the instance-level arrays only."""
    from bevy_hikari_amd.scenes import animate, synthetic_camera, synthetic_scene

    scene, sun = synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=3, sphere_rings=6, sphere_segs=8)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0, emissive_spatial_reuse=True)
    cam, lights = synthetic_camera(128, 96), hk.lights_uniform(directional=sun)
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    rels = []
    for n in range(1, 9):
        if n > 1:
            scene = animate(scene, n - 1, movers=(3, 9, 16, 19))
            for p in (gpu, cpu):
                p.update_instances(scene)
        for p in (gpu, cpu):
            p.render(cam, s, lights=lights, frame_number=n)
        a, b = gpu.output(s), cpu.output(s)
        rels.append(float(np.linalg.norm(a - b) / np.linalg.norm(b)))
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert not any(k in bad for k in GBUFFER), (n, bad)
        if n > 1:
            vel = gpu.engine.read(F.BUF_VELOCITY_UV)[..., :2]
            assert (vel != 0).any()
    assert max(rels) <= 1e-3, rels
    st = gpu.engine.stats()
    assert (st.scene_mesh_builds, st.scene_instance_builds) == (1, 8)


def test_instance_updates_in_flight_use_the_spare_slot():
    """A scene too big for the LDS copy keeps two slots of the instance-level region: eight animated frames are
    enqueued back to back - builder re-finish, upload, render, no wait in between - each update going through pinned
    staging into the slot the frames in flight do not read.  The G-buffer of the last frame (which also holds the
    previous-model velocity) must be the oracle's, bit for bit, and every update after the first must have taken the
    asynchronous route."""
    from bevy_hikari_amd.scenes import animate, synthetic_camera, synthetic_scene

    scene, sun = synthetic_scene(n_boxes=24, n_spheres=6, n_emitters=3, sphere_rings=12, sphere_segs=16)
    s = hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)
    cam, lights = synthetic_camera(160, 96), hk.lights_uniform(directional=sun)
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    movers = (2, 5, 11, 17, 23, 26, 29)
    for p in (gpu, cpu):
        p.render(cam, s, lights=lights, frame_number=1)
    cur = scene
    for n in range(2, 10):                            # GPU: no read, no wait until the end
        cur = animate(cur, n - 1, movers=movers)
        gpu.update_instances(cur)
        gpu.render(cam, s, lights=lights, frame_number=n)
    cur = animate(cur, 0, movers=movers)              # replay the same poses for the oracle (animate sets absolute poses)
    for n in range(2, 10):
        cur = animate(cur, n - 1, movers=movers)
        cpu.update_instances(cur)
        cpu.render(cam, s, lights=lights, frame_number=n)
    bad = diff_buffers(snapshot(gpu), snapshot(cpu))
    assert not any(k in bad for k in GBUFFER + ("previous_position", "previous_velocity_uv")), bad
    assert (gpu.engine.read(F.BUF_VELOCITY_UV)[..., :2] != 0).any()
    a, b = gpu.output(s), cpu.output(s)
    assert float(np.linalg.norm(a - b) / np.linalg.norm(b)) <= 1e-3
    st = gpu.engine.stats()
    assert (st.scene_mesh_builds, st.scene_instance_builds, st.scene_async_instance_uploads) == (1, 9, 8)   # (the slot has room for the previous models from the start)


def test_two_slot_scene_grows_between_frames_in_flight():
    """The same two-slot scene, with instances ADDED while frames are in flight: the update that outgrows the slots
    takes the synchronous route (device-to-device move of the mesh region behind two larger slots), the ones after it
    are asynchronous again.  Static camera, static objects apart from the additions: every buffer is bit-exact."""
    from bevy_hikari_amd.scenes import _trs, synthetic_camera, synthetic_scene

    scene, sun = synthetic_scene(n_boxes=24, n_spheres=6, n_emitters=3, sphere_rings=12, sphere_segs=16)
    s = hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)
    cam, lights = synthetic_camera(128, 80), hk.lights_uniform(directional=sun)
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    b = scene.builder
    n_frame = [0]

    def frames(k):
        for _ in range(k):
            n_frame[0] += 1
            for p in (gpu, cpu):
                p.render(cam, s, lights=lights, frame_number=n_frame[0])

    frames(2)
    for round_ in range(3):       # 40 instances per round: the first round outgrows the slots (room for +50 %), later ones may not
        for k in range(40):
            b.add_instance(0, 1 + k % 5, _trs((-4.0 + 0.2 * k, 0.3 + 0.5 * round_, 3.0), (0.1 * k, 0.2, 0.0), (0.15, 0.15, 0.15)))
        grown = b.finish()
        for p in (gpu, cpu):
            p.update_instances(grown)
        frames(2)
    bad = diff_buffers(snapshot(gpu), snapshot(cpu))
    assert bad == {}, bad
    st = gpu.engine.stats()
    assert st.scene_mesh_builds == 1 and st.scene_instance_builds == 4
    assert 1 <= st.scene_async_instance_uploads <= 2      # at least one of the three updates fitted the enlarged slots


def test_instance_growth_and_late_mesh_use():
    """Instance count grows past the instance-level slot (device-to-device move of the mesh region), then
    an instance of a mesh no earlier instance used appears (its BLAS leaf boxes must be derived).  A static
    camera and static objects: every buffer stays bit-exact."""
    from bevy_hikari_amd.scenes import _trs, synthetic_camera, synthetic_scene

    scene, sun = synthetic_scene(n_boxes=10, n_spheres=0, n_emitters=2, sphere_rings=5, sphere_segs=6)   # the sphere mesh (id 1) is unused
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam, lights = synthetic_camera(96, 64), hk.lights_uniform(directional=sun)
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    b = scene.builder

    def frame(n):
        for p in (gpu, cpu):
            p.render(cam, s, lights=lights, frame_number=n)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, (n, bad)

    frame(1)
    frame(2)
    for k in range(12):  # 12 more boxes: the instance-level arrays outgrow their slot
        b.add_instance(0, 1 + k % 5, _trs((-3.0 + 0.5 * k, 0.4, 2.5), (0.1 * k, 0.2, 0.0), (0.3, 0.4, 0.3)))
    grown = b.finish()
    assert len(grown.instances) == len(scene.instances) + 12
    for p in (gpu, cpu):
        p.update_instances(grown)
    frame(3)
    frame(4)
    assert gpu.engine.stats().scene_mesh_builds == 1
    b.add_instance(1, 2, _trs((0.5, 1.0, 0.5), (0.3, 0.1, 0.2), (0.8, 0.8, 0.8)))   # first use of the sphere mesh
    late = b.finish()
    for p in (gpu, cpu):
        p.update_instances(late)
    frame(5)
    frame(6)
    st = gpu.engine.stats()
    assert (st.scene_mesh_builds, st.scene_instance_builds) == (2, 3)


AA_BUFFERS = ("tone_mapped", "previous_tone_mapped", "previous_position", "previous_velocity_uv", "upscale_output", "taa_output",
              "previous_taa_output")
AA_CASES = {
    "smaa_ratio2_taa": dict(size=(128, 96), settings=dict(indirect_bounces=2)),                                     # the reference's defaults
    "smaa_ratio1_taa": dict(size=(72, 56), settings=dict(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)),      # 2x the window
    "smaa_odd_no_taa": dict(size=(101, 75), settings=dict(indirect_bounces=1, taa=hk.Taa.NONE)),                    # odd sizes: quads hang over the edge
    "fsr_ratio15_taa": dict(size=(90, 66), settings=dict(indirect_bounces=1, upscale=hk.Upscale.Fsr1(1.5, 0.2))),   # TAA at the scaled size, EASU + RCAS
    "fsr_ratio2_no_taa": dict(size=(101, 75), settings=dict(indirect_bounces=1, upscale=hk.Upscale.Fsr1(2.0, 0.0), taa=hk.Taa.NONE)),
    "fsr_ratio1_taa": dict(size=(64, 40), settings=dict(indirect_bounces=0, upscale=hk.Upscale.Fsr1(1.0, 1.5))),      # EASU at 1:1
}


@pytest.mark.parametrize("name", sorted(AA_CASES))
def test_antialias_bit_exact_vs_oracle(name):
    """SMAA Tu4x / TAA / FSR1 after the light path, static scene: every buffer bit-exact, frame by frame,
    through hk_frame_render(HK_FRAME_ANTIALIAS) on the GPU and dispatch by dispatch on the oracle."""
    case = AA_CASES[name]
    s = hk.HikariSettings(**case["settings"])
    cam = hk.cornell_camera(*case["size"])
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    scene = hk.load_cornell()
    for p in (gpu, cpu):
        p.set_scene(scene)
    for n in range(1, 7):
        gpu.render(cam, s, frame_number=n, antialias=True)
        cpu.render(cam, s, frame_number=n, antialias=True, by_nodes=True)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"{name} frame {n}: {bad}"
    img = gpu.final_image(s)
    assert np.isfinite(img).all() and img[..., :3].mean() > 0.05   # (the differential blend of SMAA may overshoot 1.0)


def test_antialias_kernels_under_motion_on_identical_inputs():
    """Camera and object motion exercise the reprojection, miss and clipping branches.  The light passes
    race under motion (reference behaviour), so the inputs of the AA dispatches are taken from the oracle
    and written into the GPU context: on identical inputs the three kernels must be bit-exact."""
    from bevy_hikari_amd.scenes import animate, synthetic_camera, synthetic_scene

    scene, sun = synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=3, sphere_rings=6, sphere_segs=8)
    lights = hk.lights_uniform(directional=sun)
    inputs = (F.BUF_POSITION, F.BUF_VELOCITY_UV, F.BUF_INSTANCE_MATERIAL, F.BUF_PREVIOUS_POSITION, F.BUF_PREVIOUS_VELOCITY_UV,
              F.BUF_TONE_MAPPED, F.BUF_PREVIOUS_TONE_MAPPED, F.BUF_PREVIOUS_TAA_OUTPUT)
    for s in (hk.HikariSettings(indirect_bounces=1), hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.Fsr1(1.0, 0.2))):
        gpu, cpu = hk.HikariPlugin(device=0), oracle()
        cur = scene
        for p in (gpu, cpu):
            p.set_scene(cur)
        clipped = 0
        for n in range(1, 7):
            cam = hk.Camera(hk.look_at_transform((6.4 + 0.15 * n, 4.4, 8.0 - 0.1 * n), (0.0, 0.6, 0.0)), 112, 80)
            if n > 1:
                cur = animate(cur, n - 1, movers=(3, 9, 16, 19))
                for p in (gpu, cpu):
                    p.update_instances(cur)
            cpu.render(cam, s, lights=lights, frame_number=n, antialias=True)
            # same frame on the GPU up to tone mapping (keeps sizes, uniforms and plane parity in step) ...
            gpu.render(cam, s, lights=lights, frame_number=n)
            for b in inputs:      # ... then the oracle's inputs, and only the AA dispatches
                gpu.engine.write(b, cpu.engine.read(b))
            gpu.post_process.run_antialias(s)
            for b in (F.BUF_UPSCALE_OUTPUT, F.BUF_TAA_OUTPUT, F.BUF_UPSCALE_SHARPENED):
                a, o = gpu.engine.read(b), cpu.engine.read(b)
                assert (a == o).all(), (n, b, int((a != o).any(axis=2).sum()))
            vel = cpu.engine.read(F.BUF_VELOCITY_UV)[..., :2]
            clipped += int((np.abs(vel).max(axis=2) > 1e-4).sum())
        assert clipped > 1000      # the motion branches really ran


def test_certified_division_route_changes_no_bit():
    """(k + 0.5) / size goes through a multiply + exact-residual correction that hk_resize certifies against
    the IEEE quotient for every coordinate; HK_CTX_PLAIN_DIVISION forces the IEEE sequence.  Same frames."""
    case = make_case("cornell_upscale2")
    snaps = []
    for flags in (0, F.CTX_PLAIN_DIVISION):
        p = hk.HikariPlugin(device=0, flags=flags)
        run_case(p, case)
        snaps.append(snapshot(p))
    assert diff_buffers(snaps[0], snaps[1]) == {}
    odd = hk.HikariPlugin(device=0)          # sizes that are not powers of two or multiples of eight
    odd.set_scene(case.scene)
    plain = hk.HikariPlugin(device=0, flags=F.CTX_PLAIN_DIVISION)
    plain.set_scene(case.scene)
    s = hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.Fsr1(1.3, 0.2))
    for n in (1, 2, 3):
        for p in (odd, plain):
            p.render(hk.cornell_camera(117, 83), s, frame_number=n)
    assert diff_buffers(snapshot(odd), snapshot(plain)) == {}


@pytest.mark.parametrize("name", ["cornell_upscale2", "cornell_aa_default"])
def test_host_supplied_gbuffer_gives_the_same_frames(name):
    """HK_FRAME_EXTERNAL_GBUFFER on the GPU: G-buffer planes written by the host after hk_frame_begin, the derived
    planes (depth, packed denoiser taps) rebuilt by k_derive_planes - every buffer as with the internal prepass."""
    from cases import run_case_with_host_gbuffer

    case = make_case(name)
    a, b = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0)
    run_case_with_host_gbuffer(a, b, case)
    assert diff_buffers(snapshot(a), snapshot(b)) == {}


def test_baseline_config_1_exact():
    """BASELINE config 1 as SURVEY 8d states it: Cornell 256x256 traced (ratio 1.0), 1 bounce, defaults otherwise,
    frames 1..8 from zeroed reservoirs - every buffer of every frame bit for bit."""
    s = hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)
    scene, cam = hk.load_cornell(), hk.cornell_camera(256, 256)
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    for n in range(1, 9):
        for p in (gpu, cpu):
            p.render(cam, s, frame_number=n)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, (n, bad)


def test_sixty_four_frame_sequence_stays_bit_exact():
    """The bench sequence length (frames 1..64, BASELINE config 2 at a quarter of its size): validation frames of
    both intervals, reservoir lifetimes past their cap, M-capping - no drift between the two sides at frame 64."""
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    scene, cam = hk.load_cornell(), hk.cornell_camera(480, 272)
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    for n in range(1, 65):
        for p in (gpu, cpu):
            p.render(cam, s, frame_number=n)
        if n in (16, 32, 48, 64):
            bad = diff_buffers(snapshot(gpu), snapshot(cpu))
            assert bad == {}, (n, bad)


def test_second_stream_overlap_changes_no_bit_and_joins_on_reads():
    """The frame path runs the two direct-light dispatches on a second stream (joined before demodulation);
    HK_CTX_SINGLE_STREAM keeps one stream.  Same frames; and a host that stops after the temporal stage and
    reads the sun / emissive outputs must see them complete (hk_read_buffer joins)."""
    case = make_case("yard_sun")      # both direct channels carry light, emissive spatial reuse on
    snaps = []
    for flags in (0, F.CTX_SINGLE_STREAM):
        p = hk.HikariPlugin(device=0, flags=flags)
        run_case(p, case)
        snaps.append(snapshot(p))
    assert diff_buffers(snaps[0], snaps[1]) == {}
    s = case.settings
    outs = []
    for flags in (0, F.CTX_SINGLE_STREAM):
        e = hk.Engine(device=0, flags=flags)
        e.upload_noise()
        e.upload_scene(case.scene)
        e.resize(case.camera.width, case.camera.height, s.upscale.ratio())
        for n in (1, 2, 3):
            e.frame_begin(hk.frame_uniform(s, n), case.camera.view_uniform(), case.camera.previous_view_uniform(), case.lights)
            e.frame_stage(F.STAGE_TEMPORAL, s.to_c())
            got = [e.read(b) for b in (F.BUF_RENDER0, F.BUF_RENDER0 + 1, F.BUF_RENDER0 + 2, F.BUF_VARIANCE0 + 1)]   # straight after the fork
            e.frame_stage(F.STAGE_SPATIAL, s.to_c())
            e.frame_stage(F.STAGE_POST_PROCESS, s.to_c())
        outs.append(got)
    for a, b in zip(*outs):
        assert (a.view(np.uint8) == b.view(np.uint8)).all()
    assert outs[0][1].view(np.float16).astype(np.float32)[..., :3].max() > 0


def test_frame_pipelining_changes_no_bit_in_any_frame_order():
    """Round 3: the a-trous levels of frame n run on a third stream beside frame n + 1's primary rays and light passes, the G-buffer
    planes both touch double-buffered by frame parity.  (a) The pipelined context equals the single-stream one in every buffer after a
    long back-to-back sequence; (b) frames of the SAME parity in a row (1, 3, 5 ...: the planes do not flip) and an arbitrary order of
    frame numbers take the serial order and still equal the oracle bit for bit; (c) switching a context to bands and back in the
    middle of a sequence (bands never pipeline) changes nothing."""
    case = make_case("cornell_b2")
    s, cam = case.settings, case.camera
    view, pview = cam.view_uniform(), cam.previous_view_uniform()
    snaps = []
    for flags in (0, F.CTX_SINGLE_STREAM):   # (a)
        p = hk.HikariPlugin(device=0, flags=flags)
        p.set_scene(case.scene)
        for n in range(1, 41):
            p.render(cam, s, lights=case.lights, frame_number=n)
        snaps.append(snapshot(p))
    assert diff_buffers(snaps[0], snaps[1]) == {}
    for numbers in ((1, 3, 5, 7, 9), (2, 2, 7, 4, 4, 11, 12)):   # (b)
        gpu, cpu = hk.HikariPlugin(device=0), oracle()
        for p in (gpu, cpu):
            p.set_scene(case.scene)
        for n in numbers:
            for p in (gpu, cpu):
                p.render(cam, s, lights=case.lights, frame_number=n)
        assert diff_buffers(snapshot(gpu), snapshot(cpu)) == {}, numbers
    e, ref = hk.Engine(device=0), hk.Engine(device=0, flags=F.CTX_SINGLE_STREAM)   # (c)
    for x in (e, ref):
        x.upload_noise(); x.upload_scene(case.scene); x.resize(cam.width, cam.height, s.upscale.ratio())
    for n in range(1, 13):
        f = hk.frame_uniform(s, n)
        ref.frame_render(f, view, pview, case.lights, s.to_c())
        if n in (5, 6, 9):   # two bands, both rendered by this context: together they are the whole frame
            e.frame_begin(f, view, pview, case.lights)
            for stage in (F.STAGE_TEMPORAL, F.STAGE_SPATIAL, F.STAGE_POST_PROCESS):
                for band in (0, 1):
                    e.set_band(band, 2)
                    e.frame_stage(stage, s.to_c())
            e.set_band(0, 1)
        else:
            e.frame_render(f, view, pview, case.lights, s.to_c())
    for b in (F.BUF_TONE_MAPPED, F.BUF_DENOISE_RENDER0 + 2, F.BUF_RENDER0 + 2, F.BUF_ALBEDO, F.BUF_DEPTH_GRADIENT, F.BUF_RESERVOIR0 + 6, F.BUF_RESERVOIR0 + 7):
        assert (e.read(b).view(np.uint8) == ref.read(b).view(np.uint8)).all(), b


def test_primary_ray_pipelining_changes_no_bit():
    """Round 5: the primary rays of frame n + 1 run on a fourth stream beside frame n's light passes (every plane the prepass writes
    is double-buffered by frame parity).  (a) A pipelined context equals a single-stream one in every buffer after a long back-to-back
    sequence - on Cornell (scene in LDS) and on a scene beyond LDS in the product default (the queue-based indirect pass, whose
    trace stages' tails the primary rays fill) - and the pipelined path really ran; (b) what breaks the chain takes the serial order
    and changes nothing: the anti-aliasing tail (reads the previous frame's planes), an instance update between frames, a host that
    dispatches passes itself, frames of one parity in a row."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large
    from cases import product_default_traversal

    case = make_case("cornell_b2")
    big, sun = synthetic_large(0x5EED0007, 8, 24, 48, 60, 8, 2, 6.0)
    runs = [(case.scene, case.camera, case.settings, case.lights, 0, 24),
            (big, synthetic_camera(320, 180, extent=6.0), hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0), hk.lights_uniform(directional=sun), None, 12)]
    os.environ["HK_PREPASS_PIPELINE"] = "all"   # (read by hk_create; off by default: measured slower)
    try:
        _pipelining_body(make_case, runs, oracle)
    finally:
        del os.environ["HK_PREPASS_PIPELINE"]


def _pipelining_body(make_case, runs, oracle):
    from cases import product_default_traversal

    for scene, cam, s, lights, flags, frames in runs:   # (a)
        snaps = []
        for single in (False, True):
            if flags is None:
                with product_default_traversal():
                    p = hk.HikariPlugin(device=0, flags=F.CTX_SINGLE_STREAM if single else 0)
            else:
                p = hk.HikariPlugin(device=0, flags=F.CTX_SINGLE_STREAM if single else 0)
            p.set_scene(scene)
            for n in range(1, frames + 1):
                p.render(cam, s, lights=lights, frame_number=n)
            snaps.append(snapshot(p))
            assert p.engine.prepasses_pipelined() == (0 if single else frames - 1)
        assert diff_buffers(snaps[0], snaps[1]) == {}
    # (b) Cornell with the anti-aliasing tail on some frames, by_nodes on others, repeated parities: against the oracle, frame by frame
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    aa_case = make_case("cornell_aa_default")
    for p in (gpu, cpu):
        p.set_scene(aa_case.scene)
    plan = [(1, False, False), (2, False, False), (3, True, False), (4, False, False), (5, False, True), (6, False, False), (7, False, False), (9, False, False), (10, True, False),
            (11, False, False), (12, False, False)]
    for n, aa, by_nodes in plan:
        for p in (gpu, cpu):
            p.render(aa_case.camera, aa_case.settings, lights=aa_case.lights, frame_number=n, antialias=aa, by_nodes=by_nodes)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, (n, bad)
    assert 0 < gpu.engine.prepasses_pipelined() < len(plan) - 1


# 5001: a 25-pixel-wide render image whose right-most 8x8 tiles have one valid column, all background - the store elision once took
# the tile's record id from lanes beyond the image edge (found by the round-2 sweep of 13 200 seeds)
@pytest.mark.parametrize("seed", list(range(40)) + [5001])
def test_random_settings_vs_oracle(seed):
    """Seeded sweep over the HikariSettings space (cases.random_case): bounce counts, reuse switches, validation
    intervals, reuse caps, lifetimes, denoise, upscale kind / ratio, TAA, odd image sizes, with and without the
    anti-aliasing tail - three frames each, every buffer bit for bit."""
    from cases import random_case

    case = random_case(seed)
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    for p in (gpu, cpu):
        p.set_scene(case.scene)
    for n in case.frames:
        for p in (gpu, cpu):
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, (seed, n, case.settings, (case.camera.width, case.camera.height), case.antialias, bad)


def test_one_context_pair_through_many_scenes_sizes_and_settings():
    """The same sweep with ONE pair of contexts for 120 further seeds: every case changes the scene, the window size
    (hk_resize: new buffers, zeroed reservoirs, plane parity reset), the upscale kind and the settings under a live
    context.  The camera history is dropped at each cut - a cut WITH history is camera motion, i.e. the reference's
    scatter-store race - and every buffer of every frame stays bit-exact.
    (tests/tools/fuzz_sweep.py runs the same loop over any seed range; 13 400 seeds / 40 200 frames were clean.)"""
    from cases import random_case

    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    for seed in range(40, 160):
        case = random_case(seed)
        for p in (gpu, cpu):
            p.set_scene(case.scene)
            p._previous_camera = None
        for n in case.frames:
            for p in (gpu, cpu):
                p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
            bad = diff_buffers(snapshot(gpu), snapshot(cpu))
            assert bad == {}, (seed, n, bad)


@pytest.mark.parametrize("seed", range(24))
def test_motion_is_bit_exact_once_the_race_is_resolved_like_the_oracle(seed):
    """Moving camera + moving instances (cases.motion_case), random settings and AA tail.  The reference lets the
    reprojected stores to previous_spatial race; HK_CTX_DETERMINISTIC_SCATTER parks them and lets the highest thread
    index win, which is the oracle's rule - and then EVERY buffer of EVERY frame is bit-exact under motion too: the
    race is the only thing that separates the two under motion.  (tests/tools/fuzz_sweep.py --motion --deterministic:
    2 900 sequences clean.)"""
    from cases import motion_case, run_motion_case

    case = motion_case(seed)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_DETERMINISTIC_SCATTER), oracle()

    def check(n):
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, (seed, n, bad)

    run_motion_case((gpu, cpu), case, check)
    assert (gpu.engine.read(F.BUF_VELOCITY_UV)[..., :2] != 0).any()


def test_racing_default_stays_close_under_motion():
    """Without the flag the stores race as in the reference: the G-buffer is still exact, and the image is held to the north
    star's 1e-3 relative L2 - as a FRACTION of sequences, because the oracle's pick of each race (highest thread index) is as
    arbitrary as the GPU's (arrival order) and an unlucky pick moves a 48..160-pixel image by more than that.  Measured over 300
    sequences in round 1: 7 % above 1e-3, none above 1.5e-2; with HK_CTX_DETERMINISTIC_SCATTER (test above) every byte agrees.
    The measured fraction is printed and written to gpurun_out/racing_report.json when that directory exists."""
    from cases import motion_case, run_motion_case

    rels = []
    n_seeds = 40
    for seed in range(n_seeds):
        case = motion_case(seed)
        gpu, cpu = hk.HikariPlugin(device=0), oracle()

        def check(n):
            bad = diff_buffers(snapshot(gpu), snapshot(cpu))
            assert not any(k in bad for k in GBUFFER + ("previous_position", "previous_velocity_uv")), (seed, n, bad)

        run_motion_case((gpu, cpu), case, check)
        a, b = gpu.output(case["settings"]), cpu.output(case["settings"])
        rels.append(float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-20)))
    above = [r for r in rels if r > 1e-3]
    report = {"sequences": n_seeds, "fraction_above_1e-3": len(above) / n_seeds, "median": float(np.median(rels)), "max": max(rels),
              "above": sorted(above)}
    print("racing default vs oracle:", report)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json

        with open(os.path.join(out_dir, "racing_report.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert np.median(rels) <= 1e-3 and len(above) <= 0.2 * n_seeds and max(rels) <= 5e-2, report


def test_uniform_tile_store_elision_changes_no_bit():
    """Uniform-tile store elision (hk_kernels.hpp TileMeta): waves whose 64 pixels are background skip reservoir stores that
    would rewrite the record the tile already holds.  Every buffer must stay bit-identical to the oracle through the situations
    that invalidate a tile record: background turning into geometry and back (camera pans across the box), scatter stores into
    background tiles under motion, a host write into a reservoir buffer, partial-row dispatches, a resize."""
    s = hk.HikariSettings(indirect_bounces=2, emissive_spatial_reuse=True, upscale=hk.Upscale.SMAA_TU_1_0)
    scene = hk.load_cornell()
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    w, h = 160, 96
    # static frames first (records settle), then the camera jumps sideways so that tiles change between sky and box, then back
    eyes = [(0.0, 1.0, 4.0)] * 4 + [(1.6, 1.0, 4.0)] * 3 + [(0.0, 1.0, 4.0)] * 3 + [(-1.2, 1.4, 5.0)] * 2
    n = 0
    prev_cam = None
    for eye in eyes:
        n += 1
        cam = hk.Camera(hk.look_at_transform(eye, (eye[0], 1.0, 0.0)), w, h)
        static = prev_cam is not None and eye == prev_eye
        for p in (gpu, cpu):
            p.render(cam, s, frame_number=n)
        prev_cam, prev_eye = cam, eye
        if static or n == 1:  # (a jump frame reprojects: the reference's scatter race is visible there, covered by the motion tests;
            # the previous_* planes of the frame after a jump ARE that jump frame)
            bad = {k: v for k, v in diff_buffers(snapshot(gpu), snapshot(cpu)).items() if not k.startswith("previous_")}
            assert bad == {}, (n, bad)
    # a host write into a reservoir buffer (what the fixture replays do) must drop the tile records of that buffer
    for p in (gpu, cpu):
        e = p.engine
        r = e.read(F.BUF_RESERVOIR0 + 8)
        r[:8, :16] = 0x3C003C00
        e.write(F.BUF_RESERVOIR0 + 8, r)
        e.write(F.BUF_RESERVOIR0 + 9, r)
    for it in range(3):
        n += 1
        for p in (gpu, cpu):
            p.render(prev_cam, s, frame_number=n)
        bad = {name: v for name, v in diff_buffers(snapshot(gpu), snapshot(cpu)).items() if not name.startswith("previous_") or it > 0}
        assert bad == {}, (n, bad)
    # partial-row dispatches that do not end on a tile row, then whole-frame dispatches again
    e = gpu.engine
    before = snapshot(gpu)
    e.pass_run(F.PASS_INDIRECT, 0, 0, 37)
    e.pass_run(F.PASS_INDIRECT, 0, 37, h)
    e.pass_run(F.PASS_INDIRECT_SPATIAL_REUSE, 0, 16, 61)
    e.pass_run(F.PASS_INDIRECT_SPATIAL_REUSE, 0, 0, 16)
    e.pass_run(F.PASS_INDIRECT_SPATIAL_REUSE, 0, 61, h)
    assert diff_buffers(snapshot(gpu), before) == {}
    for k in range(3):
        n += 1
        for p in (gpu, cpu):
            p.render(prev_cam, s, frame_number=n)
    bad = diff_buffers(snapshot(gpu), snapshot(cpu))
    assert bad == {}, bad
