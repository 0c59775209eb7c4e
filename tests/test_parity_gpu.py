"""Parity of the HIP path (through the C ABI) with the CPU oracle and the committed golden
fixtures.  Bar: BIT-EXACT on every screen-space buffer for static scenes (the numeric contract
makes f32 arithmetic reproducible); <= 1e-3 relative L2 (BASELINE.json north_star)
where the reference itself races (moving camera).

This file: the named and random cases against the oracle and the golden fixtures, the by-nodes path, error paths, bands on one GPU,
degenerate sizes, host-supplied G-buffers, long sequences, camera / instance motion.  Its siblings: test_parity_large_scenes_gpu.py
(BASELINE configs 3-5 at full size, the threaded and wide walks), test_parity_dynamic_scenes_gpu.py (instance / mesh edits between
frames), test_parity_antialias_gpu.py (TAA / SMAA / FSR tail), test_parity_schedules_gpu.py (streams, pipelining, store elision:
every schedule the same bytes), test_default_mode_sequence_gpu.py (the timed kernels over 32 frames)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import ALL_BUFFERS, CASE_NAMES, GBUFFER, assert_rendered_within, diff_buffers, make_case, oracle, product_default_plugin, report, run_case, snapshot
from conftest import ROOT

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")


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


def test_the_default_resolves_the_scatter_race_and_the_racing_mode_stays_close():
    """Round 6: a single context resolves the reference's write-write race on previous_spatial BY DEFAULT, in the light form
    (hikari_hip.h HK_CTX_RACING_SCATTER; hk_kernels.hpp LightTargets::det_lite), for the channels whose previous_spatial buffer has a
    reader.  So without any flag: the G-buffer AND every rendered plane of every frame equal the oracle's under camera + instance motion
    (what can still differ are reservoir buffers nothing reads: the sun / emissive channels' when emissive_spatial_reuse is off).
    With HK_CTX_RACING_SCATTER the stores race as in the reference: the G-buffer is still exact, and the image is held to the north
    star's 1e-3 relative L2 - as a FRACTION of sequences, because the oracle's pick of each race (highest thread index) is as
    arbitrary as the GPU's (arrival order).  Measured over 300 sequences in round 1: 7 % above 1e-3, none above 1.5e-2.  The measured
    fraction is printed and written to gpurun_out/racing_report.json when that directory exists."""
    from cases import motion_case, run_motion_case

    rels = []
    n_seeds = 40
    for seed in range(n_seeds):
        case = motion_case(seed)
        gpu, racing, cpu = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0, flags=F.CTX_RACING_SCATTER), oracle()

        def check(n):
            want = snapshot(cpu)
            bad = diff_buffers(snapshot(gpu), want)
            assert not [k for k in bad if not k.startswith("reservoir")], (seed, n, bad)     # the default: everything rendered, bit for bit
            bad = diff_buffers(snapshot(racing), want)
            assert not any(k in bad for k in GBUFFER + ("previous_position", "previous_velocity_uv")), (seed, n, bad)

        run_motion_case((gpu, racing, cpu), case, check)
        a, b = racing.output(case["settings"]), cpu.output(case["settings"])
        rels.append(float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-20)))
    above = [r for r in rels if r > 1e-3]
    report = {"mode": "HK_CTX_RACING_SCATTER", "sequences": n_seeds, "fraction_above_1e-3": len(above) / n_seeds, "median": float(np.median(rels)), "max": max(rels),
              "above": sorted(above)}
    print("racing mode vs oracle:", report)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json

        with open(os.path.join(out_dir, "racing_report.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert np.median(rels) <= 1e-3 and len(above) <= 0.2 * n_seeds and max(rels) <= 5e-2, report
