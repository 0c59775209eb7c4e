"""The queue-based schedule of indirect_lit_ambient (HK_CTX_WAVEFRONT, kernels_wavefront.hip): set-up / persistent trace with
lane refill / shade per bounce / final.  Bar: the same bytes in every buffer as the fused kernel, and as the oracle."""
import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import diff_buffers, make_case, motion_case, product_default_traversal, random_case, run_motion_case, snapshot

pytestmark = pytest.mark.gpu


def oracle():
    from oracle_lib import oracle_plugin

    return oracle_plugin()


@pytest.mark.parametrize("name", ["cornell_b2", "cornell_b8", "cornell_upscale2", "yard_sun", "yard_textured", "yard_no_emitters", "flight_helmet", "yard_ortho",
                                  "tiny_3x5", "background_only", "cornell_notemporal", "cornell_aa_default", "cornell_b1"])
def test_wavefront_bit_exact_vs_oracle_every_frame(name):
    """Named cases (LDS-resident and global-memory scenes, textures, ortho camera, 1 x 8 bounces; cornell_b1 / aa_default have
    fewer than two bounces and must quietly take the fused kernel)."""
    case = make_case(name)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT), oracle()
    gpu.set_scene(case.scene)
    cpu.set_scene(case.scene)
    for n in case.frames:
        for p in (gpu, cpu):
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"{name} frame {n}: {bad}"


@pytest.mark.parametrize("seed", range(24))
def test_wavefront_random_settings_vs_oracle(seed):
    case = random_case(seed)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT), oracle()
    gpu.set_scene(case.scene)
    cpu.set_scene(case.scene)
    for n in case.frames:
        for p in (gpu, cpu):
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"random{seed} frame {n}: {bad}"


@pytest.mark.parametrize("seed", range(8))
def test_wavefront_motion_with_resolved_race_vs_oracle(seed):
    """Moving camera + instances: the parked scatter stores (HK_CTX_DETERMINISTIC_SCATTER) are issued by k_wf_final here."""
    case = motion_case(seed)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_DETERMINISTIC_SCATTER | F.CTX_WAVEFRONT), oracle()

    def check(n):
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"motion{seed} frame {n}: {bad}"

    run_motion_case((gpu, cpu), case, check)


def _large(size=(320, 180), bounces=3):
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large()
    s = hk.HikariSettings(indirect_bounces=bounces, upscale=hk.Upscale.SMAA_TU_1_0)
    return scene, synthetic_camera(*size, extent=9.0), s, hk.lights_uniform(directional=sun)


def test_wavefront_sponza_class_vs_oracle():
    """Config 3 stand-in (256 k triangles, global-memory traversal, 3 bounces), reference traversal order: every buffer vs the oracle."""
    scene, cam, s, lights = _large()
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    for n in (1, 2, 3):
        for p in (gpu, cpu):
            p.render(cam, s, lights=lights, frame_number=n)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"frame {n}: {bad}"


def test_default_schedule_is_wavefront_for_large_scenes_and_equals_fused():
    """Product defaults (direction-threaded BVHs, wavefront for scenes beyond the LDS copy) against the same context forced to the
    fused kernel: identical bytes, at 1280x720 where a wave's 64 rays no longer come from one tile."""
    scene, cam, s, lights = _large(size=(1280, 720))
    with product_default_traversal():
        a, b = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0, flags=F.CTX_FUSED_INDIRECT)
    for p in (a, b):
        p.set_scene(scene)
        for n in (1, 2, 3):
            p.render(cam, s, lights=lights, frame_number=n)
    assert diff_buffers(snapshot(a), snapshot(b)) == {}
    out = a.output(s)
    assert np.isfinite(out).all() and out[..., :3].max() > 0.05


def test_wavefront_config5_4k_8_bounces_equals_fused():
    """Cornell 3840x2160, 8 bounces, both spatial passes, no denoise: 2.9 M paths through 8 shade stages; row-range independence."""
    s = hk.HikariSettings(indirect_bounces=8, emissive_spatial_reuse=True, denoise=False, upscale=hk.Upscale.SMAA_TU_1_0)
    scene, cam = hk.load_cornell(), hk.cornell_camera(3840, 2160)
    a, b = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT), hk.HikariPlugin(device=0)
    for p in (a, b):
        p.set_scene(scene)
        for n in (1, 2):
            p.render(cam, s, frame_number=n)
    ref = snapshot(b)
    assert diff_buffers(snapshot(a), ref) == {}
    e = a.engine  # the last frame's indirect dispatch again, in two row ranges (a band of a sharded frame)
    e.pass_run(F.PASS_INDIRECT, 0, 0, 1000)
    e.pass_run(F.PASS_INDIRECT, 0, 1000, 2160)
    e.pass_run(F.PASS_INDIRECT_SPATIAL_REUSE, 0, 0, 2160)
    assert diff_buffers(snapshot(a), ref) == {}


def test_wavefront_survives_resize_and_scene_change():
    """The scratch follows hk_resize; one context through several sizes and scenes stays equal to the fused one.  (Frame numbers
    restart with every scene, so the first frames reproject into another scene's reservoirs and scatter-store: the race is resolved
    the oracle's way on both sides to make the two contexts comparable.)"""
    a = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT | F.CTX_DETERMINISTIC_SCATTER)
    b = hk.HikariPlugin(device=0, flags=F.CTX_FUSED_INDIRECT | F.CTX_DETERMINISTIC_SCATTER)
    for name in ("cornell_b2", "yard_sun", "cornell_upscale2"):
        case = make_case(name)
        for p in (a, b):
            p.set_scene(case.scene)
            for n in case.frames:
                p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        assert diff_buffers(snapshot(a), snapshot(b)) == {}, name


@pytest.mark.parametrize("world,case_name", [(2, "cornell_b2"), (3, "yard_sun")])
def test_wavefront_bands_equal_single_context(tmp_path, world, case_name, monkeypatch):
    """Band-sharded frames (one rank per band, all on device 0, halos over gloo) with the wavefront schedule forced in every rank:
    each band's dispatch runs set-up / trace / shade / final over its own rows; the union equals the fused single-context frame."""
    import socket

    import torch.multiprocessing as mp

    from cases import run_case
    from test_parity_gpu import _gpu_band_worker

    monkeypatch.setenv("HIKARI_HIP_DEFAULT_CTX_FLAGS", str(F.CTX_EXACT_TRAVERSAL | F.CTX_WAVEFRONT))  # inherited by the spawned ranks
    from rendezvous import new_rendezvous

    mp.spawn(_gpu_band_worker, args=(world, new_rendezvous(), case_name, str(tmp_path)), nprocs=world, join=True)
    case = make_case(case_name)
    ref = hk.HikariPlugin(device=0, flags=F.CTX_FUSED_INDIRECT)
    run_case(ref, case)
    full = snapshot(ref)
    for rank in range(world):
        d = np.load(tmp_path / f"rank{rank}.npz")
        b0, b1 = int(d["b0"]), int(d["b1"])
        for key in d.files:
            if key in ("b0", "b1") or key.endswith("_rows"):
                continue
            assert (d[key].view(np.uint8) == full[key][b0:b1].view(np.uint8)).all(), f"rank {rank} [{b0},{b1}) differs in {key}"


def _read_wf_timeline(engine):
    import ctypes as C

    raw = np.zeros(64 * 32, dtype=np.uint64)
    engine.api.call("debug_read_wf_timeline", engine.ctx, raw.ctypes.data_as(C.POINTER(C.c_uint64)), raw.size)
    return raw.reshape(64, 32)


def test_counting_twin_of_the_wide_trace_kernel_changes_no_byte_and_counts_what_ran():
    """Round 5: HK_CTX_COUNT_WALKS runs the COUNTING twin of k_wf_trace_wide - the schedule and the walks bench.py
    times, unlike HK_CTX_COUNT_RAYS (fused replay).  Its frames equal the product's byte for byte; what it counts is consistent with the
    frame: per stage as many rays as the stage's queue held (paths alive + shadow rays), fewer rays from bounce to bounce, at least one
    record per ray, closest hits <= closest-hit rays; and the stamps are ordered (first wave in <= queue dry <= last wave out).
    HK_TIMING_TRACE_STAGES: every trace launch between its own pair of events - bounces + 1 launches per pass."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0007, 8, 24, 48, 60, 8, 2, 6.0)
    s = hk.HikariSettings(indirect_bounces=3, upscale=hk.Upscale.SMAA_TU_1_0)
    cam, lights = synthetic_camera(320, 180, extent=6.0), hk.lights_uniform(directional=sun)
    with product_default_traversal():
        plain, twin = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0, flags=F.CTX_COUNT_WALKS)
    for p in (plain, twin):
        p.set_scene(scene)
    twin.engine.set_timing_mask(1 << F.TIMING_TRACE_STAGES)
    frames = 4
    for n in range(1, frames + 1):
        for p in (plain, twin):
            p.render(cam, s, lights=lights, frame_number=n)
    assert diff_buffers(snapshot(twin), snapshot(plain)) == {}
    assert twin.engine.indirect_schedule() == "wavefront" and twin.engine.wide_walk()
    twin.engine.wait()
    st = twin.engine.stats()
    assert st.pass_launches[F.TIMING_TRACE_STAGES] == frames * (s.indirect_bounces + 1) and st.pass_ms_total[F.TIMING_TRACE_STAGES] > 0.0
    tl = _read_wf_timeline(twin.engine)     # the last frame's pass
    inv = np.uint64(0xFFFFFFFFFFFFFFFF)
    rays_before = None
    for stage in range(s.indirect_bounces + 1):
        r = tl[stage]
        records, top, tris, entries, rays, any_hit, hits, pieces = (int(v) for v in r[8:16])
        assert int(r[4]) > 0 and rays > 0, stage
        t0, tdry, tend = int(inv - r[0]), int(inv - r[1]), int(r[2])
        assert t0 <= tdry <= tend, (stage, t0, tdry, tend)
        assert records >= rays and top <= records and hits <= rays - any_hit and any_hit <= rays and entries <= records * 4
        if stage == 0:
            assert any_hit == 0 and rays <= 320 * 180          # bounce 0: one closest-hit ray per geometry pixel, no shadow rays yet
        else:
            assert rays - any_hit <= rays_before               # paths only ever end
        rays_before = rays - any_hit
    assert (tl[s.indirect_bounces + 1:63] == 0).all()          # (row 63 carries the clock rate)


def test_gather_roofs_of_128_byte_records():
    """hk_measure_gather with 128-B steps (a record of the wide walk) and the cooperative variant (mode 129): positive, finite, and the
    64-B rate above the 128-B rate for an L2-resident table (the roof is bytes: DESIGN 8.1b)."""
    e = hk.Engine(device=0)
    r64 = e.measure_gather(1 << 20, 64, 5, 128)[0] / 4 * 64
    r128 = e.measure_gather(1 << 20, 128, 5, 128)[0] / 8 * 64
    coop = e.measure_gather(1 << 20, 129, 5, 128)[0] / 8 * 64
    assert 1.0 < r128 < r64 < 2000.0 and 1.0 < coop < 2000.0, (r64, r128, coop)


def test_primary_rays_through_the_queue_change_no_byte():
    """HK_PREPASS_QUEUE=1 (round 5 experiment, measured slower and off: profiles/r05_prepass_queue_ab.txt): the primary rays of a scene beyond
    LDS walked by the trace kernel (k_primary_emit + k_wf_trace_wide + k_prepass_finish) - every buffer of every frame equal to the fused prepass."""
    import os

    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0007, 8, 24, 48, 60, 8, 2, 6.0)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam, lights = synthetic_camera(333, 187, extent=6.0), hk.lights_uniform(directional=sun)   # (odd sizes: partial tiles on both edges)
    with product_default_traversal():
        fused = hk.HikariPlugin(device=0)
        os.environ["HK_PREPASS_QUEUE"] = "1"   # (read by hk_create)
        try:
            queued = hk.HikariPlugin(device=0)
        finally:
            del os.environ["HK_PREPASS_QUEUE"]
    for p in (fused, queued):
        p.set_scene(scene)
    for n in range(1, 5):
        for p in (fused, queued):
            p.render(cam, s, lights=lights, frame_number=n)
        assert diff_buffers(snapshot(queued), snapshot(fused)) == {}, n
    assert queued.engine.wide_walk() and queued.engine.stats().wide_stack_lost == 0
