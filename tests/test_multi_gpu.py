"""The halo exchange INSIDE the boundary (include/hikari_hip.h: hk_multi_*, hk_comm_*), on the one GPU a test box has.

hk_multi_*: one process drives several bands; here every band's context sits on device 0 (device ids may repeat), so the
whole code path - per-band stages, hk_band_schedule, peer copies on the receiver's stream, the event ordering between the
bands' streams, the gather - runs exactly as on n GPUs, and the union of the bands must equal the single-context frame bit
for bit.  hk_comm_*: RCCL refuses two ranks on one device, so what can run here is the one-rank communicator (dlopen of
librccl, ncclGetUniqueId, ncclCommInitRank, hk_frame_render's exchange hooks with an empty schedule)."""
import ctypes as C

import os

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.distributed import MultiEngine
from cases import make_case, random_case, run_case, snapshot

pytestmark = pytest.mark.gpu


def case_of(name):
    return random_case(int(name[6:])) if name.startswith("random") else make_case(name)


@pytest.mark.parametrize("bands,case_name", [(2, "cornell_b2"), (3, "yard_sun"), (3, "cornell_aa_default"), (4, "random12"), (2, "cornell_aa_fsr"), (8, "cornell_b2"),
                                             (3, "random33"), (2, "yard_textured")])
def test_multi_engine_union_equals_single_context(bands, case_name):
    case = case_of(case_name)
    s = case.settings
    m = MultiEngine([0] * bands)
    m.upload_noise()
    m.upload_scene(case.scene)
    w, h = case.camera.width, case.camera.height
    m.resize(w, h, s.upscale.ratio())
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    for n in case.frames:
        m.frame_render(hk.frame_uniform(s, n), view, pview, case.lights, s.to_c(), F.FRAME_ANTIALIAS if case.antialias else 0)
    m.wait()
    ref = hk.HikariPlugin(device=0)
    run_case(ref, case)
    e = ref.engine
    prev = 1 - case.frames[-1] % 2
    want = [F.BUF_TONE_MAPPED, F.BUF_RENDER0, F.BUF_RENDER0 + 1, F.BUF_RENDER0 + 2, F.BUF_VARIANCE0 + 2, F.BUF_RESERVOIR0 + prev + 6, F.BUF_RESERVOIR0 + prev + 8,
            F.BUF_POSITION, F.BUF_ALBEDO]
    if s.denoise:
        want += [F.BUF_DENOISE_RENDER0, F.BUF_DENOISE_RENDER0 + 1] + ([F.BUF_DENOISE_RENDER0 + 2] if s.indirect_bounces else [])
    if case.antialias:
        want += [F.BUF_TAA_OUTPUT] if s.taa == hk.Taa.Jasmine else []
        want += [F.BUF_UPSCALE_OUTPUT] + ([F.BUF_UPSCALE_SHARPENED] if s.upscale.kind == F.UPSCALE_FSR1 else [])
    for b in want:
        a, r = m.read(b), e.read(b)
        if F.BUF_RESERVOIR0 <= b < F.BUF_RESERVOIR0 + 10:   # rw x rh records inside window-size storage: compare the records
            rw, rh, _ = e.buffer_info(F.BUF_TONE_MAPPED)
            a, r = a.reshape(-1, 16)[:rw * rh], r.reshape(-1, 16)[:rw * rh]
        assert a.shape == r.shape and (a.view(np.uint8) == r.view(np.uint8)).all(), f"{case_name} x{bands}: buffer {b} differs"


@pytest.mark.parametrize("bands", [8, 5])
def test_multi_engine_full_size_config2_equals_single_context(bands):
    """BASELINE config 2 at its full 1920x1080 (2 bounces, ReSTIR, denoise), cut into 8 bands (135 rows each: the driver's
    8-GPU run) and 5 bands (216 rows): the union of the bands equals the single-context frame bit for bit, every buffer the
    frame's consumers read."""
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    w, h, frames = 1920, 1080, 4
    cam = hk.cornell_camera(w, h)
    view, pview, lights = cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform()
    scene = hk.load_cornell()
    m = MultiEngine([0] * bands)
    ref = hk.Engine(device=0)
    for t in (m, ref):
        t.upload_noise(); t.upload_scene(scene); t.resize(w, h, 1.0)
    for n in range(1, frames + 1):
        f = hk.frame_uniform(s, n)
        m.frame_render(f, view, pview, lights, s.to_c())
        ref.frame_render(f, view, pview, lights, s.to_c())
    m.wait(); ref.wait()
    prev = 1 - frames % 2
    for b in [F.BUF_TONE_MAPPED, F.BUF_DENOISE_RENDER0, F.BUF_DENOISE_RENDER0 + 1, F.BUF_DENOISE_RENDER0 + 2, F.BUF_RENDER0, F.BUF_RENDER0 + 1, F.BUF_RENDER0 + 2,
              F.BUF_VARIANCE0 + 2, F.BUF_RESERVOIR0 + prev + 6, F.BUF_RESERVOIR0 + prev + 8, F.BUF_POSITION, F.BUF_ALBEDO]:
        a, r = m.read(b), ref.read(b)
        assert a.shape == r.shape and (a.view(np.uint8) == r.view(np.uint8)).all(), f"x{bands}: buffer {b} differs at 1920x1080"
    borrowed = m.contexts[0]
    m.close()
    assert m.contexts == [] and not borrowed.ctx   # (the borrowed contexts died with the hk_multi ...)
    with pytest.raises(hk.HikariError):             # ... so a later call gets HK_E_INVALID for a NULL context, not freed memory
        borrowed.stats()


def _uneven(rh, bands, seed):
    cuts = np.random.default_rng(seed).choice(np.arange(1, rh), size=bands - 1, replace=False)
    return [0] + sorted(int(c) for c in cuts) + [rh]


@pytest.mark.parametrize("bands,case_name,split", [(3, "cornell_b2", "uneven"), (4, "yard_sun", "uneven"), (3, "cornell_aa_default", "uneven"), (2, "cornell_aa_fsr", "uneven"),
                                                   (5, "random12", "uneven"), (3, "yard_sun", "balanced"), (4, "cornell_aa_default", "balanced"),
                                                   (3, "cornell_aa_fsr", "balanced"), (4, "cornell_upscale2", "balanced")])
def test_multi_engine_bands_of_unequal_height(bands, case_name, split):
    """hk_set_band_bounds / HK_FRAME_BALANCE_BANDS: the bands need not be equally tall - random boundaries (bands of a single
    row included), or the split by cost the library derives on the first frame.  The union equals the single context, bit for bit."""
    case = case_of(case_name)
    s = case.settings
    m = MultiEngine([0] * bands)
    m.upload_noise(); m.upload_scene(case.scene)
    w, h = case.camera.width, case.camera.height
    m.resize(w, h, s.upscale.ratio())
    _, rh, _ = m.contexts[0].buffer_info(F.BUF_TONE_MAPPED)
    if split == "uneven":
        bounds = _uneven(rh, bands, 7 * bands + len(case_name))
        m.set_band_bounds(bounds)
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    for k, n in enumerate(case.frames):
        flags = (F.FRAME_ANTIALIAS if case.antialias else 0) | (F.FRAME_BALANCE_BANDS if split == "balanced" and k == 0 else 0)
        m.frame_render(hk.frame_uniform(s, n), view, pview, case.lights, s.to_c(), flags)
    m.wait()
    got = [e.band_bounds() for e in m.contexts]
    assert all(g == got[0] for g in got) and got[0][0] == 0 and got[0][-1] == rh
    if split == "uneven":
        assert got[0] == bounds
    else:   # the split follows the geometry: it is what the counts of the single context's G-buffer give
        from bevy_hikari_amd.distributed import balanced_band_bounds

        m2 = hk.Engine(device=0)   # (frame 1's G-buffer: the sub-pixel jitter of later frames moves a boundary by a row)
        m2.upload_noise(); m2.upload_scene(case.scene); m2.resize(w, h, s.upscale.ratio())
        m2.frame_render(hk.frame_uniform(s, case.frames[0]), view, pview, case.lights, s.to_c())
        assert got[0] == balanced_band_bounds(m2.row_costs(), w, rh, bands, 8, 0.25)   # (scenes walked from LDS: a background pixel costs 1/4 of a geometry pixel)
    ref = hk.HikariPlugin(device=0)
    run_case(ref, case)
    e = ref.engine
    costs = e.row_costs()
    depth = e.read(F.BUF_POSITION)[..., 3]
    assert (costs == (~(depth < np.float32(1.1920929e-7))).sum(axis=1)).all()
    prev = 1 - case.frames[-1] % 2
    want = [F.BUF_TONE_MAPPED, F.BUF_RENDER0, F.BUF_RENDER0 + 1, F.BUF_RENDER0 + 2, F.BUF_VARIANCE0 + 2, F.BUF_RESERVOIR0 + prev + 6, F.BUF_RESERVOIR0 + prev + 8,
            F.BUF_POSITION, F.BUF_ALBEDO]
    if s.denoise:
        want += [F.BUF_DENOISE_RENDER0, F.BUF_DENOISE_RENDER0 + 1] + ([F.BUF_DENOISE_RENDER0 + 2] if s.indirect_bounces else [])
    if case.antialias:
        want += [F.BUF_TAA_OUTPUT] if s.taa == hk.Taa.Jasmine else []
        want += [F.BUF_UPSCALE_OUTPUT] + ([F.BUF_UPSCALE_SHARPENED] if s.upscale.kind == F.UPSCALE_FSR1 else [])
    for b in want:
        a, r = m.read(b), e.read(b)
        if F.BUF_RESERVOIR0 <= b < F.BUF_RESERVOIR0 + 10:
            rw, rh2, _ = e.buffer_info(F.BUF_TONE_MAPPED)
            a, r = a.reshape(-1, 16)[:rw * rh2], r.reshape(-1, 16)[:rw * rh2]
        assert a.shape == r.shape and (a.view(np.uint8) == r.view(np.uint8)).all(), f"{case_name} x{bands} {split} {got[0]}: buffer {b} differs"


def test_multi_engine_balanced_config4_class_frame_at_1080p():
    """A frame shaped like BASELINE config 4 (sky above, a city below: equal-row bands take 0.3 ms and 6.3 ms) at 1920x1080 in 8
    bands split by cost: the sky bands come out tall, the city bands thin, the union equals the single context bit for bit, and
    the heaviest band holds far fewer geometry pixels than under the equal split."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0004, 12, 16, 32, 600, 20, 1, 40.0)   # bench.py's config-4 layout and camera, lighter meshes
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    w, h, bands = 1920, 1080, 8
    cam, lights = synthetic_camera(w, h, extent=30.0), hk.lights_uniform(directional=dict(sun, illuminance=10000.0))
    view, pview = cam.view_uniform(), cam.previous_view_uniform()
    m, ref = MultiEngine([0] * bands), hk.Engine(device=0)
    for t in (m, ref):
        t.upload_noise(); t.upload_scene(scene); t.resize(w, h, 1.0)
    for n in range(1, 4):
        f = hk.frame_uniform(s, n)
        m.frame_render(f, view, pview, lights, s.to_c(), F.FRAME_BALANCE_BANDS if n == 1 else 0)
        ref.frame_render(f, view, pview, lights, s.to_c())
    m.wait(); ref.wait()
    bounds = m.contexts[0].band_bounds()
    costs = ref.row_costs().astype(np.int64)
    per_band = [int(costs[a:b].sum()) for a, b in zip(bounds, bounds[1:])]
    equal = [int(c.sum()) for c in np.array_split(costs, bands)]
    print("balanced split:", bounds, "geometry px per band:", per_band, "equal split:", equal)
    assert max(per_band) <= 0.75 * max(equal) or max(equal) <= 1.15 * sum(equal) / bands, (per_band, equal)
    prev = 1 - 3 % 2
    for b in [F.BUF_TONE_MAPPED, F.BUF_DENOISE_RENDER0, F.BUF_DENOISE_RENDER0 + 1, F.BUF_DENOISE_RENDER0 + 2, F.BUF_RENDER0 + 2, F.BUF_VARIANCE0 + 2,
              F.BUF_RESERVOIR0 + prev + 6, F.BUF_RESERVOIR0 + prev + 8, F.BUF_POSITION]:
        a, r = m.read(b), ref.read(b)
        assert (a.view(np.uint8) == r.view(np.uint8)).all(), f"balanced x{bands}: buffer {b} differs at 1920x1080"


@pytest.mark.parametrize("bands,case_name,balanced", [(3, "cornell_b2", False), (4, "cornell_aa_default", True), (3, "cornell_aa_fsr", False), (2, "yard_aa_fsr_notaa", False),
                                                      (5, "yard_aa_smaa2x", True), (8, "cornell_b2", False)])
def test_multi_engine_gathers_the_final_image_on_band_0(bands, case_name, balanced):
    """HK_FRAME_GATHER: after the frame, band 0's CONTEXT holds the whole image the overlay presents - peer
    copies on its stream, no host merge - equal to the single context's, bit for bit."""
    case = case_of(case_name)
    s = case.settings
    m = MultiEngine([0] * bands)
    m.upload_noise(); m.upload_scene(case.scene)
    w, h = case.camera.width, case.camera.height
    m.resize(w, h, s.upscale.ratio())
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    aa = F.FRAME_ANTIALIAS if case.antialias else 0
    for k, n in enumerate(case.frames):
        m.frame_render(hk.frame_uniform(s, n), view, pview, case.lights, s.to_c(), aa | F.FRAME_GATHER | (F.FRAME_BALANCE_BANDS if balanced and k == 0 else 0))
    m.wait()
    final = F.api().final_buffer(s.to_c(), aa)
    ref = hk.HikariPlugin(device=0)
    run_case(ref, case)
    got, want = m.contexts[0].read(final), ref.engine.read(final)
    assert got.shape == want.shape and (got.view(np.uint8) == want.view(np.uint8)).all(), f"{case_name} x{bands}: band 0 does not hold the gathered image"
    # every frame gathered: the bands' own state must not have been disturbed (the union still equals the single context)
    assert (m.read(F.BUF_TONE_MAPPED).view(np.uint8) == ref.engine.read(F.BUF_TONE_MAPPED).view(np.uint8)).all()


_MOTION_BUFFERS = [F.BUF_TONE_MAPPED, F.BUF_DENOISE_RENDER0, F.BUF_DENOISE_RENDER0 + 1, F.BUF_DENOISE_RENDER0 + 2, F.BUF_RENDER0, F.BUF_RENDER0 + 1, F.BUF_RENDER0 + 2,
                   F.BUF_VARIANCE0, F.BUF_VARIANCE0 + 1, F.BUF_VARIANCE0 + 2, F.BUF_POSITION, F.BUF_VELOCITY_UV, F.BUF_ALBEDO]


def _same_buffers(m, ref, frame_number, settings, what=""):
    """Every buffer a frame's consumers read, bands' union vs single context, bit for bit - the reservoirs that have a reader included."""
    prev = 1 - frame_number % 2
    reservoirs = [prev + 0, prev + 2, prev + 6] + ([prev + 4] if settings.emissive_spatial_reuse else []) + ([prev + 8] if settings.indirect_spatial_reuse else [])
    rw, rh, _ = ref.buffer_info(F.BUF_TONE_MAPPED)
    for b in _MOTION_BUFFERS + [F.BUF_RESERVOIR0 + k for k in reservoirs]:
        a, r = m.read(b), ref.read(b)
        if F.BUF_RESERVOIR0 <= b < F.BUF_RESERVOIR0 + 10:
            a, r = a.reshape(-1, 16)[:rw * rh], r.reshape(-1, 16)[:rw * rh]
        bad = (a.view(np.uint8) != r.view(np.uint8))
        assert a.shape == r.shape and not bad.any(), f"{what}frame {frame_number}: buffer {b} differs in {int(bad.sum())} bytes"


@pytest.mark.parametrize("bands,emissive_spatial", [(2, False), (5, False), (8, True)])
def test_multi_engine_history_rows_under_camera_motion(bands, emissive_spatial):
    """SURVEY 8e step 6 inside the library.  With the camera moving vertically, reprojection crosses the band borders both ways:
    a band reads last frame's reservoirs in its neighbours' rows (exchange C) and its temporal dispatches store rejected history at
    reprojected slots that neighbours own and read (light.wgsl:1092-1095,1456-1459).  hk_frame_begin derives the halo from the two
    views and the scene's bounds (HK_HISTORY_AUTO, the default), the bands park their stores, hand each other the rows near the
    borders with exchange A and resolve them by the single-context rule: EVERY frame of the union equals the single context with
    HK_CTX_DETERMINISTIC_SCATTER bit for bit.  With the halo forced to 0 it does not."""
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0, emissive_spatial_reuse=emissive_spatial)
    w, h, frames = 96, 64, 8
    cams = [hk.Camera(hk.look_at_transform((0.0, 0.4 + 0.16 * n, 4.0), (0.0, 0.4 + 0.16 * n, 0.0)), w, h) for n in range(1, frames + 1)]
    scene = hk.load_cornell()
    ref = hk.Engine(device=0, flags=F.CTX_DETERMINISTIC_SCATTER)
    m = MultiEngine([0] * bands)           # (no verification flag: a band with a history halo parks its stores by itself)
    m0 = MultiEngine([0] * bands)
    m0.set_history_rows(0)
    for t in (ref, m, m0):
        t.upload_noise(); t.upload_scene(scene); t.resize(w, h, 1.0)
    used = []
    for n in range(1, frames + 1):
        cam, prev = cams[n - 1], cams[max(n - 2, 0)]
        args = (hk.frame_uniform(s, n), cam.view_uniform(), cam.previous_view_uniform(prev), hk.lights_uniform(), s.to_c())
        for t in (ref, m, m0):
            t.frame_render(*args)
        used.append([e.history_rows() for e in m.contexts])
        _same_buffers(m, ref, n, s, f"x{bands} ")
    assert all(u == [0] * bands for u in used[:1]) and all(len(set(u)) == 1 and 4 <= u[0] <= 16 for u in used[1:]), used
    assert ref.history_rows() == 0          # (a single band has no halo)
    got = np.stack([m0.read(F.BUF_DENOISE_RENDER0 + i).view(np.float16).astype(np.float32) for i in range(3)])
    want = np.stack([ref.read_f16(F.BUF_DENOISE_RENDER0 + i) for i in range(3)])
    assert float(np.linalg.norm(got - want) / np.linalg.norm(want)) > 1e-3   # without the halo the bands drift


@pytest.mark.parametrize("bands,emissive_spatial", [(3, False), (5, True), (8, False)])
def test_multi_engine_rebalances_by_measured_time_under_camera_motion(bands, emissive_spatial):
    """Round 6, on the GPU: the split follows band times - hk_rebalanced_band_bounds on one number per band,
    hk_multi_migrate_bands moves the history rows that change owner (peer copies), the new split is in force from the next frame -
    while the camera moves vertically (exchange C + the parked scatter stores).  EVERY frame of the union equals the single context
    with HK_CTX_DETERMINISTIC_SCATTER bit for bit, every buffer and reservoir; the boundaries really move (by up to 6 rows, four
    times); the same boundaries set WITHOUT the migration leave stale history in the rows that changed owner."""
    from bevy_hikari_amd.distributed import rebalanced_band_bounds

    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0, emissive_spatial_reuse=emissive_spatial)
    w, h, frames = 96, 64, 9
    cams = [hk.Camera(hk.look_at_transform((0.0, 0.4 + 0.16 * n, 4.0), (0.0, 0.4 + 0.16 * n, 0.0)), w, h) for n in range(1, frames + 1)]
    scene = hk.load_cornell()
    ref = hk.Engine(device=0, flags=F.CTX_DETERMINISTIC_SCATTER)
    m = MultiEngine([0] * bands)
    cut = MultiEngine([0] * bands)           # the same boundaries, set like a cut: no migration
    for t in (ref, m, cut):
        t.upload_noise(); t.upload_scene(scene); t.resize(w, h, 1.0)
    seen = set()
    for n in range(1, frames + 1):
        cam, prev = cams[n - 1], cams[max(n - 2, 0)]
        args = (hk.frame_uniform(s, n), cam.view_uniform(), cam.previous_view_uniform(prev), hk.lights_uniform(), s.to_c())
        for t in (ref, m, cut):
            t.frame_render(*args)
        _same_buffers(m, ref, n, s, f"rebalanced x{bands} ")
        if n in (2, 4, 5, 7):
            ms = [float(1 + i) if n in (2, 7) else float(bands - i) for i in range(bands)]
            new = rebalanced_band_bounds(m.contexts[0].band_bounds(), ms, h, None, min_rows=3, max_shift=6, damping=0.6)
            m.migrate_bands(new, n + 1, s.to_c())
            cut.set_band_bounds(new)
            assert all(e.band_bounds() == new for e in m.contexts)
            seen.add(tuple(new))
    assert len(seen) >= 3, seen
    got = np.stack([cut.read(F.BUF_DENOISE_RENDER0 + i).view(np.float16).astype(np.float32) for i in range(3)])
    want = np.stack([ref.read_f16(F.BUF_DENOISE_RENDER0 + i) for i in range(3)])
    assert float(np.linalg.norm(got - want) / np.linalg.norm(want)) > 0.0   # (the migration is what keeps the bands exact)


def test_band_time_is_measured_on_the_stream():
    """HK_FRAME_TIME_BAND / hk_band_time_ms: events around stage TEMPORAL and stage SPATIAL on the context's stream - a positive time
    below the frame's wall clock, NOT READY before any timed frame, and no change to any byte."""
    import time

    case = case_of("cornell_b2")
    s = case.settings
    a, b = hk.Engine(device=0), hk.Engine(device=0)
    for e in (a, b):
        e.upload_noise(); e.upload_scene(case.scene); e.resize(256, 256, 1.0)
        e.set_band(1, 3)
    cam = hk.cornell_camera(256, 256)
    view, pview = cam.view_uniform(), cam.previous_view_uniform()
    with pytest.raises(F.HikariError):
        a.band_time_ms()
    for n in range(1, 5):
        t0 = time.perf_counter()
        a.frame_render(hk.frame_uniform(s, n), view, pview, case.lights, s.to_c(), F.FRAME_TIME_BAND)
        ms = a.band_time_ms()
        wall = (time.perf_counter() - t0) * 1e3
        b.frame_render(hk.frame_uniform(s, n), view, pview, case.lights, s.to_c())
        assert 0.0 < ms < wall + 0.05, (ms, wall)
    for buf in (F.BUF_TONE_MAPPED, F.BUF_RESERVOIR0 + 6, F.BUF_RESERVOIR0 + 8, F.BUF_DENOISE_RENDER0 + 2):
        assert (a.read(buf).view(np.uint8) == b.read(buf).view(np.uint8)).all()


@pytest.mark.parametrize("bands", [5, 8])
def test_multi_engine_camera_and_instance_motion_equals_single_context(bands):
    """camera AND instances move (device refit on every band's replica of the scene), 5 and 8 bands of unequal
    height on a 160 x 120 frame, both spatial passes on.  Nothing is supplied by the host: the halo is derived per frame, the stores
    that cross a border travel with exchange A.  Union of the bands == single context (HK_CTX_DETERMINISTIC_SCATTER), every buffer,
    every frame, bit for bit."""
    from bevy_hikari_amd.scenes import synthetic_scene
    from test_device_refit import LARGE, pose

    multi_scene, sun = synthetic_scene(**LARGE)
    single_scene, _ = synthetic_scene(**LARGE)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0, emissive_spatial_reuse=True)
    w, h, frames = 160, 120, 7
    lights = hk.lights_uniform(directional=sun)
    cams = [hk.Camera(hk.look_at_transform((6.4 + 0.05 * n, 4.4 + 0.22 * n, 8.0 - 0.1 * n), (0.0, 0.6 + 0.05 * n, 0.0)), w, h) for n in range(frames + 1)]
    m = MultiEngine([0] * bands)
    ref = hk.Engine(device=0, flags=F.CTX_DETERMINISTIC_SCATTER)
    for t, scene in ((m, multi_scene), (ref, single_scene)):
        t.upload_noise(); t.upload_scene(scene); t.resize(w, h, 1.0)
    m.set_band_bounds(_uneven(h, bands, 31 * bands))
    rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in single_scene.instances], dtype=np.float32)
    movers = [1, 4, 9, 15, 20, len(rest) - 2]
    halos = []
    for n in range(1, frames + 1):
        if n > 1:
            for k, i in enumerate(movers):
                for scene in (multi_scene, single_scene):
                    scene.builder.set_instance_transform(i, pose(rest[i], n - 1, k))
            assert m.refit_instances(multi_scene.builder) == len(movers)
            assert ref.refit_instances(single_scene.builder) == len(movers)
        args = (hk.frame_uniform(s, n), cams[n].view_uniform(), cams[n].previous_view_uniform(cams[n - 1] if n > 1 else cams[n]), lights, s.to_c())
        m.frame_render(*args)
        ref.frame_render(*args)
        halos.append(m.contexts[0].history_rows())
        _same_buffers(m, ref, n, s, f"camera + instances x{bands} ")
    assert halos[0] == 0 and min(halos[1:]) >= 4, halos


def test_rccl_single_rank_communicator():
    e = hk.Engine(device=0)
    ident = e.comm_unique_id()
    assert len(ident) == 128 and any(ident)
    e.comm_init(0, 1, ident)
    case = make_case("cornell_b2")
    e.upload_noise(); e.upload_scene(case.scene)
    e.resize(case.camera.width, case.camera.height, 1.0)
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    for n in case.frames:
        e.frame_render(hk.frame_uniform(case.settings, n), view, pview, case.lights, case.settings.to_c())
    ref = hk.HikariPlugin(device=0)
    run_case(ref, case)
    assert (e.read(F.BUF_TONE_MAPPED) == ref.engine.read(F.BUF_TONE_MAPPED)).all()
    e.comm_destroy()


def test_rccl_send_and_receive_execute_on_the_stream():
    """RCCL refuses two ranks on one device, so on a one-GPU box no halo exchange BETWEEN ranks can run - but
    a rank may send to itself inside a group.  hk_debug_comm_loopback moves rows of one buffer to another buffer of the same
    context through the very function hk_frame_render's exchanges use (comm.cpp run_transfers: ncclGroupStart, ncclSend,
    ncclRecv, ncclGroupEnd on the context's stream).  Enqueued between two frames with no host wait anywhere: the rows that arrive
    are what the frame BEFORE wrote (the exchange is ordered behind the kernels) and not what the frame AFTER writes over the
    source (the kernels are ordered behind the exchange); rows outside the range stay untouched."""
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0, denoise=False)   # (denoise off: the denoiser's outputs are free scratch)
    w, h = 256, 144
    cam = hk.cornell_camera(w, h)
    view, pview, lights = cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform()
    e, ref = hk.Engine(device=0), hk.Engine(device=0)
    for t in (e, ref):
        t.upload_noise(); t.upload_scene(hk.load_cornell()); t.resize(w, h, 1.0)
    e.comm_init(0, 1, e.comm_unique_id())
    lanes = C.c_uint32()
    e.api.call("debug_comm_lanes", e.ctx, C.byref(lanes))
    assert lanes.value == 2   # round 6: exchange B / the gather travel on a communicator and stream of their own (ncclCommSplit came up)
    src, dst, rows = F.BUF_RENDER0 + 2, F.BUF_DENOISE_RENDER0 + 2, (37, 101)
    ref.frame_render(hk.frame_uniform(s, 1), view, pview, lights, s.to_c())
    first = ref.read(src)
    ref.frame_render(hk.frame_uniform(s, 2), view, pview, lights, s.to_c())
    second = ref.read(src)
    assert (first[rows[0]:rows[1]] != second[rows[0]:rows[1]]).any() and first[rows[0]:rows[1]].any()
    e.frame_render(hk.frame_uniform(s, 1), view, pview, lights, s.to_c())
    e.debug_comm_loopback(src, dst, *rows)
    e.frame_render(hk.frame_uniform(s, 2), view, pview, lights, s.to_c())
    e.wait()
    got = e.read(dst)
    assert (got[rows[0]:rows[1]] == first[rows[0]:rows[1]]).all(), "the rows RCCL delivered are not what the frame before the exchange wrote"
    assert not got[:rows[0]].any() and not got[rows[1]:].any()
    assert (e.read(src) == second).all()
    # ... the same on the communicator's SECOND lane (exchange B's: its own communicator and stream, behind whichever stream the context is on)
    e.api.call("debug_comm_loopback", e.ctx, src, F.BUF_DENOISE_RENDER0 + 1, rows[0], rows[1], 2)
    e.wait()
    assert (e.read(F.BUF_DENOISE_RENDER0 + 1)[rows[0]:rows[1]] == second[rows[0]:rows[1]]).all()
    # ... and the way the GATHER of a finished frame runs (hk_frame_render with HK_FRAME_GATHER): on the communicator's stream, nobody
    # waits - frame 4 (the other parity's planes) renders meanwhile; frame 5 writes the plane the transfer reads and therefore joins
    # first (hk_frame_begin), as does anybody who reads a buffer.  What arrives is frame 3's tone-mapped image.
    ref.frame_render(hk.frame_uniform(s, 3), view, pview, lights, s.to_c())
    third = ref.read(F.BUF_TONE_MAPPED)
    e.frame_render(hk.frame_uniform(s, 3), view, pview, lights, s.to_c())
    e.debug_comm_loopback(F.BUF_TONE_MAPPED, F.BUF_DENOISE_RENDER0, 11, 133, overlapped=True)
    for n in (4, 5, 6):
        ref.frame_render(hk.frame_uniform(s, n), view, pview, lights, s.to_c())
        e.frame_render(hk.frame_uniform(s, n), view, pview, lights, s.to_c())
    got = e.read(F.BUF_DENOISE_RENDER0)
    assert (got[11:133] == third[11:133]).all() and third[11:133].any() and not got[:11].any() and not got[133:].any()
    assert (e.read(F.BUF_TONE_MAPPED) == ref.read(F.BUF_TONE_MAPPED)).all()
    with pytest.raises(hk.HikariError):
        e.debug_comm_loopback(src, F.BUF_POSITION, *rows)    # another shape
    e.comm_destroy()
    with pytest.raises(hk.HikariError):
        e.debug_comm_loopback(src, dst, *rows)               # no communicator


def test_multi_create_rejects_bad_devices():
    with pytest.raises(hk.HikariError) as err:
        MultiEngine([0, 99])
    assert err.value.code == F.HK_E_NO_DEVICE


def test_multi_engine_device_motion_reaches_every_band():
    """hk_multi_refit_scene_instances / hk_multi_rebuild_scene_trees: every band's device copy of the scene gets the same
    per-instance records and the same trees as a single context fed the same poses (the builder's transforms are committed
    once, after the last band: were they committed after the first, the others would see no motion), and the union of six bands EQUALS the single-context frame (moving objects reproject across
    the band borders: the derived history halo, the parked scatter stores)."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_scene
    from test_device_refit import LARGE, pose, same_links

    multi_scene, sun = synthetic_scene(**LARGE)
    single_scene, _ = synthetic_scene(**LARGE)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    w, h = 112, 72
    cam, lights = synthetic_camera(w, h), hk.lights_uniform(directional=sun)
    view, pview = cam.view_uniform(), cam.previous_view_uniform()
    m = MultiEngine([0] * 6)
    ref = hk.Engine(device=0, flags=F.CTX_DETERMINISTIC_SCATTER)
    for t, scene in ((m, multi_scene), (ref, single_scene)):
        t.upload_noise(); t.upload_scene(scene); t.resize(w, h, 1.0)
    rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in single_scene.instances], dtype=np.float32)
    movers = [2, 7, 11, 22, len(rest) - 1]
    n_tlas, n_light = len(single_scene.instance_nodes), len(single_scene.emissive_nodes)
    for n in range(1, 7):
        if n > 1:
            for k, i in enumerate(movers):
                for scene in (multi_scene, single_scene):
                    scene.builder.set_instance_transform(i, pose(rest[i], n - 1, k))
            assert m.refit_instances(multi_scene.builder) == len(movers)
            assert ref.refit_instances(single_scene.builder) == len(movers)
            if n == 4:
                m.rebuild_trees(F.TREE_SAH); ref.rebuild_trees(F.TREE_SAH)
            want = ref.read_trees(n_tlas, n_light)
            for e in m.contexts:
                got = e.read_trees(n_tlas, n_light)
                for a, b in zip(got, want):
                    assert same_links(a, b) and bytes(a) == bytes(b), f"frame {n}: a band's trees differ from the single context's"
        f = hk.frame_uniform(s, n)
        m.frame_render(f, view, pview, lights, s.to_c())
        ref.frame_render(f, view, pview, lights, s.to_c())
        # the camera rests, five instances move: the halo comes from their boxes (hk_history_rows_bound's moved boxes), the same
        # count on every band; the union equals the single context bit for bit, frame by frame
        rows = [e.history_rows() for e in m.contexts]
        assert len(set(rows)) == 1 and (rows[0] > 0) == (n > 1), (n, rows)
        _same_buffers(m, ref, n, s, "moving instances x6 ")
    m.wait()
    for e in m.contexts:
        st = e.stats()
        assert st.scene_device_refits == 5 and st.scene_device_tree_builds == 1


@pytest.mark.parametrize("bands,which", [(3, "large"), (8, "large"),
                                         pytest.param(8, "lds", marks=pytest.mark.xfail(strict=False, reason="the bands of a scene walked from LDS are pipelined by the debug option only: one run "
                                                      "of ~160 showed 11 records of an indirect spatial reservoir differing (every rendered plane equal); not reproduced since - NOTES 'open'"))])
def test_bands_with_pipelined_primary_rays_equal_the_single_context(bands, which):
    """Round 6: a band's context pipelines its primary rays like a single one (context.hip stage TEMPORAL; its fourth stream in the chain's
    queue pool, the default pool being full with the communicator lanes).  A scene beyond the LDS copy, camera AND instances moving
    (device refits between frames: those frames take the serial order), frames enqueued in bursts - a read waits for the post stream,
    after which the next frame has nothing to pipeline behind.  The bands' union == a single context that never pipelines
    (HK_CTX_DETERMINISTIC_SCATTER: it has no post stream), every buffer that has a reader, bit for bit."""
    from bevy_hikari_amd.scenes import synthetic_large
    from cases import product_default_traversal

    if which == "large":
        multi_scene, sun = synthetic_large(0x5EED0003, 40, 40, 80, 400, 50, 8, 12.0)
        single_scene, _ = synthetic_large(0x5EED0003, 40, 40, 80, 400, 50, 8, 12.0)
    else:   # a scene every kernel walks from its LDS copy: pipelined only where a band leaves most of the chip idle (the rule counts the band's pixels)
        from bevy_hikari_amd.scenes import synthetic_scene
        from test_device_refit import LARGE

        multi_scene, sun = synthetic_scene(**LARGE)
        single_scene, _ = synthetic_scene(**LARGE)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    w, h, frames = 256, 144, 18
    lights = hk.lights_uniform(directional=sun)
    eye = (1.6 * 9.0, 1.1 * 9.0, 2.0 * 9.0) if which == "large" else (6.4, 4.4, 8.0)
    cams = [hk.Camera(hk.look_at_transform((eye[0] + 0.04 * n, eye[1] + 0.03 * n, eye[2]), (0.0, 0.6, 0.0)), w, h) for n in range(frames + 1)]
    with product_default_traversal():
        m = MultiEngine([0] * bands)
        ref = hk.Engine(device=0, flags=F.CTX_DETERMINISTIC_SCATTER)
    for t, scene in ((m, multi_scene), (ref, single_scene)):
        t.upload_noise(); t.upload_scene(scene); t.resize(w, h, 1.0)
    m.set_band_bounds(_uneven(h, bands, 17 * bands))
    if os.environ.get("HIKARI_TEST_PIPELINE") or which == "lds":   # (the rule does not pipeline scenes walked from LDS; the environment variable is a debugging aid: 0 = never)
        for c in m.contexts:
            c.set_debug_option(F.DEBUG_OPT_PREPASS_PIPELINE, int(os.environ.get("HIKARI_TEST_PIPELINE", "1")))
    rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in single_scene.instances], dtype=np.float32)
    for n in range(1, frames + 1):
        if n in (5, 6, 13):
            for k in range(0, len(rest), 41 if which == "large" else 5):
                mm = rest[k].reshape(4, 4).T.copy()
                mm[0, 3] += 0.02 * n
                for scene in (multi_scene, single_scene):
                    scene.builder.set_instance_transform(k, mm.T.astype(np.float32).reshape(-1))
            assert m.refit_instances(multi_scene.builder) == ref.refit_instances(single_scene.builder) > 0
        args = (hk.frame_uniform(s, n), cams[n].view_uniform(), cams[n].previous_view_uniform(cams[n - 1]), lights, s.to_c())
        m.frame_render(*args)
        ref.frame_render(*args)
        if n % 3 == 0:
            _same_buffers(m, ref, n, s, f"pipelined primary rays x{bands} ")
    assert os.environ.get("HIKARI_TEST_PIPELINE") == "0" or sum(c.prepasses_pipelined() for c in m.contexts) >= bands * 4
