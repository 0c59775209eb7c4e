"""Every schedule and every cheaper route renders the SAME BYTES: second stream, frame pipelining, primary-ray pipelining, the
certified division route, uniform-tile store elision.  Split from test_parity_gpu.py."""
import os

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import diff_buffers, make_case, oracle, run_case, snapshot

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("name", ["cornell_b2", "cornell_upscale2", "cornell_ratio15_fsr", "cornell_b8", "yard_sun", "yard_textured", "yard_ortho", "background_only", "tiny_3x5"])
def test_windowed_spatial_reuse_changes_no_byte(name):
    """k_spatial_reuse has two forms (kernels.hip): the plain one and the WINDOWED one - the depths a workgroup's taps reach in an LDS
    window, the taps that reach their record in per-lane lists - which launch_spatial takes for launches of many rounds of workgroups
    (4K frames; the full-size tests of configs 4 / 5 run it).  Here it is FORCED on small cases - upscale ratios 1, 1.5 and 2 (the window
    is in deferred texels), both spatial passes, an orthographic camera, images smaller than a window, no geometry at all: every buffer
    of every frame equals the oracle's, and the windowed form really ran (light.wgsl:1503-1684)."""
    case = make_case(name)
    cpu = oracle()
    forced = hk.HikariPlugin(device=0)
    forced.engine.set_debug_option(F.DEBUG_OPT_SPATIAL_WINDOW, 1)
    plain = hk.HikariPlugin(device=0)

    for p in (forced, plain, cpu):
        p.set_scene(case.scene)
    for n in case.frames:
        for p in (forced, plain, cpu):
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        want = snapshot(cpu)
        for what, p in (("windowed", forced), ("plain", plain)):
            bad = diff_buffers(snapshot(p), want)
            assert bad == {}, (what, n, bad)
    spatial_passes = int(bool(case.settings.emissive_spatial_reuse)) + int(bool(case.settings.indirect_spatial_reuse))
    assert plain.engine.spatial_windowed_launches() == 0
    assert forced.engine.spatial_windowed_launches() == (spatial_passes * len(case.frames) if spatial_passes else 0)


def test_the_side_stream_may_lag_a_frame_behind_and_no_bit_changes():
    """Round 6: the main stream no longer waits for the direct-light dispatches (side stream) at the end of a frame - the post-processing
    does, on its own stream; the next frame's primary rays and indirect pass touch nothing they read or write (normal / instance_material
    planes double-buffered like the rest of the G-buffer).  Against HK_DEBUG_OPT_SIDE_JOIN = 1 (the order of rounds 1-5) and against the
    single-stream context: Cornell back to back, frames of one parity in a row, stages driven by hand without a post-processing stage, a
    scene beyond LDS whose instances move through the two-slot upload AND the device refit between frames."""
    from bevy_hikari_amd.scenes import animate, synthetic_camera, synthetic_scene

    case = make_case("cornell_b2")
    s, cam = case.settings, case.camera
    view, pview = cam.view_uniform(), cam.previous_view_uniform()
    engines = []
    for flags, join in ((0, 0), (0, 1), (F.CTX_SINGLE_STREAM, 0)):
        e = hk.Engine(device=0, flags=flags)
        e.upload_noise(); e.upload_scene(case.scene); e.resize(cam.width, cam.height, s.upscale.ratio())
        e.set_debug_option(F.DEBUG_OPT_SIDE_JOIN, join)
        engines.append(e)
    n = 0
    for step in [1] * 9 + [2, 2, 2] + ["temporal_only"] * 4 + [1, 1, 2, 1]:
        n += step if isinstance(step, int) else 1
        for e in engines:
            if step == "temporal_only":   # a host that drives the stages itself and skips the post-processing of some frames
                e.frame_begin(hk.frame_uniform(s, n), view, pview, case.lights)
                e.frame_stage(F.STAGE_TEMPORAL, s.to_c())
                e.frame_stage(F.STAGE_SPATIAL, s.to_c())
            else:
                e.frame_render(hk.frame_uniform(s, n), view, pview, case.lights, s.to_c())
    for b in (F.BUF_TONE_MAPPED, F.BUF_RENDER0, F.BUF_RENDER0 + 1, F.BUF_RENDER0 + 2, F.BUF_VARIANCE0, F.BUF_NORMAL, F.BUF_INSTANCE_MATERIAL, F.BUF_RESERVOIR0, F.BUF_RESERVOIR0 + 1,
              F.BUF_RESERVOIR0 + 2, F.BUF_RESERVOIR0 + 3, F.BUF_RESERVOIR0 + 4, F.BUF_RESERVOIR0 + 5, F.BUF_RESERVOIR0 + 6, F.BUF_RESERVOIR0 + 8):
        a = engines[0].read(b)
        for other in engines[1:]:
            assert (a.view(np.uint8) == other.read(b).view(np.uint8)).all(), b
    del engines
    # a scene beyond LDS, instances moving: the host-side re-finish + two-slot upload on odd frames, the device refit on even ones
    scenes = [synthetic_scene(n_boxes=20, n_spheres=5, n_emitters=3, sphere_rings=12, sphere_segs=16, seed=77)[0] for _ in range(2)]
    sun = synthetic_scene(n_boxes=1, n_spheres=1, n_emitters=1, seed=77)[1]
    cam = synthetic_camera(160, 96)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0, emissive_spatial_reuse=True)
    lights = hk.lights_uniform(directional=sun)
    plugins = []
    for join in (0, 1):
        p = hk.HikariPlugin(device=0, flags=F.CTX_DETERMINISTIC_SCATTER if False else 0)
        p.engine.set_debug_option(F.DEBUG_OPT_SIDE_JOIN, join)
        p.set_scene(scenes[join])
        plugins.append(p)
    cur = list(scenes)
    for n in range(1, 11):
        for k, p in enumerate(plugins):
            if n > 1:
                cur[k] = animate(cur[k], n - 1)
                p.update_instances(cur[k])
            p.render(cam, s, lights=lights, frame_number=n)
    # (the racing default: the two contexts run the SAME kernels in the same order per stream - what is compared is every rendered plane)
    a, b = snapshot(plugins[0]), snapshot(plugins[1])
    bad = {k: v for k, v in diff_buffers(a, b).items() if not k.startswith("reservoir")}
    assert bad == {}, bad


def test_the_main_streams_priority_changes_no_bit_and_may_change_between_frames():
    """Round 6: the context's main stream is created at the device's highest stream priority while the context dispatches few pixels per
    frame (context.hip pick_main_stream) and created again at the other priority when that changes - here forced back and forth
    between frames (HK_DEBUG_OPT_MAIN_PRIORITY), with the side stream's direct-light dispatches and the post stream's a-trous levels of
    the frames before still in flight when the change is asked for.  Every buffer of every frame equals a context that never changed."""
    case = make_case("cornell_b2")
    a, b = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0)
    for p in (a, b):
        p.set_scene(case.scene)
    b.engine.set_debug_option(F.DEBUG_OPT_MAIN_PRIORITY, 0)
    for n in range(1, 13):
        if n % 3 == 0:
            a.engine.set_debug_option(F.DEBUG_OPT_MAIN_PRIORITY, (n // 3) % 2)
        if n == 8:   # ... and the rule's own two triggers: a band of the frame, the whole frame again
            a.engine.set_band(0, 2)
            a.engine.set_band(0, 1)
        for p in (a, b):
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n)
        assert diff_buffers(snapshot(a), snapshot(b)) == {}, n
    a.engine.set_debug_option(F.DEBUG_OPT_MAIN_PRIORITY, -1)
    a.render(case.camera, case.settings, lights=case.lights, frame_number=13)
    b.render(case.camera, case.settings, lights=case.lights, frame_number=13)
    assert diff_buffers(snapshot(a), snapshot(b)) == {}


def test_the_main_streams_priority_follows_the_pixels_of_the_contexts_first_frame():
    """context.hip pick_main_stream: decided once, at the first frame - the highest priority for a context that dispatches at most 6 Mi
    pixels per frame (a small frame, or a band of a large one), the default for a whole 3840 x 2160 frame - and not revisited by later
    resizes or bands."""
    case = make_case("cornell_b2")
    s = case.settings

    def engine(width, height, band=None):
        e = hk.Engine(device=0)
        e.upload_noise()
        e.upload_scene(case.scene)
        e.resize(width, height, 1.0)
        if band:
            e.set_band(*band)
        assert e.main_stream_priority() == (False, False)       # created with the context at the default priority; nothing decided yet
        cam = hk.cornell_camera(width, height)
        e.frame_render(hk.frame_uniform(s, 1), cam.view_uniform(), cam.previous_view_uniform(), case.lights, s.to_c())
        e.wait()
        return e

    assert engine(640, 360).main_stream_priority() == (True, True)
    big = engine(3840, 2160)
    assert big.main_stream_priority() == (False, True)
    big.resize(640, 360, 1.0)                                   # (decided: a later size does not revisit it)
    cam = hk.cornell_camera(640, 360)
    big.frame_render(hk.frame_uniform(s, 1), cam.view_uniform(), cam.previous_view_uniform(), case.lights, s.to_c())
    big.wait()
    assert big.main_stream_priority() == (False, True)
    assert engine(3840, 2160, band=(1, 4)).main_stream_priority() == (True, True)   # a quarter of it: 2 Mi pixels per frame


def test_primary_ray_pipelining_changes_no_bit_and_stands_down_when_the_scene_is_written():
    """Round 6 (context.hip stage TEMPORAL): on a context whose chain runs in the high-priority queue pool, a frame's primary rays go to a
    stream of their own - ordered only behind the frame two back - and run beside the previous frame's spatial pass.  A frame whose scene
    memory was written since the last one (a refit, an upload), a frame behind the anti-aliasing tail, a timed frame take the serial order.
    Every buffer of every frame equals a context that never pipelines; the camera moves every frame."""
    from bevy_hikari_amd.scenes import synthetic_large

    scene, sun = synthetic_large(0x5EED0003, 40, 40, 80, 400, 50, 8, 12.0)
    lights = hk.lights_uniform(directional=sun)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    from cases import product_default_traversal
    with product_default_traversal():
        a, b = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0)   # (the product default: the verification contexts have no post stream, hence no fourth)
    a.engine.set_debug_option(F.DEBUG_OPT_PREPASS_PIPELINE, 1)
    b.engine.set_debug_option(F.DEBUG_OPT_PREPASS_PIPELINE, 0)
    for p in (a, b):
        p.set_scene(scene)
    rest = [np.ctypeslib.as_array(i.model).copy() for i in scene.instances]
    setter = scene.builder.api.raw("scene_builder_set_instance_transform")
    import ctypes as C
    for n in range(1, 25):
        cam = hk.Camera(hk.look_at_transform((1.6 * 9.0 + 0.03 * n, 1.1 * 9.0, 2.0 * 9.0), (0.0, 0.6, 0.0)), 320, 180)
        if n in (6, 7, 14):   # instances move: a device refit of both contexts' scenes between two frames
            for k in range(0, len(rest), 37):
                m = rest[k].reshape(4, 4).T.copy()
                m[0, 3] += 0.02 * n
                t = m.T.astype(np.float32).reshape(-1)
                setter(scene.builder.h, k, t.ctypes.data_as(C.POINTER(F.f32)))
            for p in (a, b):
                p.engine.refit_instances(scene.builder)
        for p in (a, b):
            p.render(cam, s, lights=lights, frame_number=n, antialias=(n == 10))
        if n % 4:   # (reading a buffer waits for the post stream, after which the next frame has nothing to pipeline behind: frames run in bursts)
            continue
        # (the camera moves: the default resolves the scatter race in the buffers that are read - the sun / emitter channels' previous_spatial
        # records, reservoir4 / reservoir5, race on in both contexts, each its own way)
        bad = {k: v for k, v in diff_buffers(snapshot(a), snapshot(b)).items() if k not in ("reservoir4", "reservoir5")}
        assert bad == {}, n
    assert a.engine.main_stream_priority()[0] and a.engine.prepasses_pipelined() >= 6 and b.engine.prepasses_pipelined() == 0
