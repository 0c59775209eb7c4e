"""The persistent schedule of the queue-based indirect pass (round 6; kernels_wavefront.hip k_wf_trace_wide<.., PATHS>): every bounce
in ONE launch, a path staying with the wave that claimed it - against the staged schedule (one trace + one shade launch per bounce;
HK_DEBUG_OPT_PERSISTENT_PATHS = 0), which the rest of the suite holds to the oracle.  Same walks, same arithmetic per bounce, same order
of a path's additions: EVERY byte of every buffer equal, whatever the timing did to the order in which a wave met its rays."""
import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import diff_buffers, product_default_traversal, snapshot

pytestmark = pytest.mark.gpu


def pair(flags=0):
    with product_default_traversal():
        one, staged = hk.HikariPlugin(device=0, flags=flags), hk.HikariPlugin(device=0, flags=flags)
    one.engine.set_debug_option(F.DEBUG_OPT_PERSISTENT_PATHS, 1)
    staged.engine.set_debug_option(F.DEBUG_OPT_PERSISTENT_PATHS, 0)
    return one, staged


@pytest.mark.parametrize("bounces,size", [(2, (640, 360)), (2, (197, 111)), (3, (320, 180)), (5, (160, 90))])
def test_city_class_scene_every_byte(bounces, size):
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)
    s = hk.HikariSettings(indirect_bounces=bounces, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = synthetic_camera(*size, extent=30.0)
    lights = hk.lights_uniform(directional=dict(sun, illuminance=10000.0))
    one, staged = pair()
    for p in (one, staged):
        p.set_scene(scene)
    for n in (1, 2, 3):
        for p in (one, staged):
            p.render(cam, s, lights=lights, frame_number=n)
        assert diff_buffers(snapshot(one), snapshot(staged)) == {}, n
    assert one.engine.indirect_schedule() == "wavefront" and one.engine.wide_walk()
    assert one.engine.stats().wide_stack_lost == 0
    out = one.output(s)
    assert np.isfinite(out).all() and out[..., :3].max() > 0.05


def test_long_walks_inside_few_large_meshes_every_byte():
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0007, 2, 160, 320, 6, 8, 2, 3.0)
    s = hk.HikariSettings(indirect_bounces=3, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = synthetic_camera(320, 180, extent=3.0)
    lights = hk.lights_uniform(directional=sun)
    one, staged = pair()
    for p in (one, staged):
        p.set_scene(scene)
    for n in (1, 2, 3):
        for p in (one, staged):
            p.render(cam, s, lights=lights, frame_number=n)
        assert diff_buffers(snapshot(one), snapshot(staged)) == {}, n
    assert one.engine.stats().wide_stack_lost == 0


def test_config3_class_scene_with_emitters_and_a_moving_camera():
    """emitters (the shading samples their meshes: the light tree's walk and the emitter's mesh tree inside the trace kernel's waves),
    the camera moving from frame to frame (reprojection, scatter stores), settings changing between frames (the planes per bounce are
    carved again for more bounces).  HK_CTX_DETERMINISTIC_SCATTER: under motion the default resolves the reference's scatter race only in the
    buffers that are read - the sun / emitter channels' previous_spatial records race on, in both contexts, each its own way."""
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large(0x5EED0003, 40, 40, 80, 400, 50, 8, 12.0)
    lights = hk.lights_uniform(directional=sun)
    one, staged = pair(F.CTX_DETERMINISTIC_SCATTER)
    for p in (one, staged):
        p.set_scene(scene)
    for n, (bounces, dx) in enumerate([(2, 0.0), (2, 0.05), (4, 0.1), (1, 0.15), (3, 0.2)], start=1):
        s = hk.HikariSettings(indirect_bounces=bounces, upscale=hk.Upscale.SMAA_TU_1_0)
        cam = hk.Camera(hk.look_at_transform((1.6 * 9.0 + dx, 1.1 * 9.0, 2.0 * 9.0), (0.0, 0.6, 0.0)), 480, 270)
        for p in (one, staged):
            p.render(cam, s, lights=lights, frame_number=n)
        assert diff_buffers(snapshot(one), snapshot(staged)) == {}, n
