"""The near-plane clip of the ray-cast G-buffer (VERDICT r05 missing 2 / next 6).

The reference RASTERISES its G-buffer with `unclipped_depth: false` (/root/reference/src/prepass.rs:242-266; perspective camera, near =
0.1, infinite reverse-Z far plane): geometry between the eye and the near plane never reaches a fragment, a triangle that straddles the
plane is cut at it.  Primary rays start at the eye, so without a clip such geometry would enter the G-buffer - in the oracle as in the
kernels, which is why no parity test could see it.  Both now trace a pixel whose nearest surface lies in front of the plane once more
from the plane (hk_prepass.hpp clip_at_near_plane, oracle pass_prepass).  Here: the contract on the oracle (CPU), then GPU == oracle."""
import math

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.scenes import _box, _quad_strip, _trs
from cases import diff_buffers, oracle, snapshot

NEAR = 0.1
EYE = (0.0, 1.0, 4.0)


def build(with_near_quad=True, with_straddler=True, rocks=0):
    """A floor, a back wall and a lamp seen from EYE down -Z; + an upright quad 0.05 in front of the eye that covers the whole view
    (must vanish) + a horizontal strip just below the eye that runs from behind the camera to 1.5 in front of it (must be cut at
    z_view = 0.1).  The two extra quads come LAST: every other instance keeps its id.  rocks > 0: enough unique triangles to leave the
    LDS copy (the wide walk's prepass)."""
    from bevy_hikari_amd.plugin import SceneBuilder, standard_material
    from bevy_hikari_amd.scenes import _rock

    b = SceneBuilder()
    box = b.add_mesh(*_box())
    qp, qn, quv = _quad_strip(2)
    quad = b.add_mesh(qp, qn, quv, None, F.TOPOLOGY_TRIANGLE_STRIP)
    grey = b.add_material(standard_material((0.7, 0.7, 0.7, 1.0), (0, 0, 0), 0.8, 0.0, 0.5))
    red = b.add_material(standard_material((0.8, 0.2, 0.2, 1.0), (0, 0, 0), 0.6, 0.0, 0.5))
    lamp = b.add_material(standard_material((0.8, 0.8, 0.8, 1.0), (1.0, 0.9, 0.8), 1.0, 0.0, 0.5))
    b.add_instance(box, grey, _trs((0, -0.25, 0), (0, 0, 0), (12, 0.5, 12)))          # 0 floor
    b.add_instance(box, red, _trs((0, 1.5, -3), (0, 0, 0), (8, 3, 0.3)))              # 1 back wall
    b.add_instance(quad, lamp, _trs((0, 3.2, 0.5), (math.pi, 0, 0), (1.5, 1, 1.5)))   # 2 lamp, facing down
    rng = np.random.default_rng(5)
    for k in range(rocks):
        mesh = b.add_mesh(*_rock(rng, 24, 48))
        b.add_instance(mesh, grey if k % 2 else red, _trs((rng.uniform(-3, 3), rng.uniform(0.4, 1.6), rng.uniform(-2, 1.5)), rng.uniform(-1, 1, 3), rng.uniform(0.5, 1.0, 3)))
    ids = {}
    if with_straddler:   # XZ strip (normal +Y), 0.03 below the eye, z from EYE.z + 0.5 (behind the camera) to EYE.z - 1.5
        ids["straddler"] = b.add_instance(quad, red, _trs((0.0, EYE[1] - 0.03, EYE[2] - 0.5), (0, 0, 0), (0.6, 1, 2.0)))
    if with_near_quad:   # the XZ quad turned upright (normal towards the eye), 0.05 in front of the eye, 0.4 x 0.4: the frustum is 0.07 x 0.04 there
        ids["near"] = b.add_instance(quad, grey, _trs((0.0, EYE[1], EYE[2] - 0.05), (math.pi / 2, 0, 0), (0.4, 1, 0.4)))
    return b.finish(), ids


def camera(w=96, h=64):
    return hk.Camera(hk.look_at_transform(EYE, (0.0, 1.0, 0.0)), w, h)


def render(plugin, scene, cam, frames=(1, 2, 3)):
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    plugin.set_scene(scene)
    for n in frames:
        plugin.render(cam, s, frame_number=n)
    plugin.engine.wait()
    return snapshot(plugin)


def view_depth(position):
    """z_view of the stored world positions (camera looks down -Z from EYE), NaN for background"""
    z = EYE[2] - position[..., 2]
    return np.where(position[..., 3] > 0.0, z, np.nan)


def test_oracle_clips_the_g_buffer_at_the_near_plane():
    cam = camera()
    full, ids = build()
    plain, _ = build(with_near_quad=False, with_straddler=False)
    a = render(oracle(), full, cam, frames=(1,))
    b = render(oracle(), plain, cam, frames=(1,))
    inst = np.floor(a["instance_material"][..., 0]).astype(int)
    geometry = a["position"][..., 3] > 0.0
    # (1) the quad in front of the near plane is nowhere - although it covers every pixel's ray
    assert not (geometry & (inst == ids["near"])).any()
    # (2) nothing in the G-buffer lies in front of the plane: clip depth = near / z_view <= 1, z_view >= near
    z = view_depth(a["position"])
    assert np.nanmin(z) >= NEAR * (1.0 - 1e-5) and a["position"][..., 3].max() <= 1.0 + 1e-5
    # (3) the strip that straddles the plane is there - beyond the plane only - and cut AT it: its nearest stored point sits on the plane
    strip = geometry & (inst == ids["straddler"])
    assert strip.sum() > 50
    assert NEAR * (1.0 - 1e-5) <= z[strip].min() <= NEAR * 1.25
    # ... in the image it ends 16.7 degrees below the axis (atan(0.03 / 0.1): inside the 22.5 degrees of the view): the rows below,
    # whose rays meet the strip in front of the plane, show the floor under it instead
    rows = np.nonzero(strip.any(axis=1))[0]
    assert rows.max() < strip.shape[0] - 3 and not strip[rows.max() + 1:].any()
    mid = strip.shape[1] // 2
    assert geometry[rows.max() + 2, mid] and inst[rows.max() + 2, mid] == 0
    # (4) every pixel the two extra quads do not own shows exactly what the scene without them shows
    # (the same surfaces and normals exactly; their positions to rounding - these pixels were traced from the plane, not from the eye)
    other = ~strip
    for name in ("normal", "instance_material"):
        assert (a[name][other].view(np.uint8) == b[name][other].view(np.uint8)).all(), name
    assert np.allclose(a["position"][other], b["position"][other], rtol=2e-5, atol=2e-5)
    assert np.allclose(a["velocity_uv"][other], b["velocity_uv"][other], rtol=1e-4, atol=1e-5)


def test_orthographic_rays_already_start_on_their_near_plane():
    cam = hk.Camera(hk.look_at_transform(EYE, (0.0, 1.0, 0.0)), 64, 48, ortho_height=3.0)
    full, ids = build()
    a = render(oracle(), full, cam, frames=(1,))
    inst = np.floor(a["instance_material"][..., 0]).astype(int)
    # bevy's orthographic near is 0: the plane through the eye - the upright quad 0.05 beyond it IS drawn, nothing behind the eye is
    assert ((a["position"][..., 3] > 0.0) & (inst == ids["near"])).any()
    assert np.nanmin(view_depth(a["position"])) >= 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("rocks", [0, 6])
def test_gpu_near_plane_clip_equals_the_oracle(rocks):
    """every buffer of three frames, bit for bit - the scene in LDS (rocks = 0: k_prepass<*, 1 / 2>) and beyond it (k_prepass<*, 0>
    under the suite's exact traversal, k_prepass<*, 4> - the wide walk - in the product default), and the clip really happened"""
    from cases import assert_rendered_within, product_default_plugin

    cam = camera()
    scene, ids = build(rocks=rocks)
    want = render(oracle(), scene, cam)
    got = render(hk.HikariPlugin(device=0), scene, cam)
    assert diff_buffers(got, want) == {}
    default = product_default_plugin()
    got_default = render(default, scene, cam)
    if rocks:
        assert default.engine.wide_walk()
    for name in ("position", "normal", "instance_material", "velocity_uv", "depth_gradient"):   # the G-buffer: the same hits, the same bits
        assert (got_default[name].view(np.uint8) == want[name].view(np.uint8)).all(), name
    assert_rendered_within(got_default, want, "product default")
    inst = np.floor(want["instance_material"][..., 0]).astype(int)
    geometry = want["position"][..., 3] > 0.0
    assert not (geometry & (inst == ids["near"])).any() and (geometry & (inst == ids["straddler"])).sum() > 50
