"""hk_history_rows_bound (include/hikari_hip.h): the history halo a band-sharded frame needs is DERIVED from the
frame's own uniforms and the scene, not supplied by the host.  The bound must dominate what the kernels do: for every pixel whose
reprojection lands on screen (the temporal dispatches neither load nor store otherwise, light.wgsl:1091-1095), the row of the
reprojected pixel is at most `rows` away from the pixel's own row.  Checked here against the oracle's own G-buffer (velocity plane)
for random camera pairs, an orthographic camera and moving instances - pure host logic against a CPU render, no GPU."""
import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.distributed import history_rows_bound


def _measured_rows(e, w, h):
    """max |row of the reprojected pixel - row| over the geometry pixels whose reprojection is on screen (ratio 1: uv = pixel centre)"""
    vel, pos = e.read(F.BUF_VELOCITY_UV), e.read(F.BUF_POSITION)
    geo = pos[..., 3] > 1.1920929e-7
    ys, xs = (np.arange(h, dtype=np.float32)[:, None] + 0.5) / h, (np.arange(w, dtype=np.float32)[None, :] + 0.5) / w
    py, px = ys - vel[..., 1], xs - vel[..., 0]
    on = geo & (np.abs(py - 0.5) <= 0.5) & (np.abs(px - 0.5) <= 0.5)
    if not on.any():
        return 0.0
    rows = np.floor(py * h) - np.arange(h)[:, None]
    return float(np.abs(rows[on]).max())


def _oracle(scene, w, h):
    from oracle_lib import oracle_engine

    e = oracle_engine()
    e.upload_noise(); e.upload_scene(scene); e.resize(w, h, 1.0)
    return e


def test_static_view_needs_no_halo_and_bad_arguments_fail():
    cam = hk.cornell_camera(96, 64)
    v, pv = cam.view_uniform(), cam.previous_view_uniform()
    assert history_rows_bound(v, pv, 64, (-1, 0, -1), (1, 2, 1)) == 0
    assert history_rows_bound(v, pv, 64, (1, 0, 0), (-1, 2, 1)) == 0          # an empty scene
    other = hk.Camera(hk.look_at_transform((0.0, 1.3, 4.0), (0.0, 1.0, 0.0)), 96, 64)
    rows = history_rows_bound(v, cam.previous_view_uniform(other), 64, (-1, 0, -1), (1, 2, 1))
    assert 4 <= rows <= 64 and rows % 4 == 0
    with pytest.raises(F.HikariError):
        history_rows_bound(v, pv, 0, (-1, 0, -1), (1, 2, 1))


@pytest.mark.parametrize("seed", range(6))
def test_bound_dominates_the_reprojection_of_every_pixel_under_camera_motion(seed):
    rng = np.random.default_rng(40 + seed)
    w, h = 96, 64
    scene = hk.load_cornell()
    e = _oracle(scene, w, h)
    mn, mx = e.scene_bounds()
    s = hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)
    prev = None
    slack = []
    for n in range(1, 7):
        eye = (rng.uniform(-0.7, 0.7), rng.uniform(0.3, 1.7), rng.uniform(2.2, 4.6))
        tgt = (rng.uniform(-0.4, 0.4), rng.uniform(0.6, 1.4), rng.uniform(-0.3, 0.3))
        if seed == 5 and n >= 3:   # straight back and forth along the view axis: the previous camera's plane approaches the near geometry
            eye, tgt = (0.0, 1.0, 3.0 + 0.4 * (n % 2)), (0.0, 1.0, 0.0)
        cam = hk.Camera(hk.look_at_transform(eye, tgt), w, h, ortho_height=(3.0 if seed == 4 else None))
        pv = cam.previous_view_uniform(prev if prev is not None else cam)
        e.frame_begin(hk.frame_uniform(s, n), cam.view_uniform(), pv, hk.lights_uniform())
        e.pass_run(F.PASS_PREPASS)
        measured = _measured_rows(e, w, h)
        bound = history_rows_bound(cam.view_uniform(), pv, h, mn, mx)
        assert bound >= measured, (n, measured, bound)
        if n > 1:
            slack.append(bound - measured)
        prev = cam
    assert min(slack) <= 16, slack   # (not vacuous: the bound is within a few rows of what some frame really needs, on a 64-row image)


def test_bound_covers_instances_that_move_on_their_own():
    """The camera rests; four instances move (their previous model differs): the bound comes from their boxes alone
    (HkMovedBox: previous_model x model^-1), and it dominates the velocity the prepass writes for them (prepass.wgsl:50,94-95)."""
    from bevy_hikari_amd.scenes import animate, synthetic_camera, synthetic_scene

    w, h = 128, 96
    scene, sun = synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=3, sphere_rings=6, sphere_segs=8)
    cam = synthetic_camera(w, h)
    e = _oracle(scene, w, h)
    s = hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)
    v, pv = cam.view_uniform(), cam.previous_view_uniform()
    movers = (3, 9, 16, 19)
    for n in range(1, 5):
        scene = animate(scene, 4 * n, movers=movers)   # big steps: several rows per frame
        e.upload_instances(scene)
        e.frame_begin(hk.frame_uniform(s, n), v, pv, hk.lights_uniform(directional=sun))
        e.pass_run(F.PASS_PREPASS)
        measured = _measured_rows(e, w, h)
        moved = []
        for i in movers:
            inst = scene.instances[i]
            model = np.ctypeslib.as_array(inst.model).reshape(4, 4).T.astype(np.float64)
            prev = np.asarray(scene.previous_transforms[i], dtype=np.float64).reshape(4, 4).T
            b = F.HkMovedBox()
            b.min[:], b.max[:] = list(inst.min), list(inst.max)
            b.previous_from_current[:] = list((prev @ np.linalg.inv(model)).T.astype(np.float32).reshape(-1))
            moved.append(b)
        mn, mx = e.scene_bounds()
        bound = history_rows_bound(v, pv, h, mn, mx, moved)
        assert measured > 0 and bound >= measured, (n, measured, bound)
        assert history_rows_bound(v, pv, h, mn, mx) == 0   # (without the moved boxes a resting camera needs nothing)
