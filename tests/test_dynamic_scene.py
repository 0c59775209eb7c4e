"""Dynamic scenes: instance-only updates (prepare_instances re-runs whenever an instance changes,
instance.rs:352-437) and the previous-transform input of the G-buffer's velocity output
(PreviousMeshUniform, instance.rs:111-128; prepass.wgsl:50,96)."""
import ctypes as C

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.scenes import animate, synthetic_camera, synthetic_scene
from oracle_lib import oracle_plugin
from test_scene_builder import check_flat_bvh


def raw(arr):
    return bytes(memoryview(arr).cast("B")) if len(arr) else b""


def models(scene):
    return np.array([np.ctypeslib.as_array(i.model).copy() for i in scene.instances], dtype=np.float32)


def small_yard():
    return synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=3, sphere_rings=6, sphere_segs=8)


def test_refinish_redoes_only_the_instance_level():
    scene, _ = small_yard()
    rest = models(scene)
    s1 = animate(scene, 1, movers=(3, 9, 16, 19))
    s2 = animate(s1, 2, movers=(3, 9, 16, 19))
    # mesh-level buffers and materials are byte-identical; mesh records of the instances too
    for name in ("vertices", "primitives", "asset_nodes", "materials"):
        assert raw(getattr(s2, name)) == raw(getattr(scene, name)), name
    for a, b in zip(scene.instances, s2.instances):
        assert (a.mesh.vertex, a.mesh.primitive, a.mesh.node_offset, a.mesh.node_count, a.material) == \
               (b.mesh.vertex, b.mesh.primitive, b.mesh.node_offset, b.mesh.node_count, b.material)
    moved = sorted(np.nonzero((models(s2) != rest).any(axis=1))[0].tolist())
    assert moved == [3, 9, 16, 19]
    # previous transforms = the pose at the finish before
    assert (s1.previous_transforms == rest).all()
    assert (s2.previous_transforms == models(s1)).all()
    # the TLAS is a valid flat BVH over the NEW world boxes, the light BVH over the new emitter spheres
    boxes = [(np.array(list(i.min)), np.array(list(i.max))) for i in s2.instances]
    check_flat_bvh(list(s2.instance_nodes), len(s2.instances), boxes)
    em = {e.instance: e for e in s2.emissives}
    assert 19 in em                                  # instance 19 is an emitter in this scene
    e0 = {e.instance: e for e in scene.emissives}[19]
    assert tuple(em[19].position) != tuple(e0.position)
    assert abs(em[19].surface_area - e0.surface_area) < 1e-5 * e0.surface_area   # rigid motion keeps the area
    # inverse-transpose really is the inverse transpose of the new model
    m = np.ctypeslib.as_array(s2.instances[9].model).reshape(4, 4).T.astype(np.float64)
    it = np.ctypeslib.as_array(s2.instances[9].inverse_transpose_model).reshape(4, 4).T.astype(np.float64)
    assert np.allclose(it.T @ m, np.eye(4), atol=1e-5)


def test_static_refinish_is_idempotent():
    scene, _ = small_yard()
    again = scene.builder.finish()
    for name in hk.SceneData.FIELDS:
        assert raw(getattr(again, name)) == raw(getattr(scene, name)), name
    assert (again.previous_transforms == models(scene)).all()


def test_builder_rejects_unknown_instance():
    scene, _ = small_yard()
    with pytest.raises(hk.HikariError) as e:
        scene.builder.set_instance_transform(10_000, np.eye(4, dtype=np.float32))
    assert e.value.code == F.HK_E_INVALID


def test_oracle_velocity_of_moved_instances():
    """Static camera: velocity must be zero on everything that did not move, and for a moved
    instance equal uv_now - uv(previous_view_proj * previous_model * local position)."""
    scene, sun = small_yard()
    cam = synthetic_camera(96, 72)
    s = hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0, taa=hk.Taa.NONE)
    cpu = oracle_plugin()
    cpu.set_scene(scene)
    cpu.render(cam, s, frame_number=1)
    v0 = cpu.engine.read(F.BUF_VELOCITY_UV)
    assert (v0[..., :2] == 0).all()
    s1 = animate(scene, 1, movers=(3, 9, 16, 19))
    cpu.update_instances(s1)
    cpu.render(cam, s, frame_number=2)
    vel = cpu.engine.read(F.BUF_VELOCITY_UV)[..., :2]
    ids = np.floor(cpu.engine.read(F.BUF_INSTANCE_MATERIAL)[..., 0]).astype(np.int64)
    pos = cpu.engine.read(F.BUF_POSITION)
    geometry = pos[..., 3] > 0
    moved = np.isin(ids, [3, 9, 16, 19]) & geometry
    assert moved.sum() > 20, "test scene must show the movers"
    assert (vel[~moved] == 0).all()
    assert (np.abs(vel[moved]).max(axis=1) > 0).mean() > 0.95
    # independent evaluation in float64
    view_proj = np.ctypeslib.as_array(cam.view_uniform().view_proj).reshape(4, 4).T.astype(np.float64)
    ys, xs = np.nonzero(moved)
    worst = 0.0
    for y, x in list(zip(ys, xs))[::7]:
        i = ids[y, x]
        m_now = np.ctypeslib.as_array(s1.instances[i].model).reshape(4, 4).T.astype(np.float64)
        m_prev = s1.previous_transforms[i].reshape(4, 4).T.astype(np.float64)
        w = np.append(pos[y, x, :3].astype(np.float64), 1.0)
        wp = m_prev @ np.linalg.solve(m_now, w)

        def uv(p):
            c = view_proj @ p
            return np.array([0.5 + 0.5 * c[0] / c[3], 0.5 - 0.5 * c[1] / c[3]])

        worst = max(worst, float(np.abs((uv(w) - uv(wp)) - vel[y, x]).max()))
    assert worst < 2e-5, worst


def test_previous_transforms_must_match_instances():
    scene, _ = small_yard()
    cpu = oracle_plugin()
    cpu.set_scene(scene)
    with pytest.raises(hk.HikariError) as e:
        cpu.engine.api.call("upload_previous_transforms", cpu.engine.ctx, np.zeros(32, np.float32).ctypes.data_as(C.POINTER(F.f32)), 2)
    assert e.value.code == F.HK_E_INVALID
