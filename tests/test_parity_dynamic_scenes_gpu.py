"""Parity while the scene changes between frames: instance transforms, instance / mesh growth, updates while frames are in flight
(mesh_material/instance.rs + light.rs prepare systems on the reference's side; device refit / rebuild here).  Split from
test_parity_gpu.py."""

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import GBUFFER, diff_buffers, oracle, snapshot

pytestmark = pytest.mark.gpu


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
