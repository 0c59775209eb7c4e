"""Instance motion on the device (hk_refit_scene_instances): the GPU redoes the per-instance / per-emitter work
of prepare_instances (instance.rs:286-420) and REFITS both trees.  The oracle is fed exactly what that must produce - the host
builder's per-instance records for the new poses, on the OLD tree shapes with every inner box re-derived as the union of the
leaves below it (numpy, here) - and every buffer of every frame must agree bit for bit."""
import ctypes as C
import math

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.plugin import SceneData
from bevy_hikari_amd.scenes import synthetic_camera, synthetic_scene
from cases import diff_buffers, product_default_traversal, snapshot

pytestmark = pytest.mark.gpu
LEAF = 0x80000000


def oracle():
    from oracle_lib import oracle_plugin

    return oracle_plugin()


def refit_nodes(nodes, boxes):
    """`nodes` (HkNode ctypes array, bvh 0.7.1 flatten_custom layout) with every navigator's box = union of the shape boxes in its
    subtree (i, exit); leaves keep the empty box the reference stores.  boxes: float32[n_shapes][2][3]."""
    out = (F.HkNode * len(nodes))()
    C.memmove(out, nodes, C.sizeof(out))
    entry = np.array([n.entry_index for n in nodes], dtype=np.uint32)
    for i, n in enumerate(nodes):
        if entry[i] >= LEAF:
            continue
        leaves = [int(entry[j] - LEAF) for j in range(i + 1, min(n.exit_index, len(nodes))) if entry[j] >= LEAF]
        mn, mx = boxes[leaves, 0].min(axis=0), boxes[leaves, 1].max(axis=0)
        for k in range(3):
            out[i].min[k], out[i].max[k] = float(mn[k]), float(mx[k])
    return out


def pose(rest, frame, k):
    m = rest.reshape(4, 4).T.astype(np.float64)
    ang = 0.07 * frame * (1 + k)
    c, s = math.cos(ang), math.sin(ang)
    rot = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)
    shift = np.eye(4)
    shift[0, 3] = 0.05 * frame * (1 if k % 2 == 0 else -1)
    shift[1, 3] = 0.02 * frame * (k % 3 == 0)
    return (shift @ m @ rot).T.astype(np.float32).reshape(-1)


def check_tree(nodes, n_shapes, boxes):
    """A well-formed flatten_custom array over n_shapes: every shape in exactly one leaf, 3n - 2 nodes, every navigator's box the
    union of the leaves in its subtree, subtree sizes consistent with the exit links."""
    assert len(nodes) == 3 * n_shapes - 2
    entry = np.array([n.entry_index for n in nodes], dtype=np.int64)
    exit_ = np.array([n.exit_index for n in nodes], dtype=np.int64)
    leaves = entry[entry >= LEAF] - LEAF
    assert sorted(leaves.tolist()) == list(range(n_shapes))
    for i in range(len(nodes)):
        if entry[i] >= LEAF:
            assert exit_[i] == i + 1
        else:
            assert entry[i] == i + 1 and i + 1 < exit_[i] <= len(nodes)
            k = int(((entry[i + 1:exit_[i]] >= LEAF)).sum())
            assert exit_[i] - (i + 1) == 3 * k - 2, "a navigator spans exactly one subtree"
    again = refit_nodes(nodes, boxes)
    for a, b in zip(again, nodes):
        if b.entry_index < LEAF:
            assert list(a.min) == list(b.min) and list(a.max) == list(b.max)


def same_links(a, b):
    return len(a) == len(b) and all(x.entry_index == y.entry_index and x.exit_index == y.exit_index for x, y in zip(a, b))


def run_refit_sequence(kw, size, movers_of_frame, flags=F.CTX_DETERMINISTIC_SCATTER, frames=5, settings=None, rebuild_on=(), rebuild_mode=F.TREE_LBVH,
                       camera=None):
    """GPU: one upload, then hk_refit_scene_instances per frame.  Oracle: the expected arrays per frame (see the module docstring).
    Moving objects make the reference's scatter-store race observable: it is resolved the oracle's way here.
    kw: keyword arguments of synthetic_scene, or a callable that makes (scene with its builder, sun or None)."""
    make = kw if callable(kw) else (lambda: synthetic_scene(**kw))
    dev_scene, sun = make()     # its builder feeds the device refit
    ref_scene, _ = make()       # a twin builder produces the host records for the same poses
    s = settings or hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam, lights = camera or synthetic_camera(*size), (hk.lights_uniform(directional=sun) if sun else hk.lights_uniform())
    gpu, cpu = hk.HikariPlugin(device=0, flags=flags), oracle()
    gpu.set_scene(dev_scene)
    cpu.set_scene(ref_scene)
    rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in ref_scene.instances], dtype=np.float32)
    current = rest.copy()
    builds = None
    topo_tlas, topo_light = ref_scene.instance_nodes, ref_scene.emissive_nodes  # the tree shapes the device holds
    for n in range(1, frames + 1):
        if n > 1:
            movers = movers_of_frame(n)
            previous = current.copy()
            for k, i in enumerate(movers):
                current[i] = pose(rest[i], n - 1, k)
                dev_scene.builder.set_instance_transform(i, current[i])
                ref_scene.builder.set_instance_transform(i, current[i])
            assert gpu.engine.refit_instances(dev_scene.builder) == len(movers)
            new = ref_scene.builder.finish()
            boxes = np.array([[list(i.min), list(i.max)] for i in new.instances], dtype=np.float32)
            eboxes = np.array([[[e.position[k] - e.radius for k in range(3)], [e.position[k] + e.radius for k in range(3)]] for e in new.emissives], dtype=np.float32)
            if n in rebuild_on:  # hk_rebuild_scene_trees: new tree shapes, built on the device; the oracle gets exactly those
                mode = rebuild_mode(n) if callable(rebuild_mode) else rebuild_mode
                gpu.engine.rebuild_trees(mode)
                topo_tlas, topo_light = gpu.engine.read_trees(len(topo_tlas), len(topo_light))
                check_tree(topo_tlas, len(new.instances), boxes)
                if len(new.emissives):
                    check_tree(topo_light, len(new.emissives), eboxes)
                if mode == F.TREE_SAH:  # the device ran the reference's own build: the host builder's tree for these poses, link for link
                    assert same_links(topo_tlas, new.instance_nodes), "instance tree differs from the host's bvh 0.7.1 build"
                    assert same_links(topo_light, new.emissive_nodes), "light tree differs from the host's bvh 0.7.1 build"
            expected = SceneData(previous_transforms=previous, vertices=ref_scene.vertices, primitives=ref_scene.primitives, asset_nodes=ref_scene.asset_nodes,
                                 materials=ref_scene.materials, instances=new.instances, instance_nodes=refit_nodes(topo_tlas, boxes),
                                 emissives=new.emissives, emissive_nodes=refit_nodes(topo_light, eboxes) if len(new.emissives) else new.emissive_nodes,
                                 alias_table=new.alias_table)
            cpu.update_instances(expected)
        for p in (gpu, cpu):
            p.render(cam, s, lights=lights, frame_number=n)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"frame {n}: {bad}"
        if n == 1:
            builds = gpu.engine.stats().scene_instance_builds  # the upload
    st = gpu.engine.stats()
    assert st.scene_device_refits == frames - 1 and st.scene_instance_builds == builds, "the instance-level arrays must not have been rebuilt on the host"
    assert st.scene_device_tree_builds == len([n for n in rebuild_on if 1 < n <= frames])
    return gpu


SMALL = dict(n_boxes=3, n_spheres=1, n_emitters=1, sphere_rings=4, sphere_segs=5)    # fits the LDS copy: one slot, refit in place
LARGE = dict(n_boxes=20, n_spheres=5, n_emitters=3, sphere_rings=12, sphere_segs=16)  # two slots: refit in the spare one


def test_refit_small_scene_vs_oracle():
    n = 1 + 3 + 1 + 1
    run_refit_sequence(SMALL, (88, 60), lambda f: [1, 4, n - 1])  # a box, the sphere and the emitter move every frame


def test_refit_large_scene_vs_oracle():
    n = 1 + 20 + 5 + 3
    movers = [2, 7, 11, 22, n - 1, n - 3]  # boxes, a sphere, two of the three emitters
    run_refit_sequence(LARGE, (120, 72), lambda f: movers if f % 2 == 0 else movers[:3] + [14])  # some rest every other frame: their `moved` flag goes


def test_refit_of_a_scene_whose_instances_shared_one_transform():
    """The Cornell box: all eight instances carry the same model matrix at upload, so a traversal transforms its ray once
    (DScene::shared_xform) - and the scene sits in ONE slot and is refit in place.  Moving one instance on the device has to
    switch the shortcut off for the frames that follow (it did not, until the end of round 2: found by reading, not by a test)."""
    scenes = lambda: (hk.load_cornell(), None)
    first = scenes()[0]
    models = {bytes(np.ctypeslib.as_array(i.model).tobytes()) for i in first.instances}
    assert len(models) == 1, "the premise of this test: one shared transform"
    run_refit_sequence(scenes, (96, 64), lambda f: [len(first.instances) - 2] if f < 4 else [1, len(first.instances) - 2], camera=hk.cornell_camera(96, 64), frames=5)


def test_refit_with_three_bounces_and_aa_tail():
    s = hk.HikariSettings(indirect_bounces=3, emissive_spatial_reuse=True, upscale=hk.Upscale.SmaaTu4x(1.5))
    run_refit_sequence(LARGE, (96, 64), lambda f: [3, 9, 26], settings=s, frames=4)


def test_device_rebuilt_trees_vs_oracle():
    """hk_rebuild_scene_trees (LBVH on the device) in the middle of a refit sequence: the trees are read back, checked for
    well-formedness and handed to the oracle - every buffer of every frame bit for bit, before and after the rebuilds."""
    n = 1 + 20 + 5 + 3
    run_refit_sequence(LARGE, (120, 72), lambda f: [2, 7, 11, 22, n - 1, n - 3], frames=6, rebuild_on=(3, 5))
    m = 1 + 3 + 1 + 1
    run_refit_sequence(SMALL, (88, 60), lambda f: [1, 4, m - 1], frames=4, rebuild_on=(2, 3))


def test_device_built_sah_trees_are_the_host_builders_trees():
    """HK_TREE_SAH: `bvh` 0.7.1's binned-SAH build run on the device must return the tree the host builder returns for the same poses
    (entry / exit links equal, node for node), and the frames must equal the oracle's on those trees."""
    n = 1 + 20 + 5 + 3
    run_refit_sequence(LARGE, (120, 72), lambda f: [2, 7, 11, 22, n - 1, n - 3], frames=6, rebuild_on=(2, 4, 5), rebuild_mode=lambda f: F.TREE_SAH if f != 4 else F.TREE_LBVH)
    m = 1 + 3 + 1 + 1
    run_refit_sequence(SMALL, (88, 60), lambda f: [1, 4, m - 1], frames=4, rebuild_on=(2, 3), rebuild_mode=F.TREE_SAH)


@pytest.mark.parametrize("n_instances,n_emitters,n_movers", [(2000, 8, 300), (20000, 1500, 3000)])
def test_device_sah_build_at_scale_matches_the_host_builder(n_instances, n_emitters, n_movers):
    """2 009 / 21 501 instances (few small meshes; the larger one with a 1 500-leaf light tree, so both trees go through the
    one-workgroup top AND the per-subtree kernel): one refit, then the SAH rebuild on the device against the host's finish()
    for the same poses."""
    from bevy_hikari_amd.scenes import synthetic_large

    scene, sun = synthetic_large(0x5EED0004, 20, 16, 32, n_instances, 50, n_emitters, 40.0)
    twin, _ = synthetic_large(0x5EED0004, 20, 16, 32, n_instances, 50, n_emitters, 40.0)
    p = hk.HikariPlugin(device=0, flags=F.CTX_EXACT_TRAVERSAL)
    p.set_scene(scene)
    cam, s = synthetic_camera(96, 64, extent=30.0), hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)
    p.render(cam, s, lights=hk.lights_uniform(directional=sun), frame_number=1)
    rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in scene.instances], dtype=np.float32)
    movers = np.random.default_rng(3).choice(len(rest), size=n_movers, replace=False)
    for k, i in enumerate(movers):
        t = pose(rest[i], 5, k)
        scene.builder.set_instance_transform(int(i), t)
        twin.builder.set_instance_transform(int(i), t)
    assert p.engine.refit_instances(scene.builder) == len(movers)
    p.engine.rebuild_trees(F.TREE_SAH)
    new = twin.builder.finish()
    tlas, light = p.engine.read_trees(len(new.instance_nodes), len(new.emissive_nodes))
    assert same_links(tlas, new.instance_nodes) and same_links(light, new.emissive_nodes)
    boxes = np.array([[list(i.min), list(i.max)] for i in new.instances], dtype=np.float32)
    check_tree(tlas, len(new.instances), boxes)


def test_device_rebuild_of_all_orderings_stays_within_tolerance():
    """Product defaults: the rebuild writes all eight direction-threaded orderings; against the reference-order context."""
    kw, size = LARGE, (160, 96)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    outs = []
    for exact in (True, False):
        scene, sun = synthetic_scene(**kw)
        cam, lights = synthetic_camera(*size), hk.lights_uniform(directional=sun)
        if exact:
            p = hk.HikariPlugin(device=0, flags=F.CTX_EXACT_TRAVERSAL)
        else:
            with product_default_traversal():
                p = hk.HikariPlugin(device=0)
        p.set_scene(scene)
        rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in scene.instances], dtype=np.float32)
        for n in range(1, 6):
            if n > 1:
                for k, i in enumerate((2, 7, 22, 28)):
                    scene.builder.set_instance_transform(i, pose(rest[i], n - 1, k))
                assert p.engine.refit_instances(scene.builder) == 4
                if n == 3:
                    p.engine.rebuild_trees(F.TREE_LBVH)
                if n == 4:
                    p.engine.rebuild_trees(F.TREE_SAH)
            p.render(cam, s, lights=lights, frame_number=n)
        assert p.engine.wide_walk() == (not exact)  # (the wide records are derived again from the refit trees: hk_wide.hpp)
        outs.append((p.output(s), snapshot(p)))
    (a, sa), (b, sb) = outs
    assert float(np.linalg.norm(a - b) / np.linalg.norm(a)) <= 1e-3
    assert (sa["position"] == sb["position"]).mean() > 0.999


@pytest.mark.parametrize("seed", range(8))
def test_refit_and_rebuild_random_sequences_vs_oracle(seed):
    """Seeded scenes (one-slot and two-slot), movers (always including emitters when there are any), settings and rebuild frames."""
    rng = np.random.default_rng(4200 + seed)
    big = bool(rng.random() < 0.5)
    kw = dict(n_boxes=int(rng.integers(2, 24)), n_spheres=int(rng.integers(0, 5)), n_emitters=int(rng.integers(0, 4)), sphere_rings=12 if big else 4,
              sphere_segs=16 if big else 5, seed=int(rng.integers(1, 1 << 30)))
    n = 1 + kw["n_boxes"] + kw["n_spheres"] + kw["n_emitters"]
    movers = sorted(set(int(i) for i in rng.choice(n, size=min(n, int(rng.integers(1, 7))), replace=False)) | ({n - 1} if kw["n_emitters"] else set()))
    s = hk.HikariSettings(indirect_bounces=int(rng.integers(0, 4)), emissive_spatial_reuse=bool(rng.random() < 0.5), denoise=bool(rng.random() < 0.7),
                          upscale=hk.Upscale.SmaaTu4x(float(rng.choice([1.0, 1.5, 2.0]))))
    rebuild_on = tuple(int(f) for f in range(2, 6) if rng.random() < 0.4)
    modes = {f: (F.TREE_SAH if rng.random() < 0.6 else F.TREE_LBVH) for f in range(2, 6)}
    run_refit_sequence(kw, (int(rng.integers(48, 130)), int(rng.integers(40, 90))), lambda f: movers if f % 3 else movers[:max(1, len(movers) // 2)], settings=s,
                       rebuild_on=rebuild_on, rebuild_mode=lambda f: modes[f])


def test_refit_with_direction_threaded_orderings_stays_within_tolerance():
    """Product defaults (eight orderings of the instance tree, all refit): against the reference-order refit of the same sequence."""
    kw, size = LARGE, (160, 96)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    outs = []
    for exact in (True, False):
        scene, sun = synthetic_scene(**kw)
        cam, lights = synthetic_camera(*size), hk.lights_uniform(directional=sun)
        if exact:
            p = hk.HikariPlugin(device=0, flags=F.CTX_EXACT_TRAVERSAL)
        else:
            with product_default_traversal():
                p = hk.HikariPlugin(device=0)
        p.set_scene(scene)
        rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in scene.instances], dtype=np.float32)
        for n in range(1, 6):
            if n > 1:
                for k, i in enumerate((2, 7, 22, 28)):
                    scene.builder.set_instance_transform(i, pose(rest[i], n - 1, k))
                assert p.engine.refit_instances(scene.builder) == 4
            p.render(cam, s, lights=lights, frame_number=n)
        outs.append((p.output(s), snapshot(p)))
    (a, sa), (b, sb) = outs
    assert float(np.linalg.norm(a - b) / np.linalg.norm(a)) <= 1e-3
    assert (sa["position"] == sb["position"]).mean() > 0.999  # the G-buffer: equal up to exact ties between two triangles


def test_refit_refuses_what_it_cannot_do():
    scene, _ = synthetic_scene(**SMALL)
    gpu = hk.HikariPlugin(device=0)
    gpu.set_scene(scene)
    other, _ = synthetic_scene(n_boxes=5, n_spheres=1, n_emitters=1, sphere_rings=4, sphere_segs=5)
    with pytest.raises(hk.HikariError):   # another instance count
        gpu.engine.refit_instances(other.builder)
    scene.builder.set_instance_transform(1, np.zeros(16, np.float32))
    with pytest.raises(hk.HikariError):   # singular transform
        gpu.engine.refit_instances(scene.builder)


@pytest.mark.parametrize("rings,segs", [(24, 40), (40, 48)])
def test_refit_of_a_moving_emissive_sphere(rings, segs):
    """examples/scene.rs:231-235: a rotating emissive SPHERE - an emitter whose area sum and alias table run over ~1 900 (LDS work
    arrays) / ~3 700 (global work arrays) triangles.  k_refit_emitters computes them with one wave per emitter; the oracle is fed
    the host builder's records for the same poses: every buffer of every frame bit for bit."""
    kw = dict(n_boxes=3, n_spheres=1, n_emitters=1, sphere_rings=rings, sphere_segs=segs, n_emissive_spheres=1)
    n = 1 + 3 + 1 + 1 + 1
    scene, _ = synthetic_scene(**kw)
    assert scene.emissives[-1].alias_table[1] >= (1800 if rings == 24 else 3300), "the premise: a many-triangle emitter"
    run_refit_sequence(kw, (96, 64), lambda f: [n - 1, 2] if f % 2 else [n - 1, n - 2], frames=4)


def _edit_sequence(kw, size, edits, settings=None, tree_mode=F.TREE_SAH):
    """edits: {frame: callable(builder, rng)} applied to BOTH builders before that frame.  Device: hk_update_scene_instances (host
    lays out the records, the device builds both trees).  Oracle: the twin builder's full finish() - the reference's path."""
    dev_scene, sun = synthetic_scene(**kw)
    ref_scene, _ = synthetic_scene(**kw)
    s = settings or hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam, lights = synthetic_camera(*size), hk.lights_uniform(directional=sun)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_DETERMINISTIC_SCATTER), oracle()
    gpu.set_scene(dev_scene)
    cpu.set_scene(ref_scene)
    builds = 0
    for n in range(1, max(edits) + 2):
        if n in edits:
            for b in (dev_scene.builder, ref_scene.builder):
                edits[n](b)
            gpu.engine.update_instances_on_device(dev_scene.builder, tree_mode)
            builds += 1
            new = ref_scene.builder.finish()
            cpu.update_instances(new)
            tlas, light = gpu.engine.read_trees(len(new.instance_nodes), len(new.emissive_nodes))
            boxes = np.array([[list(i.min), list(i.max)] for i in new.instances], dtype=np.float32)
            check_tree(tlas, len(new.instances), boxes)
            if tree_mode == F.TREE_SAH:   # the device ran the reference's own build: the host builder's tree, link for link
                assert same_links(tlas, new.instance_nodes), "instance tree differs from the host's bvh 0.7.1 build"
                assert same_links(light, new.emissive_nodes), "light tree differs from the host's bvh 0.7.1 build"
        for p in (gpu, cpu):
            p.render(cam, s, lights=lights, frame_number=n)
        if tree_mode == F.TREE_SAH:
            bad = diff_buffers(snapshot(gpu), snapshot(cpu))
            assert bad == {}, f"frame {n}: {bad}"
    assert gpu.engine.stats().scene_device_tree_builds == builds
    return gpu


def test_instances_added_removed_and_rematerialed_with_device_built_trees():
    """VERDICT r02 missing 3: the reference re-runs prepare_instances on ANY instance change (instance.rs:352-437).  Here: add two
    instances (one of them an emitter), remove one, give one another material (non-emissive -> emissive: the emitter list grows),
    move some - and after every edit hk_update_scene_instances; the trees the device builds are the host builder's (links
    compared), every buffer of every frame equals the oracle fed the host builder's full finish()."""
    kw = dict(n_boxes=6, n_spheres=2, n_emitters=2, sphere_rings=6, sphere_segs=8)

    def add_two(b):
        # (mesh ids in synthetic_scene: 0 box, 1 sphere, 2 quad strip; material ids 0..7 plain, 8.. emissive)
        b.add_instance(0, 3, _pose_matrix((0.7, 0.9, -0.4), 0.3, (0.5, 0.8, 0.4)))
        b.add_instance(2, 9, _pose_matrix((-0.8, 2.6, 0.6), math.pi, (0.6, 1.0, 0.5)))

    def remove_one(b):
        b.remove_instance(4)

    def rematerial_and_move(b):
        b.set_instance_material(2, 8)     # a box becomes an emitter
        b.set_instance_transform(5, _pose_matrix((1.2, 0.6, 1.1), -0.4, (0.7, 0.7, 0.7)))

    _edit_sequence(kw, (104, 72), {2: add_two, 3: remove_one, 4: rematerial_and_move})


def test_instance_edits_on_a_two_slot_scene_and_lbvh_trees():
    """... on a scene beyond the LDS copy (two slots: the records go to the spare slot in stream order), SAH trees compared link for
    link; and with HK_TREE_LBVH (another valid tree over the same instances: checked structurally, frames finite and lit)."""
    kw = dict(n_boxes=20, n_spheres=5, n_emitters=3, sphere_rings=12, sphere_segs=16)

    def grow(b):
        for k in range(5):
            b.add_instance(k % 2, 1 + k, _pose_matrix((-2.0 + k, 0.8, 2.0 - 0.7 * k), 0.2 * k, (0.5, 0.6, 0.5)))

    def shrink(b):
        for i in (27, 11, 3):
            b.remove_instance(i)

    _edit_sequence(kw, (120, 72), {2: grow, 4: shrink})
    gpu = _edit_sequence(kw, (120, 72), {2: grow, 3: shrink}, tree_mode=F.TREE_LBVH)
    out = gpu.output(hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0))
    assert np.isfinite(out).all() and out[..., :3].max() > 0.05


def test_stand_in_trees_only_reach_the_device_through_the_update_that_rebuilds_them():
    """hk_scene_builder_finish_instances leaves valid but NON-reference trees in the builder (for the device build of
    hk_update_scene_instances).  The plain upload paths refuse such a builder - frames from stand-in trees would differ silently in
    tie-breaks and visit order - and hk_update_scene_instances, which does take it, leaves the device with the reference's trees."""
    from bevy_hikari_amd.scenes import synthetic_scene

    scene, _ = synthetic_scene(n_boxes=9, n_spheres=2, n_emitters=3, sphere_rings=4, sphere_segs=5)
    e = hk.Engine(device=0)
    e.upload_noise(); e.upload_scene(scene); e.resize(64, 48, 1.0)
    b = scene.builder
    b.add_instance(0, 1, _pose_matrix((0.3, 0.9, -0.4), 0.4, (0.5, 0.5, 0.5)))
    b.finish(build_trees=False)
    for call in ("upload_scene", "upload_scene_instances"):
        with pytest.raises(hk.HikariError) as err:
            e.api.call(call, e.ctx, b.h)
        assert err.value.code == F.HK_E_NOT_READY and "stand-in" in str(err.value)
    e.update_instances_on_device(b)             # finishes again (stand-ins), uploads, builds both trees on the device
    full = b.finish()                           # the reference's trees for the same instances
    got = e.read_trees(len(full.instance_nodes), len(full.emissive_nodes))
    assert same_links(got[0], full.instance_nodes) and same_links(got[1], full.emissive_nodes)
    e.api.call("upload_scene_instances", e.ctx, b.h)   # a fully finished builder is welcome again

def test_a_mesh_first_instanced_by_a_later_edit_gets_its_wide_records():
    """The wide walk's mesh-tree records are derived per mesh an instance uses.  A mesh that is uploaded but not instanced (the quad
    strip of a yard without emitters) has none - until an instance edit puts one in the scene: the update must derive them then, not
    only after a mesh-level upload.  Product default traversal (wide walk + queue-based indirect pass) against the oracle, every frame."""
    from cases import assert_rendered_within

    kw = dict(n_boxes=20, n_spheres=5, n_emitters=0, sphere_rings=12, sphere_segs=16)   # beyond the LDS copy; mesh 2 (the quad strip) unused
    dev_scene, sun = synthetic_scene(**kw)
    ref_scene, _ = synthetic_scene(**kw)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam, lights = synthetic_camera(160, 96), hk.lights_uniform(directional=sun)
    with product_default_traversal():
        gpu = hk.HikariPlugin(device=0, flags=F.CTX_DETERMINISTIC_SCATTER)
    cpu = oracle()
    gpu.set_scene(dev_scene)
    cpu.set_scene(ref_scene)
    for n in (1, 2, 3, 4):
        if n == 3:   # a big quad across the yard, a metre up: most rays now meet the mesh that had no instance
            for b in (dev_scene.builder, ref_scene.builder):
                b.add_instance(2, 3, _pose_matrix((0.0, 1.0, 0.0), 0.4, (3.0, 1.0, 3.0)))
            gpu.engine.update_instances_on_device(dev_scene.builder, F.TREE_SAH)
            cpu.update_instances(ref_scene.builder.finish())
        for p in (gpu, cpu):
            p.render(cam, s, lights=lights, frame_number=n)
        assert_rendered_within(snapshot(gpu), snapshot(cpu), f"frame {n}")
    assert gpu.engine.wide_walk() and gpu.engine.indirect_schedule() == "wavefront" and gpu.engine.stats().wide_stack_lost == 0
    quad = len(ref_scene.builder.finish().instances) - 1   # (the plane holds instance + 0.5; 0 = background)
    seen = [float((p.engine.read(F.BUF_INSTANCE_MATERIAL)[..., 0] == quad + 0.5).mean()) for p in (gpu, cpu)]
    assert seen[0] == seen[1] and seen[1] > 0.005, seen   # ... and primary rays see it, the same pixels as the oracle's



def _pose_matrix(t, yaw, scale):
    c, s = math.cos(yaw), math.sin(yaw)
    m = np.array([[c * scale[0], 0, s * scale[2], t[0]], [0, scale[1], 0, t[1]], [-s * scale[0], 0, c * scale[2], t[2]], [0, 0, 0, 1]], dtype=np.float64)
    return m.T.astype(np.float32).reshape(-1)
