"""Host builders (hk_scene_builder_*): flat-BVH format invariants, brute-force equivalence of the
traversal, alias tables, strip winding, instance AABBs."""
import numpy as np

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.scenes import synthetic_scene
from oracle_lib import oracle_engine, oracle_api

import ctypes as C

LEAF = 0x80000000


def check_flat_bvh(nodes, n_shapes, shape_boxes=None):
    """`bvh` 0.7.1 flatten_custom format: 3n-2 nodes, depth-first, every shape in exactly one leaf,
    exit > own index, leaf boxes empty, navigator boxes bound their subtree."""
    n = len(nodes)
    assert n == 3 * n_shapes - 2
    seen = []
    for i, nd in enumerate(nodes):
        assert nd.exit_index > i and nd.exit_index <= n
        if nd.entry_index >= LEAF:
            seen.append(nd.entry_index - LEAF)
            assert nd.exit_index == i + 1
            assert nd.min[0] == np.inf and nd.max[0] == -np.inf
        else:
            assert nd.entry_index == i + 1
            if shape_boxes is not None:
                sub = [nodes[k].entry_index - LEAF for k in range(i + 1, nd.exit_index) if nodes[k].entry_index >= LEAF]
                assert sub, "navigator with empty subtree"
                lo = np.min([shape_boxes[s][0] for s in sub], axis=0)
                hi = np.max([shape_boxes[s][1] for s in sub], axis=0)
                assert np.allclose(lo, list(nd.min)) and np.allclose(hi, list(nd.max))
    assert sorted(seen) == list(range(n_shapes))
    # skip-link walk with "always descend" visits every leaf exactly once, in order
    i, visited = 0, []
    while i < n:
        if nodes[i].entry_index >= LEAF:
            visited.append(nodes[i].entry_index - LEAF)
            i = nodes[i].exit_index
        else:
            i = nodes[i].entry_index
    assert visited == seen


def test_cornell_counts_and_format(cornell):
    s = cornell
    assert (len(s.vertices), len(s.primitives), len(s.asset_nodes), len(s.instances), len(s.instance_nodes)) == (78, 32, 80, 8, 22)
    assert (len(s.emissives), len(s.emissive_nodes), len(s.alias_table)) == (1, 1, 2)
    boxes = [(np.array(list(i.min)), np.array(list(i.max))) for i in s.instances]
    check_flat_bvh(list(s.instance_nodes), 8, boxes)
    for inst in s.instances:
        m = inst.mesh
        nodes = list(s.asset_nodes)[m.node_offset:m.node_offset + m.node_count]
        nprim = (m.node_count + 2) // 3
        prim_boxes = []
        for k in range(nprim):
            p = np.array([list(v.position) for v in s.primitives[m.primitive + k].vertices])
            prim_boxes.append((p.min(0), p.max(0)))
        check_flat_bvh(nodes, nprim, prim_boxes)
    e = s.emissives[0]
    assert e.instance == 4 and abs(e.surface_area - 0.47 * 0.38) < 1e-6 and list(e.alias_table) == [0, 2]
    # two equal-area triangles: nothing to pour (mod.rs:338-374) -> prob 0, index self
    assert [(a.prob, a.index) for a in s.alias_table] == [(0.0, 0), (0.0, 1)]
    # radius = half diagonal + sqrt(255 * a * |rgb|), instance.rs:401-404
    diag = np.linalg.norm(np.array(list(s.instances[4].max)) - np.array(list(s.instances[4].min)))
    assert abs(e.radius - (0.5 * diag + np.sqrt(255.0 * np.sqrt(3.0)))) < 1e-4


def test_instance_aabb_seeded_at_zero():
    """instance.rs:298-305: min/max start at 0, so the box is centre +/- the per-axis extent."""
    b = hk.SceneBuilder()
    pos = np.array([[0, 0, 0], [2, 0, 0], [0, 4, 0], [2, 4, 6]], np.float32)
    mesh = b.add_mesh(pos, np.tile([0, 0, 1], (4, 1)), np.zeros((4, 2)), [0, 1, 2, 1, 2, 3])
    mat = b.add_material(hk.standard_material())
    t = np.eye(4)
    t[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    t[:3, 3] = [10, 20, 30]
    b.add_instance(mesh, mat, t.T.reshape(-1))
    s = b.finish()
    inst = s.instances[0]
    assert np.allclose(list(inst.min), [10 - 4, 20, 30]) and np.allclose(list(inst.max), [10, 22, 36])
    itm = np.array(list(inst.inverse_transpose_model)).reshape(4, 4).T
    assert np.allclose(itm, np.linalg.inv(t).T, atol=1e-6)


def test_triangle_strip_winding():  # mod.rs:432-449
    b = hk.SceneBuilder()
    pos = np.array([[0, 0, 0], [0, 0, 1], [1, 0, 0], [1, 0, 1], [2, 0, 0]], np.float32)
    mesh = b.add_mesh(pos, np.tile([0, 1, 0], (5, 1)), np.zeros((5, 2)), None, F.TOPOLOGY_TRIANGLE_STRIP)
    b.add_instance(mesh, b.add_material(hk.standard_material()), np.eye(4).reshape(-1))
    s = b.finish()
    idx = [[v.index for v in p.vertices] for p in s.primitives]
    assert idx == [[0, 1, 2], [2, 1, 3], [2, 3, 4]]
    normals = [np.cross(pos[i[1]] - pos[i[0]], pos[i[2]] - pos[i[0]]) for i in idx]
    assert all(n[1] > 0 for n in normals)  # consistent winding after the odd-triangle flip


def test_alias_table_reproduces_area_distribution():
    """mod.rs:330-376 with the sampling rule of light.wgsl:662-664."""
    rng = np.random.default_rng(3)
    b = hk.SceneBuilder()
    n = 9
    pos, idx = [], []
    for k in range(n):
        s = rng.uniform(0.2, 3.0)
        base = len(pos)
        pos += [[k * 4, 0, 0], [k * 4 + s, 0, 0], [k * 4, 0, s]]
        idx += [base, base + 1, base + 2]
    pos = np.array(pos, np.float32)
    mesh = b.add_mesh(pos, np.tile([0, 1, 0], (len(pos), 1)), np.zeros((len(pos), 2)), idx)
    mat = b.add_material(hk.standard_material(emissive_linear=(1, 1, 1)))
    b.add_instance(mesh, mat, np.diag([2.0, 1.0, 0.5, 1.0]).reshape(-1))
    s = b.finish()
    areas = np.array([0.5 * np.linalg.norm(np.cross((pos[3 * k + 1] - pos[3 * k]) * [2, 1, 0.5], (pos[3 * k + 2] - pos[3 * k]) * [2, 1, 0.5])) for k in range(n)])
    assert abs(s.emissives[0].surface_area - areas.sum()) < 1e-4
    prob = np.zeros(n)
    for i, a in enumerate(s.alias_table):
        assert 0.0 <= a.prob <= 1.0 and a.index < n
        prob[a.index] += a.prob / n
        prob[i] += (1.0 - a.prob) / n
    assert np.allclose(prob, areas / areas.sum(), atol=1e-5)


def brute_force_closest(scene, o, d):
    best = (np.inf, -1, -1)
    for ii, inst in enumerate(scene.instances):
        m = np.array(list(inst.model), np.float64).reshape(4, 4).T
        for k in range((inst.mesh.node_count + 2) // 3):
            p = np.array([list(v.position) + [1.0] for v in scene.primitives[inst.mesh.primitive + k].vertices]) @ m.T
            p = p[:, :3]
            ab, ac = p[1] - p[0], p[2] - p[0]
            u_vec = np.cross(d, ac)
            det = np.dot(ab, u_vec)
            if abs(det) < 1e-12:
                continue
            ao = o - p[0]
            u = np.dot(ao, u_vec) / det
            v = np.dot(d, np.cross(ao, ab)) / det
            t = np.dot(ac, np.cross(ao, ab)) / det
            if u < 0 or v < 0 or u + v > 1 or t <= 1e-6:
                continue
            if t < best[0]:
                best = (t, ii, inst.mesh.primitive + k)
    return best


def test_traversal_equals_brute_force():
    scene, _ = synthetic_scene(n_boxes=10, n_spheres=3, n_emitters=2, sphere_rings=5, sphere_segs=6)
    eng = oracle_engine()
    eng.upload_scene(scene)
    rng = np.random.default_rng(11)
    fp = lambda a: a.ctypes.data_as(C.POINTER(F.f32))
    hits = 0
    for _ in range(300):
        o = rng.uniform(-6, 6, 3).astype(np.float32); o[1] = abs(o[1]) + 2.0
        tgt = rng.uniform(-3, 3, 3); tgt[1] = rng.uniform(0, 1.5)
        d = (tgt - o); d = (d / np.linalg.norm(d)).astype(np.float32)
        inst, prim, t, uv = F.u32(), F.u32(), F.f32(), (F.f32 * 2)()
        oracle_api().dll.orc_kat_trace(eng.ctx, fp(o), fp(d), np.float32(3.4e38), np.float32(0.0), 0xFFFFFFFF, C.byref(inst), C.byref(prim), C.byref(t), uv)
        bt, bi, bp = brute_force_closest(scene, o.astype(np.float64), d.astype(np.float64))
        if bi < 0:
            assert inst.value == 0xFFFFFFFF
            continue
        hits += 1
        assert inst.value != 0xFFFFFFFF and abs(t.value - bt) < 1e-3 * max(1.0, bt)
    assert hits > 150


def _walk_always(nodes):
    """skip-link walk that descends everywhere: the leaf visiting order of a flat BVH"""
    i, out = 0, []
    while i < len(nodes):
        if nodes[i].entry_index >= LEAF:
            out.append(nodes[i].entry_index - LEAF)
            i = nodes[i].exit_index
        else:
            i = nodes[i].entry_index
    return out


def test_direction_threaded_flattenings(cornell):
    """hk_bvh_rethread (the eight per-octant flattenings kept for large scenes): every flattening is a well-formed flat BVH over
    the SAME leaves and boxes; octant 7 visits the leaves in exactly the reverse order of octant 0; and along the axis that
    separates two siblings, the sibling visited first is the one a ray of that direction sign meets first."""
    scene, _ = synthetic_scene(n_boxes=30, n_spheres=6, n_emitters=3)
    api = F.api()
    for nodes in (list(scene.instance_nodes), list(cornell.instance_nodes), list(scene.asset_nodes)[scene.instances[1].mesh.node_offset:][:scene.instances[1].mesh.node_count]):
        n = len(nodes)
        src = (F.HkNode * n)(*nodes)
        n_shapes = (n + 2) // 3
        orders = []
        for oct in range(8):
            out = (F.HkNode * n)()
            api.call("bvh_rethread", src, n, oct, out)
            got = list(out)
            for i, nd in enumerate(got):  # structure
                assert i < nd.exit_index <= n and (nd.entry_index >= LEAF or nd.entry_index == i + 1)
            assert sorted(_walk_always(got)) == list(range(n_shapes))
            # the multiset of (box, leaf id / navigator) is unchanged: same nodes, only re-linked
            key = lambda nd: (tuple(nd.min), tuple(nd.max), nd.entry_index if nd.entry_index >= LEAF else -1)
            assert sorted(map(key, got)) == sorted(map(key, nodes))
            # sibling order: navigator a at i and its sibling b at exit(a) (when b is a navigator ending where the parent ends)
            for i, a in enumerate(got):
                j = a.exit_index
                if a.entry_index >= LEAF or j >= n or got[j].entry_index >= LEAF:
                    continue
                parent_end = got[j].exit_index
                if i > 0 and not (got[i - 1].entry_index == i and got[i - 1].exit_index == parent_end) and i != 0:
                    continue  # (a, b) are siblings only if a is the first child of the node in front of it (or of the root)
                ca = np.array(list(a.min)) + np.array(list(a.max))
                cb = np.array(list(got[j].min)) + np.array(list(got[j].max))
                axis = int(np.argmax(np.abs(ca - cb)))
                if ca[axis] != cb[axis]:
                    first_is_lower = ca[axis] < cb[axis]
                    assert first_is_lower == (not (oct >> axis) & 1), (oct, i, j, axis)
            orders.append(_walk_always(got))
        assert orders[7] == orders[0][::-1]
        # ordering 0 of a tree whose left children are all "lower" is not required to equal the reference order, but re-threading
        # octant 0 twice is idempotent
        out0 = (F.HkNode * n)()
        api.call("bvh_rethread", src, n, 0, out0)
        out00 = (F.HkNode * n)()
        api.call("bvh_rethread", out0, n, 0, out00)
        assert bytes(out0) == bytes(out00)
    bad = (F.HkNode * 3)(*list(cornell.instance_nodes)[:3])
    import pytest
    with pytest.raises(F.HikariError):
        api.call("bvh_rethread", bad, 3, 0, (F.HkNode * 3)())


def test_finish_instances_lays_out_the_records_with_stand_in_trees_and_edits_keep_the_history():
    """hk_scene_builder_finish_instances (for hk_update_scene_instances: the device builds the real trees): every per-instance /
    per-emitter record equals hk_scene_builder_finish's, the two trees are VALID flatten_custom trees of the final size over
    the same shapes (3n - 2 nodes, every shape in one leaf, navigator boxes = union of their leaves).  remove_instance /
    set_instance_material edit the declared set; the transform history follows the instances."""
    import ctypes as C

    from bevy_hikari_amd.scenes import synthetic_scene
    from test_device_refit import check_tree

    scene, _ = synthetic_scene(n_boxes=9, n_spheres=2, n_emitters=3, sphere_rings=4, sphere_segs=5)
    twin, _ = synthetic_scene(n_boxes=9, n_spheres=2, n_emitters=3, sphere_rings=4, sphere_segs=5)
    b, t = scene.builder, twin.builder
    moved = np.ctypeslib.as_array(scene.instances[5].model).copy()
    moved[12] += 0.5
    for x in (b, t):
        x.set_instance_transform(5, moved)
        x.remove_instance(3)                  # instance 5 becomes instance 4
        x.set_instance_material(2, 8)         # a plain box becomes an emitter (material 8 is emissive in synthetic_scene)
        x.add_instance(0, 1, np.eye(4, dtype=np.float32).reshape(-1))
    full, light = t.finish(), b.finish(build_trees=False)
    n = len(full.instances)
    assert n == len(scene.instances) and len(light.instances) == n and len(full.emissives) == len(scene.emissives) + 1
    for name in ("instances", "emissives", "alias_table"):
        fa, la = getattr(full, name), getattr(light, name)
        if name in ("instances", "emissives"):   # (node_index = the leaf's position in the tree: differs with the tree, unused by the shaders)
            strip = lambda arr: [bytes(bytearray(x))[:C.sizeof(type(x)) - 0] for x in arr]
            for x, y in zip(fa, la):
                x.node_index = y.node_index = 0
        assert bytes(fa) == bytes(la), name
    assert bytes(full.previous_transforms) == bytes(light.previous_transforms)
    # the moved instance's previous transform is its pose at the first finish - although its index changed from 5 to 4
    prev = np.frombuffer(bytes(light.previous_transforms), dtype=np.float32).reshape(-1, 16)
    assert (prev[4] == np.ctypeslib.as_array(scene.instances[5].model)).all() and (np.ctypeslib.as_array(light.instances[4].model) == moved).all()
    boxes = np.array([[list(i.min), list(i.max)] for i in light.instances], dtype=np.float32)
    check_tree(light.instance_nodes, n, boxes)
    eboxes = np.array([[[e.position[k] - e.radius for k in range(3)], [e.position[k] + e.radius for k in range(3)]] for e in light.emissives], dtype=np.float32)
    check_tree(light.emissive_nodes, len(light.emissives), eboxes)
