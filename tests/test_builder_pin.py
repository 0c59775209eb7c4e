"""The product's host builders (`hk_scene_builder_*`: BLAS / TLAS / light BVH through the `bvh` 0.7.1 algorithm, instance
AABBs, alias tables, emissive list) pinned from OUTSIDE the product.

Every GPU-vs-oracle parity test feeds both sides the product builder's output, so a wrong restatement of `bvh` 0.7.1
inside scene_builder.cpp (tree shape -> closest-hit tie-breaks, any-hit identity, the light BVH's visit order of
light.wgsl:638-645) would be invisible to them.  tests/bvh071.py restates the same published algorithms a second time,
independently, in numpy f32; here its arrays must equal the C++ builder's BYTE FOR BYTE on Cornell, the yard scenes and
the reference's FlightHelmet asset."""
import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import plugin, scenes
from bevy_hikari_amd import _ffi as F

import bvh071


class RecordingBuilder(plugin.SceneBuilder):
    """hk.SceneBuilder that also keeps what the host handed it."""

    def __init__(self):
        super().__init__()
        self.meshes, self.instances, self.materials = [], [], []

    def add_mesh(self, positions, normals, uvs, indices=None, topology=F.TOPOLOGY_TRIANGLE_LIST):
        self.meshes.append((np.array(positions, np.float32).reshape(-1, 3), None if indices is None else np.array(indices, np.int64).reshape(-1), topology))
        return super().add_mesh(positions, normals, uvs, indices, topology)

    def add_material(self, material):
        m = super().add_material(material)
        self.materials.append(tuple(material.emissive))
        return m

    def add_instance(self, mesh_id, material_id, transform):
        self.instances.append((mesh_id, material_id, np.array(transform, np.float32).reshape(-1)))
        return super().add_instance(mesh_id, material_id, transform)


def record(loader, monkeypatch):
    made = []

    def factory():
        b = RecordingBuilder()
        made.append(b)
        return b

    monkeypatch.setattr(plugin, "SceneBuilder", factory)
    monkeypatch.setattr(scenes, "SceneBuilder", factory)
    out = loader()
    scene = out[0] if isinstance(out, tuple) else out
    return scene, made[-1]


def nodes_of(arr):
    dt = np.dtype([("min", np.float32, 3), ("entry", np.uint32), ("max", np.float32, 3), ("exit", np.uint32)])
    return np.frombuffer(bytes(arr), dt) if len(arr) else np.zeros(0, dt)


def canon(x):
    """Rust's f32::min / f32::max (what `AABB::join` / `grow` use) leave the sign of a zero result unspecified when the
    operands are -0 and +0 (IEEE minNum / maxNum; LLVM picks by operand order) - a box face at -0 or +0 is the same
    box for every ray.  The comparison is bytewise on everything else: -0 is mapped to +0 on both sides."""
    x = np.array(x, np.float32, copy=True)
    x[x == 0] = 0.0
    return x.tobytes()


def same_nodes(a, b, what):
    assert len(a) == len(b), f"{what}: {len(a)} vs {len(b)} nodes"
    a, b = a.copy(), b.copy()
    for arr in (a, b):
        arr["min"][arr["min"] == 0] = 0.0
        arr["max"][arr["max"] == 0] = 0.0
    bad = np.nonzero((np.frombuffer(a.tobytes(), np.uint8).reshape(len(a), 32) != np.frombuffer(b.tobytes(), np.uint8).reshape(len(b), 32)).any(1))[0]
    assert bad.size == 0, f"{what}: {bad.size} differing nodes, first {bad[0]}: {a[bad[0]]} vs {b[bad[0]]}"


LOADERS = {
    "cornell": lambda: hk.load_cornell(),
    "yard": lambda: scenes.synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=3, sphere_rings=6, sphere_segs=8),
    "yard_default": lambda: scenes.synthetic_scene(),
    "flight_helmet": lambda: scenes.flight_helmet_scene(),
}


@pytest.mark.parametrize("name", list(LOADERS))
def test_builder_arrays_equal_an_independent_restatement(name, monkeypatch):
    scene, rec = record(LOADERS[name], monkeypatch)
    # ---- BLAS of every mesh (mod.rs:413-459): concatenated in mesh order (mesh.rs:106-166)
    want_nodes, tris, node_off, prim_off, off_n, off_p = [], [], [], [], 0, 0
    for pos, idx, topo in rec.meshes:
        tri = bvh071.mesh_primitives(pos, idx, strip=topo == F.TOPOLOGY_TRIANGLE_STRIP)
        nodes = bvh071.blas(pos, tri)
        tris.append(tri)
        want_nodes.append(nodes)
        node_off.append((off_n, len(nodes)))
        prim_off.append(off_p)
        off_n += len(nodes)
        off_p += len(tri)
    same_nodes(nodes_of(scene.asset_nodes), np.concatenate(want_nodes), f"{name} BLAS nodes")
    # the primitives themselves: positions + vertex indices, mesh after mesh
    got_prims = np.frombuffer(bytes(scene.primitives), np.dtype([("p", np.float32, 3), ("i", np.uint32)])).reshape(-1, 3)
    k = 0
    for (pos, _, _), tri in zip(rec.meshes, tris):
        assert (got_prims["i"][k:k + len(tri)] == tri).all() and (got_prims["p"][k:k + len(tri)] == pos[tri]).all()
        k += len(tri)
    assert k == len(got_prims)
    # ---- instances: world AABB (instance.rs:286-310) and mesh slices
    boxes = []
    for inst, (mesh_id, material_id, transform) in zip(scene.instances, rec.instances):
        lo, hi = bvh071.instance_aabb(rec.meshes[mesh_id][0], transform)
        assert canon(inst.min[:]) == canon(lo) and canon(inst.max[:]) == canon(hi), (list(inst.min), lo, list(inst.max), hi)
        assert (inst.mesh.node_offset, inst.mesh.node_count) == node_off[mesh_id] and inst.mesh.primitive == prim_off[mesh_id]
        assert inst.material == material_id and np.array(inst.model[:], np.float32).tobytes() == transform.tobytes()
        boxes.append((lo, hi))
    # ---- TLAS (instance.rs:365-370)
    bmin, bmax = np.array([b[0] for b in boxes]), np.array([b[1] for b in boxes])
    same_nodes(nodes_of(scene.instance_nodes), bvh071.flatten(bvh071.build(bmin, bmax)), f"{name} TLAS nodes")
    # ---- emissive list, alias tables (instance.rs:380-421, mod.rs:318-376), light BVH (instance.rs:422-428)
    want_alias, want_em = [], []
    for i, (mesh_id, material_id, transform) in enumerate(rec.instances):
        e = np.array(rec.materials[material_id], np.float32)
        length = np.sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2])  # glam Vec3::length = sqrt(dot)
        intensity = np.float32(255.0) * e[3] * length
        if not intensity > 0.0:
            continue
        areas = bvh071.primitive_areas(rec.meshes[mesh_id][0], tris[mesh_id], transform)
        table = bvh071.alias_table(areas)
        lo, hi = boxes[i]
        position = np.float32(0.5) * (hi + lo)
        d = hi - lo
        radius = np.float32(0.5) * np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + np.sqrt(intensity)
        want_em.append(dict(position=position, radius=radius, instance=i, alias=(len(want_alias), len(table)), area=bvh071.sum_f32(areas)))
        want_alias += table
    assert len(scene.emissives) == len(want_em)
    for got, want in zip(scene.emissives, want_em):
        assert canon(got.position[:]) == canon(want["position"])
        assert np.float32(got.radius).tobytes() == np.float32(want["radius"]).tobytes(), (got.radius, want["radius"])
        assert got.instance == want["instance"] and tuple(got.alias_table) == want["alias"]
        assert np.float32(got.surface_area).tobytes() == np.float32(want["area"]).tobytes()
    assert len(scene.alias_table) == len(want_alias)
    for got, (prob, index) in zip(scene.alias_table, want_alias):
        assert np.float32(got.prob).tobytes() == np.float32(prob).tobytes() and got.index == index
    if want_em:
        emin = np.array([w["position"] - w["radius"] for w in want_em], np.float32)  # Bounded for GpuEmissive, mod.rs:239-246
        emax = np.array([w["position"] + w["radius"] for w in want_em], np.float32)
        same_nodes(nodes_of(scene.emissive_nodes), bvh071.flatten(bvh071.build(emin, emax)), f"{name} light BVH nodes")
    else:
        assert len(scene.emissive_nodes) == 0


def test_restated_bvh_on_degenerate_inputs():
    """The `split_axis_size < EPSILON` branch (all centroids coincide: the index list is halved) and a single shape."""
    bmin = np.zeros((5, 3), np.float32)
    bmax = np.ones((5, 3), np.float32)
    flat = bvh071.flatten(bvh071.build(bmin, bmax))
    assert len(flat) == 3 * 5 - 2
    leaves = [int(n["entry"]) - bvh071.LEAF for n in flat if n["entry"] >= bvh071.LEAF]
    assert leaves == [0, 1, 2, 3, 4]  # halving keeps the order
    one = bvh071.flatten(bvh071.build(bmin[:1], bmax[:1]))
    assert len(one) == 1 and one[0]["entry"] == bvh071.LEAF and one[0]["exit"] == 1 and one[0]["min"][0] == np.inf
    b = hk.SceneBuilder()
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]] * 5, np.float32)
    mesh = b.add_mesh(pos, np.tile([0, 0, 1], (15, 1)), np.zeros((15, 2)), np.arange(15))
    b.add_instance(mesh, b.add_material(hk.standard_material()), np.eye(4).reshape(-1))
    s = b.finish()
    tri = bvh071.mesh_primitives(pos, np.arange(15))
    same_nodes(nodes_of(s.asset_nodes), bvh071.blas(pos, tri), "coincident triangles")
