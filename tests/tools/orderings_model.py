#!/usr/bin/env python3
"""What do fewer direction-threaded orderings cost in walk length?  (CPU model, no GPU.)

The product keeps EIGHT flattenings of every tree for scenes beyond the LDS copy (hk_bvh_rethread: one per sign pattern of the ray
direction).  Their nodes are 8 x the footprint of one flattening - config 3: 197 MB against 32 MB of L2 - and the wavefront trace
kernel is bound by its memory system.  Keeping only the orderings of `octant & mask` (mask 3: the z sign is ignored, 4 copies;
mask 1: 2 copies; mask 0: the reference order) shrinks the footprint; this script measures the other side of the trade on the
host: node steps and triangle tests of closest-hit walks through one mesh's BLAS, rays from random directions, each walking the
ordering of (its octant & mask).  A vectorised stackless walker over the very arrays hk_scene_builder / hk_bvh_rethread produce.

Usage: python tests/tools/orderings_model.py [n_rays]      (prints one JSON line)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bevy_hikari_amd as hk  # noqa: E402
from bevy_hikari_amd import _ffi as F  # noqa: E402
from bevy_hikari_amd.scenes import synthetic_large  # noqa: E402

LEAF = 0x80000000
EPS = np.float32(1.1920929e-07)


def arrays(nodes):
    n = len(nodes)
    raw = np.frombuffer(bytes(nodes), dtype=np.uint8).reshape(n, C.sizeof(F.HkNode))
    f = raw.view(np.float32).reshape(n, -1)
    u = raw.view(np.uint32).reshape(n, -1)
    # HkNode: min[3], entry_index, max[3], exit_index  (include/hikari_hip.h)
    return f[:, 0:3].copy(), f[:, 4:7].copy(), u[:, 3].copy(), u[:, 7].copy()


def walk(orderings, tri, origin, direction, which):
    """Closest-hit walks of all rays at once; ray r walks orderings[which[r]].  Returns (node steps, triangle tests, hit distance)."""
    n = len(origin)
    inv = (np.float32(1.0) / direction).astype(np.float32)
    idx = np.zeros(n, dtype=np.int64)
    best = np.full(n, np.float32(3.4e38), dtype=np.float32)
    steps = np.zeros(n, dtype=np.int64)
    tests = np.zeros(n, dtype=np.int64)
    count = len(orderings[0][0])
    mn = np.stack([o[0] for o in orderings]); mx = np.stack([o[1] for o in orderings])
    entry = np.stack([o[2] for o in orderings]).astype(np.int64); exit_ = np.stack([o[3] for o in orderings]).astype(np.int64)
    v0, v1, v2 = tri
    alive = np.arange(n)
    while len(alive):
        w, i = which[alive], idx[alive]
        e = entry[w, i]
        leaf = e >= LEAF
        steps[alive] += 1
        # slab test against the node's box (leaves: the flatten_custom leaf has an empty box; the triangle test decides)
        o, iv = origin[alive], inv[alive]
        t1 = (mn[w, i] - o) * iv
        t2 = (mx[w, i] - o) * iv
        tmin = np.minimum(t1, t2).max(axis=1)
        tmax = np.maximum(t1, t2).min(axis=1)
        hit_box = (tmax >= tmin) & (tmax >= 0) & (tmin < best[alive])
        nxt = np.where(leaf | ~hit_box, exit_[w, i], e)
        if leaf.any():
            la = alive[leaf]
            t = (e[leaf] - LEAF).astype(np.int64)
            tests[la] += 1
            a, b, c = v0[t], v1[t], v2[t]
            ab, ac = b - a, c - a
            d = direction[la]
            u_vec = np.cross(d, ac)
            det = (ab * u_vec).sum(axis=1)
            ok = np.abs(det) >= EPS
            inv_det = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0).astype(np.float32)
            ao = origin[la] - a
            uu = (ao * u_vec).sum(axis=1) * inv_det
            v_vec = np.cross(ao, ab)
            vv = (d * v_vec).sum(axis=1) * inv_det
            dist = (ac * v_vec).sum(axis=1) * inv_det
            ok &= (uu >= 0) & (uu <= 1) & (vv >= 0) & (uu + vv <= 1) & (dist > EPS) & (dist < best[la])
            best[la] = np.where(ok, dist, best[la])
        idx[alive] = nxt
        alive = alive[nxt < count]
    return steps, tests, best


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    scene, _ = synthetic_large()   # config 3's scene: 40 unique rock meshes of 6 400 triangles
    api = F.api()
    # the largest BLAS of the scene
    node_count, k = max((i.mesh.node_count, k) for k, i in enumerate(scene.instances))
    mesh = scene.instances[k].mesh
    src = (F.HkNode * mesh.node_count).from_buffer_copy(bytes(scene.asset_nodes)[mesh.node_offset * C.sizeof(F.HkNode):(mesh.node_offset + mesh.node_count) * C.sizeof(F.HkNode)])
    n_tri = (mesh.node_count + 2) // 3  # flatten_custom: 3n - 2 nodes over n triangles; a leaf's id is local to the mesh
    prim = np.frombuffer(bytes(scene.primitives), dtype=np.uint8).reshape(len(scene.primitives), C.sizeof(F.HkPrimitive)).view(np.float32)
    prim = prim[mesh.primitive:mesh.primitive + n_tri]   # per triangle: 3 x (position xyz, index)
    tri = (prim[:, 0:3].copy(), prim[:, 4:7].copy(), prim[:, 8:11].copy())
    orderings = []
    for octant in range(8):
        out = (F.HkNode * mesh.node_count)()
        api.call("bvh_rethread", src, mesh.node_count, octant, out)
        orderings.append(arrays(out))
    reference = arrays(src)
    lo, hi = np.minimum.reduce(tri[0]), np.maximum.reduce(tri[0])
    centre, radius = (lo + hi) / 2, float(np.linalg.norm(hi - lo))
    rng = np.random.default_rng(1)
    d = rng.normal(size=(n_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    target = (centre + rng.uniform(-0.35, 0.35, size=(n_rays, 3)) * (hi - lo)).astype(np.float32)
    origin = (target - d * np.float32(radius)).astype(np.float32)
    octant = ((d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2))
    res = {"mesh_triangles": int(n_tri), "mesh_nodes": int(mesh.node_count), "rays": n_rays}
    base = None
    for name, mask in (("8 orderings", 7), ("4 orderings (x, y signs)", 3), ("2 orderings (x sign)", 1)):
        s, t, dist = walk(orderings, tri, origin, d, octant & mask)
        if base is None:
            base = (s.mean(), dist)
        res[name] = {"node_steps": round(float(s.mean()), 1), "triangle_tests": round(float(t.mean()), 2), "steps_vs_8": round(float(s.mean() / base[0]), 3),
                     "hits": int((dist < 3e38).sum())}
    s, t, dist = walk([reference], tri, origin, d, np.zeros(n_rays, dtype=np.int64))
    res["reference order"] = {"node_steps": round(float(s.mean()), 1), "triangle_tests": round(float(t.mean()), 2), "steps_vs_8": round(float(s.mean() / base[0]), 3),
                              "hits": int((dist < 3e38).sum())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
