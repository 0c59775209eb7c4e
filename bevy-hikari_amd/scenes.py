"""Seeded synthetic scenes standing in for the reference's missing assets (scene.gltf,
City/scene.bin and the Low Poly glbs are absent from the checkout).  Pure data generation; all
acceleration structures are built by the library (SceneBuilder)."""
import math

import numpy as np

from . import _ffi as F
from .plugin import SceneBuilder, standard_material


def _box(sx=1.0, sy=1.0, sz=1.0):
    """24-vertex box centred at the origin with per-face normals/uvs, 12 triangles (triangle list)."""
    p, n, uv, idx = [], [], [], []
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)),
             ((0, -1, 0), (1, 0, 0), (0, 0, 1)), ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0))]
    half = np.array([sx, sy, sz]) * 0.5
    for nrm, u, v in faces:
        nrm, u, v = np.array(nrm, float), np.array(u, float), np.array(v, float)
        base = len(p)
        for a, b in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            p.append((nrm + a * u + b * v) * half)
            n.append(nrm)
            uv.append(((a + 1) / 2, (b + 1) / 2))
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.array(p, np.float32), np.array(n, np.float32), np.array(uv, np.float32), np.array(idx, np.uint32)


def _quad_strip(nx=4):
    """A unit XZ quad (normal +Y) tessellated as ONE triangle strip of 2*nx triangles (exercises the
    strip winding rule, mod.rs:432-449)."""
    p, n, uv = [], [], []
    for i in range(nx + 1):
        x = i / nx - 0.5
        for z in (0.5, -0.5):
            p.append((x, 0.0, z))
            n.append((0.0, 1.0, 0.0))
            uv.append((i / nx, z + 0.5))
    return np.array(p, np.float32), np.array(n, np.float32), np.array(uv, np.float32)


def _sphere(rings=8, segs=12):
    p, n, uv, idx = [], [], [], []
    for r in range(rings + 1):
        th = math.pi * r / rings
        for s in range(segs + 1):
            ph = 2 * math.pi * s / segs
            d = (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
            p.append(tuple(0.5 * c for c in d))
            n.append(d)
            uv.append((s / segs, r / rings))
    for r in range(rings):
        for s in range(segs):
            a = r * (segs + 1) + s
            b = a + segs + 1
            idx += [a, a + 1, b, a + 1, b + 1, b]
    return np.array(p, np.float32), np.array(n, np.float32), np.array(uv, np.float32), np.array(idx, np.uint32)


def _trs(t, angles, s):
    ax, ay, az = angles
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    m = np.eye(4)
    m[:3, :3] = (ry @ rx @ rz) * np.asarray(s, float)[None, :]
    m[:3, 3] = t
    return m.T.astype(np.float32).reshape(-1)  # column-major


def synthetic_scene(seed=0x5EED0003, n_boxes=24, n_spheres=6, n_emitters=4, extent=4.0, sphere_rings=8, sphere_segs=12):
    """A room-less yard: ground slab, random boxes and spheres with rotated / non-uniformly scaled
    instances of a few shared meshes, `n_emitters` emissive strip-quads above it.  Returns
    (SceneData, suggested sun dict)."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    box = b.add_mesh(*_box())
    sp = _sphere(sphere_rings, sphere_segs)
    sphere = b.add_mesh(*sp)
    qp, qn, quv = _quad_strip(4)
    quad = b.add_mesh(qp, qn, quv, None, F.TOPOLOGY_TRIANGLE_STRIP)
    mats = [b.add_material(standard_material(tuple(rng.uniform(0.2, 0.9, 3)) + (1.0,), (0, 0, 0), float(rng.uniform(0.3, 1.0)),
                                             float(rng.choice([0.0, 0.0, 1.0])), 0.5)) for _ in range(8)]
    emat = [b.add_material(standard_material((0.8, 0.8, 0.8, 1.0), tuple(rng.uniform(0.2, 1.0, 3)), 1.0, 0.0, 0.5)) for _ in range(max(1, n_emitters))]
    # ground
    b.add_instance(box, mats[0], _trs((0, -0.25, 0), (0, 0, 0), (2.5 * extent, 0.5, 2.5 * extent)))
    for _ in range(n_boxes):
        t = (rng.uniform(-extent, extent), rng.uniform(0.2, 1.5), rng.uniform(-extent, extent))
        b.add_instance(box, mats[int(rng.integers(1, 8))], _trs(t, rng.uniform(-0.6, 0.6, 3), rng.uniform(0.3, 1.4, 3)))
    for _ in range(n_spheres):
        t = (rng.uniform(-extent, extent), rng.uniform(0.4, 1.8), rng.uniform(-extent, extent))
        b.add_instance(sphere, mats[int(rng.integers(1, 8))], _trs(t, rng.uniform(-1, 1, 3), rng.uniform(0.5, 1.5, 3)))
    for i in range(n_emitters):
        t = (rng.uniform(-extent, extent) * 0.8, rng.uniform(2.2, 3.2), rng.uniform(-extent, extent) * 0.8)
        # flipped so the strip's +Y normal faces down
        b.add_instance(quad, emat[i], _trs(t, (math.pi + rng.uniform(-0.3, 0.3), rng.uniform(-1, 1), 0.0), (rng.uniform(0.4, 1.0), 1.0, rng.uniform(0.4, 1.0))))
    sun = dict(color=(1.0, 0.96, 0.9), illuminance=20000.0, direction_to_light=(0.35, 0.8, 0.45))
    return b.finish(), sun


def synthetic_camera(width, height, extent=4.0):
    from .plugin import Camera, look_at_transform

    return Camera(look_at_transform((1.6 * extent, 1.1 * extent, 2.0 * extent), (0.0, 0.6, 0.0)), width, height)
