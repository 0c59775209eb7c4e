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


def _test_textures(rng):
    """Four small procedural RGBA8 images covering the sampler variants (sRGB / linear format,
    bilinear / nearest, repeat / mirror / clamp)."""
    yy, xx = np.mgrid[0:16, 0:32]
    checker = np.where(((xx // 4 + yy // 4) % 2)[..., None] == 0, np.array([230, 60, 40, 255]), np.array([40, 90, 220, 255])).astype(np.uint8)
    stripes = np.zeros((8, 8, 4), np.uint8)
    stripes[..., 3] = 255
    stripes[:, ::2, :3] = (200, 200, 190)
    stripes[:, 1::2, :3] = (90, 120, 60)
    noise = rng.integers(0, 256, (16, 16, 4), dtype=np.uint8)
    glow = np.zeros((4, 16, 4), np.uint8)
    glow[..., 3] = 255
    glow[..., 0] = np.linspace(40, 255, 16).astype(np.uint8)[None, :]
    glow[..., 1] = 180
    glow[..., 2] = np.linspace(255, 30, 16).astype(np.uint8)[None, :]
    return [dict(rgba=checker, srgb=True, linear=True, address_u=F.ADDRESS_REPEAT, address_v=F.ADDRESS_REPEAT),
            dict(rgba=stripes, srgb=True, linear=False, address_u=F.ADDRESS_MIRROR_REPEAT, address_v=F.ADDRESS_CLAMP_TO_EDGE),
            dict(rgba=noise, srgb=False, linear=True, address_u=F.ADDRESS_CLAMP_TO_EDGE, address_v=F.ADDRESS_MIRROR_REPEAT),
            dict(rgba=glow, srgb=True, linear=True, address_u=F.ADDRESS_REPEAT, address_v=F.ADDRESS_CLAMP_TO_EDGE)]


def synthetic_scene(seed=0x5EED0003, n_boxes=24, n_spheres=6, n_emitters=4, extent=4.0, sphere_rings=8, sphere_segs=12, textured=False, n_emissive_spheres=0):
    """A room-less yard: ground slab, random boxes and spheres with rotated / non-uniformly scaled
    instances of a few shared meshes, `n_emitters` emissive strip-quads above it.  Returns
    (SceneData, suggested sun dict).  textured=True binds four procedural textures (base colour,
    metallic, occlusion, emissive; light.wgsl:749-793)."""
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
    textures = []
    if textured:
        textures = _test_textures(rng)

        def mat(base=(1, 1, 1, 1), emissive=(0, 0, 0), rough=0.6, metallic=0.0, **ids):
            m = standard_material(base, emissive, rough, metallic, 0.5)
            for k, v in ids.items():
                setattr(m, k, v)
            return b.add_material(m)

        mats[0] = mat((0.9, 0.9, 0.9, 1), base_color_texture=1)                                   # ground: nearest / mirror / clamp
        mats[1] = mat((1, 1, 1, 1), base_color_texture=0)                                         # checker, bilinear repeat
        mats[2] = mat((0.8, 0.7, 0.6, 1), rough=0.4, metallic=1.0, metallic_roughness_texture=2)  # metallic *= noise.r
        mats[3] = mat((0.7, 0.8, 0.9, 1), base_color_texture=0, occlusion_texture=2)
        emat[0] = mat((0.8, 0.8, 0.8, 1), (1.0, 0.9, 0.8), 1.0, emissive_texture=3)
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
    for i in range(n_emissive_spheres):  # examples/scene.rs:231-235: an emissive sphere - an emitter with as many triangles as the sphere mesh
        t = (rng.uniform(-extent, extent) * 0.6, rng.uniform(1.6, 2.4), rng.uniform(-extent, extent) * 0.6)
        b.add_instance(sphere, emat[i % len(emat)], _trs(t, rng.uniform(-1, 1, 3), rng.uniform(0.25, 0.5, 3)))
    sun = dict(color=(1.0, 0.96, 0.9), illuminance=20000.0, direction_to_light=(0.35, 0.8, 0.45))
    scene = b.finish()
    scene.textures = textures
    scene.builder = b  # kept for dynamic-scene use: builder.set_instance_transform(i, ..) + builder.finish()
    return scene, sun


def animate(scene, frame, movers=(3, 9, 26, 31)):
    """Move a few instances of a synthetic_scene (two boxes, a sphere, an emitter) to their pose at
    `frame` and re-finish the builder.  Returns the new SceneData (meshes and materials unchanged;
    previous_transforms = the poses of the previous call).  Frame 0 is the rest pose."""
    b = scene.builder
    if not hasattr(scene, "rest_pose"):
        scene.rest_pose = np.array([np.ctypeslib.as_array(inst.model).copy() for inst in scene.instances], dtype=np.float32)
    for k, i in enumerate(movers):
        if i >= len(scene.rest_pose):
            continue
        m = scene.rest_pose[i].reshape(4, 4).T.astype(np.float64)  # column-major -> math layout
        ang = 0.05 * frame * (1 + k)
        c, s_ = math.cos(ang), math.sin(ang)
        rot = np.array([[c, 0, s_, 0], [0, 1, 0, 0], [-s_, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)
        shift = np.eye(4)
        shift[0, 3] = 0.04 * frame * (1 if k % 2 == 0 else -1)
        shift[1, 3] = 0.02 * frame * (k % 3 == 0)
        t = shift @ m @ rot  # spin about the local Y axis, drift in world space
        b.set_instance_transform(i, t.T.astype(np.float32).reshape(-1))
    new = b.finish()
    new.textures = getattr(scene, "textures", [])
    new.builder, new.rest_pose = b, scene.rest_pose
    return new


def synthetic_camera(width, height, extent=4.0):
    from .plugin import Camera, look_at_transform

    return Camera(look_at_transform((1.6 * extent, 1.1 * extent, 2.0 * extent), (0.0, 0.6, 0.0)), width, height)


def _rock(rng, rings, segs):
    """A sphere displaced by a few random low-frequency lobes: a unique closed mesh per seed."""
    p, n, uv, idx = _sphere(rings, segs)
    d = p / np.linalg.norm(p, axis=1, keepdims=True)
    disp = np.ones(len(p))
    for _ in range(6):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        disp += rng.uniform(0.05, 0.25) * np.sin(rng.uniform(1.5, 5.0) * (d @ axis) + rng.uniform(0, 6.28))
    p = (d * (0.5 * disp)[:, None]).astype(np.float32)
    # smooth normals from the displaced surface (area-weighted face normals)
    nrm = np.zeros_like(p, dtype=np.float64)
    tri = idx.reshape(-1, 3)
    fn = np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
    for k in range(3):
        np.add.at(nrm, tri[:, k], fn)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = np.where(ln > 1e-12, nrm / np.maximum(ln, 1e-12), d)
    return p, nrm.astype(np.float32), uv, idx


def synthetic_large(seed=0x5EED0003, n_meshes=40, rings=40, segs=80, n_instances=400, n_materials=50, n_emitters=8, extent=12.0):
    """Sponza-class stand-in (~260 k unique triangles, ~400 instances, 50
    materials, 8 emissive quads, sun 100 000 lux).  `rings x segs x 2` triangles per unique mesh.
    City-class (config 4): synthetic_large(0x5EED0004, 60, 80, 160, 2000, 50, 1, 40.0)."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    box = b.add_mesh(*_box())
    rocks = [b.add_mesh(*_rock(rng, rings, segs)) for _ in range(n_meshes)]
    qp, qn, quv = _quad_strip(4)
    quad = b.add_mesh(qp, qn, quv, None, F.TOPOLOGY_TRIANGLE_STRIP)
    mats = [b.add_material(standard_material(tuple(rng.uniform(0.15, 0.9, 3)) + (1.0,), (0, 0, 0), float(rng.uniform(0.25, 1.0)),
                                             float(rng.choice([0.0, 0.0, 0.0, 1.0])), 0.5)) for _ in range(n_materials)]
    emat = [b.add_material(standard_material((0.8, 0.8, 0.8, 1.0), tuple(rng.uniform(0.3, 1.0, 3)), 1.0, 0.0, 0.5)) for _ in range(max(1, n_emitters))]
    b.add_instance(box, mats[0], _trs((0, -0.25, 0), (0, 0, 0), (2.5 * extent, 0.5, 2.5 * extent)))
    for i in range(n_instances):
        t = (rng.uniform(-extent, extent), rng.uniform(0.3, 2.5), rng.uniform(-extent, extent))
        s = rng.uniform(0.6, 2.2, 3)
        b.add_instance(rocks[int(rng.integers(0, n_meshes))], mats[int(rng.integers(1, n_materials))], _trs(t, rng.uniform(-3.1, 3.1, 3), s))
    for i in range(n_emitters):
        t = (rng.uniform(-extent, extent) * 0.8, rng.uniform(3.5, 5.0), rng.uniform(-extent, extent) * 0.8)
        b.add_instance(quad, emat[i], _trs(t, (math.pi + rng.uniform(-0.3, 0.3), rng.uniform(-1, 1), 0.0), (rng.uniform(0.8, 2.0), 1.0, rng.uniform(0.8, 2.0))))
    sun = dict(color=(1.0, 0.96, 0.9), illuminance=100000.0, direction_to_light=(0.35, 0.8, 0.45))
    scene = b.finish()
    scene.builder = b
    return scene, sun


def flight_helmet_scene():
    """The reference's textured glTF asset (assets/models/FlightHelmet, flattened by
    tools/make_fixtures.py: 6 meshes, 94 722 triangles, base-colour + occlusion/roughness/metal
    textures at 256^2) on a ground slab under one emissive quad and a sun.  Materials are what
    bevy_gltf 0.9.1 builds: factors, texture ids, reflectance 0.5; glTF samplers are linear + repeat.
    Returns (SceneData with .textures, sun dict, Camera factory)."""
    import os

    from .plugin import ASSETS, Camera, look_at_transform

    d = np.load(os.path.join(ASSETS, "flight_helmet.npz"))
    b = SceneBuilder()
    textures = [dict(rgba=d["textures"][i], srgb=bool(d["texture_srgb"][i]), address_u=F.ADDRESS_REPEAT, address_v=F.ADDRESS_REPEAT, linear=True)
                for i in range(len(d["textures"]))]
    mats = []
    for row in d["materials"]:
        m = standard_material(tuple(row[:4]), (0, 0, 0), float(row[4]), float(row[5]), 0.5)
        m.base_color_texture, m.metallic_roughness_texture, m.occlusion_texture = int(row[6]), int(row[7]), int(row[8])
        mats.append(b.add_material(m))
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for inst in d["instances"]:
        k = int(inst[0])
        mesh = b.add_mesh(d[f"mesh{k}_positions"], d[f"mesh{k}_normals"], d[f"mesh{k}_uvs"], d[f"mesh{k}_indices"])
        b.add_instance(mesh, mats[int(d[f"mesh{k}_material"][0])], inst[1:17])
        t = inst[1:17].reshape(4, 4).T
        p = d[f"mesh{k}_positions"] @ t[:3, :3].T + t[:3, 3]
        lo, hi = np.minimum(lo, p.min(axis=0)), np.maximum(hi, p.max(axis=0))
    centre, extent = 0.5 * (lo + hi), float((hi - lo).max())
    ground = b.add_material(standard_material((0.55, 0.55, 0.6, 1.0), (0, 0, 0), 0.7, 0.0, 0.5))
    lamp = b.add_material(standard_material((0.8, 0.8, 0.8, 1.0), (1.0, 0.95, 0.85), 1.0, 0.0, 0.5))
    box = b.add_mesh(*_box())
    qp, qn, quv = _quad_strip(2)
    quad = b.add_mesh(qp, qn, quv, None, F.TOPOLOGY_TRIANGLE_STRIP)
    b.add_instance(box, ground, _trs((centre[0], lo[1] - 0.05 * extent, centre[2]), (0, 0, 0), (4 * extent, 0.1 * extent, 4 * extent)))
    b.add_instance(quad, lamp, _trs((centre[0] + 0.4 * extent, hi[1] + 0.6 * extent, centre[2] + 0.5 * extent), (math.pi, 0.3, 0.0),
                                    (0.6 * extent, 1.0, 0.6 * extent)))
    scene = b.finish()
    scene.textures = textures
    scene.builder = b
    sun = dict(color=(1.0, 0.96, 0.9), illuminance=15000.0, direction_to_light=(0.4, 0.8, 0.5))

    def camera(width, height):
        eye = (centre[0] + 0.9 * extent, centre[1] + 0.35 * extent, centre[2] + 1.5 * extent)
        return Camera(look_at_transform(eye, tuple(centre)), width, height)

    return scene, sun, camera
