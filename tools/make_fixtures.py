#!/usr/bin/env python3
"""Generate the data fixtures this repo ships, from the read-only reference checkout.

Run once in the build container (``/root/reference`` does not exist on the GPU box):

    python tools/make_fixtures.py

Outputs (committed):
  bevy-hikari_amd/assets/noise_rgba8_16x64x64.bin
      the 16 blue-noise tiles the reference embeds (src/lib.rs:189-219 loads
      src/noise/LDR_RGBA_{0..15}.png with is_srgb=false), decoded with Pillow to
      raw RGBA8, tile-major [16][64][64][4].  This is the path's RNG source.
  bevy-hikari_amd/assets/cornell.json
      assets/models/cornell.glb (the scene examples/cornell.rs:37-41 spawns) flattened to
      plain arrays: per mesh positions/normals/uvs/indices, per node the world transform,
      per material the glTF factors.  No reference *source code* is copied - these are the
      reference's data assets, which a drop-in for the path has to consume unchanged.
  bevy-hikari_amd/assets/flight_helmet.npz
      assets/models/FlightHelmet: the reference's textured glTF asset, geometry
      unchanged, textures box-filtered to 256^2.
"""
import json
import os
import struct
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "bevy-hikari_amd", "assets")


def make_noise():
    from PIL import Image

    tiles = []
    for i in range(16):
        im = Image.open(os.path.join(REF, "src", "noise", f"LDR_RGBA_{i}.png"))
        assert im.mode == "RGBA" and im.size == (64, 64), (im.mode, im.size)
        tiles.append(np.asarray(im, dtype=np.uint8))
    arr = np.stack(tiles)  # [16][64(y)][64(x)][4]
    assert arr.shape == (16, 64, 64, 4)
    arr.tofile(os.path.join(OUT, "noise_rgba8_16x64x64.bin"))
    print("noise:", arr.shape, arr.dtype, "mean", float(arr.mean()))


_COMP = {5120: "b", 5121: "B", 5122: "h", 5123: "H", 5125: "I", 5126: "f"}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def quat_to_mat(q):
    x, y, z, w = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ],
        dtype=np.float64,
    )


def node_local(n):
    m = np.eye(4)
    if "matrix" in n:
        return np.array(n["matrix"], dtype=np.float64).reshape(4, 4).T
    r = quat_to_mat(n.get("rotation", [0, 0, 0, 1]))
    s = np.array(n.get("scale", [1, 1, 1]), dtype=np.float64)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = n.get("translation", [0, 0, 0])
    return m


def make_cornell():
    d = open(os.path.join(REF, "assets", "models", "cornell.glb"), "rb").read()
    magic, ver, length = struct.unpack("<III", d[:12])
    assert magic == 0x46546C67
    off = 12
    clen, ctype = struct.unpack("<II", d[off : off + 8])
    off += 8
    j = json.loads(d[off : off + clen])
    off += clen
    blen, btype = struct.unpack("<II", d[off : off + 8])
    off += 8
    blob = d[off : off + blen]

    def accessor(i):
        a = j["accessors"][i]
        bv = j["bufferViews"][a["bufferView"]]
        fmt = _COMP[a["componentType"]]
        n = _NCOMP[a["type"]]
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        arr = np.frombuffer(blob, dtype="<" + fmt, count=a["count"] * n, offset=start)
        return arr.reshape(a["count"], n) if n > 1 else arr

    meshes = []
    for m in j["meshes"]:
        assert len(m["primitives"]) == 1
        p = m["primitives"][0]
        assert p.get("mode", 4) == 4  # triangle list
        meshes.append(
            {
                "name": m["name"],
                "material": p["material"],
                "positions": accessor(p["attributes"]["POSITION"]).astype(np.float32).tolist(),
                "normals": accessor(p["attributes"]["NORMAL"]).astype(np.float32).tolist(),
                "uvs": accessor(p["attributes"]["TEXCOORD_0"]).astype(np.float32).tolist(),
                "indices": accessor(p["indices"]).astype(np.uint32).tolist(),
            }
        )

    # Depth-first walk of the default scene; the order mesh-bearing nodes are met is the
    # entity spawn order, which is the instance order (instance.rs:231-239 BTreeMap<Entity,..>).
    instances = []

    def walk(idx, parent):
        n = j["nodes"][idx]
        world = parent @ node_local(n)
        if "mesh" in n:
            instances.append(
                {
                    "name": n.get("name", ""),
                    "mesh": n["mesh"],
                    # column-major 16 floats, f32-rounded (bevy GlobalTransform::compute_matrix)
                    "transform": world.T.astype(np.float32).reshape(-1).tolist(),
                }
            )
        for c in n.get("children", []):
            walk(c, world)

    for root in j["scenes"][j.get("scene", 0)]["nodes"]:
        walk(root, np.eye(4))

    materials = []
    for m in j["materials"]:
        pbr = m.get("pbrMetallicRoughness", {})
        materials.append(
            {
                "name": m["name"],
                "base_color_factor": pbr.get("baseColorFactor", [1, 1, 1, 1]),
                "metallic_factor": pbr.get("metallicFactor", 1.0),
                "roughness_factor": pbr.get("roughnessFactor", 1.0),
                "emissive_factor": m.get("emissiveFactor", [0, 0, 0]),
                "double_sided": m.get("doubleSided", False),
            }
        )

    out = {
        "source": "assets/models/cornell.glb (bevy-hikari v0.3.15)",
        "meshes": meshes,
        "materials": materials,
        "instances": instances,
    }
    with open(os.path.join(OUT, "cornell.json"), "w") as f:
        json.dump(out, f, indent=None, separators=(",", ":"))
    ntri = sum(len(m["indices"]) // 3 for m in meshes)
    nv = sum(len(m["positions"]) for m in meshes)
    print(f"cornell: {len(meshes)} meshes, {ntri} tris, {nv} verts, {len(materials)} materials, {len(instances)} instances")


def make_cornell_bin():
    """Binary twin of cornell.json (same numbers) for hosts without a JSON parser (examples/cornell.cpp):
    'HKSC' u32 version, n_meshes, n_materials, n_instances; per mesh: n_vertices, n_indices, material,
    positions f32x3[], normals f32x3[], uvs f32x2[], indices u32[]; per material: base_color f32x4,
    emissive f32x3, roughness, metallic; per instance: mesh u32, transform f32x16 (column-major)."""
    j = json.load(open(os.path.join(OUT, "cornell.json")))
    out = bytearray(struct.pack("<4sIIII", b"HKSC", 1, len(j["meshes"]), len(j["materials"]), len(j["instances"])))
    for m in j["meshes"]:
        pos, nrm = np.asarray(m["positions"], np.float32), np.asarray(m["normals"], np.float32)
        uv, idx = np.asarray(m["uvs"], np.float32), np.asarray(m["indices"], np.uint32)
        out += struct.pack("<III", len(pos), len(idx), m["material"]) + pos.tobytes() + nrm.tobytes() + uv.tobytes() + idx.tobytes()
    for m in j["materials"]:
        out += np.asarray(list(m["base_color_factor"]) + list(m["emissive_factor"]) + [m["roughness_factor"], m["metallic_factor"]], np.float32).tobytes()
    for i in j["instances"]:
        out += struct.pack("<I", i["mesh"]) + np.asarray(i["transform"], np.float32).tobytes()
    open(os.path.join(OUT, "cornell.hkscene"), "wb").write(out)
    print("cornell.hkscene:", len(out), "bytes")


def make_flight_helmet(tex_size=256):
    """assets/models/FlightHelmet (glTF + .bin + 2048^2 PNGs) -> bevy-hikari_amd/assets/flight_helmet.npz.
    Geometry is kept as is (6 single-primitive meshes, 46 k triangles, u16 indices widened to u32);
    the base-colour (sRGB) and occlusion/roughness/metal (linear) images are box-filtered to
    tex_size^2 RGBA8 to keep the fixture small; normal maps are dropped (the reference binds them but
    never samples them in the light passes).  Material = what bevy_gltf 0.9.1 load_material produces:
    base_color / metallic / roughness factors (defaults 1), textures by index, reflectance 0.5."""
    from PIL import Image

    base = os.path.join(REF, "assets", "models", "FlightHelmet")
    j = json.load(open(os.path.join(base, "FlightHelmet.gltf")))
    blob = open(os.path.join(base, j["buffers"][0]["uri"]), "rb").read()

    def accessor(i):
        a = j["accessors"][i]
        bv = j["bufferViews"][a["bufferView"]]
        fmt, n = _COMP[a["componentType"]], _NCOMP[a["type"]]
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0)
        assert stride in (0, n * np.dtype(fmt).itemsize), "interleaved views are not expected in this asset"
        arr = np.frombuffer(blob, dtype="<" + fmt, count=a["count"] * n, offset=start)
        return arr.reshape(a["count"], n) if n > 1 else arr

    out = {}
    images, image_index = [], {}

    def image(tex_idx, srgb):
        src = j["textures"][tex_idx]["source"]
        if src not in image_index:
            im = Image.open(os.path.join(base, j["images"][src]["uri"])).convert("RGBA")
            im = im.resize((tex_size, tex_size), Image.BOX)
            image_index[src] = len(images)
            images.append((np.asarray(im, dtype=np.uint8), srgb, j["images"][src]["uri"]))
        return image_index[src]

    mats = []
    for m in j["materials"]:
        pbr = m.get("pbrMetallicRoughness", {})
        mats.append(list(pbr.get("baseColorFactor", [1, 1, 1, 1])) + [pbr.get("roughnessFactor", 1.0), pbr.get("metallicFactor", 1.0),
                    image(pbr["baseColorTexture"]["index"], True), image(pbr["metallicRoughnessTexture"]["index"], False),
                    image(m["occlusionTexture"]["index"], False)])
    out["materials"] = np.asarray(mats, dtype=np.float32)  # base rgba, roughness, metallic, base tex, metallic-roughness tex, occlusion tex
    out["textures"] = np.stack([im for im, _, _ in images])
    out["texture_srgb"] = np.asarray([s for _, s, _ in images], dtype=np.uint8)
    instances = []

    def walk(idx, parent):
        n = j["nodes"][idx]
        world = parent @ node_local(n)
        if "mesh" in n:
            instances.append([n["mesh"]] + world.T.astype(np.float32).reshape(-1).tolist())
        for c in n.get("children", []):
            walk(c, world)

    for root in j["scenes"][j.get("scene", 0)]["nodes"]:
        walk(root, np.eye(4))
    out["instances"] = np.asarray(instances, dtype=np.float32)  # mesh id, 16 floats column-major
    ntri = 0
    for k, m in enumerate(j["meshes"]):
        assert len(m["primitives"]) == 1
        p = m["primitives"][0]
        assert p.get("mode", 4) == 4
        out[f"mesh{k}_positions"] = accessor(p["attributes"]["POSITION"]).astype(np.float32)
        out[f"mesh{k}_normals"] = accessor(p["attributes"]["NORMAL"]).astype(np.float32)
        out[f"mesh{k}_uvs"] = accessor(p["attributes"]["TEXCOORD_0"]).astype(np.float32)
        out[f"mesh{k}_indices"] = accessor(p["indices"]).astype(np.uint32)
        out[f"mesh{k}_material"] = np.asarray([p["material"]], dtype=np.uint32)
        ntri += len(out[f"mesh{k}_indices"]) // 3
    path = os.path.join(OUT, "flight_helmet.npz")
    np.savez_compressed(path, **out)
    print(f"flight_helmet: {len(j['meshes'])} meshes, {ntri} tris, {len(images)} textures {tex_size}^2, {os.path.getsize(path) / 1e6:.2f} MB")


def make_cornell_screenshot():
    """assets/screenshots/cornell.png - the ONE output of the reference itself that ships with it (800x600, taken
    from examples/cornell.rs with the orbit camera dollied in and the short box's material edited in the inspector) -
    box-filtered to 200x150 RGB8 for tests/test_reference_screenshot.py.  Not a golden vector (no settings, frame
    count or camera pose are recorded), but it does pin the conventions no oracle of ours could: projection, scene
    transform, emitter strength, tone mapping and display encoding."""
    from PIL import Image

    im = Image.open(os.path.join(REF, "assets", "screenshots", "cornell.png")).convert("RGB").resize((200, 150), Image.BOX)
    out = os.path.join(ROOT, "tests", "golden", "reference_cornell_screenshot_200x150.npz")
    np.savez_compressed(out, rgb=np.asarray(im, dtype=np.uint8))
    print("cornell screenshot:", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; fixtures are already committed")
    os.makedirs(OUT, exist_ok=True)
    make_noise()
    make_cornell()
    make_cornell_bin()
    make_flight_helmet()
    make_cornell_screenshot()
