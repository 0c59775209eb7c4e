"""Host-side mirror of the reference's operator interface for the path.

The reference is a Rust Bevy plugin (no Rust toolchain exists in this image, so the host side
above the C ABI is Python; INTEGRATION.md shows the Rust binding).  Names, fields, defaults and
call order follow cryscan/bevy-hikari v0.3.15:

  HikariSettings / Taa / Upscale / HikariUniversalSettings   src/lib.rs:373-513
  graph.NAME + node names                                     src/lib.rs:43-51
  PrepassNode / LightNode / PostProcessNode  .run()           src/prepass.rs:769, src/light.rs:590, src/post_process.rs:1140
  FrameCounter                                                src/view.rs:75-103
  HikariPlugin                                                src/lib.rs:95-370

All arithmetic of the path runs in libhikari_hip.so; this module only marshals.
"""
import ctypes as C
import dataclasses
import json
import math
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _ffi as F

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

WORKGROUP_SIZE = 8        # lib.rs:53
NOISE_TEXTURE_COUNT = 16  # lib.rs:54


class graph:  # lib.rs:43-51
    NAME = "hikari"

    class node:
        PREPASS = "hikari_prepass"
        LIGHT = "hikari_light"
        POST_PROCESS = "hikari_post_process"
        OVERLAY = "hikari_overlay"


class Taa:  # lib.rs:466-472
    Jasmine = F.TAA_JASMINE
    NONE = F.TAA_NONE


@dataclass(frozen=True)
class Upscale:  # lib.rs:474-513
    kind: int = F.UPSCALE_SMAA_TU4X
    ratio_: float = 2.0
    sharpness_: float = 0.0

    @staticmethod
    def Fsr1(ratio, sharpness):
        return Upscale(F.UPSCALE_FSR1, ratio, sharpness)

    @staticmethod
    def SmaaTu4x(ratio):
        return Upscale(F.UPSCALE_SMAA_TU4X, ratio, 0.0)

    def ratio(self):
        return min(max(self.ratio_, 1.0), 2.0)

    def sharpness(self):
        return self.sharpness_ if self.kind == F.UPSCALE_FSR1 else 0.0


Upscale.SMAA_TU_1_0 = Upscale.SmaaTu4x(1.0)
Upscale.SMAA_TU_2_0 = Upscale.SmaaTu4x(2.0)


@dataclass
class HikariUniversalSettings:  # lib.rs:373-389
    build_mesh_acceleration_structure: bool = True
    build_instance_acceleration_structure: bool = True


@dataclass
class HikariSettings:  # lib.rs:400-455 (field order and defaults)
    direct_validate_interval: int = 3
    emissive_validate_interval: int = 5
    max_temporal_reuse_count: int = 50
    max_spatial_reuse_count: int = 800
    max_reservoir_lifetime: float = 100.0
    solar_angle: float = 0.046
    indirect_bounces: int = 1
    max_indirect_luminance: float = 10.0
    clear_color: tuple = (0.4, 0.4, 0.4, 1.0)
    temporal_reuse: bool = True
    emissive_spatial_reuse: bool = False
    indirect_spatial_reuse: bool = True
    denoise: bool = True
    taa: int = Taa.Jasmine
    upscale: Upscale = Upscale.SMAA_TU_2_0

    def to_c(self):
        s = F.HkSettings()
        s.direct_validate_interval = self.direct_validate_interval
        s.emissive_validate_interval = self.emissive_validate_interval
        s.max_temporal_reuse_count = self.max_temporal_reuse_count
        s.max_spatial_reuse_count = self.max_spatial_reuse_count
        s.max_reservoir_lifetime = self.max_reservoir_lifetime
        s.solar_angle = self.solar_angle
        s.indirect_bounces = self.indirect_bounces
        s.max_indirect_luminance = self.max_indirect_luminance
        s.clear_color[:] = list(self.clear_color)
        s.temporal_reuse = int(self.temporal_reuse)
        s.emissive_spatial_reuse = int(self.emissive_spatial_reuse)
        s.indirect_spatial_reuse = int(self.indirect_spatial_reuse)
        s.denoise = int(self.denoise)
        s.taa = self.taa
        s.upscale_kind = self.upscale.kind
        s.upscale_ratio = self.upscale.ratio_
        s.upscale_sharpness = self.upscale.sharpness_
        return s


# ---------------------------------------------------------------------------------------------
# 4x4 helpers in plain IEEE doubles, column-major flat lists m[c*4+r] (glam conventions).  Written
# with explicit loops in a fixed order so that the C++ host mirror (include/hikari.hpp) produces the
# SAME f32 uniforms - no BLAS/LAPACK whose summation order could differ in the last bit.
# ---------------------------------------------------------------------------------------------
def _mul4(a, b):
    o = [0.0] * 16
    for c in range(4):
        for r in range(4):
            s = 0.0
            for k in range(4):
                s += a[k * 4 + r] * b[c * 4 + k]
            o[c * 4 + r] = s
    return o


def look_at_transform(eye, target, up=(0.0, 1.0, 0.0)):
    """Transform::from_translation(eye).looking_at(target, up) as a column-major camera-to-world matrix."""
    eye, target, up = ([float(x) for x in v] for v in (eye, target, up))

    def norm(v):
        l = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        return [v[0] / l, v[1] / l, v[2] / l]

    def cross(a, b):
        return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]

    f = norm([target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]])
    r = norm(cross(f, up))
    u = cross(r, f)
    return [r[0], r[1], r[2], 0.0, u[0], u[1], u[2], 0.0, -f[0], -f[1], -f[2], 0.0, eye[0], eye[1], eye[2], 1.0]


def _f32(m):
    return np.asarray(m, dtype=np.float64).astype(np.float32)


@dataclass
class Camera:
    """Camera3dBundle with bevy's default PerspectiveProjection (fov pi/4, near 0.1, infinite reverse-Z), or -
    with ortho_height - bevy 0.9's OrthographicProjection (ScalingMode::FixedVertical(ortho_height), near 0,
    far 1000, reverse-Z): the `view.projection[3].w == 1.0` branch of light.wgsl:714-727,1040."""
    transform: list  # column-major 4x4 camera-to-world (16 doubles)
    width: int
    height: int
    fov: float = 0.78539816339744830962
    near: float = 0.1
    ortho_height: Optional[float] = None
    ortho_far: float = 1000.0

    def projection(self):  # Mat4::perspective_infinite_reverse_rh / Mat4::orthographic_rh(l, r, b, t, far, near)
        if self.ortho_height is not None:
            half_h = 0.5 * self.ortho_height
            half_w = half_h * float(self.width) / float(self.height)
            rcp_w, rcp_h, r = 1.0 / (2.0 * half_w), 1.0 / (2.0 * half_h), 1.0 / (self.ortho_far - 0.0)
            m = [0.0] * 16
            m[0], m[5], m[10] = 2.0 * rcp_w, 2.0 * rcp_h, r
            m[12], m[13], m[14], m[15] = 0.0, 0.0, r * self.ortho_far, 1.0
            return m
        f = 1.0 / math.tan(0.5 * self.fov)
        aspect = float(self.width) / float(self.height)
        m = [0.0] * 16
        m[0], m[5], m[11], m[14] = f / aspect, f, -1.0, self.near
        return m

    def inverse_projection(self):
        p = self.projection()
        if self.ortho_height is not None:  # diagonal + translation in z
            m = [0.0] * 16
            m[0], m[5], m[10], m[14], m[15] = 1.0 / p[0], 1.0 / p[5], 1.0 / p[10], -p[14] / p[10], 1.0
            return m
        m = [0.0] * 16
        m[0], m[5], m[11], m[14] = 1.0 / p[0], 1.0 / p[5], 1.0 / p[14], -1.0
        return m

    def inverse_view(self):  # rigid inverse: [R^T | -R^T t]
        t = [float(x) for x in np.asarray(self.transform, dtype=np.float64).reshape(-1)]
        m = [0.0] * 16
        for c in range(3):
            for r in range(3):
                m[c * 4 + r] = t[r * 4 + c]
        for r in range(3):
            m[12 + r] = -(t[r * 4 + 0] * t[12] + t[r * 4 + 1] * t[13] + t[r * 4 + 2] * t[14])
        m[15] = 1.0
        return m

    def view_uniform(self):  # bevy_render 0.9.1 ViewUniform
        t = [float(x) for x in np.asarray(self.transform, dtype=np.float64).reshape(-1)]
        proj, inv_view, inv_proj = self.projection(), self.inverse_view(), self.inverse_projection()
        v = F.HkView()
        v.view_proj[:] = _f32(_mul4(proj, inv_view))
        v.inverse_view_proj[:] = _f32(_mul4(t, inv_proj))  # (P * V^-1)^-1 = V * P^-1
        v.view[:] = _f32(t)
        v.inverse_view[:] = _f32(inv_view)
        v.projection[:] = _f32(proj)
        v.inverse_projection[:] = _f32(inv_proj)
        v.world_position[:] = _f32(t[12:15])
        v.viewport[:] = [0.0, 0.0, float(self.width), float(self.height)]
        return v

    def previous_view_uniform(self, previous: Optional["Camera"] = None):
        cam = previous or self
        v = cam.view_uniform()
        p = F.HkPreviousView()
        p.view_proj[:] = list(v.view_proj)
        p.inverse_view_proj[:] = list(v.inverse_view_proj)
        return p


def lights_uniform(directional=None, ambient_color=(1.0, 1.0, 1.0), ambient_brightness=0.05):
    """bevy AmbientLight default (white, 0.05) and an optional DirectionalLight
    dict(color=(r,g,b) linear, illuminance=lux, direction_to_light=(x,y,z)); bevy scales the colour
    by illuminance / 4800 (exposure of the default camera) before upload."""
    l = F.HkLights()
    l.ambient_color[:] = [c * ambient_brightness for c in ambient_color] + [ambient_brightness]
    if directional:
        scale = directional.get("illuminance", 100000.0) / 4800.0  # bevy_pbr 0.9.1 light.rs exposure
        col = [c * scale for c in directional.get("color", (1.0, 1.0, 1.0))]
        l.directional_color[:] = col + [1.0]
        d = np.asarray(directional["direction_to_light"], dtype=np.float64)
        d = d / np.linalg.norm(d)
        l.direction_to_light[:] = d.astype(np.float32)
        l.n_directional_lights = 1
    return l


# ---------------------------------------------------------------------------------------------
# scene description -> builder
# ---------------------------------------------------------------------------------------------
def linear_to_srgb(c):
    c = float(c)
    return 12.92 * c if c <= 0.0031308 else 1.055 * (c ** (1.0 / 2.4)) - 0.055


def standard_material(base_color_linear=(1, 1, 1, 1), emissive_linear=(0, 0, 0), perceptual_roughness=0.5, metallic=0.0,
                      reflectance=0.5, nonlinear=False):
    """StandardMaterial -> GpuStandardMaterial as material.rs:168-199 does it: `Color -> Vec4` goes
    through bevy 0.9's `From<Color> for Vec4` = as_rgba_f32().  bevy_gltf 0.9.1 builds the colours
    with `Color::rgba(factor)`, for which as_rgba_f32() is the identity, so glTF factors reach the
    GPU buffer unchanged (default).  `nonlinear=True` applies the linear->sRGB encoding that the same
    conversion performs for `Color::rgba_linear` inputs.  Either way these values are INPUTS of the
    boundary; the library never converts colours."""
    m = F.HkMaterial()
    enc = linear_to_srgb if nonlinear else float
    bc = list(base_color_linear) + [1.0] * (4 - len(base_color_linear))
    m.base_color[:] = [enc(bc[0]), enc(bc[1]), enc(bc[2]), float(bc[3])]
    em = list(emissive_linear)
    m.emissive[:] = [enc(em[0]), enc(em[1]), enc(em[2]), 1.0]
    m.base_color_texture = m.emissive_texture = m.metallic_roughness_texture = F.NO_TEXTURE
    m.normal_map_texture = m.occlusion_texture = F.NO_TEXTURE
    m.perceptual_roughness, m.metallic, m.reflectance = perceptual_roughness, metallic, reflectance
    return m


class SceneData:
    """The nine storage buffers of bind group 2 (mesh_material_bindings.wgsl:5-22) as ctypes arrays."""

    FIELDS = ("vertices", "primitives", "asset_nodes", "materials", "instances", "instance_nodes", "emissives", "emissive_nodes",
              "alias_table")

    def __init__(self, previous_transforms=None, **arrays):
        for k in self.FIELDS:
            setattr(self, k, arrays[k])
        #: float32[n_instances][16] or None: PreviousMeshUniform::transform per instance (instance.rs:111-128)
        self.previous_transforms = previous_transforms

    def upload(self, api, ctx):
        n = lambda a: len(a)
        api.call("upload_meshes", ctx, self.vertices, n(self.vertices), self.primitives, n(self.primitives), self.asset_nodes, n(self.asset_nodes))
        api.call("upload_materials", ctx, self.materials, n(self.materials))
        self.upload_instances(api, ctx)

    def upload_instances(self, api, ctx):
        """The instance-level buffers only (what prepare_instances rewrites when something moves)."""
        n = lambda a: len(a)
        api.call("upload_instances", ctx, self.instances, n(self.instances), self.instance_nodes, n(self.instance_nodes), self.emissives,
                 n(self.emissives), self.emissive_nodes, n(self.emissive_nodes), self.alias_table, n(self.alias_table))
        if self.previous_transforms is not None:
            pt = np.ascontiguousarray(self.previous_transforms, dtype=np.float32).reshape(-1, 16)
            assert len(pt) == n(self.instances)
            api.call("upload_previous_transforms", ctx, pt.ctypes.data_as(C.POINTER(F.f32)), len(pt))


class SceneBuilder:
    """hk_scene_builder_*: the Prepare-stage host work (mesh -> BLAS, TLAS, emissives, alias tables)."""

    def __init__(self):
        self.api = F.api()
        self.h = C.c_void_p()
        self.api.call("scene_builder_create", C.byref(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.api.raw("scene_builder_destroy")(self.h)
            self.h = None

    def add_mesh(self, positions, normals, uvs, indices=None, topology=F.TOPOLOGY_TRIANGLE_LIST):
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        uv = np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 2)
        out = F.u32()
        fp = lambda a: a.ctypes.data_as(C.POINTER(F.f32))
        if indices is None:
            self.api.call("scene_builder_add_mesh", self.h, fp(pos), fp(nrm), fp(uv), len(pos), None, 0, topology, C.byref(out))
        else:
            idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
            self.api.call("scene_builder_add_mesh", self.h, fp(pos), fp(nrm), fp(uv), len(pos), idx.ctypes.data_as(C.POINTER(F.u32)), len(idx),
                          topology, C.byref(out))
        return out.value

    def add_material(self, material):
        out = F.u32()
        self.api.call("scene_builder_add_material", self.h, C.byref(material), C.byref(out))
        return out.value

    def add_instance(self, mesh_id, material_id, transform):
        t = np.ascontiguousarray(transform, dtype=np.float32).reshape(-1)
        out = F.u32()
        self.api.call("scene_builder_add_instance", self.h, mesh_id, material_id, t.ctypes.data_as(C.POINTER(F.f32)), C.byref(out))
        return out.value

    def set_instance_transform(self, instance_id, transform):
        """Move an instance; the next finish() redoes the instance-level work only."""
        t = np.ascontiguousarray(transform, dtype=np.float32).reshape(-1)
        self.api.call("scene_builder_set_instance_transform", self.h, instance_id, t.ctypes.data_as(C.POINTER(F.f32)))

    def remove_instance(self, instance_id):
        """Take an instance out of the scene (ids of later instances shift down by one)."""
        self.api.call("scene_builder_remove_instance", self.h, instance_id)

    def set_instance_material(self, instance_id, material_id):
        self.api.call("scene_builder_set_instance_material", self.h, instance_id, material_id)

    def finish(self, build_trees=True):
        """hk_scene_builder_finish; build_trees=False: hk_scene_builder_finish_instances (stand-in trees, for a device-side build)."""
        self.api.call("scene_builder_finish" if build_trees else "scene_builder_finish_instances", self.h)
        arrays = {}
        for name, typ in (("vertices", F.HkVertex), ("primitives", F.HkPrimitive), ("asset_nodes", F.HkNode), ("materials", F.HkMaterial),
                          ("instances", F.HkInstance), ("instance_nodes", F.HkNode), ("emissives", F.HkEmissive),
                          ("emissive_nodes", F.HkNode), ("alias_table", F.HkAliasEntry)):
            p, n = C.POINTER(typ)(), F.u32()
            self.api.call("scene_builder_" + name, self.h, C.byref(p), C.byref(n))
            arr = (typ * n.value)()
            if n.value:
                C.memmove(arr, p, n.value * C.sizeof(typ))
            arrays[name] = arr
        p, n = C.POINTER(F.f32)(), F.u32()
        self.api.call("scene_builder_previous_transforms", self.h, C.byref(p), C.byref(n))
        prev = np.ctypeslib.as_array(p, shape=(n.value, 16)).copy() if n.value else np.zeros((0, 16), np.float32)
        return SceneData(previous_transforms=prev, **arrays)


def load_cornell(nonlinear_colors=False):
    """examples/cornell.rs:37-41: the glTF scene, flattened by tools/make_fixtures.py."""
    with open(os.path.join(ASSETS, "cornell.json")) as f:
        j = json.load(f)
    b = SceneBuilder()
    mats = []
    for m in j["materials"]:
        # bevy_gltf 0.9.1 load_material: base_color = Color::rgba_linear(factor), emissive = Color::rgb_linear(factor),
        # perceptual_roughness = roughness factor, metallic = metallic factor, reflectance default 0.5
        mats.append(b.add_material(standard_material(m["base_color_factor"], m["emissive_factor"], m["roughness_factor"],
                                                     m["metallic_factor"], 0.5, nonlinear_colors)))
    meshes = [b.add_mesh(m["positions"], m["normals"], m["uvs"], m["indices"]) for m in j["meshes"]]
    for inst in j["instances"]:
        b.add_instance(meshes[inst["mesh"]], mats[j["meshes"][inst["mesh"]]["material"]], inst["transform"])
    scene = b.finish()
    scene.builder = b  # kept for moving instances: builder.set_instance_transform(i, ..) + Engine.refit_instances(builder) / builder.finish()
    return scene


def cornell_camera(width, height):  # examples/cornell.rs:49-50
    return Camera(look_at_transform((0.0, 1.0, 4.0), (0.0, 1.0, 0.0)), width, height)


def load_noise():
    path = os.path.join(ASSETS, "noise_rgba8_16x64x64.bin")
    data = np.fromfile(path, dtype=np.uint8)
    assert data.size == 16 * 64 * 64 * 4
    return data


# ---------------------------------------------------------------------------------------------
# engine: one hk_ctx
# ---------------------------------------------------------------------------------------------
_BUF_DTYPES = {16: (np.float32, 4), 4: (np.uint32, 1), 8: (np.uint16, 4), 64: (np.uint32, 16)}


class Engine:
    """One context of the C ABI (`hk_ctx`).  `api` defaults to the product library."""

    def __init__(self, api=None, device=0, flags=0):
        self.api = api or F.api()
        self.ctx = C.c_void_p()
        self.api.call("create", device, flags | F.DEFAULT_CTX_FLAGS, C.byref(self.ctx))
        self.width = self.height = 0
        self.owned = True
        self.generation = 0  # bumped by resize(): holders of device pointers / views compare it (distributed.BandRenderer)

    @classmethod
    def borrowed(cls, api, ctx):
        """An Engine over a context somebody else owns (hk_multi_context): never destroyed from here."""
        e = cls.__new__(cls)
        e.api, e.ctx, e.owned, e.generation, e.width, e.height = api, ctx, False, 0, 0, 0
        return e

    def close(self):
        if self.ctx and getattr(self, "owned", True):
            self.api.raw("destroy")(self.ctx)
        self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- Prepare stage
    def upload_scene(self, scene: SceneData):
        self.upload_textures(getattr(scene, "textures", []))
        scene.upload(self.api, self.ctx)

    def upload_instances(self, scene: SceneData):
        """Instance-level update of a scene whose meshes and materials are already uploaded."""
        scene.upload_instances(self.api, self.ctx)

    def refit_instances(self, builder):
        """Instance motion on the device (hk_refit_scene_instances): the poses set on `builder` since the last upload / refit go to
        the GPU, which redoes the per-instance work and refits both trees.  Returns the number of instances that moved."""
        moved = C.c_uint32()
        self.api.call("refit_scene_instances", self.ctx, builder.h, C.byref(moved))
        return moved.value

    def update_instances_on_device(self, builder, mode=F.TREE_SAH):
        """Instances added / removed / re-materialed on `builder`: hk_update_scene_instances (host lays out the records, the device
        builds both trees)."""
        self.api.call("update_scene_instances", self.ctx, builder.h, mode)

    def rebuild_trees(self, mode=F.TREE_SAH):
        """New instance tree and light tree over the current boxes, built on the device (hk_rebuild_scene_trees): F.TREE_SAH = the
        reference's own binned-SAH tree, F.TREE_LBVH = the quick Morton-order tree."""
        self.api.call("rebuild_scene_trees", self.ctx, mode)

    def read_trees(self, n_instance_nodes, n_emissive_nodes):
        """(instance_nodes, emissive_nodes) as the device holds them, in the reference layout (test hook)."""
        a, b = (F.HkNode * max(1, n_instance_nodes))(), (F.HkNode * max(1, n_emissive_nodes))()
        self.api.call("debug_read_trees", self.ctx, a, n_instance_nodes, b, n_emissive_nodes)
        return (F.HkNode * n_instance_nodes).from_buffer_copy(bytes(a)[:n_instance_nodes * C.sizeof(F.HkNode)]), \
               (F.HkNode * n_emissive_nodes).from_buffer_copy(bytes(b)[:n_emissive_nodes * C.sizeof(F.HkNode)])

    def upload_textures(self, images):
        """images: list of dict(rgba=uint8[h][w][4], srgb=bool, address_u/address_v=F.ADDRESS_*, linear=bool) -
        the `textures` / `samplers` binding arrays (mod.rs:760-782).  Material *_texture ids index this list."""
        descs = (F.HkImageDesc * max(len(images), 1))()
        keep = []
        for d, im in zip(descs, images):
            a = np.ascontiguousarray(im["rgba"], dtype=np.uint8)
            assert a.ndim == 3 and a.shape[2] == 4
            keep.append(a)
            d.rgba8, d.height, d.width = a.ctypes.data, a.shape[0], a.shape[1]
            d.is_srgb, d.filter_linear = int(im.get("srgb", True)), int(im.get("linear", True))
            d.address_u, d.address_v = im.get("address_u", F.ADDRESS_REPEAT), im.get("address_v", F.ADDRESS_REPEAT)
        self.api.call("upload_textures", self.ctx, descs, len(images))

    def upload_noise(self, noise=None):
        noise = load_noise() if noise is None else np.ascontiguousarray(noise, dtype=np.uint8)
        self.api.call("upload_noise", self.ctx, noise.ctypes.data, noise.size)

    def resize(self, width, height, upscale_ratio=1.0):
        self.api.call("resize", self.ctx, width, height, upscale_ratio)
        self.width, self.height = width, height
        self._bounds = None   # (hk_resize drops an explicit band split: it was in rows of the old render image)
        self.generation += 1  # every buffer was freed and reallocated: device pointers / views of an older generation are dead

    # -- per frame
    def frame_begin(self, frame, view, previous_view, lights):
        self.api.call("frame_begin", self.ctx, C.byref(frame), C.byref(view), C.byref(previous_view), C.byref(lights))

    def set_view_options(self, taa, upscale_kind, upscale_sharpness=0.0):
        self.api.call("set_view_options", self.ctx, taa, upscale_kind, upscale_sharpness)

    def pass_run(self, pass_id, arg=0, row_begin=0, row_end=0):
        self.api.call("pass_run", self.ctx, pass_id, arg, row_begin, row_end)

    def frame_stage(self, stage, settings_c, flags=0):
        self.api.call("frame_stage", self.ctx, stage, C.byref(settings_c), flags)

    def frame_render(self, frame, view, previous_view, lights, settings_c, flags=0):
        self.api.call("frame_render", self.ctx, C.byref(frame), C.byref(view), C.byref(previous_view), C.byref(lights), C.byref(settings_c), flags)

    def wait(self):
        self.api.call("frame_wait", self.ctx)

    def set_band(self, index, count):
        self.api.call("set_band", self.ctx, index, count)
        self._band = (index, count)

    def band_count(self):
        if "get_band" in self.api._fns:
            n = F.u32(0)
            self.api.call("get_band", self.ctx, None, C.byref(n))
            return n.value
        return getattr(self, "_band", (0, 1))[1]

    def set_band_bounds(self, bounds=None):
        """hk_set_band_bounds: band i renders scaled render rows [bounds[i], bounds[i + 1]); None = the equal split."""
        self._bounds = None if bounds is None else [int(b) for b in bounds]
        if bounds is None:
            self.api.call("set_band_bounds", self.ctx, None, 0)
            return
        arr = (C.c_uint32 * len(bounds))(*[int(b) for b in bounds])
        self.api.call("set_band_bounds", self.ctx, arr, len(bounds))

    def band_bounds(self):
        """hk_get_band_bounds: the split in force (band_count + 1 scaled render rows), explicit or equal."""
        n = self.band_count() + 1
        if "get_band_bounds" not in self.api._fns:   # (the oracle behind the same table)
            return getattr(self, "_bounds", None)
        out = (C.c_uint32 * n)()
        self.api.call("get_band_bounds", self.ctx, out, n)
        return [int(x) for x in out]

    def balance_bands(self, min_rows=0):
        """hk_balance_bands (after frame_begin): ray-cast the whole frame's primary rays, count geometry pixels per row, split the
        render rows by cost and keep the split.  Returns the boundaries - the same on every rank that does this for the frame."""
        _w, rh, _b = self.buffer_info(F.BUF_TONE_MAPPED)
        n = self.band_count() + 1
        if "balance_bands" in self.api._fns:
            out = (C.c_uint32 * n)()
            self.api.call("balance_bands", self.ctx, min_rows, out, n)
            return [int(x) for x in out]
        # (a library without the composite - the oracle behind the same ctypes table in the CPU tests: the same three steps)
        from .distributed import balanced_band_bounds

        if n == 2:
            return [0, rh]
        self.pass_run(F.PASS_PREPASS)
        w, _h, _b = self.buffer_info(F.BUF_POSITION)
        bounds = balanced_band_bounds(self.row_costs(), w, rh, n - 1, max(1, min(min_rows or 8, rh // (n - 1))), 0.25)   # (the CPU test scenes are LDS-sized)
        self.set_band_bounds(bounds)
        return bounds

    def migrate_bands(self, new_bounds, next_frame_number, settings_c):
        """hk_migrate_bands (needs a communicator): the history rows that change owner travel as one RCCL exchange, then the new split is
        set.  Every rank calls it with the same arguments between two frames.  new_bounds None = the equal split."""
        if new_bounds is None:
            self.api.call("migrate_bands", self.ctx, None, 0, int(next_frame_number), C.byref(settings_c))
        else:
            arr = (C.c_uint32 * len(new_bounds))(*[int(b) for b in new_bounds])
            self.api.call("migrate_bands", self.ctx, arr, len(new_bounds), int(next_frame_number), C.byref(settings_c))

    def band_time_ms(self):
        """hk_band_time_ms: the band's own GPU time (stage TEMPORAL + stage SPATIAL, without the waits for its neighbours' halos) in the
        last frame rendered with F.FRAME_TIME_BAND."""
        ms = C.c_float()
        self.api.call("band_time_ms", self.ctx, C.byref(ms))
        return float(ms.value)

    def row_costs(self):
        """hk_row_costs: geometry pixels per full-size row of the frame most recently begun (numpy uint32[height])."""
        _w, h, _b = self.buffer_info(F.BUF_POSITION)
        out = np.zeros(h, dtype=np.uint32)
        self.api.call("row_costs", self.ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)), h)
        return out

    # -- buffers
    def buffer_info(self, buf):
        w, h, bpp = F.u32(), F.u32(), F.u32()
        self.api.call("buffer_info", self.ctx, buf, C.byref(w), C.byref(h), C.byref(bpp))
        return w.value, h.value, bpp.value

    def _buffer_layout(self, buf):
        w, h, bpp = self.buffer_info(buf)
        if buf in (F.BUF_DEPTH_GRADIENT, F.BUF_INSTANCE_MATERIAL):
            dt, k = np.float32, 2
        elif bpp == 4 and buf != F.BUF_NORMAL:
            dt, k = np.float32, 1
        else:
            dt, k = _BUF_DTYPES[bpp]
        return h, w, k, dt

    def read(self, buf):
        """Raw contents as a numpy array [h, w, k] (f32 for 16/8-byte float formats, u16 for rgba16f, u32 otherwise)."""
        h, w, k, dt = self._buffer_layout(buf)
        out = np.empty((h, w, k), dtype=dt)
        self.api.call("read_buffer", self.ctx, buf, out.ctypes.data, out.nbytes)
        return out

    def shape_buffer(self, buf, raw_bytes):
        """Raw bytes of `buf` (e.g. from hk_multi_read_buffer) as the array Engine.read would return."""
        h, w, k, dt = self._buffer_layout(buf)
        return np.frombuffer(raw_bytes, dtype=dt).reshape(h, w, k).copy()

    def read_f16(self, buf):
        return self.read(buf).view(np.float16).astype(np.float32)

    def write(self, buf, array):
        a = np.ascontiguousarray(array)
        self.api.call("write_buffer", self.ctx, buf, a.ctypes.data, a.nbytes)

    def device_ptr(self, buf):
        p, n = C.c_void_p(), C.c_size_t()
        self.api.call("device_ptr", self.ctx, buf, C.byref(p), C.byref(n))
        return p.value, n.value

    def allocated_bytes(self, buf):
        return self.device_ptr(buf)[1]

    # -- halo exchange inside the library (one process per GPU; bevy-hikari_amd/distributed.py does the rendezvous)
    def comm_available(self):
        """hk_comm_available as (ok, reason): never raises, so that every rank reaches the agreement step."""
        try:
            self.api.call("comm_available", self.ctx)
            return True, ""
        except Exception as err:  # HikariError from the library; KeyError / AttributeError behind a library without RCCL (the oracle)
            return False, repr(err)

    def comm_unique_id(self):
        ident = (C.c_uint8 * 128)()
        self.api.call("comm_unique_id", ident)
        return bytes(ident)

    def comm_init(self, rank, n_ranks, ident):
        buf = (C.c_uint8 * 128).from_buffer_copy(ident)
        self.api.call("comm_init", self.ctx, rank, n_ranks, buf)

    def comm_set_history_rows(self, rows):
        self.set_history_rows(rows)

    def set_history_rows(self, rows=F.HISTORY_AUTO):
        """hk_set_history_rows: rows of last frame's reservoirs a band fetches from its neighbours before TEMPORAL (and, doubled,
        of the parked scatter stores before SPATIAL).  F.HISTORY_AUTO (the default of a context): derived per frame."""
        self.api.call("set_history_rows", self.ctx, int(rows))

    def history_rows(self):
        """hk_history_rows: the count in force for the frame most recently begun (0 for a single band or a static view)."""
        n = F.u32(0)
        self.api.call("history_rows", self.ctx, C.byref(n))
        return n.value

    def scene_bounds(self):
        mn, mx = (F.f32 * 3)(), (F.f32 * 3)()
        self.api.call("scene_bounds", self.ctx, mn, mx)
        return list(mn), list(mx)

    def comm_gather(self, buffer, root=0):
        """hk_comm_gather: rank `root` collects every other rank's rows of `buffer` (RCCL, on the context's stream)."""
        self.api.call("comm_gather", self.ctx, buffer, root)

    def debug_comm_loopback(self, src_buffer, dst_buffer, row_begin, row_end, overlapped=False):
        """hk_debug_comm_loopback (hikari_hip_debug.h): rows of one buffer to the same rows of another through ncclSend / ncclRecv
        to the context's own rank, on the context's stream."""
        self.api.call("debug_comm_loopback", self.ctx, src_buffer, dst_buffer, row_begin, row_end, 1 if overlapped else 0)

    def comm_destroy(self):
        self.api.call("comm_destroy", self.ctx)

    def stats(self):
        s = F.HkStats()
        self.api.call("get_stats", self.ctx, C.byref(s))
        return s

    def reset_stats(self):
        self.api.call("reset_stats", self.ctx)

    def stream(self):
        """The HIP stream the context enqueues on (hk_stream), as an integer handle."""
        h = C.c_void_p()
        self.api.call("stream", self.ctx, C.byref(h))
        return h.value or 0

    def set_stream(self, hip_stream_ptr):
        """Run all subsequent work on a host-owned HIP stream (e.g. torch.cuda.current_stream().cuda_stream)."""
        self.api.call("set_stream", self.ctx, C.c_void_p(hip_stream_ptr))

    def set_timing_mask(self, mask):
        self.api.call("set_timing_mask", self.ctx, mask)

    def indirect_schedule(self):
        """'fused' or 'wavefront': the schedule indirect_lit_ambient takes for the frame most recently begun (HK_CTX_WAVEFRONT)."""
        v = C.c_uint32()
        self.api.call("indirect_schedule", self.ctx, C.byref(v))
        return "wavefront" if v.value else "fused"

    def traversal_mode(self):
        """('reference' | 'threaded' | 'one-level', stored direction orderings): hk_traversal_mode."""
        v, n = C.c_uint32(), C.c_uint32()
        self.api.call("traversal_mode", self.ctx, C.byref(v), C.byref(n))
        return ("reference", "threaded", "one-level")[v.value & 0xFF], n.value

    def wide_walk(self):
        """True when the closest-hit walks of the uploaded scene take the wide records (HK_TRAVERSAL_WIDE; HK_CTX_NO_WIDE_WALK)."""
        v = C.c_uint32()
        self.api.call("traversal_mode", self.ctx, C.byref(v), None)
        return bool(v.value & 0x100)

    def set_debug_option(self, option, value):
        """hikari_hip_debug.h hk_debug_set_option (F.DEBUG_OPT_*): the switches of tests and A/B tools - the library reads no environment variable."""
        self.api.call("debug_set_option", self.ctx, option, int(value))

    def main_stream_priority(self):
        """(high, decided): whether the context's own main stream was created at the device's highest priority, and whether the rule has
        decided yet - it does at the context's first frame (hikari_hip_debug.h hk_debug_main_stream_priority)."""
        v = F.u32()
        self.api.call("debug_main_stream_priority", self.ctx, C.byref(v))
        return bool(v.value & 1), bool(v.value & 2)

    def prepasses_pipelined(self):
        """frames whose primary rays ran on their own stream beside the previous frame's spatial pass (hk_debug_main_stream_priority, bits 8..27)"""
        v = F.u32()
        self.api.call("debug_main_stream_priority", self.ctx, C.byref(v))
        return int(v.value >> 8)

    def spatial_windowed_launches(self):
        """spatial_reuse launches that took the windowed form of the kernel (hikari_hip_debug.h; F.DEBUG_OPT_SPATIAL_WINDOW)."""
        n = C.c_uint64()
        self.api.call("debug_spatial_windowed_launches", self.ctx, C.byref(n))
        return int(n.value)

    def measure_hbm(self, bytes_per_array=1 << 30, reps=8):
        """Empirical HBM ceiling: (copy GB/s, triad GB/s) of grid-stride float4 streams over arrays too big for the Infinity Cache."""
        cp, tr = C.c_double(), C.c_double()
        self.api.call("measure_hbm", self.ctx, bytes_per_array, reps, C.byref(cp), C.byref(tr))
        return cp.value, tr.value

    def measure_gather(self, footprint_bytes, bytes_per_step=32, waves_per_simd=7, steps=512, workgroups=0):
        """hk_measure_gather: (G wave-level loads / s, GB/s) of dependent divergent gathers over `footprint_bytes` of records."""
        a, b = C.c_double(), C.c_double()
        self.api.call("measure_gather", self.ctx, int(footprint_bytes), bytes_per_step, waves_per_simd, steps, workgroups, C.byref(a), C.byref(b))
        return a.value, b.value

    def measure_valu(self, iters=2048):
        """VALU issue ceiling: {waves per SIMD: 1e9 wave64 instructions per second} for 1, 2, 4, 8 resident waves per SIMD."""
        r = (C.c_double * 4)()
        self.api.call("measure_valu", self.ctx, iters, r)
        return {1 << k: r[k] for k in range(4)}

    def debug_math(self, op, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        fp = lambda a: a.ctypes.data_as(C.POINTER(F.f32))
        yy = None if y is None else np.ascontiguousarray(y, dtype=np.float32)
        self.api.call("debug_math", self.ctx, op, fp(x), None if yy is None else fp(yy), fp(out), x.size)
        return out


# ---------------------------------------------------------------------------------------------
# the reference's nodes and plugin
# ---------------------------------------------------------------------------------------------
class FrameCounter:  # view.rs:75-103: inserted as 0 on a new camera, +1 every frame
    def __init__(self, value=0):
        self.value = value

    def tick(self):
        self.value += 1
        return self.value


def frame_uniform(settings: HikariSettings, frame_number: int):
    """FrameUniform::extract_component (view.rs:141-193), evaluated by the library."""
    f = F.HkFrame()
    s = settings.to_c()
    F.api().call("frame_from_settings", C.byref(s), frame_number, C.byref(f))
    return f


class _Node:
    IN_VIEW = "view"  # light.rs:572

    def __init__(self, engine: Engine):
        self.engine = engine


class PrepassNode(_Node):  # prepass.rs:736-852
    def run(self, settings: HikariSettings):
        self.engine.set_view_options(settings.taa, settings.upscale.kind, settings.upscale.sharpness_)
        self.engine.pass_run(F.PASS_PREPASS)


class LightNode(_Node):  # light.rs:557-703
    def run(self, settings: HikariSettings):
        e = self.engine
        e.pass_run(F.PASS_FULL_SCREEN_ALBEDO)                      # light.rs:646-653
        e.pass_run(F.PASS_DIRECT_LIT)                              # light.rs:656-688, render[0], reservoirs (0,4)
        e.pass_run(F.PASS_DIRECT_EMISSIVE)                         # render[1], reservoirs (2,4)
        if settings.emissive_spatial_reuse:                        # light.rs:675,689-697
            e.pass_run(F.PASS_EMISSIVE_SPATIAL_REUSE)
        e.pass_run(F.PASS_INDIRECT)                                # render[2], reservoirs (6,8)
        if settings.indirect_spatial_reuse:                        # light.rs:676
            e.pass_run(F.PASS_INDIRECT_SPATIAL_REUSE)


class PostProcessNode(_Node):  # post_process.rs:1107-1312
    def run(self, settings: HikariSettings, antialias=False):
        e = self.engine
        if settings.denoise:                                       # post_process.rs:1190-1224
            channels = 2 if settings.indirect_bounces == 0 else 3  # post_process.rs:949-954
            for ch in range(channels):
                e.pass_run(F.PASS_DEMODULATION, ch)
                for level in range(4):
                    e.pass_run(F.PASS_DENOISE_L0 + level, ch)
        e.pass_run(F.PASS_TONE_MAPPING, int(settings.denoise))     # post_process.rs:1226-1234
        if antialias:
            self.run_antialias(settings)

    def run_antialias(self, settings: HikariSettings):
        e = self.engine
        e.set_view_options(settings.taa, settings.upscale.kind, settings.upscale.sharpness_)
        if settings.upscale.kind == F.UPSCALE_SMAA_TU4X:           # post_process.rs:1236-1258
            e.pass_run(F.PASS_SMAA_TU4X)
            e.pass_run(F.PASS_SMAA_TU4X_EXTRAPOLATE)
        if settings.taa == Taa.Jasmine:                            # post_process.rs:1260-1275
            e.pass_run(F.PASS_TAA_JASMINE)
        if settings.upscale.kind == F.UPSCALE_FSR1:                # post_process.rs:1277-1308
            e.pass_run(F.PASS_FSR_EASU)
            e.pass_run(F.PASS_FSR_RCAS)


class HikariPlugin:
    """App::add_plugin(HikariPlugin): owns the context, uploads the noise tiles at start-up
    (lib.rs:189-219) and renders one camera with the `hikari` sub-graph order
    PREPASS -> LIGHT -> POST_PROCESS (lib.rs:252-367)."""

    def __init__(self, device=0, universal_settings: Optional[HikariUniversalSettings] = None, flags=0, api=None):
        self.universal_settings = universal_settings or HikariUniversalSettings()
        self.engine = Engine(api=api, device=device, flags=flags)
        self.engine.upload_noise()
        self.prepass, self.light, self.post_process = PrepassNode(self.engine), LightNode(self.engine), PostProcessNode(self.engine)
        self.counter = FrameCounter(0)
        self._size = None
        self._previous_camera = None

    def set_scene(self, scene: SceneData):
        self.engine.upload_scene(scene)

    def update_instances(self, scene: SceneData):
        """Instances moved (prepare_instances, instance.rs:352-437): rewrite the instance-level buffers only."""
        self.engine.upload_instances(scene)

    def render(self, camera: Camera, settings: HikariSettings, lights=None, frame_number=None, by_nodes=False, antialias=False):
        """One frame of the camera's render graph.  Returns the frame number used.  antialias=True also runs the
        SMAA Tu4x / TAA dispatches of PostProcessNode::run (the north-star frame ends at tone mapping)."""
        size = (camera.width, camera.height, settings.upscale.ratio())
        if size != self._size:  # prepare_light_textures, light.rs:342-363: reallocate + zero on size change
            self.engine.resize(*size)
            self._size = size
        n = self.counter.tick() if frame_number is None else frame_number
        frame = frame_uniform(settings, n)
        view = camera.view_uniform()
        pview = camera.previous_view_uniform(self._previous_camera)
        lights = lights or lights_uniform()
        if by_nodes:
            self.engine.frame_begin(frame, view, pview, lights)
            self.prepass.run(settings)
            self.light.run(settings)
            self.post_process.run(settings, antialias)
        else:
            self.engine.frame_render(frame, view, pview, lights, settings.to_c(), F.FRAME_ANTIALIAS if antialias else 0)
        self._previous_camera = camera
        return n

    def final_image(self, settings: HikariSettings):
        """What OverlayNode samples (overlay.rs:226-231), as f32 [H][W][4]."""
        if settings.upscale.kind == F.UPSCALE_SMAA_TU4X:
            buf = F.BUF_TAA_OUTPUT if settings.taa == Taa.Jasmine else F.BUF_UPSCALE_OUTPUT
        else:                                                       # upscale_output[1]: EASU then RCAS
            buf = F.BUF_UPSCALE_SHARPENED
        return self.engine.read_f16(buf)

    def output(self, settings: HikariSettings):
        """The three radiance channels the tone-mapping pass sums (tone_mapping.wgsl:25-27), as f32 [3][H][W][4]."""
        base = F.BUF_DENOISE_RENDER0 if settings.denoise else F.BUF_RENDER0
        return np.stack([self.engine.read_f16(base + i) for i in range(3)])
