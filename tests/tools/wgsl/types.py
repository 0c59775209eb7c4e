"""Type descriptors (zero values, WGSL memory layout, decode / encode of buffers) and the texture builtins of the
WGSL -> Python translation.  TEST INFRASTRUCTURE ONLY (see runtime.py).

Sampling follows the numeric contract the oracle documents: texel centres at (i + 0.5) / size,
nearest = floor(uv * size), bilinear weights in f32 as mix(mix(t00, t10, fx), mix(t01, t11, fx), fy), textureGather
order (u-,v+) (u+,v+) (u+,v-) (u-,v-); rgba16float stores round to nearest even."""
import numpy as np

from . import runtime as R

f32, i32, u32, V = R.f32, R.i32, R.u32, R.V


def _round_up(x, a):
    return (x + a - 1) // a * a


class Scalar:
    def __init__(self, name):
        self.name, self.size, self.align = name, 4, 4
        self.np = {"f32": np.float32, "i32": np.int32, "u32": np.uint32, "bool": np.uint32}[name]

    def zero(self):
        return {"f32": R.F0, "i32": i32(0), "u32": u32(0), "bool": False}[self.name]

    def decode(self, buf, off):
        return buf[off:off + 4].view(self.np)[0]

    def encode(self, buf, off, v):
        buf[off:off + 4] = np.array([v], dtype=self.np).view(np.uint8)


class Vec:
    def __init__(self, n, ty):
        self.n, self.elem = int(n), Scalar(ty)
        self.size = 4 * self.n
        self.align = {2: 8, 3: 16, 4: 16}[self.n]

    def zero(self):
        z = self.elem.zero()
        return V(z for _ in range(self.n))

    def decode(self, buf, off):
        return V(buf[off:off + self.size].view(self.elem.np))

    def encode(self, buf, off, v):
        buf[off:off + self.size] = np.array(list(v), dtype=self.elem.np).view(np.uint8)


class Mat:
    def __init__(self, c, r):
        self.c, self.r = int(c), int(r)
        self.col = Vec(self.r, "f32")
        self.stride = _round_up(self.col.size, self.col.align)
        self.size, self.align = self.c * self.stride, self.col.align

    def zero(self):
        return R.M(self.col.zero() for _ in range(self.c))

    def decode(self, buf, off):
        return R.M(self.col.decode(buf, off + i * self.stride) for i in range(self.c))

    def encode(self, buf, off, m):
        for i in range(self.c):
            self.col.encode(buf, off + i * self.stride, m[i])


class Array:
    def __init__(self, elem, n):
        self.elem, self.n = elem, n
        self.stride = _round_up(elem.size, elem.align)
        self.size = None if n is None else self.stride * n
        self.align = elem.align

    def zero(self):
        return [self.elem.zero() for _ in range(self.n)]

    def decode(self, buf, off):
        if self.n is None:
            return BufferArray(self.elem, self.stride, buf, off)
        return [self.elem.decode(buf, off + i * self.stride) for i in range(self.n)]

    def encode(self, buf, off, v):
        for i, x in enumerate(v):
            self.elem.encode(buf, off + i * self.stride, x)


class BufferArray:
    """runtime-sized array<T> living in a storage buffer: decoded on access, written through"""

    def __init__(self, elem, stride, buf, off):
        self.elem, self.stride, self.buf, self.off = elem, stride, buf, off

    def __len__(self):
        return (len(self.buf) - self.off) // self.stride

    def __getitem__(self, i):
        i = int(i)
        if i < 0 or i >= len(self):      # robust buffer access: out-of-bounds reads return zero
            return self.elem.zero()
        return self.elem.decode(self.buf, self.off + i * self.stride)

    def __setitem__(self, i, v):
        i = int(i)
        if 0 <= i < len(self):
            self.elem.encode(self.buf, self.off + i * self.stride, v)


class StructBase:
    __slots__ = ()

    def copy_value(self):
        c = self.__class__.__new__(self.__class__)
        for n in self.FIELDS:
            object.__setattr__(c, n, R.cp(getattr(self, n)))
        return c

    def set_value(self, other):
        for n in self.FIELDS:
            object.__setattr__(self, n, R.cp(getattr(other, n)))

    @classmethod
    def make(cls, *args):
        c = cls.__new__(cls)
        for n, a in zip(cls.FIELDS, args):
            setattr(c, n, R.cp(a))
        return c

    def __repr__(self):
        return self.__class__.__name__ + "(" + ", ".join(f"{n}={getattr(self, n)!r}" for n in self.FIELDS) + ")"


class Struct:
    def __init__(self, cls, fields):
        self.cls, self.fields = cls, fields
        off, align, self.offsets = 0, 4, []
        for name, ty in fields:
            off = _round_up(off, ty.align)
            self.offsets.append(off)
            align = max(align, ty.align)
            off = None if ty.size is None else off + ty.size
        self.align = align
        self.size = None if off is None else _round_up(off, align)

    def zero(self):
        c = self.cls.__new__(self.cls)
        for name, ty in self.fields:
            setattr(c, name, ty.zero())
        return c

    def decode(self, buf, off):
        c = self.cls.__new__(self.cls)
        for (name, ty), o in zip(self.fields, self.offsets):
            setattr(c, name, ty.decode(buf, off + o))
        return c

    def encode(self, buf, off, v):
        for (name, ty), o in zip(self.fields, self.offsets):
            ty.encode(buf, off + o, getattr(v, name))


class Opaque:
    size = align = None

    def __init__(self, name):
        self.name = name


def _mk(n, ty):
    return R.make_vec(n, ty)


vec2f32, vec3f32, vec4f32 = _mk(2, "f32"), _mk(3, "f32"), _mk(4, "f32")
vec2i32, vec3i32, vec4i32 = _mk(2, "i32"), _mk(3, "i32"), _mk(4, "i32")
vec2u32, vec3u32, vec4u32 = _mk(2, "u32"), _mk(3, "u32"), _mk(4, "u32")
vec2bool, vec3bool, vec4bool = _mk(2, "bool"), _mk(3, "bool"), _mk(4, "bool")
mat2x2, mat3x3, mat4x4 = R.make_mat(2, 2), R.make_mat(3, 3), R.make_mat(4, 4)


# ---------------------------------------------------------------- textures and samplers
class Sampler:
    """address: "clamp" | "repeat" | "mirror" (per axis); noise=True: the blue-noise lookup, which the oracle evaluates as
    floor(fract(uv) * size) (hk_oracle.cpp noise_fetch)"""

    def __init__(self, linear=False, address="clamp", address_v=None, noise=False):
        self.linear, self.address, self.address_v, self.noise = linear, address, address_v or address, noise


class Texture:
    """2-D texture, one mip level.  `data`: float32 (or uint32 for u32 textures) [h][w][4]; `store_f16`: values written by
    textureStore are rounded to binary16 first (rgba16float storage textures)."""

    def __init__(self, data, store_f16=False, integer=False):
        self.data, self.store_f16, self.integer = data, store_f16, integer
        self.h, self.w = data.shape[:2]

    def texel(self, x, y):
        t = self.data[y, x]
        return V(t) if not self.integer else V(u32(v) for v in t)


def _zero4(tex):
    return V((u32(0),) * 4) if tex.integer else V((R.F0,) * 4)


def texture_dimensions(tex, level=0):
    return V((i32(tex.w), i32(tex.h)))


def texture_num_levels(tex):
    return i32(1)


def texture_load(tex, coords, level=None):
    x, y = int(coords[0]), int(coords[1])
    if x < 0 or y < 0 or x >= tex.w or y >= tex.h:
        return _zero4(tex)       # out-of-bounds loads return zero
    return tex.texel(x, y)


def texture_store(tex, coords, value):
    x, y = int(coords[0]), int(coords[1])
    if x < 0 or y < 0 or x >= tex.w or y >= tex.h:
        return
    v = [f32(c) for c in value]
    if tex.store_f16:
        v = [R.round_f16(c) for c in v]
    tex.data[y, x, :len(v)] = v


def _wrap(i, n, address):
    if address == "repeat":
        return i % n
    if address == "mirror":
        p = 2 * n
        m = i % p
        return m if m < n else p - 1 - m
    return min(max(i, 0), n - 1)


def texture_sample_level(tex, sampler, uv, level):
    w, h = f32(tex.w), f32(tex.h)
    if not sampler.linear:
        if sampler.noise:
            x = int(np.floor(R.w_fract(uv[0]) * w)) % tex.w
            y = int(np.floor(R.w_fract(uv[1]) * h)) % tex.h
        else:
            x = _wrap(int(np.floor(uv[0] * w)), tex.w, sampler.address)
            y = _wrap(int(np.floor(uv[1] * h)), tex.h, sampler.address_v)
        return tex.texel(x, y)
    px, py = uv[0] * w - f32(0.5), uv[1] * h - f32(0.5)
    fx0, fy0 = f32(np.floor(px)), f32(np.floor(py))
    fx, fy = px - fx0, py - fy0
    ix, iy = int(fx0), int(fy0)
    x0, x1 = _wrap(ix, tex.w, sampler.address), _wrap(ix + 1, tex.w, sampler.address)
    y0, y1 = _wrap(iy, tex.h, sampler.address_v), _wrap(iy + 1, tex.h, sampler.address_v)
    top = R.w_mix(tex.texel(x0, y0), tex.texel(x1, y0), fx)
    bot = R.w_mix(tex.texel(x0, y1), tex.texel(x1, y1), fx)
    return R.w_mix(top, bot, fy)


def texture_gather(component, tex, sampler, uv):
    w, h = f32(tex.w), f32(tex.h)
    px, py = uv[0] * w - f32(0.5), uv[1] * h - f32(0.5)
    ix, iy = int(np.floor(px)), int(np.floor(py))
    x0, x1 = _wrap(ix, tex.w, "clamp"), _wrap(ix + 1, tex.w, "clamp")
    y0, y1 = _wrap(iy, tex.h, "clamp"), _wrap(iy + 1, tex.h, "clamp")
    c = int(component)
    return V((tex.texel(x0, y1)[c], tex.texel(x1, y1)[c], tex.texel(x1, y0)[c], tex.texel(x0, y0)[c]))


def array_length(a):
    return u32(len(a))


# derivative builtins of fragment shaders: the harness that runs a fragment function sets these (default: flat)
DERIVATIVES = {"dpdx": lambda v: R.F0 * v, "dpdy": lambda v: R.F0 * v}


def dpdx(v):
    return DERIVATIVES["dpdx"](v)


def dpdy(v):
    return DERIVATIVES["dpdy"](v)
