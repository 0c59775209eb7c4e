"""Runtime of the WGSL -> Python translation (tests/tests/tools/wgsl): value types and builtins.

TEST INFRASTRUCTURE ONLY.  Executes the reference's OWN shader source (src/shaders/*.wgsl) one invocation at a time,
every f32 operation rounded once (numpy.float32 scalars), so that the CPU oracle - a hand restatement of the same
shaders - can be pinned against a mechanical execution of the text it restates.  What WGSL leaves to the
implementation is bound to the numeric contract of oracle/hk_oracle_math.h (header comment there): fma chains in
dot / cross / matrix * vector, minNum / maxNum, normalize(v) = v * (1 / sqrt(dot)), the polynomial sin / cos / exp /
exp2 / log2 (called through the oracle library so that there is exactly one implementation of them) and the
constant-exponent pow reductions."""
import ctypes as C
import ctypes.util

import numpy as np

f32, i32, u32 = np.float32, np.int32, np.uint32
np.seterr(all="ignore")
_libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.fmaf.argtypes, _libm.fmaf.restype = [C.c_float, C.c_float, C.c_float], C.c_float


def fma(a, b, c):
    return f32(_libm.fmaf(float(a), float(b), float(c)))


_contract = None   # set by bind_contract(): op code, x, y -> f32 through orc_debug_math


def bind_contract(fn):
    global _contract
    _contract = fn


F0, F1 = f32(0.0), f32(1.0)


class V(tuple):
    """vecN<T>: immutable, elementwise arithmetic with scalar broadcast, swizzles."""
    __slots__ = ()
    __array_ufunc__ = None      # numpy scalars must defer to __radd__ / __rmul__ ... instead of building an ndarray

    def _bin(self, o, op):
        if isinstance(o, V):
            return V(op(a, b) for a, b in zip(self, o))
        return V(op(a, o) for a in self)

    def _rbin(self, o, op):
        return V(op(o, a) for a in self)

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._rbin(o, lambda a, b: a + b)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._rbin(o, lambda a, b: a - b)
    def __mul__(self, o):
        if isinstance(o, M):
            return o.rmul_vec(self)
        return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._rbin(o, lambda a, b: a * b)
    def __truediv__(self, o): return self._bin(o, div)
    def __rtruediv__(self, o): return self._rbin(o, div)
    def __mod__(self, o): return self._bin(o, mod)
    def __neg__(self): return V(-a for a in self)
    def __and__(self, o): return self._bin(o, lambda a, b: a & b)
    def __or__(self, o): return self._bin(o, lambda a, b: a | b)
    def __xor__(self, o): return self._bin(o, lambda a, b: a ^ b)
    def __lshift__(self, o): return self._bin(o, shl)
    def __rshift__(self, o): return self._bin(o, shr)
    def __invert__(self): return V(not a for a in self) if isinstance(self[0], (bool, np.bool_)) else V(~a for a in self)
    def __lt__(self, o): return self._bin(o, lambda a, b: bool(a < b))
    def __le__(self, o): return self._bin(o, lambda a, b: bool(a <= b))
    def __gt__(self, o): return self._bin(o, lambda a, b: bool(a > b))
    def __ge__(self, o): return self._bin(o, lambda a, b: bool(a >= b))
    def __eq__(self, o): return self._bin(o, lambda a, b: bool(a == b))
    def __ne__(self, o): return self._bin(o, lambda a, b: bool(a != b))
    __hash__ = None
    def lt(self, o): return self._bin(o, lambda a, b: bool(a < b))
    def le(self, o): return self._bin(o, lambda a, b: bool(a <= b))
    def gt(self, o): return self._bin(o, lambda a, b: bool(a > b))
    def ge(self, o): return self._bin(o, lambda a, b: bool(a >= b))
    def eq(self, o): return self._bin(o, lambda a, b: bool(a == b))
    def ne(self, o): return self._bin(o, lambda a, b: bool(a != b))

    def __getattr__(self, name):
        try:
            idx = [_SWZ[ch] for ch in name]
        except KeyError:
            raise AttributeError(name) from None
        if len(idx) == 1:
            return self[idx[0]]
        return V(self[i] for i in idx)

    def with_(self, name, value):
        """the vector with components `name` replaced (assignment to a swizzle / component of a var)"""
        out = list(self)
        if len(name) == 1:
            out[_SWZ[name]] = value
        else:
            for ch, v in zip(name, value):
                out[_SWZ[ch]] = v
        return V(out)

    def with_index(self, i, value):
        out = list(self)
        out[int(i)] = value
        return V(out)


_SWZ = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


def _flatten(args):
    out = []
    for a in args:
        if isinstance(a, V):
            out.extend(a)
        else:
            out.append(a)
    return out


def make_vec(n, ty):
    conv = {"f32": to_f32, "i32": to_i32, "u32": to_u32, "bool": bool}[ty]

    def ctor(*args):
        flat = _flatten(args)
        if len(flat) == 0:
            return V(conv(0) for _ in range(n))
        if len(flat) == 1:
            return V(conv(flat[0]) for _ in range(n))
        assert len(flat) == n, (n, ty, args)
        return V(conv(a) for a in flat)
    return ctor


def to_f32(x):
    return f32(x)


def to_i32(x):
    if isinstance(x, (np.floating, float)):
        x = float(x)
        if x != x:
            return i32(0)
        x = max(-2147483648.0, min(2147483647.0, float(np.trunc(x))))   # WGSL: f32 -> i32 saturates
        return i32(int(x))
    return i32(np.int64(int(x)).astype(np.int32)) if not isinstance(x, np.int32) else x


def to_u32(x):
    if isinstance(x, (np.floating, float)):
        x = float(x)
        if x != x:
            return u32(0)
        x = max(0.0, min(4294967295.0, float(np.trunc(x))))
        return u32(int(x))
    return u32(int(x) & 0xFFFFFFFF)


def convert(ty):
    return {"f32": to_f32, "i32": to_i32, "u32": to_u32, "bool": bool}[ty]


def vconvert(ty, x):
    c = convert(ty)
    return V(c(a) for a in x) if isinstance(x, V) else c(x)


def bitcast(ty, x):
    def one(a):
        return np.array([a]).view({"f32": np.float32, "i32": np.int32, "u32": np.uint32}[ty])[0]
    return V(one(a) for a in x) if isinstance(x, V) else one(x)


class M:
    """matCxR<f32>: tuple of C column vectors."""
    __slots__ = ("cols",)
    __array_ufunc__ = None

    def __init__(self, cols):
        self.cols = tuple(cols)

    def __getitem__(self, i):
        return self.cols[int(i)]

    def copy(self):
        return self

    def __mul__(self, o):
        if isinstance(o, V):            # M * v, contract: per component fma(c3,v.w, fma(c2,v.z, fma(c1,v.y, c0*v.x)))
            rows = len(self.cols[0])
            out = []
            for r in range(rows):
                acc = self.cols[0][r] * o[0]
                for c in range(1, len(self.cols)):
                    acc = fma(self.cols[c][r], o[c], acc)
                out.append(acc)
            return V(out)
        if isinstance(o, M):
            return M(self * c for c in o.cols)
        return M(c * o for c in self.cols)

    def rmul_vec(self, v):              # v * M = (dot(v, c0), dot(v, c1), ...)
        return V(dot(v, c) for c in self.cols)

    def __rmul__(self, o):
        return M(c * o for c in self.cols)


def make_mat(c, r):
    def ctor(*args):
        if len(args) == 0:
            return M(V(F0 for _ in range(r)) for _ in range(c))
        if all(isinstance(a, V) for a in args) and len(args) == c:
            return M(V(f32(x) for x in a) for a in args)
        flat = _flatten(args)
        assert len(flat) == c * r
        return M(V(f32(x) for x in flat[i * r:(i + 1) * r]) for i in range(c))
    return ctor


def transpose(m):
    c, r = len(m.cols), len(m.cols[0])
    return M(V(m.cols[j][i] for j in range(c)) for i in range(r))


# ---------------------------------------------------------------- scalar helpers used by the translation
def div(a, b):
    if isinstance(a, V) or isinstance(b, V):
        return (a if isinstance(a, V) else V(a for _ in b))._bin(b, div)
    if isinstance(a, (np.floating, float)) or isinstance(b, (np.floating, float)):
        return f32(a) / f32(b)
    if isinstance(a, np.uint32) or isinstance(b, np.uint32):
        return u32(int(a) // int(b)) if int(b) != 0 else u32(int(a))          # WGSL: x / 0 = x
    ia, ib = int(a), int(b)
    if ib == 0:
        return i32(ia)
    q = abs(ia) // abs(ib)
    return i32(q if (ia < 0) == (ib < 0) else -q)


def mod(a, b):
    if isinstance(a, V) or isinstance(b, V):
        return (a if isinstance(a, V) else V(a for _ in b))._bin(b, mod)
    if isinstance(a, (np.floating, float)) or isinstance(b, (np.floating, float)):
        a, b = f32(a), f32(b)
        return a - b * f32(np.trunc(a / b))
    if isinstance(a, np.uint32) or isinstance(b, np.uint32):
        return u32(int(a) % int(b)) if int(b) != 0 else u32(0)
    ia, ib = int(a), int(b)
    if ib == 0:
        return i32(0)
    r = abs(ia) % abs(ib)
    return i32(-r if ia < 0 else r)


def shl(a, b):
    v = (int(a) << (int(b) & 31)) & 0xFFFFFFFF
    return u32(v) if isinstance(a, np.uint32) else i32(v - (1 << 32) if v >= (1 << 31) else v)


def shr(a, b):
    return u32(int(a) >> (int(b) & 31)) if isinstance(a, np.uint32) else i32(int(a) >> (int(b) & 31))


def cmp(op, a, b):
    if isinstance(a, V) or isinstance(b, V):
        a = a if isinstance(a, V) else V(a for _ in b)
        return getattr(a, op)(b)
    return {"lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b, "eq": a == b, "ne": a != b}[op]


def lnot(a):
    return V(not x for x in a) if isinstance(a, V) else (not a)


def neg(a):
    return -a


class Box:
    """the target of a `ptr<function, T>` whose pointee is a scalar / vector / matrix (immutable here): the caller boxes the
    variable for the call and writes it back afterwards; structs are passed by reference instead"""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


def cp(x):
    """WGSL value semantics: structs and arrays are copied on assignment / argument passing / return"""
    if x.__class__ is V or isinstance(x, (np.generic, bool, int, float, M)):
        return x
    if isinstance(x, list):
        return [cp(e) for e in x]
    c = getattr(x, "copy_value", None)
    return c() if c else x


# ---------------------------------------------------------------- builtins
def _map(fn):
    def g(*a):
        if any(isinstance(x, V) for x in a):
            n = max(len(x) for x in a if isinstance(x, V))
            a = [x if isinstance(x, V) else V(x for _ in range(n)) for x in a]
            return V(fn(*xs) for xs in zip(*a))
        return fn(*a)
    return g


def _fmin(a, b):   # IEEE minNum with -0 < +0
    if isinstance(a, (np.floating, float)) or isinstance(b, (np.floating, float)):
        a, b = f32(a), f32(b)
        if a != a: return b
        if b != b: return a
        if a == b: return a if np.signbit(a) else b
        return a if a < b else b
    return a if a < b else b


def _fmax(a, b):
    if isinstance(a, (np.floating, float)) or isinstance(b, (np.floating, float)):
        a, b = f32(a), f32(b)
        if a != a: return b
        if b != b: return a
        if a == b: return b if np.signbit(a) else a
        return a if a > b else b
    return a if a > b else b


w_min, w_max = _map(_fmin), _map(_fmax)
w_abs = _map(lambda a: abs(a) if not isinstance(a, (np.floating, float)) else f32(np.fabs(f32(a))))
w_floor = _map(lambda a: f32(np.floor(f32(a))))
w_ceil = _map(lambda a: f32(np.ceil(f32(a))))
w_round = _map(lambda a: f32(np.rint(f32(a))))
w_trunc = _map(lambda a: f32(np.trunc(f32(a))))
w_sqrt = _map(lambda a: f32(np.sqrt(f32(a))))
w_fract = _map(lambda a: f32(a) - f32(np.floor(f32(a))))
w_clamp = _map(lambda x, lo, hi: _fmin(_fmax(x, lo), hi))
w_saturate = _map(lambda x: _fmin(_fmax(f32(x), F0), F1))
w_mix = _map(lambda a, b, t: f32(a) * (F1 - f32(t)) + f32(b) * f32(t))
w_sign = _map(lambda x: F1 if x > 0 else (f32(-1.0) if x < 0 else F0))
w_step = _map(lambda e, x: F1 if x >= e else F0)
w_sin = _map(lambda x: _contract(0, x, 0.0))
w_cos = _map(lambda x: _contract(1, x, 0.0))
w_exp = _map(lambda x: _contract(2, x, 0.0))
w_exp2 = _map(lambda x: _contract(3, x, 0.0))
w_log2 = _map(lambda x: _contract(4, x, 0.0))
w_inverse_sqrt = _map(lambda a: F1 / f32(np.sqrt(f32(a))))


def _pow(x, y):
    x, y = f32(x), f32(y)
    # the contract's constant-exponent reductions (hk_oracle_math.h: pow2_, pow5_, pow16_, pow_quarter_)
    if y == f32(2.0):
        return x * x
    if y == f32(5.0):
        x2 = x * x
        return (x2 * x2) * x
    if y == f32(16.0):
        x2 = x * x; x4 = x2 * x2; x8 = x4 * x4
        return x8 * x8
    if y == f32(0.25):
        return f32(np.sqrt(f32(np.sqrt(x))))
    return _contract(5, x, y)


w_pow = _map(_pow)


def select(f, t, c):
    if isinstance(c, V):
        f = f if isinstance(f, V) else V(f for _ in c)
        t = t if isinstance(t, V) else V(t for _ in c)
        return V(b if k else a for a, b, k in zip(f, t, c))
    return t if c else f


def w_any(v):
    return any(v) if isinstance(v, V) else bool(v)


def w_all(v):
    return all(v) if isinstance(v, V) else bool(v)


def dot(a, b):
    acc = a[0] * b[0]
    for i in range(1, len(a)):
        acc = fma(a[i], b[i], acc) if isinstance(acc, np.floating) else acc + a[i] * b[i]
    return acc


def cross(a, b):
    return V((fma(a[1], b[2], -(a[2] * b[1])), fma(a[2], b[0], -(a[0] * b[2])), fma(a[0], b[1], -(a[1] * b[0]))))


def length(a):
    return f32(np.sqrt(dot(a, a))) if isinstance(a, V) else f32(np.fabs(a))


def distance(a, b):
    return length(a - b)


def normalize(a):
    s = F1 / f32(np.sqrt(dot(a, a)))
    return a * s


def reflect(i, n):
    return i - n * (f32(2.0) * dot(n, i))


# ---------------------------------------------------------------- f16 and pack / unpack (the WGSL formulas)
def f32_to_f16_bits(x):
    return int(np.array([f32(x)]).astype(np.float16).view(np.uint16)[0])


def f16_bits_to_f32(h):
    return f32(np.array([h], dtype=np.uint16).view(np.float16)[0])


def round_f16(x):
    return f16_bits_to_f32(f32_to_f16_bits(x))


def pack2x16float(v):
    return u32(f32_to_f16_bits(v[0]) | (f32_to_f16_bits(v[1]) << 16))


def unpack2x16float(u):
    u = int(u)
    return V((f16_bits_to_f32(u & 0xFFFF), f16_bits_to_f32(u >> 16)))


def _unorm16(x):
    return int(np.floor(f32(0.5) + f32(65535.0) * _fmin(_fmax(f32(x), F0), F1)))


def pack2x16unorm(v):
    return u32(_unorm16(v[0]) | (_unorm16(v[1]) << 16))


def unpack2x16unorm(u):
    u = int(u)
    return V((f32(u & 0xFFFF) / f32(65535.0), f32(u >> 16) / f32(65535.0)))


def _snorm8(x):
    return int(np.floor(f32(0.5) + f32(127.0) * _fmin(_fmax(f32(x), f32(-1.0)), F1))) & 0xFF


def pack4x8snorm(v):
    return u32(_snorm8(v[0]) | (_snorm8(v[1]) << 8) | (_snorm8(v[2]) << 16) | (_snorm8(v[3]) << 24))


def _unsnorm8(b):
    b = b - 256 if b >= 128 else b
    return _fmax(f32(b) / f32(127.0), f32(-1.0))


def unpack4x8snorm(u):
    u = int(u)
    return V(_unsnorm8((u >> s) & 0xFF) for s in (0, 8, 16, 24))


def pack4x8unorm(v):
    q = [int(np.floor(f32(0.5) + f32(255.0) * _fmin(_fmax(f32(x), F0), F1))) for x in v]
    return u32(q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24))


def unpack4x8unorm(u):
    u = int(u)
    return V(f32((u >> s) & 0xFF) / f32(255.0) for s in (0, 8, 16, 24))
