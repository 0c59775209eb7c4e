"""The numeric contract: oracle math == libm within a few ulp (CPU), and the HIP
library's device math == the oracle's bit for bit (GPU)."""
import ctypes as C

import numpy as np
import pytest

from bevy_hikari_amd import _ffi as F
from oracle_lib import oracle_api

OPS = {"sin": 0, "cos": 1, "exp": 2, "exp2": 3, "log2": 4, "pow": 5, "min": 6, "max": 7, "f16": 8, "div": 9, "sqrt": 10}


def oracle_math(op, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    fp = lambda a: a.ctypes.data_as(C.POINTER(F.f32))
    yy = None if y is None else np.ascontiguousarray(y, dtype=np.float32)
    oracle_api().call("debug_math", None, OPS[op], fp(x), None if yy is None else fp(yy), fp(out), x.size)
    return out


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def inputs(rng, n=200_000):
    return {
        "sin": (rng.uniform(-10.0, 10.0, n),),
        "cos": (rng.uniform(-10.0, 10.0, n),),
        "exp": (rng.uniform(-90.0, 20.0, n),),
        "exp2": (rng.uniform(-130.0, 30.0, n),),
        "log2": (np.exp(rng.uniform(-60.0, 60.0, n)),),
        "pow": (rng.uniform(0.0, 2.0, n), rng.choice([0.25, 2.0, 5.0, 16.0], n)),
        "min": (rng.choice([-0.0, 0.0, 1.0, -1.0, np.nan, np.inf, -np.inf, 0.5], n), rng.choice([-0.0, 0.0, 1.0, -1.0, np.nan, np.inf, 2.0], n)),
        "max": (rng.choice([-0.0, 0.0, 1.0, -1.0, np.nan, np.inf, -np.inf, 0.5], n), rng.choice([-0.0, 0.0, 1.0, -1.0, np.nan, np.inf, 2.0], n)),
        "f16": (np.concatenate([rng.uniform(-70000, 70000, n // 2), np.exp(rng.uniform(-30, 12, n // 2))]),),
        "div": (rng.normal(0, 100, n), np.concatenate([rng.normal(0, 10, n // 2), np.exp(rng.uniform(-40, 40, n // 2))])),
        "sqrt": (np.exp(rng.uniform(-80, 80, n)),),
    }


def test_oracle_math_close_to_libm():
    rng = np.random.default_rng(1)
    data = inputs(rng, 100_000)
    x = data["sin"][0].astype(np.float32)
    assert np.abs(oracle_math("sin", x) - np.sin(x.astype(np.float64))).max() < 2.5e-7
    assert np.abs(oracle_math("cos", x) - np.cos(x.astype(np.float64))).max() < 2.5e-7
    x = data["exp"][0].astype(np.float32)
    x = x[x > -85.0]  # below that the result is an f32 denormal
    ref = np.exp(x.astype(np.float64))
    assert (np.abs(oracle_math("exp", x) - ref) / ref).max() < 4e-7
    x = data["exp2"][0].astype(np.float32)
    x = x[x > -125]
    ref = np.exp2(x.astype(np.float64))
    assert (np.abs(oracle_math("exp2", x) - ref) / ref).max() < 3e-7
    x = data["log2"][0].astype(np.float32)
    ref = np.log2(x.astype(np.float64))
    assert np.abs(oracle_math("log2", x) - ref).max() / 1.0 < 2e-5  # absolute; relative error below
    big = np.abs(ref) > 1
    assert (np.abs(oracle_math("log2", x) - ref)[big] / np.abs(ref[big])).max() < 3e-7
    xb, yb = (a.astype(np.float32) for a in data["pow"])
    ref = np.power(xb.astype(np.float64), yb.astype(np.float64))
    got = oracle_math("pow", xb, yb)
    ok = ref > 1e-30
    assert (np.abs(got - ref)[ok] / ref[ok]).max() < 4e-6  # WGSL: pow inherits exp2(y*log2(x))
    assert (got[xb == 0] == 0).all()


def test_oracle_f16_roundtrip_matches_numpy():
    rng = np.random.default_rng(2)
    x = inputs(rng, 200_000)["f16"][0].astype(np.float32)
    x = np.concatenate([x, np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 6e-8, 5.96e-8, 2.98e-8, 2.99e-8, np.inf, -np.inf], np.float32)])
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).astype(np.float32)
    got = oracle_math("f16", x)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()


def test_oracle_div_sqrt_are_ieee():
    rng = np.random.default_rng(3)
    d = inputs(rng, 100_000)
    a, b = (v.astype(np.float32) for v in d["div"])
    assert (oracle_math("div", a, b).view(np.uint32) == (a / b).view(np.uint32)).all()
    s = d["sqrt"][0].astype(np.float32)
    assert (oracle_math("sqrt", s).view(np.uint32) == np.sqrt(s).view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("op", list(OPS))
def test_device_math_bit_exact_vs_oracle(op):
    import bevy_hikari_amd as hk

    rng = np.random.default_rng(7)
    args = [a.astype(np.float32) for a in inputs(rng)[op]]
    eng = hk.Engine(device=0)
    got = eng.debug_math(OPS[op], *args)
    want = oracle_math(op, *args)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        i = np.nonzero(~same)[0][:5]
        raise AssertionError(f"{op}: {(~same).sum()} of {same.size} differ, e.g. x={[a[i] for a in args]} gpu={got[i]} cpu={want[i]}")


def edge_inputs(op, rng, n=400_000):
    """Where exp_ / exp2_ leave the normal range: results that are subnormal (one rounding in the final scaling - the device does it
    with v_ldexp_f32, the contract writes it as two multiplications by powers of two), the flush-to-zero and overflow thresholds
    and their float neighbours, and the specials."""
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e30, -1e30, 3e38, -3e38, 1e-45, -1e-45, 1e-38, -1e-38], dtype=np.float32)

    def around(v, k=64):
        out = [np.float32(v)]
        for _ in range(k):
            out.append(np.nextafter(out[-1], np.float32(np.inf)))
        lo = np.float32(v)
        for _ in range(k):
            lo = np.nextafter(lo, np.float32(-np.inf))
            out.append(lo)
        return np.array(out, dtype=np.float32)

    if op == "exp":
        return (np.concatenate([rng.uniform(-106.0, -85.0, n), rng.uniform(86.0, 90.0, n // 4), around(-103.972084045410), around(88.72283905206835),
                                around(-87.33654), special]).astype(np.float32),)
    if op == "exp2":
        return (np.concatenate([rng.uniform(-152.0, -124.0, n), rng.uniform(125.0, 129.0, n // 4), around(-150.0), around(128.0), around(-126.0), around(-149.5),
                                np.arange(-152, 130).astype(np.float32), np.arange(-152, 130).astype(np.float32) + 0.5, special]).astype(np.float32),)
    # pow = exp2(y * log2(x)): tiny and huge results
    x = np.concatenate([np.exp(rng.uniform(-80.0, 80.0, n)), special[:4]]).astype(np.float32)
    y = np.concatenate([rng.uniform(-2.0, 2.0, n), np.array([0.0, 1.0, -1.0, 2.0])]).astype(np.float32)
    return x, y


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["exp", "exp2", "pow"])
def test_device_exp_family_at_the_edges_of_the_range(op):
    import bevy_hikari_amd as hk

    args = edge_inputs(op, np.random.default_rng(11))
    got = hk.Engine(device=0).debug_math(OPS[op], *args)
    want = oracle_math(op, *args)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        i = np.nonzero(~same)[0][:5]
        raise AssertionError(f"{op}: {(~same).sum()} of {same.size} differ, e.g. x={[a[i] for a in args]} gpu={got[i]} cpu={want[i]}")
    if op != "pow":
        assert ((np.abs(want) < 1.17e-38) & (want != 0)).sum() > 10_000, "the sweep must reach subnormal results"


NORM_OPS = {"unorm16": (14, 65536), "snorm8": (15, 256), "unorm8": (20, 256)}


def oracle_norm(op, n):
    x = np.arange(n, dtype=np.float32)
    out = np.empty_like(x)
    fp = lambda a: a.ctypes.data_as(C.POINTER(F.f32))
    oracle_api().call("debug_math", None, op, fp(x), None, fp(out), x.size)
    return out


def test_oracle_norm_decodes_are_plain_ieee_divisions():
    """unpack2x16unorm / unpack4x8snorm / the unorm8 texel decode, for EVERY input of the format."""
    u = np.arange(65536, dtype=np.float32)
    assert (oracle_norm(14, 65536).view(np.uint32) == (u / np.float32(65535.0)).view(np.uint32)).all()
    b = np.arange(256).astype(np.uint8).view(np.int8).astype(np.float32)
    assert (oracle_norm(15, 256).view(np.uint32) == np.maximum(b / np.float32(127.0), np.float32(-1.0)).view(np.uint32)).all()
    assert (oracle_norm(20, 256).view(np.uint32) == (np.arange(256, dtype=np.float32) / np.float32(255.0)).view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(NORM_OPS))
def test_device_norm_decodes_exhaustive(name):
    """The device replaces x / 65535, x / 127, x / 255 by a 3-instruction multiply + exact-residual correction
    (hk_device_math.hpp div_norm); it must equal the IEEE quotient for every input of the format."""
    import bevy_hikari_amd as hk

    op, n = NORM_OPS[name]
    got = hk.Engine(device=0).debug_math(op, np.arange(n, dtype=np.float32))
    assert (got.view(np.uint32) == oracle_norm(op, n).view(np.uint32)).all()
