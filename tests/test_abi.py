"""The C-ABI shared library loads without a GPU and exports every symbol include/hikari_hip.h
declares; host-side mirrors of reference logic give the reference's values."""
import ctypes as C
import os
import re

import pytest

from bevy_hikari_amd import _ffi as F
from conftest import ROOT, has_gpu

HEADER = os.path.join(ROOT, "include", "hikari_hip.h")
DEBUG_HEADER = os.path.join(ROOT, "include", "hikari_hip_debug.h")


def header_functions(path=HEADER):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hk_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    api = F.api()
    names = header_functions()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(api.dll, n)]
    assert not missing, missing
    # and the binding table covers the header (no silently unbound entry points)
    assert set(names) == set(F.DECLARED_SYMBOLS), set(names) ^ set(F.DECLARED_SYMBOLS)
    assert api.abi_version() == 8
    # the boundary a host binds carries no test or measurement hooks: those live in hikari_hip_debug.h (same library)
    assert not [n for n in names if n.startswith(("hk_debug_", "hk_measure_"))]
    debug = header_functions(DEBUG_HEADER)
    assert set(debug) == set(F.DECLARED_DEBUG_SYMBOLS) and all(hasattr(api.dll, n) for n in debug), set(debug) ^ set(F.DECLARED_DEBUG_SYMBOLS)


def test_struct_layouts_are_std430():
    assert C.sizeof(F.HkVertex) == 32 and C.sizeof(F.HkPrimitive) == 48 and C.sizeof(F.HkNode) == 32
    assert C.sizeof(F.HkInstance) == 176 and C.sizeof(F.HkMaterial) == 80 and C.sizeof(F.HkEmissive) == 64
    assert C.sizeof(F.HkFrame) == 256 and C.sizeof(F.HkView) == 416 and C.sizeof(F.HkPreviousView) == 128
    assert F.HkMaterial.emissive.offset == 32 and F.HkMaterial.reflectance.offset == 64 and F.HkMaterial.occlusion_texture.offset == 72
    assert F.HkEmissive.alias_table.offset == 40 and F.HkEmissive.surface_area.offset == 48
    assert F.HkFrame.number.offset == 192 and F.HkFrame.upscale_ratio.offset == 240
    assert F.HkInstance.model.offset == 32 and F.HkInstance.inverse_transpose_model.offset == 96 and F.HkInstance.mesh.offset == 160


def test_settings_default_matches_reference():  # lib.rs:435-455
    s = F.HkSettings()
    F.api().call("settings_default", C.byref(s))
    assert (s.direct_validate_interval, s.emissive_validate_interval, s.max_temporal_reuse_count, s.max_spatial_reuse_count) == (3, 5, 50, 800)
    assert s.max_reservoir_lifetime == 100.0 and abs(s.solar_angle - 0.046) < 1e-9 and s.indirect_bounces == 1 and s.max_indirect_luminance == 10.0
    assert (s.temporal_reuse, s.emissive_spatial_reuse, s.indirect_spatial_reuse, s.denoise) == (1, 0, 1, 1)
    assert s.taa == F.TAA_JASMINE and s.upscale_kind == F.UPSCALE_SMAA_TU4X and s.upscale_ratio == 2.0
    assert [round(c, 6) for c in s.clear_color] == [0.4, 0.4, 0.4, 1.0]


def test_frame_uniform_constants():  # view.rs:125-193
    s = F.HkSettings()
    F.api().call("settings_default", C.byref(s))
    s.upscale_ratio = 3.0
    f = F.HkFrame()
    F.api().call("frame_from_settings", C.byref(s), 41, C.byref(f))
    k = [[f.kernel[c][r] for r in range(3)] for c in range(3)]
    assert k == [[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]]
    assert abs(f.halton[1][1] - 0.666667) < 1e-6 and abs(f.halton[7][3] - 0.259259) < 1e-6 and f.halton[0][2] == 0.5
    assert f.number == 41 and f.upscale_ratio == 2.0  # Upscale::ratio clamps to [1,2]
    assert (f.direct_validate_interval, f.emissive_validate_interval, f.indirect_bounces, f.temporal_reuse) == (3, 5, 1, 1)


def test_scaled_size_and_bands():
    w, h = F.u32(), F.u32()
    F.api().call("scaled_size", 1920, 1080, 2.0, C.byref(w), C.byref(h))
    assert (w.value, h.value) == (960, 540)
    F.api().call("scaled_size", 101, 77, 1.5, C.byref(w), C.byref(h))
    assert (w.value, h.value) == (68, 52)  # ceil, light.rs:319
    covered = []
    for i in range(8):
        b0, b1 = F.u32(), F.u32()
        F.api().call("band_rows", 1080, i, 8, C.byref(b0), C.byref(b1))
        covered += list(range(b0.value, b1.value))
    assert covered == list(range(1080))


def test_band_plan_halo_rows():
    s = F.HkSettings()
    F.api().call("settings_default", C.byref(s))
    s.indirect_bounces = 2
    from bevy_hikari_amd.distributed import halo_plan

    ops = halo_plan(1920, 1080, 1.0, 3, 8, F.STAGE_SPATIAL, 17, s)
    # frame 17: cur = 1, previous = 0 -> temporal indirect output is reservoir 6; 20 rows from each neighbour
    assert sorted((o.buffer, o.peer, o.row_begin, o.row_end) for o in ops) == [(F.BUF_RESERVOIR0 + 6, 2, 385, 405), (F.BUF_RESERVOIR0 + 6, 4, 540, 560)]
    assert all(o.row_bytes == 1920 * 64 for o in ops)
    ops = halo_plan(1920, 1080, 1.0, 0, 8, F.STAGE_POST_PROCESS, 17, s)
    got = sorted((o.buffer, o.peer, o.row_begin, o.row_end) for o in ops)
    assert got == sorted([(F.BUF_RENDER0 + c, 1, 135, 150) for c in range(3)] + [(F.BUF_VARIANCE0 + c, 1, 135, 151) for c in range(3)])
    assert halo_plan(1920, 1080, 1.0, 0, 1, F.STAGE_SPATIAL, 17, s) == []


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_create_fails_loudly_without_gpu():
    ctx = C.c_void_p()
    with pytest.raises(F.HikariError) as e:
        F.api().call("create", 0, 0, C.byref(ctx))
    assert e.value.code == F.HK_E_NO_DEVICE and "no CPU fallback" in str(e.value)
