"""Host-side constants of the reference read from its Rust text (no Rust toolchain here: they are parsed, not executed) and
compared with what the C ABI produces: the 3x3 a-trous kernel and the 16-entry Halton table of FrameUniform (view.rs:125-140),
HikariSettings::default() (lib.rs:435-455), the workgroup size and noise tile count (lib.rs).  Skips where the reference is
not mounted."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is not mounted here")


def numbers(text):
    return [float(x) for x in re.findall(r"-?\d+\.\d+", text)]


def test_frame_uniform_kernel_and_halton_table():
    src = open(os.path.join(REF, "view.rs")).read()
    kernel = numbers(re.search(r"const KERNEL: Mat3 = Mat3 \{(.*?)\};", src, re.S).group(1))
    halton = numbers(re.search(r"const HALTON: \[Vec4; 8\] = \[(.*?)\];", src, re.S).group(1))
    assert len(kernel) == 9 and len(halton) == 32
    f = hk.frame_uniform(hk.HikariSettings(), 7)
    assert [float(f.kernel[c][r]) for c in range(3) for r in range(3)] == [float(np.float32(v)) for v in kernel]      # Mat3 columns x_axis, y_axis, z_axis
    assert [float(f.halton[i][k]) for i in range(8) for k in range(4)] == [float(np.float32(v)) for v in halton]
    assert f.number == 7


def test_settings_default():
    src = open(os.path.join(REF, "lib.rs")).read()
    body = re.search(r"impl Default for HikariSettings \{.*?Self \{(.*?)\n        \}", src, re.S).group(1)
    want = dict(re.findall(r"(\w+): ([\w.:()\s,]+?),\n", body))
    s = F.HkSettings()
    F.api().call("settings_default", C.byref(s))
    for name in ("direct_validate_interval", "emissive_validate_interval", "max_temporal_reuse_count", "max_spatial_reuse_count", "indirect_bounces"):
        assert getattr(s, name) == int(want[name]), name
    for name in ("max_reservoir_lifetime", "solar_angle", "max_indirect_luminance"):
        assert float(getattr(s, name)) == float(np.float32(float(want[name]))), name
    for name in ("temporal_reuse", "emissive_spatial_reuse", "indirect_spatial_reuse", "denoise"):
        assert bool(getattr(s, name)) == (want[name] == "true"), name
    assert [float(c) for c in s.clear_color] == [float(np.float32(v)) for v in numbers(want["clear_color"])] + [1.0]
    # Taa::default() = Jasmine, Upscale::default() = SmaaTu4x { ratio: 2.0 } (lib.rs:466-496)
    assert re.search(r"#\[default\]\s*Jasmine", src) and (s.taa, s.upscale_kind, s.upscale_ratio) == (F.TAA_JASMINE, F.UPSCALE_SMAA_TU4X, 2.0)
    assert re.search(r"SMAA_TU_2_0: Self = Self::SmaaTu4x \{ ratio: 2\.0 \}", src) or re.search(r"SmaaTu4x \{ ratio: 2\.0 \}", src)


def test_workgroup_size_noise_count_and_shader_constants():
    lib = open(os.path.join(REF, "lib.rs")).read()
    assert int(re.search(r"pub const WORKGROUP_SIZE: u32 = (\d+);", lib).group(1)) == 8
    assert int(re.search(r"pub const NOISE_TEXTURE_COUNT: usize = (\d+);", lib).group(1)) == 16
    light = open(os.path.join(REF, "shaders", "light.wgsl")).read()
    consts = dict(re.findall(r"^let (\w+): \w+ = ([^;]+);", light, re.M))
    assert (float(consts["RAY_BIAS"]), float(consts["DISTANCE_MAX"]), float(consts["MAX_VARIANCE"])) == (0.02, 65535.0, 10.0)
