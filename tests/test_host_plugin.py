"""The Python host mirrors the reference's plugin interface (names, defaults, dispatch order)."""
import dataclasses

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F


def test_settings_fields_and_defaults():  # lib.rs:400-455
    names = [f.name for f in dataclasses.fields(hk.HikariSettings)]
    assert names == ["direct_validate_interval", "emissive_validate_interval", "max_temporal_reuse_count", "max_spatial_reuse_count",
                     "max_reservoir_lifetime", "solar_angle", "indirect_bounces", "max_indirect_luminance", "clear_color", "temporal_reuse",
                     "emissive_spatial_reuse", "indirect_spatial_reuse", "denoise", "taa", "upscale"]
    s = hk.HikariSettings()
    assert s.upscale == hk.Upscale.SMAA_TU_2_0 and s.upscale.ratio() == 2.0 and s.taa == hk.Taa.Jasmine
    assert hk.Upscale.Fsr1(3.0, 0.5).ratio() == 2.0 and hk.Upscale.SmaaTu4x(0.5).ratio() == 1.0 and hk.Upscale.SmaaTu4x(1.5).sharpness() == 0.0
    assert hk.graph.NAME == "hikari" and hk.graph.node.LIGHT == "hikari_light" and hk.graph.node.POST_PROCESS == "hikari_post_process"
    u = hk.HikariUniversalSettings()
    assert u.build_mesh_acceleration_structure and u.build_instance_acceleration_structure


class _Recorder:
    def __init__(self):
        self.calls = []

    def pass_run(self, p, arg=0, row_begin=0, row_end=0):
        self.calls.append((F.PASS_NAMES[p], arg))

    def set_view_options(self, *a):
        pass


def test_node_dispatch_order():
    """light.rs:646-697 and post_process.rs:1190-1234."""
    rec = _Recorder()
    s = hk.HikariSettings(indirect_bounces=2, emissive_spatial_reuse=True)
    hk.PrepassNode(rec).run(s)
    hk.LightNode(rec).run(s)
    hk.PostProcessNode(rec).run(s)
    names = [c[0] for c in rec.calls]
    assert names[:7] == ["prepass", "full_screen_albedo", "direct_lit", "direct_emissive", "emissive_spatial_reuse", "indirect_lit_ambient",
                         "indirect_spatial_reuse"]
    per_channel = ["demodulation", "denoise_l0", "denoise_l1", "denoise_l2", "denoise_l3"]
    assert names[7:] == per_channel * 3 + ["tone_mapping"]
    assert [c[1] for c in rec.calls[7:22]] == [0] * 5 + [1] * 5 + [2] * 5
    rec = _Recorder()
    s = hk.HikariSettings(indirect_bounces=0, indirect_spatial_reuse=False, denoise=True)
    hk.LightNode(rec).run(s)
    hk.PostProcessNode(rec).run(s)
    names = [c[0] for c in rec.calls]
    assert "indirect_spatial_reuse" not in names and names.count("demodulation") == 2  # post_process.rs:949-954
    rec = _Recorder()
    hk.PostProcessNode(rec).run(hk.HikariSettings(denoise=False))
    assert rec.calls == [("tone_mapping", 0)]


def test_camera_uniform_is_bevy_reverse_z():
    cam = hk.cornell_camera(256, 256)
    v = cam.view_uniform()
    assert list(v.world_position) == [0.0, 1.0, 4.0] and v.projection[15] == 0.0 and abs(v.projection[14] - 0.1) < 1e-7 and v.projection[11] == -1.0
    import numpy as np

    vp = np.array(v.view_proj, dtype=np.float64).reshape(4, 4).T
    clip = vp @ np.array([0.0, 1.0, 0.0, 1.0])  # the look-at target, 4 units away
    assert abs(clip[2] / clip[3] - 0.1 / 4.0) < 1e-7 and abs(clip[0]) < 1e-7 and abs(clip[1]) < 1e-7
    ivp = np.array(v.inverse_view_proj, dtype=np.float64).reshape(4, 4).T
    assert np.allclose(ivp @ vp, np.eye(4), atol=1e-5)
