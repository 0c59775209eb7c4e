"""SMAA Tu4x and TAA (smaa.wgsl, taa.wgsl, post_process.rs:1236-1272): CPU-side
checks of the oracle restatement and of the host logic around the double-buffered planes."""
import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from oracle_lib import oracle_plugin

S, U = hk.HikariSettings, hk.Upscale


def f16(a):
    return a.view(np.float16).astype(np.float32)


def test_buffer_sizes_follow_the_reference():
    cpu = oracle_plugin()
    cpu.set_scene(hk.load_cornell())
    # default: SMAA Tu4x at ratio 2 -> the upscaled image is the window size (post_process.rs:718-722)
    cpu.render(hk.cornell_camera(101, 75), S(), frame_number=1, antialias=True)
    assert cpu.engine.buffer_info(F.BUF_TONE_MAPPED)[:2] == (51, 38)
    assert cpu.engine.buffer_info(F.BUF_UPSCALE_OUTPUT)[:2] == (101, 75)
    assert cpu.engine.buffer_info(F.BUF_TAA_OUTPUT)[:2] == (101, 75)
    # SMAA Tu4x at ratio 1: 2x the window
    cpu.render(hk.cornell_camera(40, 24), S(upscale=U.SMAA_TU_1_0), frame_number=2, antialias=True)
    assert cpu.engine.buffer_info(F.BUF_UPSCALE_OUTPUT)[:2] == (80, 48)
    assert cpu.engine.buffer_info(F.BUF_TAA_OUTPUT)[:2] == (80, 48)
    # FSR1 kind: TAA runs at the scaled size (post_process.rs:723-733)
    cpu.render(hk.cornell_camera(90, 66), S(upscale=U.Fsr1(1.5, 0.2)), frame_number=3, antialias=True)
    assert cpu.engine.buffer_info(F.BUF_TAA_OUTPUT)[:2] == cpu.engine.buffer_info(F.BUF_TONE_MAPPED)[:2] == (60, 44)


def test_previous_planes_are_last_frames_planes():
    cpu = oracle_plugin()
    cpu.set_scene(hk.load_cornell())
    s = S(upscale=U.SMAA_TU_1_0)
    keep = {}
    for n in range(1, 5):
        eye = (0.05 * n, 1.0, 4.0)
        cpu.render(hk.Camera(hk.look_at_transform(eye, (0.0, 1.0, 0.0)), 48, 32), s, frame_number=n, antialias=True)
        now = {b: cpu.engine.read(b) for b in (F.BUF_POSITION, F.BUF_VELOCITY_UV, F.BUF_TONE_MAPPED, F.BUF_TAA_OUTPUT)}
        if keep:
            for cur, prev in ((F.BUF_POSITION, F.BUF_PREVIOUS_POSITION), (F.BUF_VELOCITY_UV, F.BUF_PREVIOUS_VELOCITY_UV),
                              (F.BUF_TONE_MAPPED, F.BUF_PREVIOUS_TONE_MAPPED), (F.BUF_TAA_OUTPUT, F.BUF_PREVIOUS_TAA_OUTPUT)):
                assert (cpu.engine.read(prev).view(np.uint8) == keep[cur].view(np.uint8)).all(), (n, prev)
        keep = now
    # the same frame number again (e.g. a second hk_frame_begin to refresh uniforms) must not rotate the planes
    before = cpu.engine.read(F.BUF_PREVIOUS_POSITION)
    cam = hk.Camera(hk.look_at_transform((0.2, 1.0, 4.0), (0.0, 1.0, 0.0)), 48, 32)
    cpu.engine.frame_begin(hk.frame_uniform(s, 4), cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform())
    assert (cpu.engine.read(F.BUF_PREVIOUS_POSITION) == before).all()


def test_smaa_places_the_current_sample_on_the_frame_diagonal():
    cpu = oracle_plugin()
    cpu.set_scene(hk.load_cornell())
    s = S(taa=hk.Taa.NONE)     # ratio 2
    for n in (1, 2, 3):
        cpu.render(hk.cornell_camera(64, 48), s, frame_number=n, antialias=True)
        tm = f16(cpu.engine.read(F.BUF_TONE_MAPPED))
        up = f16(cpu.engine.read(F.BUF_UPSCALE_OUTPUT))
        j = 0 if n % 2 == 0 else 1           # smaa.wgsl:75-77
        assert (up[j::2, j::2, :3] == tm[..., :3]).all()
        assert (up[j::2, j::2, 3] == 1.0).all()
        assert np.isfinite(up).all()
    # static scene and camera, no misses: the other diagonal holds last frame's samples (velocity 0 -> remix weight 0)
    prev = f16(cpu.engine.read(F.BUF_PREVIOUS_TONE_MAPPED))
    geometry = cpu.engine.read(F.BUF_POSITION)[1::2, 1::2, 3][:prev.shape[0], :prev.shape[1]] > 0
    k = 1 - j
    same = (up[k::2, k::2, :3] == prev[..., :3]).all(axis=2)
    assert same[geometry].mean() > 0.97


def test_taa_blend_of_a_static_view_is_exact():
    """No motion: the Catmull-Rom taps collapse onto the texel centre (weights 1, 0, 0, 0, 0), so
    taa = mix(previous_taa, current, 0.1 / ratio) exactly wherever no miss clips the history."""
    cpu = oracle_plugin()
    cpu.set_scene(hk.load_cornell())
    s = S(upscale=U.Fsr1(1.0, 0.2))    # TAA straight on the tone-mapped image
    cam = hk.cornell_camera(72, 56)
    for n in range(1, 5):
        cpu.render(cam, s, frame_number=n, antialias=True)
    prev, cur, out = (f16(cpu.engine.read(b)) for b in (F.BUF_PREVIOUS_TAA_OUTPUT, F.BUF_TONE_MAPPED, F.BUF_TAA_OUTPUT))
    t = np.float32(0.1) / np.float32(1.0)
    p = np.clip(prev[..., :3], 0, 1).astype(np.float32)
    want = (p * (np.float32(1.0) - t) + cur[..., :3] * t).astype(np.float16).astype(np.float32)
    interior = np.zeros(out.shape[:2], bool)
    interior[2:-2, 2:-2] = True
    agree = (want == out[..., :3]).all(axis=2)
    assert agree[interior].mean() > 0.9      # silhouettes take the clipping branch
    assert (out[..., 3] == cur[..., 3]).all()


def test_antialias_band_plan():
    """Exchange D: tone-mapped rows (4 with SMAA Tu4x, 1 for TAA alone) and last frame's TAA rows around the band."""
    from bevy_hikari_amd.distributed import halo_plan

    assert halo_plan(1920, 1080, 1.0, 0, 1, F.STAGE_ANTIALIAS, 5, S().to_c()) == []
    ops = halo_plan(1920, 1080, 2.0, 1, 4, F.STAGE_ANTIALIAS, 5, S().to_c())          # 960x540 traced, bands of 135 rows
    got = sorted((o.buffer, o.peer, o.row_begin, o.row_end, o.row_bytes) for o in ops)
    assert got == [(F.BUF_TONE_MAPPED, 0, 131, 135, 960 * 8), (F.BUF_TONE_MAPPED, 2, 270, 274, 960 * 8),
                   (F.BUF_PREVIOUS_TAA_OUTPUT, 0, 266, 270, 1920 * 8), (F.BUF_PREVIOUS_TAA_OUTPUT, 2, 540, 544, 1920 * 8)]
    fsr = halo_plan(1920, 1080, 1.5, 0, 2, F.STAGE_ANTIALIAS, 5, S(upscale=U.Fsr1(1.5, 0.2)).to_c())   # TAA at the scaled size
    assert sorted((o.buffer, o.row_begin, o.row_end) for o in fsr) == [(F.BUF_TONE_MAPPED, 360, 361), (F.BUF_PREVIOUS_TAA_OUTPUT, 360, 364)]
    moving = halo_plan(1920, 1080, 2.0, 0, 2, F.STAGE_ANTIALIAS | (3 << 8), 5, S().to_c())              # 3 rows of motion
    assert sorted((o.buffer, o.row_begin, o.row_end) for o in moving) == [(F.BUF_TONE_MAPPED, 270, 274), (F.BUF_PREVIOUS_TONE_MAPPED, 270, 277),
                                                                           (F.BUF_PREVIOUS_TAA_OUTPUT, 540, 550)]
