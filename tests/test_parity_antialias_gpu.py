"""Parity of the anti-aliasing tail (taa.wgsl, smaa.wgsl, FSR1 EASU / RCAS: SURVEY 8f-4) with the oracle, bit for bit.  Split from
test_parity_gpu.py."""

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import diff_buffers, oracle, snapshot

pytestmark = pytest.mark.gpu


AA_BUFFERS = ("tone_mapped", "previous_tone_mapped", "previous_position", "previous_velocity_uv", "upscale_output", "taa_output",
              "previous_taa_output")
AA_CASES = {
    "smaa_ratio2_taa": dict(size=(128, 96), settings=dict(indirect_bounces=2)),                                     # the reference's defaults
    "smaa_ratio1_taa": dict(size=(72, 56), settings=dict(indirect_bounces=1, upscale=hk.Upscale.SMAA_TU_1_0)),      # 2x the window
    "smaa_odd_no_taa": dict(size=(101, 75), settings=dict(indirect_bounces=1, taa=hk.Taa.NONE)),                    # odd sizes: quads hang over the edge
    "fsr_ratio15_taa": dict(size=(90, 66), settings=dict(indirect_bounces=1, upscale=hk.Upscale.Fsr1(1.5, 0.2))),   # TAA at the scaled size, EASU + RCAS
    "fsr_ratio2_no_taa": dict(size=(101, 75), settings=dict(indirect_bounces=1, upscale=hk.Upscale.Fsr1(2.0, 0.0), taa=hk.Taa.NONE)),
    "fsr_ratio1_taa": dict(size=(64, 40), settings=dict(indirect_bounces=0, upscale=hk.Upscale.Fsr1(1.0, 1.5))),      # EASU at 1:1
}


@pytest.mark.parametrize("name", sorted(AA_CASES))
def test_antialias_bit_exact_vs_oracle(name):
    """SMAA Tu4x / TAA / FSR1 after the light path, static scene: every buffer bit-exact, frame by frame,
    through hk_frame_render(HK_FRAME_ANTIALIAS) on the GPU and dispatch by dispatch on the oracle."""
    case = AA_CASES[name]
    s = hk.HikariSettings(**case["settings"])
    cam = hk.cornell_camera(*case["size"])
    gpu, cpu = hk.HikariPlugin(device=0), oracle()
    scene = hk.load_cornell()
    for p in (gpu, cpu):
        p.set_scene(scene)
    for n in range(1, 7):
        gpu.render(cam, s, frame_number=n, antialias=True)
        cpu.render(cam, s, frame_number=n, antialias=True, by_nodes=True)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"{name} frame {n}: {bad}"
    img = gpu.final_image(s)
    assert np.isfinite(img).all() and img[..., :3].mean() > 0.05   # (the differential blend of SMAA may overshoot 1.0)


def test_antialias_kernels_under_motion_on_identical_inputs():
    """Camera and object motion exercise the reprojection, miss and clipping branches.  The light passes
    race under motion (reference behaviour), so the inputs of the AA dispatches are taken from the oracle
    and written into the GPU context: on identical inputs the three kernels must be bit-exact."""
    from bevy_hikari_amd.scenes import animate, synthetic_camera, synthetic_scene

    scene, sun = synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=3, sphere_rings=6, sphere_segs=8)
    lights = hk.lights_uniform(directional=sun)
    inputs = (F.BUF_POSITION, F.BUF_VELOCITY_UV, F.BUF_INSTANCE_MATERIAL, F.BUF_PREVIOUS_POSITION, F.BUF_PREVIOUS_VELOCITY_UV,
              F.BUF_TONE_MAPPED, F.BUF_PREVIOUS_TONE_MAPPED, F.BUF_PREVIOUS_TAA_OUTPUT)
    for s in (hk.HikariSettings(indirect_bounces=1), hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.Fsr1(1.0, 0.2))):
        gpu, cpu = hk.HikariPlugin(device=0), oracle()
        cur = scene
        for p in (gpu, cpu):
            p.set_scene(cur)
        clipped = 0
        for n in range(1, 7):
            cam = hk.Camera(hk.look_at_transform((6.4 + 0.15 * n, 4.4, 8.0 - 0.1 * n), (0.0, 0.6, 0.0)), 112, 80)
            if n > 1:
                cur = animate(cur, n - 1, movers=(3, 9, 16, 19))
                for p in (gpu, cpu):
                    p.update_instances(cur)
            cpu.render(cam, s, lights=lights, frame_number=n, antialias=True)
            # same frame on the GPU up to tone mapping (keeps sizes, uniforms and plane parity in step) ...
            gpu.render(cam, s, lights=lights, frame_number=n)
            for b in inputs:      # ... then the oracle's inputs, and only the AA dispatches
                gpu.engine.write(b, cpu.engine.read(b))
            gpu.post_process.run_antialias(s)
            for b in (F.BUF_UPSCALE_OUTPUT, F.BUF_TAA_OUTPUT, F.BUF_UPSCALE_SHARPENED):
                a, o = gpu.engine.read(b), cpu.engine.read(b)
                assert (a == o).all(), (n, b, int((a != o).any(axis=2).sum()))
            vel = cpu.engine.read(F.BUF_VELOCITY_UV)[..., :2]
            clipped += int((np.abs(vel).max(axis=2) > 1e-4).sum())
        assert clipped > 1000      # the motion branches really ran
