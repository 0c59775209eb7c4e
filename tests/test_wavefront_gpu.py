"""The queue-based schedule of indirect_lit_ambient (HK_CTX_WAVEFRONT, kernels_wavefront.hip): set-up / persistent trace with
lane refill / shade per bounce / final.  Bar: the same bytes in every buffer as the fused kernel, and as the oracle."""
import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import diff_buffers, make_case, motion_case, product_default_traversal, random_case, run_motion_case, snapshot

pytestmark = pytest.mark.gpu


def oracle():
    from oracle_lib import oracle_plugin

    return oracle_plugin()


@pytest.mark.parametrize("name", ["cornell_b2", "cornell_b8", "cornell_upscale2", "yard_sun", "yard_textured", "yard_no_emitters", "flight_helmet", "yard_ortho",
                                  "tiny_3x5", "background_only", "cornell_notemporal", "cornell_aa_default", "cornell_b1"])
def test_wavefront_bit_exact_vs_oracle_every_frame(name):
    """Named cases (LDS-resident and global-memory scenes, textures, ortho camera, 1 x 8 bounces; cornell_b1 / aa_default have
    fewer than two bounces and must quietly take the fused kernel)."""
    case = make_case(name)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT), oracle()
    gpu.set_scene(case.scene)
    cpu.set_scene(case.scene)
    for n in case.frames:
        for p in (gpu, cpu):
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"{name} frame {n}: {bad}"


@pytest.mark.parametrize("seed", range(24))
def test_wavefront_random_settings_vs_oracle(seed):
    case = random_case(seed)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT), oracle()
    gpu.set_scene(case.scene)
    cpu.set_scene(case.scene)
    for n in case.frames:
        for p in (gpu, cpu):
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"random{seed} frame {n}: {bad}"


@pytest.mark.parametrize("seed", range(8))
def test_wavefront_motion_with_resolved_race_vs_oracle(seed):
    """Moving camera + instances: the parked scatter stores (HK_CTX_DETERMINISTIC_SCATTER) are issued by k_wf_final here."""
    case = motion_case(seed)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_DETERMINISTIC_SCATTER | F.CTX_WAVEFRONT), oracle()

    def check(n):
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"motion{seed} frame {n}: {bad}"

    run_motion_case((gpu, cpu), case, check)


def _large(size=(320, 180), bounces=3):
    from bevy_hikari_amd.scenes import synthetic_camera, synthetic_large

    scene, sun = synthetic_large()
    s = hk.HikariSettings(indirect_bounces=bounces, upscale=hk.Upscale.SMAA_TU_1_0)
    return scene, synthetic_camera(*size, extent=9.0), s, hk.lights_uniform(directional=sun)


def test_wavefront_sponza_class_vs_oracle():
    """Config 3 stand-in (256 k triangles, global-memory traversal, 3 bounces), reference traversal order: every buffer vs the oracle."""
    scene, cam, s, lights = _large()
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT), oracle()
    for p in (gpu, cpu):
        p.set_scene(scene)
    for n in (1, 2, 3):
        for p in (gpu, cpu):
            p.render(cam, s, lights=lights, frame_number=n)
        bad = diff_buffers(snapshot(gpu), snapshot(cpu))
        assert bad == {}, f"frame {n}: {bad}"


def test_default_schedule_is_wavefront_for_large_scenes_and_equals_fused():
    """Product defaults (direction-threaded BVHs, wavefront for scenes beyond the LDS copy) against the same context forced to the
    fused kernel: identical bytes, at 1280x720 where a wave's 64 rays no longer come from one tile."""
    scene, cam, s, lights = _large(size=(1280, 720))
    with product_default_traversal():
        a, b = hk.HikariPlugin(device=0), hk.HikariPlugin(device=0, flags=F.CTX_FUSED_INDIRECT)
    for p in (a, b):
        p.set_scene(scene)
        for n in (1, 2, 3):
            p.render(cam, s, lights=lights, frame_number=n)
    assert diff_buffers(snapshot(a), snapshot(b)) == {}
    out = a.output(s)
    assert np.isfinite(out).all() and out[..., :3].max() > 0.05


def test_wavefront_config5_4k_8_bounces_equals_fused():
    """Cornell 3840x2160, 8 bounces, both spatial passes, no denoise: 2.9 M paths through 8 shade stages; row-range independence."""
    s = hk.HikariSettings(indirect_bounces=8, emissive_spatial_reuse=True, denoise=False, upscale=hk.Upscale.SMAA_TU_1_0)
    scene, cam = hk.load_cornell(), hk.cornell_camera(3840, 2160)
    a, b = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT), hk.HikariPlugin(device=0)
    for p in (a, b):
        p.set_scene(scene)
        for n in (1, 2):
            p.render(cam, s, frame_number=n)
    ref = snapshot(b)
    assert diff_buffers(snapshot(a), ref) == {}
    e = a.engine  # the last frame's indirect dispatch again, in two row ranges (a band of a sharded frame)
    e.pass_run(F.PASS_INDIRECT, 0, 0, 1000)
    e.pass_run(F.PASS_INDIRECT, 0, 1000, 2160)
    e.pass_run(F.PASS_INDIRECT_SPATIAL_REUSE, 0, 0, 2160)
    assert diff_buffers(snapshot(a), ref) == {}


def test_wavefront_survives_resize_and_scene_change():
    """The scratch follows hk_resize; one context through several sizes and scenes stays equal to the fused one.  (Frame numbers
    restart with every scene, so the first frames reproject into another scene's reservoirs and scatter-store: the race is resolved
    the oracle's way on both sides to make the two contexts comparable.)"""
    a = hk.HikariPlugin(device=0, flags=F.CTX_WAVEFRONT | F.CTX_DETERMINISTIC_SCATTER)
    b = hk.HikariPlugin(device=0, flags=F.CTX_FUSED_INDIRECT | F.CTX_DETERMINISTIC_SCATTER)
    for name in ("cornell_b2", "yard_sun", "cornell_upscale2"):
        case = make_case(name)
        for p in (a, b):
            p.set_scene(case.scene)
            for n in case.frames:
                p.render(case.camera, case.settings, lights=case.lights, frame_number=n, antialias=case.antialias)
        assert diff_buffers(snapshot(a), snapshot(b)) == {}, name


@pytest.mark.parametrize("world,case_name", [(2, "cornell_b2"), (3, "yard_sun")])
def test_wavefront_bands_equal_single_context(tmp_path, world, case_name, monkeypatch):
    """Band-sharded frames (one rank per band, all on device 0, halos over gloo) with the wavefront schedule forced in every rank:
    each band's dispatch runs set-up / trace / shade / final over its own rows; the union equals the fused single-context frame."""
    import socket

    import torch.multiprocessing as mp

    from cases import run_case
    from test_parity_gpu import _gpu_band_worker

    monkeypatch.setenv("HIKARI_HIP_DEFAULT_CTX_FLAGS", str(F.CTX_EXACT_TRAVERSAL | F.CTX_WAVEFRONT))  # inherited by the spawned ranks
    from rendezvous import new_rendezvous

    mp.spawn(_gpu_band_worker, args=(world, new_rendezvous(), case_name, str(tmp_path)), nprocs=world, join=True)
    case = make_case(case_name)
    ref = hk.HikariPlugin(device=0, flags=F.CTX_FUSED_INDIRECT)
    run_case(ref, case)
    full = snapshot(ref)
    for rank in range(world):
        d = np.load(tmp_path / f"rank{rank}.npz")
        b0, b1 = int(d["b0"]), int(d["b1"])
        for key in d.files:
            if key in ("b0", "b1") or key.endswith("_rows"):
                continue
            assert (d[key].view(np.uint8) == full[key][b0:b1].view(np.uint8)).all(), f"rank {rank} [{b0},{b1}) differs in {key}"
