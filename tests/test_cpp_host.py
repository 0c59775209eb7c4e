"""The compiled host layer: include/hikari.hpp (C++ mirror of HikariPlugin / HikariSettings / the
three nodes) and examples/cornell (twin of the reference's examples/cornell.rs)."""
import os
import subprocess

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from conftest import ROOT, has_gpu

EXAMPLE = os.path.join(ROOT, "examples", "cornell")


def run(*args):
    return subprocess.run([EXAMPLE, *args], cwd=ROOT, capture_output=True, text=True, timeout=300)


def test_describe_matches_reference_constants():
    r = run("--describe")
    assert r.returncode == 0, r.stderr
    assert "graph=hikari nodes=hikari_prepass,hikari_light,hikari_post_process,hikari_overlay workgroup=8 noise=16" in r.stdout
    assert "defaults_match_library=1 ratio=2.0 abi=8" in r.stdout      # C++ HikariSettings{} == hk_settings_default (lib.rs:435-455)
    assert "tlas_nodes=22 emissives=1" in r.stdout                      # same builder result as the Python path


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_example_fails_loudly_without_gpu():
    r = run("--size", "32", "32", "--frames", "1")
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("by_nodes", [False, True])
def test_cpp_host_renders_the_same_frame_as_the_python_host(tmp_path, by_nodes):
    raw = tmp_path / "tm.bin"
    args = ["--size", "96", "64", "--frames", "5", "--bounces", "2", "--ratio", "1.0", "--raw", str(raw)] + (["--by-nodes"] if by_nodes else [])
    r = run(*args)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(raw, dtype=np.uint16).reshape(64, 96, 4)
    p = hk.HikariPlugin(device=0)
    p.set_scene(hk.load_cornell())
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    for n in range(1, 6):
        p.render(hk.cornell_camera(96, 64), s, frame_number=n)
    want = p.engine.read(F.BUF_TONE_MAPPED)
    assert (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("by_nodes", [False, True])
def test_cpp_host_antialias_matches_the_python_host(tmp_path, by_nodes):
    """--antialias: SMAA Tu4x (ratio 2 -> window size) + TAA through the C++ PostProcessNode / hk_frame_render(HK_FRAME_ANTIALIAS)."""
    raw = tmp_path / "aa.bin"
    args = ["--size", "96", "64", "--frames", "5", "--bounces", "1", "--ratio", "2.0", "--antialias", "--raw", str(raw)] + (["--by-nodes"] if by_nodes else [])
    r = run(*args)
    assert r.returncode == 0, r.stderr
    assert "output size 96x64" in r.stdout
    got = np.fromfile(raw, dtype=np.uint16).reshape(64, 96, 4)
    p = hk.HikariPlugin(device=0)
    p.set_scene(hk.load_cornell())
    s = hk.HikariSettings(indirect_bounces=1)
    for n in range(1, 6):
        p.render(hk.cornell_camera(96, 64), s, frame_number=n, antialias=True)
    assert (got == p.engine.read(F.BUF_TAA_OUTPUT)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("by_nodes", [False, True])
def test_cpp_host_fsr_matches_the_python_host(tmp_path, by_nodes):
    """--fsr: TAA at the scaled size, then FSR1 EASU + RCAS to the window; the overlay image is upscale_output[1]."""
    raw = tmp_path / "fsr.bin"
    args = ["--size", "96", "64", "--frames", "4", "--bounces", "1", "--fsr", "1.5", "0.2", "--antialias", "--raw", str(raw)] + (["--by-nodes"] if by_nodes else [])
    r = run(*args)
    assert r.returncode == 0, r.stderr
    assert "output size 96x64" in r.stdout
    got = np.fromfile(raw, dtype=np.uint16).reshape(64, 96, 4)
    p = hk.HikariPlugin(device=0)
    p.set_scene(hk.load_cornell())
    s = hk.HikariSettings(indirect_bounces=1, upscale=hk.Upscale.Fsr1(1.5, 0.2))
    for n in range(1, 5):
        p.render(hk.cornell_camera(96, 64), s, frame_number=n, antialias=True)
    assert (got == p.engine.read(F.BUF_UPSCALE_SHARPENED)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("bands,antialias,balance,gather", [(2, False, False, False), (3, True, False, False), (3, False, True, False), (4, True, True, True), (3, False, False, True)])
def test_cpp_host_multi_gpu_bands_equal_the_single_context_frame(tmp_path, bands, antialias, balance, gather):
    """--devices 0,0[,0]: the one-process multi-GPU path (hk_multi_*: bands + peer-copy halo exchanges ordered by events) with
    every band on device 0; the gathered image equals the single-context frame bit for bit."""
    raw = tmp_path / "multi.bin"
    args = ["--size", "96", "64", "--frames", "5", "--bounces", "2", "--ratio", "2.0" if antialias else "1.0", "--devices", ",".join(["0"] * bands), "--raw", str(raw)]
    r = run(*(args + (["--antialias"] if antialias else []) + (["--balance"] if balance else []) + (["--gather"] if gather else [])))   # --gather: --raw reads band 0's context alone
    assert r.returncode == 0, r.stderr
    assert f"on {bands} bands" in r.stdout
    if balance:   # HikariMultiGpuPlugin::balance_bands_on_next_frame: the split follows the box (thin bands through it), not the row count
        rh = 32 if antialias else 64
        bounds = [int(x) for x in r.stdout.split("band bounds:")[1].split("\n")[0].split()]
        assert len(bounds) == bands + 1 and bounds[0] == 0 and bounds[-1] == rh and bounds == sorted(set(bounds))
    got = np.fromfile(raw, dtype=np.uint16).reshape(64, 96, 4)
    p = hk.HikariPlugin(device=0)
    p.set_scene(hk.load_cornell())
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_2_0 if antialias else hk.Upscale.SMAA_TU_1_0)
    for n in range(1, 6):
        p.render(hk.cornell_camera(96, 64), s, frame_number=n, antialias=antialias)
    assert (got == p.engine.read(F.BUF_TAA_OUTPUT if antialias else F.BUF_TONE_MAPPED)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("rebuild_at", [0, 3])
def test_cpp_host_animates_through_the_device_refit(tmp_path, rebuild_at):
    """--animate: the compiled host moves two boxes per frame through SceneBuilder::set_instance_transform + HikariPlugin::refit_instances
    (hk_refit_scene_instances; --rebuild-at: hk_rebuild_scene_trees at that frame) - frame for frame what the Python host renders with
    the same calls."""
    raw = tmp_path / "anim.bin"
    args = ["--size", "96", "64", "--frames", "5", "--bounces", "2", "--ratio", "1.0", "--animate", "--deterministic", "--raw", str(raw)] + (["--rebuild-at", str(rebuild_at)] if rebuild_at else [])
    r = run(*args)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(raw, dtype=np.uint16).reshape(64, 96, 4)
    scene = hk.load_cornell()
    rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in scene.instances], dtype=np.float32)
    p = hk.HikariPlugin(device=0, flags=F.CTX_DETERMINISTIC_SCATTER)  # moving objects: the scatter-store race is resolved the same way on both sides
    p.set_scene(scene)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    for n in range(1, 6):
        if n > 1:
            for i, sign in ((6, 1.0), (7, -1.0)):
                m = rest[i].copy()
                m[12] = rest[i][12] + np.float32(0.01) * np.float32(n - 1) * np.float32(sign)
                scene.builder.set_instance_transform(i, m)
            assert p.engine.refit_instances(scene.builder) == 2
            if n == rebuild_at:
                p.engine.rebuild_trees()  # HK_TREE_SAH, the C++ mirror's default too
        p.render(hk.cornell_camera(96, 64), s, frame_number=n)
    want = p.engine.read(F.BUF_TONE_MAPPED)
    assert (got == want).all()
    assert p.engine.stats().scene_device_refits == 4
