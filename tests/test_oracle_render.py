"""Oracle frame-level checks: committed golden fixtures, determinism across thread counts,
row-range independence (the property band sharding relies on)."""
import hashlib
import os

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import CASE_NAMES, diff_buffers, make_case, run_case, snapshot
from conftest import ROOT
from oracle_lib import default_threads, oracle_plugin, set_threads

GOLDEN = os.path.join(ROOT, "tests", "golden")


def check_against_golden(snap, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    for key in g.files:
        if key.startswith("sha256_"):
            got = np.frombuffer(hashlib.sha256(snap[key[7:]].tobytes()).digest(), dtype=np.uint8)
            assert (got == g[key]).all(), f"{name}: buffer {key[7:]} differs from the golden fixture"
    for key in ("tone_mapped", "denoise_render2", "variance2"):
        assert (snap[key].view(np.uint8) == g[key].view(np.uint8)).all()
    return g


@pytest.mark.parametrize("name", CASE_NAMES)
def test_oracle_matches_golden(name):
    case = make_case(name)
    p = oracle_plugin()
    run_case(p, case)
    snap = snapshot(p)
    g = check_against_golden(snap, name)
    st = p.engine.stats()
    assert [st.rays_primary, st.rays_tlas, st.rays_blas] == list(g["rays"])
    out = p.output(case.settings)
    assert np.isfinite(out).all() and (name == "background_only" or out[..., :3].max() > 0.05)


def test_oracle_is_thread_count_invariant():
    case = make_case("cornell_b2")
    snaps = []
    for nt in (1, max(2, default_threads())):
        set_threads(nt)
        p = oracle_plugin()
        run_case(p, case)
        snaps.append(snapshot(p))
    set_threads(default_threads())
    assert diff_buffers(snaps[0], snaps[1]) == {}


def test_pass_rows_are_independent():
    """Running every pass over two row ranges gives the same frame as one full dispatch."""
    case = make_case("cornell_b2")
    s = case.settings
    full, split = oracle_plugin(), oracle_plugin()
    for p in (full, split):
        p.set_scene(case.scene)
        p.engine.resize(case.camera.width, case.camera.height, s.upscale.ratio())
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    h = case.camera.height
    cut = 29
    for n in (1, 2, 3):
        frame = hk.frame_uniform(s, n)
        for p, ranges in ((full, [(0, 0)]), (split, [(0, cut), (cut, h)])):
            e = p.engine
            e.frame_begin(frame, view, pview, case.lights)
            e.set_view_options(s.taa, s.upscale.kind, s.upscale.sharpness_)
            order = [(F.PASS_PREPASS, 0), (F.PASS_FULL_SCREEN_ALBEDO, 0), (F.PASS_DIRECT_LIT, 0), (F.PASS_DIRECT_EMISSIVE, 0), (F.PASS_INDIRECT, 0),
                     (F.PASS_INDIRECT_SPATIAL_REUSE, 0)]
            for ch in range(3):
                order += [(F.PASS_DEMODULATION, ch)] + [(F.PASS_DENOISE_L0 + l, ch) for l in range(4)]
            order += [(F.PASS_TONE_MAPPING, 1)]
            for pid, arg in order:
                for r0, r1 in ranges:
                    e.pass_run(pid, arg, r0, r1)
    assert diff_buffers(snapshot(full), snapshot(split)) == {}


def test_nodes_equal_frame_render():
    """PrepassNode/LightNode/PostProcessNode.run (reference order, emissive spatial before indirect)
    == hk_frame_render (stage order)."""
    case = make_case("cornell_upscale2")
    a, b = oracle_plugin(), oracle_plugin()
    for p, by_nodes in ((a, False), (b, True)):
        p.set_scene(case.scene)
        for n in case.frames:
            p.render(case.camera, case.settings, lights=case.lights, frame_number=n, by_nodes=by_nodes, antialias=case.antialias)
    assert diff_buffers(snapshot(a), snapshot(b)) == {}


@pytest.mark.parametrize("name", ["cornell_upscale2", "cornell_aa_default"])
def test_host_supplied_gbuffer_gives_the_same_frames(name):
    """HK_FRAME_EXTERNAL_GBUFFER (a host that keeps its raster prepass): same frames as the internal prepass,
    including the previous-frame planes the anti-aliasing tail reads."""
    from cases import run_case_with_host_gbuffer

    case = make_case(name)
    a, b = oracle_plugin(), oracle_plugin()
    run_case_with_host_gbuffer(a, b, case)
    assert diff_buffers(snapshot(a), snapshot(b)) == {}
