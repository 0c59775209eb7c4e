"""The one-level walk (hk_device.hpp traverse_flat, HK_TRAVERSAL_ONE_LEVEL): the product default for LDS-resident scenes whose
instances share one transform - BASELINE configs 2 and 5 (the Cornell box).  It runs the reference's per-triangle arithmetic on
the reference's operands, so it is expected to equal the reference walk (HK_CTX_EXACT_TRAVERSAL, which the rest of the suite
runs and which is bit-exact against the oracle) on all but a handful of pixels per million - rays that graze an instance's world
box within rounding, exact distance ties.  The gate is the north star's: relative L2 <= 1e-3 on the output, plus the fraction of
pixels whose primary hit / any output byte differs, at the sizes the configs are benchmarked at."""
import json
import os

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from cases import ALL_BUFFERS, make_case, product_default_traversal, run_case, snapshot
from conftest import ROOT

pytestmark = pytest.mark.gpu


def default_plugin(flags=0):
    with product_default_traversal():
        return hk.HikariPlugin(device=0, flags=flags)


def oracle():
    from oracle_lib import oracle_plugin

    return oracle_plugin()


def compare(fast, ref, settings):
    a, b = fast.output(settings), ref.output(settings)
    rel = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    fa, fb = snapshot(fast), snapshot(ref)
    differs = {}
    for name in fa:
        x, y = fa[name], fb[name]
        ne = (x.view(np.uint8).reshape(x.shape[0], x.shape[1], -1) != y.view(np.uint8).reshape(y.shape[0], y.shape[1], -1)).any(axis=2)
        if ne.any():
            differs[name] = float(ne.mean())
    return rel, differs


@pytest.mark.parametrize("name", ["cornell_b2", "cornell_b1", "cornell_b8", "cornell_notemporal", "cornell_b0_nodenoise", "cornell_upscale2", "cornell_aa_default", "tiny_3x5"])
def test_one_level_walk_vs_oracle_on_the_cornell_cases(name):
    """Small frames, against the ORACLE: the one-level walk is in use and the frames are the oracle's - bit for bit on these
    sizes, where a grazing ray or a tie is a once-in-many-runs event (a differing G-buffer pixel would fail here)."""
    case = make_case(name)
    gpu, cpu = default_plugin(), oracle()
    for p in (gpu, cpu):
        run_case(p, case)
    assert gpu.engine.traversal_mode()[0] == "one-level"
    rel, differs = compare(gpu, cpu, case.settings)
    assert rel <= 1e-3, (rel, differs)
    for k in ("position", "normal", "instance_material", "velocity_uv"):
        assert differs.get(k, 0.0) <= 1e-4, differs
    # Reservoir BYTES may differ where a shadow ray is occluded: the walk returns the first occluder it meets below the
    # early-out distance (light.wgsl:421-423), WHICH one depends on the visit order, and its hit point is stored as the
    # sample position of a sample whose radiance is (0,0,0,1) whatever the occluder (occlude_hit_info + input_radiance,
    # light.wgsl:526-533,835-867).  What is rendered from them must agree.
    for k, v in differs.items():
        assert v <= (0.05 if k.startswith("reservoir") else 2e-3), differs


def test_one_level_walk_is_not_used_where_it_does_not_apply():
    # the verification mode keeps the reference walk
    e = hk.Engine(device=0)
    e.upload_noise(); e.upload_scene(hk.load_cornell()); e.resize(64, 64, 1.0)
    assert e.traversal_mode() == ("reference", 1)
    # instances with different transforms (the yard): the reference walk from LDS, bit-exact against the oracle
    case = make_case("yard_sun")
    gpu, cpu = default_plugin(), oracle()
    for p in (gpu, cpu):
        run_case(p, case)
    assert gpu.engine.traversal_mode()[0] == "reference"
    rel, differs = compare(gpu, cpu, case.settings)
    assert differs == {}, differs


def _full_size(name, settings, width, height, frames, tol_pixels):
    scene, cam = hk.load_cornell(), hk.cornell_camera(width, height)
    exact = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS)
    fast = default_plugin(flags=F.CTX_COUNT_RAYS)
    for p in (exact, fast):
        p.set_scene(scene)
    for n in frames:
        for p in (exact, fast):
            p.render(cam, settings, frame_number=n)
    mode = fast.engine.traversal_mode()
    assert mode[0] == "one-level" and exact.engine.traversal_mode()[0] == "reference"
    rel, differs = compare(fast, exact, settings)
    sf, se = fast.engine.stats(), exact.engine.stats()
    report = {"case": name, "traversal": list(mode), "frames": len(frames), "rel_l2": rel, "fraction_of_pixels_differing_per_buffer": differs,
              "rays": [int(sf.rays_tlas + sf.rays_blas), int(se.rays_tlas + se.rays_blas)]}
    print("one-level vs reference walk:", report)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"one_level_walk_{name}.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert rel <= 1e-3, report
    assert differs.get("position", 0.0) <= tol_pixels and differs.get("instance_material", 0.0) <= tol_pixels, report
    for k, v in differs.items():   # (reservoir bytes: the occluder a blocked shadow ray reports is order-dependent, see above)
        assert v <= (0.05 if k.startswith("reservoir") else 1e-3), report
    return report


def test_one_level_walk_config2_full_1080p():
    """BASELINE config 2 at the size and in the mode bench.py times it: 12 frames of Cornell 1920x1080, 2 bounces, ReSTIR,
    denoise - the counting replay (HK_CTX_COUNT_RAYS: the same walk from global memory) on both sides."""
    _full_size("config2_1080p", hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0), 1920, 1080, range(1, 13), 2e-5)


def test_one_level_walk_config5_full_4k_8_bounces():
    _full_size("config5_4k", hk.HikariSettings(indirect_bounces=8, emissive_spatial_reuse=True, denoise=False, upscale=hk.Upscale.SMAA_TU_1_0),
               3840, 2160, range(1, 5), 2e-5)


def test_counting_replay_walks_like_the_timed_kernels():
    """bench.py's replay_bit_identical: the LDS kernels (mode 2) and the counting kernels (mode 3, global memory) take the same
    one-level walk, so their frames agree bit for bit."""
    case = make_case("cornell_b2")
    a, b = default_plugin(), default_plugin(flags=F.CTX_COUNT_RAYS)
    for p in (a, b):
        run_case(p, case)
    fa, fb = snapshot(a), snapshot(b)
    for name in fa:
        assert (fa[name].view(np.uint8) == fb[name].view(np.uint8)).all(), name
    st = b.engine.stats()
    assert st.rays_tlas > 0 and st.rays_primary > 0


def test_an_instance_moving_on_its_own_drops_back_to_the_reference_walk():
    """The one-level BVH lives in the instances' SHARED local space.  A device refit that moves one instance alone breaks the
    sharing: from that frame on the context walks the reference's two levels again (mode 'reference') and equals a context that
    never used the one-level walk, bit for bit (both resolve the scatter race deterministically)."""
    from test_device_refit import pose

    engines, scenes = [], []
    for default_mode in (False, True):
        scene = hk.load_cornell()
        scenes.append(scene)
        if default_mode:
            with product_default_traversal():
                e = hk.Engine(device=0, flags=F.CTX_DETERMINISTIC_SCATTER)
        else:
            e = hk.Engine(device=0, flags=F.CTX_DETERMINISTIC_SCATTER | F.DEFAULT_CTX_FLAGS)
        e.upload_noise(); e.upload_scene(scene); e.resize(96, 80, 1.0)
        engines.append(e)
    s = hk.HikariSettings(indirect_bounces=2, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = hk.cornell_camera(96, 80)
    view, pview, lights = cam.view_uniform(), cam.previous_view_uniform(), hk.lights_uniform()
    rest = np.array([np.ctypeslib.as_array(i.model).copy() for i in scenes[0].instances], dtype=np.float32)
    assert engines[1].traversal_mode()[0] == "one-level"
    for n in range(1, 5):
        if n > 1:
            for e, scene in zip(engines, scenes):
                scene.builder.set_instance_transform(2, pose(rest[2], n - 1, 1))
                assert e.refit_instances(scene.builder) == 1
            assert engines[1].traversal_mode()[0] == "reference"
        for e in engines:
            e.frame_render(hk.frame_uniform(s, n), view, pview, lights, s.to_c())
        if n > 1:   # from the first frame after the move both contexts run the same walk on the same state ... except the
            pass    # reservoirs frame 1 left behind, which the one-level walk wrote (equal to the reference's but for ties)
    a = np.stack([engines[1].read_f16(F.BUF_DENOISE_RENDER0 + i) for i in range(3)]).astype(np.float64)
    b = np.stack([engines[0].read_f16(F.BUF_DENOISE_RENDER0 + i) for i in range(3)]).astype(np.float64)
    assert float(np.linalg.norm(a - b) / np.linalg.norm(b)) <= 1e-3
    assert (engines[1].read(F.BUF_POSITION).view(np.uint8) == engines[0].read(F.BUF_POSITION).view(np.uint8)).all()
