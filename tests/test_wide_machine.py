"""The trace stage's bookkeeping on the CPU (tests/wide_machine.py: a port of the state machine of hk_wide.hpp / k_wf_trace_wide):
stack base / marker / tombstones, hand-over of instance-tree and mesh-tree entries, helpers helped in turn, merge into the root,
published distance - random schedules of a dry wave against brute force, closest-hit and any-hit rays, exact ties included."""
import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd.scenes import synthetic_scene
from wide_machine import U32_MAX, Wave
from wide_model import Scene, brute_force


@pytest.fixture(scope="module")
def models():
    yard, _ = synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=2, sphere_rings=6, sphere_segs=8)
    return {"cornell": Scene(hk.load_cornell()), "yard": Scene(yard)}


def rays(seed, n, extent):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-extent, extent, (n, 3)) + np.array([0.0, extent * 0.5, 0.0])
    d = rng.normal(size=(n, 3))
    return o, d / np.linalg.norm(d, axis=1, keepdims=True)


def as_key(hit):
    return None if hit[2] == U32_MAX else (hit[1], hit[2])


@pytest.mark.parametrize("name,extent", [("cornell", 1.2), ("yard", 5.0)])
@pytest.mark.parametrize("lanes,share_steps,give", [(2, 0, 1.0), (8, 0, 1.0), (8, 2, 0.5), (16, 1, 0.7)])
def test_closest_hits_of_a_dry_wave(models, name, extent, lanes, share_steps, give):
    """One or two rays in a wave of 2..16 lanes, the others idle: whatever is handed over, to whom, and when."""
    sc = models[name]
    o, d = rays(61, 36, extent)
    lost = 0
    stats = {}
    for k in range(0, len(o), 2):
        batch = [(o[k], d[k], np.inf, 0.0, U32_MAX)] + ([(o[k + 1], d[k + 1], np.inf, 0.0, U32_MAX)] if lanes > 2 else [])
        w = Wave(sc, lanes, batch, np.random.default_rng(k), share_min=1, share_steps=share_steps, give_probability=give)
        res, turns = w.run()
        for i, r in enumerate(batch):
            want = brute_force(sc, r[0], r[1])
            got = res[i]
            assert (got[0] if want[1] is not None else np.inf, as_key(got)) == (want[0] if want[1] is not None else np.inf, None if want[1] is None else want[1][2:]), (k, i, got, want)
        lost += sum(l.lost for l in w.lanes)
        for key, v in w.stats.items():
            stats[key] = stats.get(key, 0) + v
    assert lost == 0
    # the paths this is about were taken: both kinds of entries handed over, and (in waves with lanes to spare) helpers helped in turn
    assert stats["instance_tree_entries"] > 0 and stats["mesh_tree_entries"] > 0 and (lanes == 2 or stats["from_helpers"] > 0), stats


@pytest.mark.parametrize("name,extent", [("cornell", 1.2), ("yard", 5.0)])
def test_any_hit_rays_of_a_dry_wave(models, name, extent):
    """Shadow rays: a limit, an early-out distance below which the first hit ends the piece, an excluded instance."""
    sc = models[name]
    o, d = rays(67, 40, extent)
    rng = np.random.default_rng(9)
    n_occ = 0
    for k in range(len(o)):
        t_max = float(rng.uniform(0.3, 2.5 * extent))
        early = t_max * float(rng.uniform(0.0, 1.0))
        w = Wave(sc, 8, [(o[k], d[k], t_max, early, U32_MAX)], np.random.default_rng(k), share_min=1, share_steps=int(rng.integers(0, 3)), give_probability=0.8)
        res, _ = w.run()
        want = brute_force(sc, o[k], d[k], t_max)[1] is not None
        assert (res[0][1] != U32_MAX) == want, (k, res[0], want)
        n_occ += want
    assert 5 < n_occ < len(o) - 5


def test_exact_ties_in_a_dry_wave():
    b = hk.SceneBuilder()
    quad_p = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32)
    mesh = b.add_mesh(quad_p, np.tile(np.array([[0, 1, 0]], dtype=np.float32), (4, 1)), np.zeros((4, 2), dtype=np.float32), np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32))
    mat = b.add_material(hk.standard_material((0.8, 0.8, 0.8, 1.0), (0, 0, 0), 0.5, 0.0, 0.5))
    for _ in range(5):
        b.add_instance(mesh, mat, np.eye(4, dtype=np.float32))
    sc = Scene(b.finish())
    o, d = np.array([0.3, 2.0, 0.2]), np.array([0.0, -1.0, 0.0])
    want = brute_force(sc, o, d)
    for seed in range(40):
        w = Wave(sc, 8, [(o, d, np.inf, 0.0, U32_MAX)], np.random.default_rng(seed), share_min=1, share_steps=0, give_probability=0.6)
        res, _ = w.run()
        assert (res[0][0], as_key(res[0])) == (want[0], want[1][2:]), (seed, res[0], want)
