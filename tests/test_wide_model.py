"""The wide walk's algorithm on the CPU (tests/wide_model.py), on trees the product's own host builder made: the records cover every
leaf exactly once, the nearest-first walk loses no candidate (== brute force over every triangle), a walk split into pieces that are
walked separately and merged (the trace stage's work sharing) gives the unsplit walk's result, and the closest hit is the skip-link
walk's - exact ties included (round 5: the leaves' ranks in the reference's flattening decide them).  No GPU: this is the design of csrc/hk_wide.hpp, held to ground truth where the device's bits are not needed."""
import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd.scenes import synthetic_scene
from wide_model import LEAF, Scene, brute_force, build_wide, nodes_of, walk_skip_link, walk_wide


def scenes():
    yard, _ = synthetic_scene(n_boxes=14, n_spheres=4, n_emitters=2, sphere_rings=6, sphere_segs=8)
    return {"cornell": hk.load_cornell(), "yard": yard}


@pytest.fixture(scope="module")
def models():
    return {k: Scene(v) for k, v in scenes().items()}


def rays(seed, n, extent):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-extent, extent, (n, 3)) + np.array([0.0, extent * 0.5, 0.0])
    d = rng.normal(size=(n, 3))
    return o, d / np.linalg.norm(d, axis=1, keepdims=True)


@pytest.mark.parametrize("name", ["cornell", "yard"])
def test_records_cover_every_leaf_once_and_reach_every_inner_node(name):
    sc = scenes()[name]
    trees = [nodes_of(sc.instance_nodes)]
    an = nodes_of(sc.asset_nodes)
    for i in sc.instances:
        sl = slice(i.mesh.node_offset, i.mesh.node_offset + i.mesh.node_count)
        trees.append(tuple(a[sl] for a in an))
    for mn, mx, en, ex in trees:
        rec = build_wide(mn, mx, en, ex)
        count = len(en)
        leaves = sorted(int(e - LEAF) for e in en if e >= LEAF)
        seen, todo, inner = [], [count - 1], set()
        while todo:
            slot = todo.pop()
            assert slot not in inner, "a record reached twice"
            inner.add(slot)
            assert 1 <= len(rec[slot]) <= 4
            for _, _, link in rec[slot]:
                if link >= LEAF:
                    seen.append(link - LEAF)
                else:
                    assert en[link] < LEAF and link in rec   # an inner node's own slot holds its record
                    todo.append(link)
        assert sorted(seen) == leaves                         # every leaf exactly once
        # a record's boxes are the tree's own boxes of those nodes: nothing is recomputed
        for slot, r in rec.items():
            for bmn, bmx, link in r:
                g = link if link < LEAF else int(np.nonzero(en == link)[0][0])
                assert (bmn == mn[g]).all() and (bmx == mx[g]).all()


@pytest.mark.parametrize("name,extent", [("cornell", 1.2), ("yard", 5.0)])
def test_the_walk_loses_no_candidate(models, name, extent):
    sc = models[name]
    o, d = rays(11, 120, extent)
    n_hits = 0
    for k in range(len(o)):
        want = brute_force(sc, o[k], d[k])
        got = walk_wide(sc, o[k], d[k])
        assert (got[0], got[1]) == want, (k, got, want)
        n_hits += want[1] is not None
    assert n_hits > 30


@pytest.mark.parametrize("name,extent", [("cornell", 1.2), ("yard", 5.0)])
@pytest.mark.parametrize("steal_after", [0, 2, 5])
def test_a_split_walk_gives_the_unsplit_walks_result(models, name, extent, steal_after):
    """Work sharing: after `steal_after` records every turn hands the BOTTOM entry of the stack to a helper, which walks it from the
    giver's closest distance (and is split again the same way); results merged under the tie rule."""
    sc = models[name]
    o, d = rays(23, 60, extent)
    more = 0
    for k in range(len(o)):
        whole = walk_wide(sc, o[k], d[k])
        split = walk_wide(sc, o[k], d[k], steal_after=steal_after)
        assert (split[0], split[1]) == (whole[0], whole[1]), (k, split, whole)
        more += split[2] - whole[2]
    assert more >= 0   # (helpers start from a stale distance: never fewer records, usually more - what the published distance is for)


@pytest.mark.parametrize("name,extent", [("cornell", 1.2), ("yard", 5.0)])
def test_same_closest_hit_as_the_reference_walk_and_fewer_dependent_steps(models, name, extent):
    sc = models[name]
    o, d = rays(37, 80, extent)
    wide_steps = ref_steps = 0
    for k in range(len(o)):
        w = walk_wide(sc, o[k], d[k])
        r = walk_skip_link(sc, o[k], d[k])
        # the reference's hit: distance AND triangle (round 5: ties go to the leaf the reference's walk meets first - Scene.key)
        assert w[0] == r[0] and (None if w[1] is None else w[1][2:]) == r[1], (k, w, r)
        wide_steps += w[2]
        ref_steps += r[2]
    assert wide_steps < 0.7 * ref_steps                      # two levels per fetch, nearest first


def test_ties_do_not_depend_on_the_order():
    """Coincident quads (the same mesh instanced three times under the same transform): every hit ties exactly.  The rule is the
    REFERENCE's - the candidate its stackless walk meets first (the leaf of smallest rank) - whichever order the wide walk visits the
    children in, and however it is split: the wide walk, brute force under the rule and the reference's own walk agree."""
    b = hk.SceneBuilder()
    quad_p = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32)
    quad_n = np.tile(np.array([[0, 1, 0]], dtype=np.float32), (4, 1))
    quad_uv = np.zeros((4, 2), dtype=np.float32)
    mesh = b.add_mesh(quad_p, quad_n, quad_uv, np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32))
    mat = b.add_material(hk.standard_material((0.8, 0.8, 0.8, 1.0), (0, 0, 0), 0.5, 0.0, 0.5))
    eye = np.eye(4, dtype=np.float32)
    for _ in range(3):
        b.add_instance(mesh, mat, eye)
    sc = Scene(b.finish())
    for k, (ox, oz) in enumerate([(0.3, 0.2), (-0.5, 0.4), (0.1, -0.7)]):
        o, d = np.array([ox, 2.0, oz]), np.array([0.0, -1.0, 0.0])
        want = brute_force(sc, o, d)
        ref = walk_skip_link(sc, o, d)
        assert want[1] is not None and want[1][0] == min(sc.tlas_rank.values()) and (want[0], want[1][2:]) == (ref[0], ref[1])   # the first instance leaf of the reference's order wins every tie
        for steal in (None, 0, 1):
            got = walk_wide(sc, o, d, steal_after=steal)
            assert (got[0], got[1]) == want


@pytest.mark.parametrize("name,extent", [("cornell", 1.2), ("yard", 5.0)])
def test_pieces_in_any_order_with_a_published_distance_give_the_same_hit(models, name, extent):
    """The lanes of a dry wave: pieces split off at random moments, advanced in a random order, pruning with the closest distance any
    of them has published - five different schedules per ray, all equal to brute force."""
    from wide_model import walk_wide_concurrent

    sc = models[name]
    o, d = rays(41, 40, extent)
    split = 0
    for k in range(len(o)):
        want = brute_force(sc, o[k], d[k])
        for seed in range(5):
            got = walk_wide_concurrent(sc, o[k], d[k], np.random.default_rng(1000 * k + seed))
            assert (got[0], got[1]) == want, (k, seed, got, want)
            split += got[2] > 1
    assert split > 100


def test_pieces_in_any_order_on_exact_ties():
    """... and on the scene where every hit ties three ways."""
    from wide_model import walk_wide_concurrent

    b = hk.SceneBuilder()
    quad_p = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32)
    mesh = b.add_mesh(quad_p, np.tile(np.array([[0, 1, 0]], dtype=np.float32), (4, 1)), np.zeros((4, 2), dtype=np.float32), np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32))
    mat = b.add_material(hk.standard_material((0.8, 0.8, 0.8, 1.0), (0, 0, 0), 0.5, 0.0, 0.5))
    for _ in range(4):
        b.add_instance(mesh, mat, np.eye(4, dtype=np.float32))
    sc = Scene(b.finish())
    o, d = np.array([0.3, 2.0, 0.2]), np.array([0.0, -1.0, 0.0])
    want = brute_force(sc, o, d)
    assert want[1][0] == min(sc.tlas_rank.values()) and want[1][2:] == walk_skip_link(sc, o, d)[1]
    for seed in range(60):
        got = walk_wide_concurrent(sc, o, d, np.random.default_rng(seed), steal_probability=0.7)
        assert (got[0], got[1]) == want, (seed, got)


@pytest.mark.parametrize("name,extent", [("cornell", 1.2), ("yard", 5.0)])
def test_any_hit_rays_occluded_or_not_whatever_the_schedule(models, name, extent):
    """A shadow ray (a limit below which anything occludes): occluded iff ANY candidate lies below the limit - the unsplit walk, split
    walks with stale limits and pieces in random order all say what brute force says."""
    from wide_model import walk_wide_concurrent

    sc = models[name]
    o, d = rays(53, 50, extent)
    rng = np.random.default_rng(5)
    n_occluded = 0
    for k in range(len(o)):
        t_max = float(rng.uniform(0.3, 2.5 * extent))
        want = brute_force(sc, o[k], d[k], t_max)[1] is not None
        assert (walk_wide(sc, o[k], d[k], t_max)[1] is not None) == want
        assert (walk_wide(sc, o[k], d[k], t_max, steal_after=1)[1] is not None) == want
        for seed in range(3):
            assert (walk_wide_concurrent(sc, o[k], d[k], np.random.default_rng(77 * k + seed), t_max)[1] is not None) == want
        n_occluded += want
    assert 5 < n_occluded < len(o) - 5
