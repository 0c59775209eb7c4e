"""A CPU model of the wide walk (csrc/hk_wide.hpp, kernels_wavefront.hip k_build_wide / k_wf_trace_wide): the derivation of the 128-B
records from a flatten_custom tree, the nearest-first walk with a stack over both levels of a scene, the order-independent tie rule,
and the splitting of a walk into pieces that are walked separately and merged (the work sharing of a dry wave).  TEST INFRASTRUCTURE:
float64 / float32 numpy arithmetic of its own - it checks the ALGORITHM (no candidate is lost, a split walk gives the unsplit walk's
result, ties do not depend on the order), not the device's bits; tests/test_wide_model.py holds it to brute force."""
import numpy as np

LEAF = 0x80000000
NONE = 0xFFFFFFFF
LEAVE = 0xFFFFFFFE


def nodes_of(arr):
    """HkNode ctypes array -> (min[n,3], max[n,3], entry[n], exit[n])."""
    n = len(arr)
    mn = np.array([[a.min[0], a.min[1], a.min[2]] for a in arr], dtype=np.float64).reshape(n, 3)
    mx = np.array([[a.max[0], a.max[1], a.max[2]] for a in arr], dtype=np.float64).reshape(n, 3)
    en = np.array([a.entry_index for a in arr], dtype=np.int64)
    ex = np.array([a.exit_index for a in arr], dtype=np.int64)
    return mn, mx, en, ex


def build_wide(mn, mx, entry, exit_):
    """k_build_wide: one record per INNER node of one flatten_custom tree, at the node's own slot; the root's record (which
    flatten_custom does not store) in the last slot.  A record = up to four (box min, box max, link); link = LEAF | id, or the slot of
    an inner node.  Returns {slot: [(min, max, link), ...]}."""
    count = len(entry)
    records = {}
    for x in range(count):
        is_root = x + 1 == count
        if not is_root and entry[x] >= LEAF:
            continue
        rec = []

        def add(g):
            rec.append((mn[g], mx[g], int(entry[g]) if entry[g] >= LEAF else g))

        if is_root:
            first, limit = 0, count
        else:
            first, limit = x + 1, int(exit_[x])
        if is_root and count == 1:
            add(0)
        else:
            c = first
            for _side in range(2):
                if c >= limit:
                    break
                if entry[c] >= LEAF:
                    add(c)
                else:
                    cl, g = int(exit_[c]), c + 1
                    for _gs in range(2):
                        if g >= cl:
                            break
                        add(g)
                        g = int(exit_[g])
                c = int(exit_[c])
        assert len(rec) <= 4
        records[x] = rec
    return records


def slab(mn, mx, o, inv):
    """intersects_aabb, light.wgsl:344-362: entry distance or None."""
    with np.errstate(invalid="ignore", over="ignore"):
        t1, t2 = (mn - o) * inv, (mx - o) * inv
    lo, hi = np.fmin(t1, t2), np.fmax(t1, t2)
    t_min, t_max = lo.max(), hi.min()
    return float(t_min) if (t_max >= t_min and t_max >= 0.0) else None


def triangle(o, d, v0, v1, v2):
    """Moeller-Trumbore as intersects_triangle does it (light.wgsl:364-398): distance or inf."""
    ab, ac = v1 - v0, v2 - v0
    u_vec = np.cross(d, ac)
    det = float(np.dot(ab, u_vec))
    if abs(det) < 1e-12:
        return np.inf
    inv_det = 1.0 / det
    ao = o - v0
    u = float(np.dot(ao, u_vec)) * inv_det
    if u < 0.0 or u > 1.0:
        return np.inf
    v_vec = np.cross(ao, ab)
    v = float(np.dot(d, v_vec)) * inv_det
    if v < 0.0 or u + v > 1.0:
        return np.inf
    t = float(np.dot(ac, v_vec)) * inv_det
    return t if t > 1e-6 else np.inf


class Scene:
    """The two levels of a SceneData on the CPU: instance tree + records, per-instance mesh tree + records, triangles, transforms."""

    def __init__(self, scene):
        self.tlas = nodes_of(scene.instance_nodes)
        self.tlas_wide = build_wide(*self.tlas)
        an = nodes_of(scene.asset_nodes)
        self.instances = []
        prims = np.array([[[v.position[0], v.position[1], v.position[2]] for v in p.vertices] for p in scene.primitives], dtype=np.float64)
        for i in scene.instances:
            m = i.mesh
            sl = slice(m.node_offset, m.node_offset + m.node_count)
            tree = tuple(a[sl] for a in an)
            model = np.array(list(i.model), dtype=np.float64).reshape(4, 4).T   # column-major in the struct
            # the rank of a leaf = its position in the reference's flattening: the order in which the reference's stackless walk meets
            # the leaves (hk_kernels.hpp WideTrees; k_build_wide writes them)
            rank = {int(e - LEAF): pos for pos, e in enumerate(tree[2]) if e >= LEAF}
            self.instances.append({"tree": tree, "wide": build_wide(*tree), "inverse": np.linalg.inv(model), "primitive": int(m.primitive), "rank": rank})
        self.tlas_rank = {int(e - LEAF): pos for pos, e in enumerate(self.tlas[2]) if e >= LEAF}
        self.prims = prims

    def key(self, inst, ident):
        """What decides a tie between two candidates at exactly the same distance - the REFERENCE's rule (round 5): the leaf its walk
        meets first, i.e. (rank of the instance's leaf, rank of the triangle's leaf inside the instance) - followed by the identity
        (instance, primitive) the walk reports."""
        return (self.tlas_rank[inst], self.instances[inst]["rank"][ident], inst, self.instances[inst]["primitive"] + ident)


BETTER = lambda d, key, best: d < best[0] or (d == best[0] and best[1] is not None and key < best[1])   # wide_triangle's tie rule


def walk_wide(sc, origin, direction, t_max=np.inf, start=None, bound=None, steal_after=None):
    """The nearest-first walk over the records.  Returns (distance, (instance, primitive) or None, records visited).
    start: [(level, instance, link)] pending entries instead of the root (a helper's piece); steal_after: after that many records the
    BOTTOM entry of the stack is handed to a helper (recursively, through this function) whenever there is one - a dry wave's work
    sharing; helpers start from the giver's closest distance and their result is merged under the tie rule."""
    o, d = np.asarray(origin, np.float64), np.asarray(direction, np.float64)
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    # a helper starts from the giver's closest distance; if that distance is a HIT's, a candidate at exactly that distance must still be
    # accepted here (it may carry the smaller key): the piece's own limit is the next float above it (hk_wide.hpp: the hand-over)
    best = [t_max if bound is None else min(t_max, bound), None]
    visited = 0
    stack = list(start) if start is not None else [("t", None, len(sc.tlas[2]) - 1)]   # entries: (level, instance, link); link = slot or LEAF | id
    helpers = []

    def local(inst):
        inv_m = sc.instances[inst]["inverse"]
        lo = inv_m @ np.append(o, 1.0)
        ld = (inv_m @ np.append(d, 0.0))[:3]
        with np.errstate(divide="ignore"):
            return lo[:3] / lo[3], ld, 1.0 / ld

    while stack:
        if steal_after is not None and visited >= steal_after and len(stack) >= 2:
            helpers.append((stack.pop(0), best[0], best[1] is not None))   # the bottom entry (the farthest pending subtree) + the giver's closest distance NOW
        level, inst, link = stack.pop()
        if link >= LEAF:
            ident = link - LEAF
            if level == "t":      # an instance leaf: enter its mesh tree at the root record
                stack.append(("b", ident, len(sc.instances[ident]["tree"][2]) - 1))
            else:                 # a triangle of instance `inst`
                lo, ld, _ = local(inst)
                p = sc.prims[sc.instances[inst]["primitive"] + ident]
                t = triangle(lo, ld, p[0], p[1], p[2])
                key = sc.key(inst, ident)
                if t <= best[0] and BETTER(t, key, best):
                    best[0], best[1] = t, key
            continue
        visited += 1
        if level == "t":
            rec, ro, rinv = sc.tlas_wide[link], o, inv
        else:
            lo, _, linv = local(inst)
            rec, ro, rinv = sc.instances[inst]["wide"][link], lo, linv
        hits = []
        for mn, mx, child in rec:
            t = slab(mn, mx, ro, rinv)
            if t is not None and t <= best[0]:     # (<=: a candidate that ties is still opened)
                hits.append((t, child))
        hits.sort(key=lambda h: -h[0])              # farthest first onto the stack: the nearest is popped next
        for _, child in hits:
            stack.append((level, inst, child))
    for h, h_bound, h_is_hit in helpers:   # (a helper never sees what the giver finds later: the stalest limit the device can have)
        hd, hk, hv = walk_wide(sc, o, d, t_max, start=[h], bound=(np.nextafter(h_bound, np.inf) if h_is_hit else h_bound), steal_after=steal_after)
        visited += hv
        if hk is not None and BETTER(hd, hk, best):
            best[0], best[1] = hd, hk
    return best[0], best[1], visited


def walk_skip_link(sc, origin, direction, t_max=np.inf):
    """The reference's stackless two-level walk (light.wgsl:400-486): `<` pruning, the first candidate met keeps a tie."""
    o, d = np.asarray(origin, np.float64), np.asarray(direction, np.float64)
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    best, key, steps = t_max, None, 0
    mn, mx, en, ex = sc.tlas
    i = 0
    while i < len(en):
        steps += 1
        if en[i] >= LEAF:
            inst = int(en[i] - LEAF)
            I = sc.instances[inst]
            lo = I["inverse"] @ np.append(o, 1.0)
            lo, ld = lo[:3] / lo[3], (I["inverse"] @ np.append(d, 0.0))[:3]
            with np.errstate(divide="ignore"):
                linv = 1.0 / ld
            bmn, bmx, ben, bex = I["tree"]
            j = 0
            while j < len(ben):
                steps += 1
                if ben[j] >= LEAF:
                    p = sc.prims[I["primitive"] + int(ben[j] - LEAF)]
                    t = triangle(lo, ld, p[0], p[1], p[2])
                    if t < best:
                        best, key = t, (inst, I["primitive"] + int(ben[j] - LEAF))
                    j = int(bex[j])
                else:
                    t = slab(bmn[j], bmx[j], lo, linv)
                    j = j + 1 if (t is not None and t < best) else int(bex[j])
            i = int(ex[i])
        else:
            t = slab(mn[i], mx[i], o, inv)
            i = i + 1 if (t is not None and t < best) else int(ex[i])
    return best, key, steps


def brute_force(sc, origin, direction, t_max=np.inf):
    """Every triangle of every instance: the closest hit under the tie rule (the reference's: Scene.key)."""
    o, d = np.asarray(origin, np.float64), np.asarray(direction, np.float64)
    best = [t_max, None]
    for inst, I in enumerate(sc.instances):
        lo = I["inverse"] @ np.append(o, 1.0)
        lo, ld = lo[:3] / lo[3], (I["inverse"] @ np.append(d, 0.0))[:3]
        n_tris = int(np.sum(I["tree"][2] >= LEAF))
        for k in range(n_tris):
            p = sc.prims[I["primitive"] + k]
            t = triangle(lo, ld, p[0], p[1], p[2])
            key = sc.key(inst, k)
            if t <= best[0] and BETTER(t, key, best):
                best[0], best[1] = t, key
    return best[0], best[1]


def walk_wide_concurrent(sc, origin, direction, rng, t_max=np.inf, steal_probability=0.5):
    """The same walk as a set of PIECES that advance one stack entry at a time in a random order - the lanes of a dry wave - with what the
    device shares between them: a piece is created by taking the bottom entry off another piece's stack (its limit: the giver's closest
    distance at that moment, one float up if it is a hit's), every piece prunes with min(own closest distance, the closest distance ANY
    piece has published) under `<=`, accepts into its OWN hit only, and the hits are merged under the tie rule when all are done."""
    o, d = np.asarray(origin, np.float64), np.asarray(direction, np.float64)
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    shared = [np.inf]   # share_best: the closest distance any piece has found

    def local(inst):
        inv_m = sc.instances[inst]["inverse"]
        lo = inv_m @ np.append(o, 1.0)
        ld = (inv_m @ np.append(d, 0.0))[:3]
        with np.errstate(divide="ignore"):
            return lo[:3] / lo[3], ld, 1.0 / ld

    pieces = [{"best": [t_max, None], "stack": [("t", None, len(sc.tlas[2]) - 1)]}]
    done = []
    while pieces:
        k = int(rng.integers(len(pieces)))
        p = pieces[k]
        if len(p["stack"]) >= 1 and rng.random() < steal_probability and len(pieces) < 64:
            entry = p["stack"].pop(0)
            b = p["best"]
            pieces.append({"best": [np.nextafter(b[0], np.inf) if b[1] is not None else b[0], None], "stack": [entry]})
        if not p["stack"]:
            done.append(pieces.pop(k))
            continue
        level, inst, link = p["stack"].pop()
        best = p["best"]
        if link >= LEAF:
            ident = link - LEAF
            if level == "t":
                p["stack"].append(("b", ident, len(sc.instances[ident]["tree"][2]) - 1))
            else:
                lo, ld, _ = local(inst)
                tri = sc.prims[sc.instances[inst]["primitive"] + ident]
                t = triangle(lo, ld, tri[0], tri[1], tri[2])
                key = sc.key(inst, ident)
                if t <= best[0] and BETTER(t, key, best):
                    best[0], best[1] = t, key
                    shared[0] = min(shared[0], t)
            continue
        if level == "t":
            rec, ro, rinv = sc.tlas_wide[link], o, inv
        else:
            lo, _, linv = local(inst)
            rec, ro, rinv = sc.instances[inst]["wide"][link], lo, linv
        bound = min(best[0], shared[0])
        hits = [(t, child) for mn, mx, child in rec for t in [slab(mn, mx, ro, rinv)] if t is not None and t <= bound]
        hits.sort(key=lambda h: -h[0])
        for _, child in hits:
            p["stack"].append((level, inst, child))
    out = [t_max, None]
    for p in done:
        b = p["best"]
        if b[1] is not None and BETTER(b[0], b[1], out):
            out[0], out[1] = b[0], b[1]
    return out[0], out[1], len(done)
