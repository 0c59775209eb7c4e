"""A line-by-line CPU port of the wide walk's STATE MACHINE as the trace stage runs it in a dry wave (csrc/hk_wide.hpp wide_begin /
wide_node / wide_triangle / wide_enter; kernels_wavefront.hip k_wf_trace_wide: hand-over, merge, published distance, every phase per
turn): the same fields (cur, sp, base, mark, blas_base, in_blas, intersected, cur_instance, hit, limit, root, helpers), the same
stack discipline (LEAVE marker, bottom entries handed over with base++ / tombstones), the same phases.  TEST INFRASTRUCTURE: where
tests/wide_model.py checks the algorithm, this checks the bookkeeping - against brute force, on random schedules."""
import numpy as np

from wide_model import LEAF, triangle

NONE, LEAVE, U32_MAX = 0xFFFFFFFF, 0xFFFFFFFE, 0xFFFFFFFF
IDLE, NODE, TRI, ENTRY, WAIT, HELPED = 0, 1, 2, 3, 4, 5
STACK = 124


def tie_goes_to(sc, instance, primitive, best_instance, best_primitive):
    """hk_wide.hpp wide_tie_goes_to: the reference's rule - the candidate its stackless walk meets first, by the leaves' ranks."""
    if instance != best_instance:
        return sc.tlas_rank[instance] < sc.tlas_rank[best_instance]
    I = sc.instances[instance]
    return I["rank"][primitive - I["primitive"]] < I["rank"][best_primitive - I["primitive"]]


def slab_t(mn, mx, o, inv):
    with np.errstate(invalid="ignore", over="ignore"):
        t1, t2 = (mn - o) * inv, (mx - o) * inv
    lo, hi = np.fmin(t1, t2), np.fmax(t1, t2)
    t_min, t_max = lo.max(), hi.min()
    return float(t_min) if (t_max >= t_min and t_max >= 0.0) else np.inf


class Lane:
    def __init__(self, index):
        self.index = index
        self.phase = IDLE
        self.stack = [NONE] * STACK
        self.lost = 0

    # ---- hk_wide.hpp
    def put(self, at, e):
        if at < STACK:
            self.stack[at] = e
        elif e != NONE:
            self.lost += 1

    def get(self, at):
        return self.stack[at] if at < STACK else NONE

    def push(self, e):
        self.put(self.sp, e)
        self.sp += 1

    def pop(self):
        self.sp -= 1
        return self.get(self.sp)

    def begin(self, sc, origin, direction, max_distance, early, exclude):
        self.origin, self.direction = np.asarray(origin, np.float64), np.asarray(direction, np.float64)
        with np.errstate(divide="ignore"):
            self.inv_direction = 1.0 / self.direction
        self.early, self.exclude = early, exclude
        self.hit = [max_distance, U32_MAX, U32_MAX]   # distance, instance, primitive
        self.limit = np.inf
        self.cur = len(sc.tlas[2]) - 1
        self.sp = self.base = self.mark = self.blas_base = 0
        self.mesh = None
        self.prim_base = self.cur_instance = 0
        self.in_blas = self.intersected = False
        self.co, self.cinv, self.ld = self.origin, self.inv_direction, self.direction

    def node(self, sc):
        """wide_node: returns (phase, pending)."""
        if self.cur == NONE:
            if self.sp == self.base:
                return IDLE, 0
            e = self.pop()
            if e == LEAVE:
                if self.intersected:
                    self.hit[1] = self.cur_instance
                    if self.hit[0] < self.early:
                        return IDLE, 0
                self.in_blas = False
                self.co, self.cinv = self.origin, self.inv_direction
                return NODE, 0
            if e == NONE:
                return NODE, 0
            if e >= LEAF:
                pending = e - LEAF
                if self.in_blas:
                    return TRI, pending
                return (ENTRY, pending) if pending != self.exclude else (NODE, 0)
            self.cur = e
        rec = self.mesh["wide"][self.cur] if self.in_blas else sc.tlas_wide[self.cur]
        bound = min(self.hit[0], self.limit)
        t, link = [np.inf] * 4, [NONE] * 4
        for c, (mn, mx, child) in enumerate(rec):
            tb = slab_t(mn, mx, self.co, self.cinv)
            if tb <= bound and tb != np.inf:
                t[c], link[c] = tb, child
        for a, b in ((0, 1), (2, 3), (0, 2), (1, 3), (1, 2)):   # the 5-comparator network
            if t[b] < t[a]:
                t[a], t[b] = t[b], t[a]
                link[a], link[b] = link[b], link[a]
        for c in (3, 2, 1):
            if link[c] != NONE:
                self.push(link[c])
        self.cur = NONE
        if link[0] == NONE:
            return NODE, 0
        if link[0] >= LEAF:
            pending = link[0] - LEAF
            if self.in_blas:
                return TRI, pending
            return (ENTRY, pending) if pending != self.exclude else (NODE, 0)
        self.cur = link[0]
        return NODE, 0

    def triangle(self, sc, pending):
        prim = self.prim_base + pending
        p = sc.prims[prim]
        d = triangle(self.co, self.ld, p[0], p[1], p[2])
        closer = d < self.hit[0]
        if d == self.hit[0] and self.hit[2] != U32_MAX:
            best_instance = self.cur_instance if self.intersected else self.hit[1]
            closer = tie_goes_to(sc, self.cur_instance, prim, best_instance, self.hit[2])
        if closer:
            self.hit[0], self.hit[2] = d, prim
            self.intersected = True
            if d < self.early:
                self.hit[1] = self.cur_instance
                return IDLE
        return NODE

    def enter(self, sc, instance):
        I = sc.instances[instance]
        lo = I["inverse"] @ np.append(self.origin, 1.0)
        self.co = lo[:3] / lo[3]
        self.ld = (I["inverse"] @ np.append(self.direction, 0.0))[:3]
        with np.errstate(divide="ignore"):
            self.cinv = 1.0 / self.ld
        self.mark = self.sp
        self.push(LEAVE)
        self.blas_base = self.sp
        self.mesh = I
        self.cur = len(I["tree"][2]) - 1
        self.prim_base = I["primitive"]
        self.cur_instance = instance
        self.in_blas = True
        self.intersected = False


class Wave:
    """k_wf_trace_wide's loop for one DRY wave: `rays` = [(origin, direction, max_distance, early, exclude)] claimed by the first
    lanes, the rest idle; every turn: merge, hand-over, then node / triangle / entry phases for every parked lane."""

    def __init__(self, sc, n_lanes, rays, rng, share_min=1, share_steps=0, give_probability=1.0):
        self.sc, self.rng = sc, rng
        self.lanes = [Lane(i) for i in range(n_lanes)]
        self.help = [0] * n_lanes
        self.best = [np.inf] * n_lanes
        self.results = {}
        self.stats = {"instance_tree_entries": 0, "mesh_tree_entries": 0, "from_helpers": 0, "tombstones_skipped": 0}
        self.share_min, self.share_steps, self.give_probability = share_min, share_steps, give_probability
        for i, r in enumerate(rays):
            l = self.lanes[i]
            l.begin(sc, *r)
            l.root, l.steps, l.tag, l.phase, l.pending = i, 0, i, NODE, 0

    def piece_done(self, l):
        if l.root != l.index:
            l.phase = HELPED
        elif self.help[l.index] != 0:
            l.phase = WAIT
        else:
            self.results[l.tag] = tuple(l.hit)
            l.phase = IDLE

    def turn(self):
        lanes = self.lanes
        # merge
        for h in lanes:
            if h.phase != HELPED:
                continue
            r = lanes[h.root]
            self.help[r.index] -= 1
            if h.hit[1] != U32_MAX:
                mine_inst = r.cur_instance if r.intersected else r.hit[1]
                closer = h.hit[0] < r.hit[0]
                if h.hit[0] == r.hit[0] and r.hit[2] != U32_MAX:
                    closer = tie_goes_to(self.sc, h.hit[1], h.hit[2], mine_inst, r.hit[2])
                if closer:
                    r.hit = list(h.hit)
                    r.intersected = False
            h.phase = IDLE
        for l in lanes:
            if l.phase == WAIT and self.help[l.index] == 0:
                self.results[l.tag] = tuple(l.hit)
                l.phase = IDLE
        idle = [l for l in lanes if l.phase == IDLE]
        if len(idle) == len(lanes):
            return False
        # hand-over
        if len(idle) >= self.share_min:
            givers = []
            for l in lanes:
                if l.phase not in (NODE, TRI, ENTRY) or l.steps < self.share_steps or self.rng.random() > self.give_probability:
                    continue
                tlas_top = l.mark if l.in_blas else l.sp
                give_at, give_blas = None, False
                if l.base < tlas_top:
                    give_at = l.base
                elif l.in_blas and l.blas_base < l.sp:
                    give_at, give_blas = l.blas_base, True
                if give_at is None:
                    continue
                link = l.get(give_at)
                if link in (NONE, LEAVE):
                    self.stats["tombstones_skipped"] += 1
                    if give_blas:
                        l.blas_base += 1
                    else:
                        l.base += 1
                    continue
                givers.append((l, give_at, give_blas, link))
            for (g, give_at, give_blas, link), t in zip(givers, idle):
                ctx = dict(link=link, tag=g.tag, root=g.root, bound=g.hit[0], is_hit=g.hit[2] != U32_MAX, origin=g.origin, direction=g.direction, early=g.early,
                           exclude=g.exclude, in_blas=give_blas, mesh=g.mesh, prim_base=g.prim_base, cur_instance=g.cur_instance,
                           co=g.co if give_blas else g.origin, ld=g.ld if give_blas else g.direction)
                self.help[g.root] += 1
                self.stats["mesh_tree_entries" if give_blas else "instance_tree_entries"] += 1
                self.stats["from_helpers"] += g.root != g.index
                if give_blas:
                    g.put(give_at, NONE)
                    g.blas_base += 1
                else:
                    g.base += 1
                t.tag, t.root = ctx["tag"], ctx["root"]
                t.origin, t.direction = ctx["origin"], ctx["direction"]
                with np.errstate(divide="ignore"):
                    t.inv_direction = 1.0 / t.direction
                t.early, t.exclude = ctx["early"], ctx["exclude"]
                t.hit = [np.nextafter(ctx["bound"], np.inf) if ctx["is_hit"] else ctx["bound"], U32_MAX, U32_MAX]
                t.limit = np.inf
                t.in_blas, t.mesh, t.prim_base, t.cur_instance = ctx["in_blas"], ctx["mesh"], ctx["prim_base"], ctx["cur_instance"]
                t.co, t.ld = ctx["co"], ctx["ld"]
                with np.errstate(divide="ignore"):
                    t.cinv = (1.0 / t.ld) if t.in_blas else t.inv_direction
                t.intersected = False
                t.cur = NONE
                t.sp = t.base = t.mark = t.blas_base = 0
                if t.in_blas:
                    t.push(LEAVE)
                    t.blas_base = 1
                t.push(ctx["link"])
                t.steps, t.phase, t.pending = 0, NODE, 0
        # every phase, every parked lane (a dry wave)
        for l in lanes:
            if l.phase == NODE:
                l.limit = self.best[l.root]
                l.steps += 1
                l.phase, l.pending = l.node(self.sc)
                if l.phase == IDLE:
                    self.piece_done(l)
        for l in lanes:
            if l.phase == TRI:
                before = l.hit[0]
                l.phase = l.triangle(self.sc, l.pending)
                if l.hit[0] < before:
                    self.best[l.root] = min(self.best[l.root], l.hit[0])
                if l.phase == IDLE:
                    self.piece_done(l)
        for l in lanes:
            if l.phase == ENTRY:
                l.enter(self.sc, l.pending)
                l.phase = NODE
        return True

    def run(self, max_turns=100000):
        n = 0
        while self.turn():
            n += 1
            assert n < max_turns, "the wave does not end"
        return self.results, n
