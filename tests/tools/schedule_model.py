#!/usr/bin/env python3
"""What could a different SCHEDULE of indirect_lit_ambient buy on the Cornell frame?  

The CPU oracle records, per pixel, the traversal work of every ray of the indirect pass (orc_debug_record_steps: node steps,
triangle tests, instance entries per walk).  A wave issues an instruction for all 64 lanes whatever their number, so a schedule
is priced in issued wave-instructions: a NODE step costs 35, a TRIANGLE test 60, an instance ENTRY 100 (ISA of k_indirect), the
shade sections 250-900.  In lock step (the fused kernel) an iteration costs 35 + 60 [any lane tests a triangle] + 100 [any lane
enters an instance].  Priced here, on the same rays:

  fused            one pixel per lane, 8x8 tiles (what kernels.hip k_indirect does)
  sorted_bound     the same kernel with the pixels of a 32x32 block dealt to waves by their TRUE total work (an oracle no real
                   key can beat): the ceiling of any re-binning of pixels to lanes
  megakernel       persistent waves, a lane takes the next pixel when its own is done, every iteration the wave runs the phase
                   (node / triangle / entry / one of the shade sections) with the most lanes waiting
  trace_lockstep   rays compacted into a queue, 64 at a time in lock step (a wavefront tracer without lane refill)
  trace_refill     ... with lane refill, triangle tests and entries inline (kernels_wavefront.hip before lanes parked)
  trace_parked     ... with lanes parked at hit leaves and the fullest phase run (kernels_wavefront.hip k_wf_trace)

Usage: python tests/tools/schedule_model.py [width height]   -> JSON (profiles/r02_schedule_model.json)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bevy_hikari_amd as hk
from oracle_lib import oracle_plugin

C_NODE, C_TRI, C_ENTRY = 35, 60, 100
C_P, C_A, C_B, C_C, C_F = 250, 600, 450, 250, 900   # prologue; after the closest hit; shadow set-up; after the shadow ray; temporal tail
OVH = 8                                              # phase selection per iteration of a state-machine loop
NODE, TRI, ENTRY, PH_P, PH_A, PH_B, PH_C, PH_F = range(8)
COST = [C_NODE, C_TRI, C_ENTRY, C_P, C_A, C_B, C_C, C_F]
rng = np.random.default_rng(1)


def record(w, h, bounces=2, frames=12):
    p = oracle_plugin()
    p.set_scene(hk.load_cornell())
    s = hk.HikariSettings(indirect_bounces=bounces, upscale=hk.Upscale.SMAA_TU_1_0)
    cam = hk.cornell_camera(w, h)
    rec = np.zeros((h, w, 3 * bounces, 3), dtype=np.uint16)
    fn = p.engine.api.dll.orc_debug_record_steps
    fn.argtypes, fn.restype = [C.c_void_p, C.c_void_p, C.c_uint32], C.c_int
    for n in range(1, frames + 1):
        if n == frames:
            assert fn(p.engine.ctx, rec.ctypes.data, 3 * bounces) == 0
        p.render(cam, s, frame_number=n)
    fn(p.engine.ctx, None, 0)
    return rec.astype(np.int64)


def ray_events(n, t, e):
    """One walk as a sequence of events: n node steps with t triangle tests and e entries after random ones of them."""
    if n == 0:
        return []
    ev = [NODE] * n
    pos = sorted(rng.integers(0, n, size=t + e).tolist(), reverse=True)
    kinds = [TRI] * t + [ENTRY] * e
    rng.shuffle(kinds)
    for q, k in zip(pos, kinds):
        ev.insert(q + 1, k)
    return ev


def pixel_walks(r):
    """Per bounce: (closest walk, emitter-BLAS walk, shadow walk or None)."""
    out = []
    for b in range(len(r) // 3):
        c, eb, s = r[3 * b], r[3 * b + 1], r[3 * b + 2]
        if c[0] == 0:
            break
        out.append((ray_events(*c), ray_events(*eb), ray_events(*s) if s[0] > 0 else None))
    return out


def lockstep(walks):
    """Issued instructions of one lock-step loop over the walks of a wave (each iteration: node step + the rare events any lane has)."""
    its = []
    for ev in walks:
        it = []
        for e in ev:
            if e == NODE:
                it.append(0)
            else:
                it[-1] |= 1 if e == TRI else 2
        its.append(it)
    cost = 0
    for k in range(max((len(i) for i in its), default=0)):
        fl = 0
        for i in its:
            if k < len(i):
                fl |= i[k]
        cost += C_NODE + (C_TRI if fl & 1 else 0) + (C_ENTRY if fl & 2 else 0)
    return cost


def useful(walks):
    return sum(COST[e] for ev in walks for e in ev)


def fused_cost(pixels, rec):
    """One wave of the fused kernel over `pixels` (<= 64)."""
    W = [pixel_walks(rec[y, x]) for (y, x) in pixels]
    cost, use = C_P + C_F, (C_P + C_F) * len(W)
    for b in range(max(len(w) for w in W)):
        live = [w[b] for w in W if len(w) > b]
        for k, c in ((0, C_A), (1, 0), (2, C_C)):
            walks = [l[k] for l in live if l[k] is not None]
            if k == 2 and walks:
                cost += C_B
                use += C_B * len(walks)
            cost += lockstep(walks)
            use += useful(walks)
            if c:
                cost += c
                use += c * len(live)
    return cost, use


def state_machine(items, fetch_cost, refill_min, phases):
    """A persistent wave: `items` = event lists, handed to idle lanes; every iteration the phase of `phases` with most lanes runs."""
    q = iter(items)
    lanes, pos = [None] * 64, [0] * 64
    cost = use = 0
    live = True

    def refill(i):
        nonlocal live
        try:
            lanes[i], pos[i] = next(q), 0
        except StopIteration:
            lanes[i], live = None, False

    for i in range(64):
        refill(i)
    while True:
        cnt = [0] * 8
        idle = [i for i in range(64) if lanes[i] is None]
        for i in range(64):
            if lanes[i] is not None:
                cnt[lanes[i][pos[i]]] += 1
        if sum(cnt) == 0:
            break
        if live and len(idle) >= refill_min:
            cost += fetch_cost
            for i in idle:
                if live:
                    refill(i)
            continue
        if phases == "inline":  # node step with whatever rare events the lanes at nodes run into this iteration
            ph_cost, done = C_NODE, set()
            fl = 0
            for i in range(64):
                if lanes[i] is None:
                    continue
                ev = lanes[i]
                use += COST[ev[pos[i]]]
                pos[i] += 1
                while pos[i] < len(ev) and ev[pos[i]] != NODE:
                    fl |= 1 if ev[pos[i]] == TRI else 2
                    use += COST[ev[pos[i]]]
                    pos[i] += 1
                if pos[i] == len(ev):
                    lanes[i] = None
            cost += ph_cost + (C_TRI if fl & 1 else 0) + (C_ENTRY if fl & 2 else 0)
            continue
        ph = int(np.argmax(cnt))
        cost += COST[ph] + OVH
        use += COST[ph] * cnt[ph]
        for i in range(64):
            if lanes[i] is not None and lanes[i][pos[i]] == ph:
                pos[i] += 1
                if pos[i] == len(lanes[i]):
                    lanes[i] = None
    return cost, use


def pixel_events(r):
    ev = [PH_P]
    for c, eb, s in pixel_walks(r):
        ev += c + [PH_A] + eb
        if s is not None:
            ev += [PH_B] + s
        ev.append(PH_C)
    return ev + [PH_F]


if __name__ == "__main__":
    w, h = (int(a) for a in (sys.argv[1:3] + ["960", "544"][len(sys.argv) - 1:]))
    rec = record(w, h)
    active = rec.reshape(h, w, -1).any(axis=2)
    tiles = []
    for ty in range(h // 8):
        for tx in range(w // 8):
            px = [(ty * 8 + j, tx * 8 + i) for j in range(8) for i in range(8) if active[ty * 8 + j, tx * 8 + i]]
            if px:
                tiles.append(px)
    out = {"size": [w, h], "bounces": 2, "geometry_pixels": float(active.mean()), "costs": dict(node=C_NODE, triangle=C_TRI, entry=C_ENTRY, shade=[C_P, C_A, C_B, C_C, C_F])}
    sel = [tiles[i] for i in rng.permutation(len(tiles))[:200]]
    c = u = 0
    for t in sel:
        a, b = fused_cost(t, rec)
        c, u = c + a, u + b
    out["fused"] = {"issued_per_pixel": c / sum(len(t) for t in sel), "lane_utilisation": u / (64 * c)}
    # ceiling of re-binning pixels to lanes: 32x32 blocks, pixels dealt to waves by their true total work
    blocks = [(by, bx) for by in range(0, h - 31, 32) for bx in range(0, w - 31, 32) if active[by:by + 32, bx:bx + 32].mean() > 0.9]
    cb = cs = 0
    work = lambda p: sum(rec[p[0], p[1], s, 0] * C_NODE + rec[p[0], p[1], s, 1] * C_TRI + rec[p[0], p[1], s, 2] * C_ENTRY for s in range(6))
    for by, bx in [blocks[i] for i in rng.permutation(len(blocks))[:6]]:
        px = [(by + j, bx + i) for j in range(32) for i in range(32) if active[by + j, bx + i]]
        tl = {}
        for p_ in px:
            tl.setdefault(((p_[0] - by) // 8, (p_[1] - bx) // 8), []).append(p_)
        cb += sum(fused_cost(t, rec)[0] for t in tl.values())
        srt = sorted(px, key=work)
        cs += sum(fused_cost(srt[i:i + 64], rec)[0] for i in range(0, len(srt), 64))
    out["sorted_bound"] = {"speedup_over_fused": cb / cs}
    allpx = [p_ for t in tiles for p_ in t]
    start = int(rng.integers(0, len(allpx) - 4000))
    chunk = allpx[start:start + 3000]
    c, u = state_machine([pixel_events(rec[y, x]) for (y, x) in chunk], C_P, 1, "max")
    out["megakernel"] = {"issued_per_pixel": c / len(chunk), "lane_utilisation": u / (64 * c)}
    for slot, name in ((0, "closest_bounce0"), (2, "shadow_bounce0"), (3, "closest_bounce1")):
        rays = [ray_events(*rec[y, x][slot]) for (y, x) in chunk if rec[y, x][slot][0] > 0]
        ideal = useful(rays) / 64
        ls = sum(lockstep(rays[i:i + 64]) for i in range(0, len(rays), 64))
        cr, _ = state_machine(rays, 40, 8, "inline")
        cp, _ = state_machine(rays, 40, 8, "max")
        out["trace_" + name] = {"rays": len(rays), "lockstep": ideal / ls, "refill": ideal / cr, "parked": ideal / cp}
    print(json.dumps(out, indent=1))
