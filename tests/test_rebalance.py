"""Bands of equal MEASURED time (round 6): the controller hk_rebalanced_band_bounds and the migration plan, pure
host logic of the product library (no GPU).

The controller is run against band-time models built from RECORDED measurements - the per-band times and boundaries of
profiles/r05_final_band_probe.json (every band of an 8-way split of BASELINE configs 2 / 4 rendered alone on an MI355X), whose cost per
row differs 3x between the sky and the city rows of config 4 and whose bands carry a fixed cost that does not shrink with their rows
(profiles/r06_band_anatomy_config{2,4}.json) - and must bring max / mean band time under 1.05 in a handful of steps without ever
producing an invalid split.  The migration plan must hand every row that changes owner to its new owner exactly once."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.distributed import band_migration_schedule, rebalanced_band_bounds
from conftest import ROOT


def recorded_model(config, key):
    """row -> cost density (ms per row) and the fixed cost per band of a recorded probe: band i of the probe took band_ms[i] for rows
    [bounds[i], bounds[i+1]); a third of a Cornell band's time and a fifth of a city band's does not depend on its rows (the anatomy)."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_final_band_probe.json")))["configs"][str(config)]
    row = d["bands"][key]
    rows = 1080 if config == 2 else 2160
    bounds = row["bounds"] or [rows * i // 8 for i in range(9)]
    fixed = (0.08 if config == 2 else 0.45)
    density = np.zeros(rows)
    for i, ms in enumerate(row["band_ms"]):
        density[bounds[i]:bounds[i + 1]] = max(ms - fixed, 0.02) / (bounds[i + 1] - bounds[i])
    return density, fixed, rows


def band_times(bounds, density, fixed):
    return [fixed + float(density[bounds[i]:bounds[i + 1]].sum()) for i in range(len(bounds) - 1)]


def valid(bounds, rows, min_rows):
    return bounds[0] == 0 and bounds[-1] == rows and all(b - a >= min_rows for a, b in zip(bounds, bounds[1:]))


@pytest.mark.parametrize("config,key,start", [(4, "8", "equal"), (4, "8_balanced", "recorded"), (2, "8", "equal"), (4, "4", "equal")])
def test_controller_converges_on_recorded_band_times(config, key, start):
    density, fixed, rows = recorded_model(config, key)
    n = int(key.split("_")[0])
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_final_band_probe.json")))["configs"][str(config)]["bands"][key]
    bounds = (d["bounds"] if start == "recorded" and d["bounds"] else [rows * i // n for i in range(n + 1)])
    ratios = []
    for step in range(8):
        ms = band_times(bounds, density, fixed)
        ratios.append(max(ms) / (sum(ms) / n))
        bounds = rebalanced_band_bounds(bounds, ms, rows, None, min_rows=8, max_shift=0, damping=0.6)
        assert valid(bounds, rows, 8), bounds
    ms = band_times(bounds, density, fixed)
    final = max(ms) / (sum(ms) / n)
    assert final <= 1.05, (ratios, final)
    assert final <= ratios[0] + 1e-9            # never worse than where it started
    # at the fixed point nothing moves any more (a boundary may dither by a row: the times are step functions of the rows)
    again = rebalanced_band_bounds(bounds, ms, rows, None, 8, 0, 0.6)
    assert max(abs(a - b) for a, b in zip(again, bounds)) <= 2


def test_a_prior_inside_the_bands_makes_the_first_step_better():
    """row_weight: a band that is half sky and half city tells the controller where INSIDE it the time was spent"""
    density, fixed, rows = recorded_model(4, "8")
    bounds = [rows * i // 8 for i in range(9)]
    ms = band_times(bounds, density, 0.0)
    flat = rebalanced_band_bounds(bounds, ms, rows, None, 8, 0, 1.0)
    informed = rebalanced_band_bounds(bounds, ms, rows, density + 1e-6, 8, 0, 1.0)
    spread = lambda b: (lambda t: max(t) / (sum(t) / len(t)))(band_times(b, density, 0.0))
    assert spread(informed) <= 1.02          # with the true distribution as the prior one full step lands on the optimum
    assert spread(informed) <= spread(flat) + 1e-9


def test_controller_contract():
    eq = [0, 270, 540, 810, 1080]
    assert rebalanced_band_bounds(eq, [1, 1, 1, 1], 1080) == eq                                # equal times: nothing moves
    assert rebalanced_band_bounds(None, [2.0, 2.0, 2.0, 2.0], 1080) == eq                       # None = the equal split
    assert rebalanced_band_bounds(eq, [1, 0, 1, 1], 1080) == eq                                # a band without a time: the split stays
    assert rebalanced_band_bounds(eq, [1, float("nan"), 1, 1], 1080) == eq
    moved = rebalanced_band_bounds(eq, [1, 2, 3, 4], 1080, damping=1.0)
    assert moved == [0, 473, 720, 911, 1080]                                                   # piecewise-linear inverse of the cumulative time
    half = rebalanced_band_bounds(eq, [1, 2, 3, 4], 1080, damping=0.5)
    assert all(abs((a + b) / 2 - h) <= 0.5 for a, b, h in zip(eq, moved, half))                # damping = the fraction of the way
    assert max(abs(a - b) for a, b in zip(rebalanced_band_bounds(eq, [1, 2, 3, 4], 1080, max_shift=5, damping=1.0), eq)) == 5
    squeezed = rebalanced_band_bounds([0, 10, 20, 30, 40], [100, 1, 1, 1], 40, min_rows=8, damping=1.0)
    assert valid(squeezed, 40, 8)
    api = F.api()
    out = (C.c_uint32 * 5)()
    b = (C.c_uint32 * 5)(0, 300, 200, 810, 1080)                                                # not increasing
    ms = (C.c_float * 4)(1, 1, 1, 1)
    with pytest.raises(F.HikariError):
        api.call("rebalanced_band_bounds", b, ms, 4, 1080, None, 8, 0, 0.5, out)
    with pytest.raises(F.HikariError):
        api.call("rebalanced_band_bounds", (C.c_uint32 * 5)(*eq), ms, 4, 1080, None, 8, 0, 0.0, out)   # damping must be > 0


@pytest.mark.parametrize("seed", range(12))
def test_migration_hands_every_row_that_changes_owner_to_its_new_owner_exactly_once(seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2, 9))
    w, h = int(rng.integers(8, 200)), int(rng.integers(n * 4, 400))
    ratio = float(rng.choice([1.0, 1.5, 2.0]))
    s = hk.HikariSettings(indirect_bounces=int(rng.integers(0, 3)), emissive_spatial_reuse=bool(rng.random() < 0.5), indirect_spatial_reuse=bool(rng.random() < 0.7),
                          upscale=hk.Upscale.SmaaTu4x(ratio)).to_c()
    rw, rh = F.u32(), F.u32()
    F.api().call("scaled_size", w, h, ratio, C.byref(rw), C.byref(rh))
    rw, rh = rw.value, rh.value
    if rh < n:
        pytest.skip("fewer rows than bands")

    def split():
        if rng.random() < 0.2:
            return None
        cuts = sorted(rng.choice(np.arange(1, rh), size=n - 1, replace=False).tolist())
        return [0] + [int(c) for c in cuts] + [rh]

    def rows_of(bounds, i):
        if bounds is None:
            base, rem = divmod(rh, n)
            b0 = i * base + min(i, rem)
            return b0, b0 + base + (1 if i < rem else 0)
        return bounds[i], bounds[i + 1]

    old, new = split(), split()
    frame = int(rng.integers(1, 100))
    per_rank = [band_migration_schedule(w, h, ratio, old, new, r, n, frame, s) for r in range(n)]
    cur = frame % 2
    expect = {F.BUF_RESERVOIR0 + cur + t for t in (0, 2, 6)}
    if s.emissive_spatial_reuse:
        expect.add(F.BUF_RESERVOIR0 + cur + 4)
    if s.indirect_spatial_reuse:
        expect.add(F.BUF_RESERVOIR0 + cur + 8)
    row_bytes = rw * 64
    for r in range(n):
        n0, n1 = rows_of(new, r)
        o0, o1 = rows_of(old, r)
        for buf in expect:
            got = np.zeros(rh, dtype=int)
            for t in per_rank[r]:
                if t.is_recv and t.buffer == buf:
                    assert t.offset % row_bytes == 0 and t.bytes % row_bytes == 0
                    a, b = t.offset // row_bytes, (t.offset + t.bytes) // row_bytes
                    p0, p1 = rows_of(old, t.peer)
                    assert p0 <= a and b <= p1 and t.peer != r            # from the band that owned them
                    got[a:b] += 1
            want = np.zeros(rh, dtype=int)
            want[n0:n1] = 1
            want[max(n0, o0):min(n1, o1)] = 0                             # what it owned before stays where it is
            assert (got == want).all(), (r, buf)
        assert {t.buffer for t in per_rank[r]} <= expect
    # every receive has its send, in the same global order on both sides (what a transport that pairs by issue order needs)
    for a in range(n):
        for b in range(n):
            if a == b:
                continue
            sends = [(t.buffer, t.offset, t.bytes) for t in per_rank[a] if not t.is_recv and t.peer == b]
            recvs = [(t.buffer, t.offset, t.bytes) for t in per_rank[b] if t.is_recv and t.peer == a]
            assert sends == recvs
    if old == new or (old is None and new is None):
        assert all(len(p) == 0 for p in per_rank)
