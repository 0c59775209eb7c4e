"""Property tests of the halo plan (hk_band_plan_for, host logic of the band-sharded frame): for any window size,
upscale ratio, band count and settings, every op names rows the peer OWNS and this band does not, ops of one buffer
never overlap, together with the own band they form one contiguous row range, and that range is the band grown by a
footprint that does not depend on which band asks (except where the image border cuts it)."""
import numpy as np
from hypothesis import given, settings, strategies as st

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.distributed import halo_plan

S, U = hk.HikariSettings, hk.Upscale

sizes = st.tuples(st.integers(64, 4096), st.integers(64, 2400))
ratios = st.sampled_from([1.0, 1.25, 1.5, 2.0])
flags = st.tuples(st.booleans(), st.booleans(), st.booleans(), st.booleans(), st.booleans(), st.integers(0, 3))


def make_settings(ratio, fl):
    fsr, taa, emissive, indirect, denoise, bounces = fl
    return S(upscale=U.Fsr1(ratio, 0.2) if fsr else U.SmaaTu4x(ratio), taa=hk.Taa.Jasmine if taa else hk.Taa.NONE,
             emissive_spatial_reuse=emissive, indirect_spatial_reuse=indirect, denoise=denoise, indirect_bounces=bounces)


def band(rows, i, n, bounds=None):
    if bounds is not None:   # explicit boundaries (hk_band_plan_bounds): in scaled render rows
        assert rows == bounds[-1]
        return bounds[i], bounds[i + 1]
    base, rem = divmod(rows, n)
    b0 = i * base + min(i, rem)
    return b0, b0 + base + (1 if i < rem else 0)


def random_bounds(seed, rows, world):
    """None (the equal split) for seed 0, else world + 1 strictly increasing boundaries of [0, rows] (bands of >= 1 row)."""
    if seed == 0 or rows <= world:
        return None
    cuts = np.random.default_rng(seed).choice(np.arange(1, rows), size=world - 1, replace=False)
    return [0] + sorted(int(c) for c in cuts) + [rows]


def buffer_rows(width, height, ratio, s, buf):
    """(rows of the buffer, rows per scaled render row) as the library sizes it for these settings"""
    rw, rh = F.u32(), F.u32()
    F.api().call("scaled_size", width, height, ratio, rw, rh)
    if buf in (F.BUF_TAA_OUTPUT, F.BUF_PREVIOUS_TAA_OUTPUT) and s.upscale.kind == F.UPSCALE_SMAA_TU4X:
        return int(np.ceil(np.float32(height) * (np.float32(1.0) / np.float32(max(1.0, min(2.0, ratio)))) * np.float32(2.0))), 2
    return rh.value, 1


@settings(max_examples=150, deadline=None)
@given(sizes, ratios, st.integers(2, 8), flags, st.integers(1, 64), st.integers(0, 6),
       st.sampled_from([F.STAGE_TEMPORAL, F.STAGE_SPATIAL, F.STAGE_POST_PROCESS, F.STAGE_ANTIALIAS, F.STAGE_UPSCALE]), st.integers(0, 5))
def test_plans_name_foreign_rows_once_and_contiguously(size, ratio, world, fl, frame, history, stage, split):
    width, height = size
    s = make_settings(ratio, fl)
    sc = s.to_c()
    bounds = random_bounds(split, buffer_rows(width, height, ratio, s, F.BUF_TONE_MAPPED)[0], world)
    stage_arg = stage | ((history << 8) if stage in (F.STAGE_TEMPORAL, F.STAGE_ANTIALIAS) else 0)
    per_band = []
    for i in range(world):
        ops = halo_plan(width, height, ratio, i, world, stage_arg, frame, sc, bounds)
        per_band.append(ops)
        by_buffer = {}
        for o in ops:
            assert o.peer != i and 0 <= o.peer < world and o.row_begin < o.row_end
            rows, scale = buffer_rows(width, height, ratio, s, o.buffer)
            rh = buffer_rows(width, height, ratio, s, F.BUF_TONE_MAPPED)[0]
            p0, p1 = band(rh, o.peer, world, bounds)
            own0, own1 = band(rh, i, world, bounds)
            lo, hi = scale * p0, (rows if p1 == rh else min(rows, scale * p1))
            assert lo <= o.row_begin and o.row_end <= hi, "rows the peer does not own"
            assert o.row_end <= scale * own0 or o.row_begin >= min(rows, scale * own1), "rows this band owns itself"
            assert o.row_bytes > 0 and o.row_bytes % 4 == 0
            by_buffer.setdefault(o.buffer, []).append((o.row_begin, o.row_end, scale * own0, min(rows, scale * own1)))
        for buf, spans in by_buffer.items():
            spans.sort()
            for (a0, a1, _, _), (b0, b1, _, _) in zip(spans, spans[1:]):
                assert a1 <= b0, "overlapping ops"
            own0, own1 = spans[0][2], spans[0][3]
            if stage != F.STAGE_UPSCALE:          # (exchange E feeds the WINDOW rows of the band, which need not touch its render rows)
                covered = sorted([(a, b) for a, b, _, _ in spans] + [(own0, own1)])
                for (a0, a1), (b0, b1) in zip(covered, covered[1:]):
                    assert a1 == b0, f"gap between the halo and the band in buffer {buf}"
    # symmetry: what the interior bands fetch below equals what they fetch above (same footprint on both sides)
    for i in range(1, world - 1):
        for buf in {o.buffer for o in per_band[i]}:
            rows, scale = buffer_rows(width, height, ratio, s, buf)
            rh = buffer_rows(width, height, ratio, s, F.BUF_TONE_MAPPED)[0]
            own0, own1 = band(rh, i, world, bounds)
            below = sum(o.row_end - o.row_begin for o in per_band[i] if o.buffer == buf and o.row_end <= scale * own0)
            above = sum(o.row_end - o.row_begin for o in per_band[i] if o.buffer == buf and o.row_begin >= scale * own1)
            if stage != F.STAGE_UPSCALE and scale * own0 >= below + 8 and rows - scale * own1 >= above + 8 and below and above:   # (narrow neighbours: the halo spans several bands, same total)
                assert below == above, (buf, below, above)


@settings(max_examples=60, deadline=None)
@given(sizes, ratios, st.integers(1, 8))
def test_band_rows_partition_the_image(size, ratio, world):
    rw, rh = F.u32(), F.u32()
    F.api().call("scaled_size", size[0], size[1], ratio, rw, rh)
    prev = 0
    for i in range(world):
        b0, b1 = F.u32(), F.u32()
        F.api().call("band_rows", rh.value, i, world, b0, b1)
        assert b0.value == prev and b1.value > b0.value and (b0.value, b1.value) == band(rh.value, i, world)
        prev = b1.value
    assert prev == rh.value


@settings(max_examples=80, deadline=None)
@given(sizes, ratios, st.integers(2, 8), flags, st.integers(1, 64), st.integers(0, 6),
       st.sampled_from([F.STAGE_TEMPORAL, F.STAGE_SPATIAL, F.STAGE_POST_PROCESS, F.STAGE_ANTIALIAS, F.STAGE_UPSCALE]), st.integers(0, 5))
def test_schedules_pair_up_in_issue_order(size, ratio, world, fl, frame, history, stage, split):
    """hk_band_schedule is what the transports execute (RCCL inside the library, peer copies in hk_multi, gloo in the tests).
    RCCL pairs the sends and receives of two ranks BY ISSUE ORDER, so for every ordered pair (a -> b) the sequence of sends a
    issues to b must be, element for element (buffer, offset, bytes), the sequence of receives b issues from a; and the
    receives of a rank are exactly its halo plan."""
    from bevy_hikari_amd.distributed import band_schedule

    width, height = size
    s = make_settings(ratio, fl)
    sc = s.to_c()
    bounds = random_bounds(split, buffer_rows(width, height, ratio, s, F.BUF_TONE_MAPPED)[0], world)
    stage_arg = stage | ((history << 8) if stage in (F.STAGE_TEMPORAL, F.STAGE_ANTIALIAS) else 0)
    sched = [band_schedule(width, height, ratio, r, world, stage_arg, frame, sc, bounds) for r in range(world)]
    for a in range(world):
        for b in range(world):
            if a == b:
                continue
            sends = [(t.buffer, t.offset, t.bytes) for t in sched[a] if not t.is_recv and t.peer == b]
            recvs = [(t.buffer, t.offset, t.bytes) for t in sched[b] if t.is_recv and t.peer == a]
            assert sends == recvs, (a, b)
    for r in range(world):
        plan = sorted((o.buffer, o.peer, o.row_begin * o.row_bytes, (o.row_end - o.row_begin) * o.row_bytes) for o in halo_plan(width, height, ratio, r, world, stage_arg, frame, sc, bounds))
        assert plan == sorted((t.buffer, t.peer, t.offset, t.bytes) for t in sched[r] if t.is_recv)
        assert all(t.peer != r and t.bytes > 0 for t in sched[r])


@settings(max_examples=100, deadline=None)
@given(st.integers(16, 4096), st.integers(64, 2400), st.integers(1, 8), st.integers(1, 32), st.integers(0, 2 ** 31))
def test_balanced_bounds_are_valid_and_balance_the_cost(width, rows, world, min_rows, seed):
    """hk_balanced_band_bounds: strictly increasing boundaries from 0 to the render rows, every band at least min_rows tall,
    and no band's cost exceeds the mean by more than the two heaviest rows + what min_rows forces."""
    from bevy_hikari_amd.distributed import balanced_band_bounds

    if min_rows * world > rows:
        min_rows = 1
    if world > rows:
        return
    rng = np.random.default_rng(seed)
    cost_rows = int(rng.integers(rows // 2 + 1, 2 * rows))
    # a frame like config 4: empty sky on top, dense geometry below, noise in between
    cost = np.where(np.arange(cost_rows) < cost_rows * rng.uniform(0, 0.7), 0, rng.integers(0, width + 1, cost_rows)).astype(np.uint32)
    bg = [0.0, 1.0 / 6.0, 0.01][seed % 3]
    b = balanced_band_bounds(cost, width, rows, world, min_rows, bg)
    assert b[0] == 0 and b[-1] == rows and len(b) == world + 1
    assert all(b1 - b0 >= min_rows for b0, b1 in zip(b, b[1:]))
    per_row = np.array([cost[y * cost_rows // rows] + width * (bg or 1.0 / 16.0) for y in range(rows)])
    band_cost = [per_row[b0:b1].sum() for b0, b1 in zip(b, b[1:])]
    slack = 2 * per_row.max() + min_rows * per_row.max() * world
    assert max(band_cost) <= per_row.sum() / world + slack
    # and it is what every rank derives from the same counts (deterministic)
    assert b == balanced_band_bounds(cost.copy(), width, rows, world, min_rows, bg)


def test_bad_bounds_are_refused():
    import pytest

    sc = S().to_c()
    for bad in ([0, 10, 10, 64], [1, 10, 20, 64], [0, 10, 20, 63], [0, 30, 20, 64]):
        with pytest.raises(F.HikariError, match="band bounds"):
            halo_plan(64, 64, 1.0, 0, 3, F.STAGE_SPATIAL, 1, sc, bad)
    assert halo_plan(64, 64, 1.0, 0, 3, F.STAGE_SPATIAL, 1, sc, [0, 10, 20, 64]) is not None


@settings(max_examples=120, deadline=None)
@given(sizes, ratios, st.integers(1, 8), st.integers(0, 7), st.integers(0, 5),
       st.sampled_from([F.BUF_TONE_MAPPED, F.BUF_DENOISE_RENDER0, F.BUF_TAA_OUTPUT, F.BUF_UPSCALE_OUTPUT, F.BUF_UPSCALE_SHARPENED, F.BUF_POSITION, F.BUF_RESERVOIR0 + 3]),
       st.sampled_from([F.UPSCALE_FSR1, F.UPSCALE_SMAA_TU4X]))
def test_gather_schedules_pair_up_and_partition_the_buffer(size, ratio, world, root, split, buf, kind):
    """hk_band_gather_schedule: band r's only transfer is the send the root's r-th receive expects (same offset, same
    bytes), and the root's own rows plus what it receives tile the buffer's rows exactly once, whatever the buffer's kind."""
    from bevy_hikari_amd.distributed import band_gather_schedule

    width, height = size
    root %= world
    rw, rh = F.u32(), F.u32()
    F.api().call("scaled_size", width, height, ratio, rw, rh)
    bounds = random_bounds(split, rh.value, world)
    sched = [band_gather_schedule(width, height, ratio, kind, r, world, root, buf, bounds) for r in range(world)]
    recvs = [(t.peer, t.offset, t.bytes) for t in sched[root]]
    assert all(t.is_recv and t.peer != root for t in sched[root]) and [p for p, _, _ in recvs] == sorted(p for p, _, _ in recvs)
    for r in range(world):
        if r == root:
            continue
        assert len(sched[r]) <= 1 and all((not t.is_recv) and t.peer == root for t in sched[r])
        mine = [(t.offset, t.bytes) for t in sched[r]]
        assert mine == [(o, b) for p, o, b in recvs if p == r], (r, mine, recvs)
    # the rows the root does not receive are its own band: the spans tile [0, buffer bytes) with at most one gap, the root's
    spans = sorted((o, o + b) for _, o, b in recvs)
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0, "overlapping receives"
    gaps = [(a1, b0) for (a0, a1), (b0, b1) in zip([(0, 0)] + spans, spans + [(None, None)]) if b0 is not None and b0 > a1]
    assert len(gaps) <= 1, (gaps, spans)
