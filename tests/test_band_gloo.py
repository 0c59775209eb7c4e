"""Multi-rank path on CPU: world_size 2 and 3 over gloo.  Each rank renders its band with the
product's BandRenderer / halo plan (hk_band_plan_for) and torch.distributed P2P; the compute behind
the C ABI is the oracle here (no GPU in this container), which is exactly what makes this a test
of the sharding + exchange logic: the union of the bands must equal the single-rank frame bit for
bit, for a static camera."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    from rendezvous import new_rendezvous

    return new_rendezvous()   # (not a port any more: a file:// rendezvous token)


def _worker(rank, world, port, case_name, out_dir, transport=None, split=None, gather=False):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from rendezvous import init_gloo

    init_gloo(rank, world, port)   # (`port`: a file:// rendezvous token, tests/rendezvous.py)
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from bevy_hikari_amd.distributed import BandRenderer
    from cases import ALL_BUFFERS, make_case
    from oracle_lib import oracle_engine, set_threads

    set_threads(2)
    case = make_case(case_name)
    s = case.settings
    e = oracle_engine()
    e.upload_noise()
    e.upload_scene(case.scene)
    w, h = case.camera.width, case.camera.height
    e.resize(w, h, s.upscale.ratio())
    if transport == "rccl":
        # no RCCL behind the oracle.  Default: every rank raises (nobody is left waiting in a rendezvous, nobody silently
        # takes a slower transport); with fallback="host" every rank agrees on the host-staged exchange.
        with pytest.raises(RuntimeError, match="RCCL halo transport did not come up on every rank"):
            BandRenderer(e, rank, world, backend_device="cpu", transport="rccl")
        r = BandRenderer(e, rank, world, backend_device="cpu", transport="rccl", fallback="host")
        assert r.transport.startswith("host (rccl unavailable"), r.transport
    else:
        r = BandRenderer(e, rank, world, backend_device="cpu", transport=transport)
    _, rh, _ = e.buffer_info(F.BUF_TONE_MAPPED)
    if split == "uneven":   # explicit boundaries (hk_set_band_bounds): thin first band, fat last one
        if world <= 3:
            fractions = np.array([0.0, 0.11, 0.37, 0.52, 1.0])[[0, 1, 2, 4] if world == 3 else [0, 2, 4]]
            r.set_bounds([int(round(f * rh)) for f in fractions])
        else:   # many thin bands: the 20-row halos span several of them
            cuts = np.random.default_rng(world).choice(np.arange(1, rh), size=world - 1, replace=False)
            r.set_bounds([0] + sorted(int(c) for c in cuts) + [rh])
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    for k, n in enumerate(case.frames):
        r.render(hk.frame_uniform(s, n), view, pview, case.lights, s, w, h, balance=(split == "balanced" and k == 0),
                 antialias=case.antialias and gather, gather=gather and n == case.frames[-1])
    if split == "balanced":   # every rank derived the split from its own full-frame primary rays: the same one
        mine = torch.tensor(r.bounds, dtype=torch.int64)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        assert all((t == mine).all() for t in everyone), everyone
        assert r.bounds == e.band_bounds() and r.bounds[0] == 0 and r.bounds[-1] == rh
    if gather:   # rank 0 holds the whole final image after the last frame
        if rank == 0:
            from bevy_hikari_amd.distributed import _final_buffer

            np.save(os.path.join(out_dir, "gathered.npy"), e.read(_final_buffer(s, case.antialias)))
        dist.barrier()
        dist.destroy_process_group()
        return
    b0, b1 = r.band(rh)
    want = [F.BUF_TONE_MAPPED, F.BUF_DENOISE_RENDER0, F.BUF_DENOISE_RENDER0 + 1, F.BUF_DENOISE_RENDER0 + 2, F.BUF_RENDER0 + 2, F.BUF_VARIANCE0 + 2]
    cur, prev = case.frames[-1] % 2, 1 - case.frames[-1] % 2
    want += [F.BUF_RESERVOIR0 + prev + 6, F.BUF_RESERVOIR0 + prev + 8, F.BUF_RESERVOIR0 + prev + 2]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), b0=b0, b1=b1, **{ALL_BUFFERS[b]: e.read(b)[b0:b1] for b in want if e.buffer_info(b)[1] == rh})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case_name,transport,split", [(2, "cornell_b2", None, None), (3, "yard_sun", None, None), (2, "cornell_b2", "rccl", None),
                                                             (3, "cornell_b2", None, "uneven"), (2, "yard_sun", None, "uneven"),
                                                             (3, "yard_sun", None, "balanced"), (2, "cornell_upscale2", None, "balanced"),
                                                             (8, "cornell_b2", None, None), (6, "yard_sun", None, "balanced"), (7, "cornell_b2", None, "uneven")])
def test_bands_equal_single_rank(tmp_path, world, case_name, transport, split):
    """split: bands of unequal height - explicit boundaries, or the cost-balanced split every rank derives on the first frame.
    transport "rccl" where RCCL cannot come up (here: the oracle has no communicator): the ranks agree BEFORE anyone enters
    the rendezvous - all of them raise, or with fallback="host" all of them stage the halos through host memory, and the frame
    is the same."""
    from cases import make_case, run_case, snapshot
    from oracle_lib import oracle_plugin

    port = _free_port()
    mp.spawn(_worker, args=(world, port, case_name, str(tmp_path), transport, split), nprocs=world, join=True)
    case = make_case(case_name)
    ref = oracle_plugin()
    run_case(ref, case)
    full = snapshot(ref)
    rows = 0
    for rank in range(world):
        d = np.load(tmp_path / f"rank{rank}.npz")
        b0, b1 = int(d["b0"]), int(d["b1"])
        rows += b1 - b0
        for key in d.files:
            if key in ("b0", "b1"):
                continue
            a = full[key]
            if a.shape[0] != full["tone_mapped"].shape[0]:
                continue
            assert (d[key].view(np.uint8) == a[b0:b1].view(np.uint8)).all(), f"rank {rank} band [{b0},{b1}) differs in {key}"
    assert rows == full["tone_mapped"].shape[0]


def _moving_cameras(n_frames, w, h):
    import bevy_hikari_amd as hk

    # vertical motion: reprojection crosses rows, i.e. the band border
    return [hk.Camera(hk.look_at_transform((0.0, 0.4 + 0.16 * n, 4.0), (0.0, 0.4 + 0.16 * n, 0.0)), w, h) for n in range(1, n_frames + 1)]


def _motion_worker(rank, world, port, history_rows, out_dir, settings_kw):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from rendezvous import init_gloo

    init_gloo(rank, world, port)   # (`port`: a file:// rendezvous token, tests/rendezvous.py)
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from bevy_hikari_amd.distributed import BandRenderer
    from oracle_lib import oracle_engine, set_threads

    set_threads(2)
    s = hk.HikariSettings(upscale=hk.Upscale.SMAA_TU_1_0, **settings_kw)
    w, h, frames = 96, 64, 8
    e = oracle_engine()
    e.upload_noise()
    e.upload_scene(hk.load_cornell())
    e.resize(w, h, 1.0)
    r = BandRenderer(e, rank, world, backend_device="cpu")
    cams = _moving_cameras(frames, w, h)
    used = []
    for n in range(1, frames + 1):
        cam, prev = cams[n - 1], cams[max(n - 2, 0)]
        r.render(hk.frame_uniform(s, n), cam.view_uniform(), cam.previous_view_uniform(prev), hk.lights_uniform(), s, w, h,
                 history_rows=history_rows)   # None: derived by the library from the two views and the scene's bounds
        used.append(e.history_rows())
    b0, b1 = r.band(h)
    out = {f"d{i}": e.read(F.BUF_DENOISE_RENDER0 + i)[b0:b1] for i in range(3)}
    for k in (6, 7, 8, 9):   # the indirect channel's temporal and spatial reservoirs: the band's own records
        out[f"r{k}"] = e.read(F.BUF_RESERVOIR0 + k).reshape(-1, 16)[b0 * w:b1 * w]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), b0=b0, b1=b1, used=np.array(used), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,settings_kw", [(2, dict(indirect_bounces=2)), (5, dict(indirect_bounces=2)),
                                               (3, dict(indirect_bounces=1, emissive_spatial_reuse=True)), (7, dict(indirect_bounces=2, emissive_spatial_reuse=True))])
def test_history_halo_for_a_moving_camera(tmp_path, world, settings_kw):
    """SURVEY 8e step 6.  With the camera moving, reprojection crosses the band borders twice: a band READS last frame's
    reservoirs in its neighbours' rows (exchange C, HK_STAGE_TEMPORAL_WITH_HISTORY) and its temporal dispatches STORE rejected
    history at reprojected slots its neighbours own - and read (light.wgsl:1092-1095,1456-1459).  The bands park those stores, hand
    each other the rows near the borders with exchange A (HK_STAGE_SPATIAL_WITH_HISTORY) and resolve them by the single-rank rule;
    the halo is the library's own bound (hk_history_rows_bound), not a number the host supplies.  The union of the bands is then
    the single-rank frame bit for bit - rendered channels and reservoirs; without a halo (history_rows = 0) it is not."""
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from oracle_lib import oracle_plugin

    s = hk.HikariSettings(upscale=hk.Upscale.SMAA_TU_1_0, **settings_kw)
    ref = oracle_plugin()
    ref.set_scene(hk.load_cornell())
    for n, cam in enumerate(_moving_cameras(8, 96, 64), start=1):
        ref.render(cam, s, frame_number=n)
    want = ref.output(s)
    want_res = {k: ref.engine.read(F.BUF_RESERVOIR0 + k).reshape(-1, 16)[:96 * 64] for k in (6, 7, 8, 9)}
    err = {}
    for rows in (0, None):
        out = tmp_path / f"rows{rows}"
        out.mkdir()
        mp.spawn(_motion_worker, args=(world, _free_port(), rows, str(out), settings_kw), nprocs=world, join=True)
        got = np.zeros_like(want)
        same_reservoirs = True
        for rank in range(world):
            d = np.load(out / f"rank{rank}.npz")
            b0, b1 = int(d["b0"]), int(d["b1"])
            for i in range(3):
                got[i, b0:b1] = d[f"d{i}"].view(np.float16).astype(np.float32)
            for k in (6, 7, 8, 9):
                same_reservoirs = same_reservoirs and bool((d[f"r{k}"] == want_res[k][b0 * 96:b1 * 96]).all())
            if rows is None:   # frame 1 reprojects onto itself (previous view = view); from frame 2 on the 0.16-unit steps need a halo
                assert d["used"][0] == 0 and (d["used"][1:] >= 4).all() and (d["used"][1:] <= 16).all(), d["used"]
        err[rows] = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        if rows is None:
            assert err[rows] == 0.0 and same_reservoirs, (err, same_reservoirs)
    assert err[0] > 1e-3, err   # (the halo is what makes the difference)


def _rebalance_worker(rank, world, port, migrate, out_dir, settings_kw):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from rendezvous import init_gloo

    init_gloo(rank, world, port)
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from bevy_hikari_amd.distributed import BandRenderer, rebalanced_band_bounds
    from oracle_lib import oracle_engine, set_threads

    set_threads(2)
    s = hk.HikariSettings(upscale=hk.Upscale.SMAA_TU_1_0, **settings_kw)
    w, h, frames = 96, 64, 8
    e = oracle_engine()
    e.upload_noise()
    e.upload_scene(hk.load_cornell())
    e.resize(w, h, 1.0)
    r = BandRenderer(e, rank, world, backend_device="cpu")
    cams = _moving_cameras(frames, w, h)
    history = []
    for n in range(1, frames + 1):
        cam, prev = cams[n - 1], cams[max(n - 2, 0)]
        r.render(hk.frame_uniform(s, n), cam.view_uniform(), cam.previous_view_uniform(prev), hk.lights_uniform(), s, w, h, time_band=True)
        assert r.band_time_ms() > 0.0
        if n in (2, 4, 5, 7):
            # "measured" times that push the boundaries down, up, and down again by a few rows (the real ones of a 96 x 64 frame on the
            # CPU oracle are noise): what is under test is the path - one all-gather, the same controller on every rank, the migration
            fake = float(1 + rank) if n in (2, 7) else float(world - rank)
            if migrate:
                r.rebalance(fake, n + 1, s, w, h, damping=0.6, max_shift=6, min_rows=4)
            else:   # the same boundaries WITHOUT moving the history rows (what HK_FRAME_BALANCE_BANDS documents as a cut)
                t = torch.zeros(world, dtype=torch.float32)
                t[rank] = fake
                dist.all_reduce(t)
                r.set_bounds(rebalanced_band_bounds(r.bounds, [float(x) for x in t], h, None, 4, 6, 0.6))
            history.append(list(r.bounds))
    b0, b1 = r.band(h)
    out = {f"d{i}": e.read(F.BUF_DENOISE_RENDER0 + i)[b0:b1] for i in range(3)}
    for k in (6, 7, 8, 9):
        out[f"r{k}"] = e.read(F.BUF_RESERVOIR0 + k).reshape(-1, 16)[b0 * w:b1 * w]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), b0=b0, b1=b1, history=np.array(history), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,settings_kw", [(2, dict(indirect_bounces=2)), (3, dict(indirect_bounces=1, emissive_spatial_reuse=True)), (5, dict(indirect_bounces=2))])
def test_moving_camera_through_the_rebalancer(tmp_path, world, settings_kw):
    """Round 6: the split follows MEASURED band times - every rank contributes one float, all evaluate
    hk_rebalanced_band_bounds, the history rows that change owner travel (hk_band_migration_schedule) - while the camera moves, i.e.
    with exchange C and the parked scatter stores in play.  The union of the bands still equals the single-rank frame bit for bit,
    rendered channels and reservoirs; moving the boundaries WITHOUT the migration does not."""
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from oracle_lib import oracle_plugin

    s = hk.HikariSettings(upscale=hk.Upscale.SMAA_TU_1_0, **settings_kw)
    ref = oracle_plugin()
    ref.set_scene(hk.load_cornell())
    for n, cam in enumerate(_moving_cameras(8, 96, 64), start=1):
        ref.render(cam, s, frame_number=n)
    want = ref.output(s)
    want_res = {k: ref.engine.read(F.BUF_RESERVOIR0 + k).reshape(-1, 16)[:96 * 64] for k in (6, 7, 8, 9)}
    err = {}
    for migrate in (True, False):
        out = tmp_path / f"migrate{int(migrate)}"
        out.mkdir()
        mp.spawn(_rebalance_worker, args=(world, _free_port(), migrate, str(out), settings_kw), nprocs=world, join=True)
        got = np.zeros_like(want)
        same_reservoirs = True
        cover = np.zeros(64, dtype=int)
        for rank in range(world):
            d = np.load(out / f"rank{rank}.npz")
            b0, b1 = int(d["b0"]), int(d["b1"])
            cover[b0:b1] += 1
            for i in range(3):
                got[i, b0:b1] = d[f"d{i}"].view(np.float16).astype(np.float32)
            for k in (6, 7, 8, 9):
                same_reservoirs = same_reservoirs and bool((d[f"r{k}"] == want_res[k][b0 * 96:b1 * 96]).all())
            hist = d["history"]
            assert (hist == np.load(out / "rank0.npz")["history"]).all()          # every rank derived the same boundaries
            assert len({tuple(b) for b in hist.tolist()}) >= 3, hist              # ... and they really moved, more than once
        assert (cover == 1).all()
        err[migrate] = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        if migrate:
            assert err[migrate] == 0.0 and same_reservoirs, (err, same_reservoirs)
    assert err[False] > 0.0, err   # (stale history in the rows that changed owner: the migration is what makes the difference)


def _aa_worker(rank, world, port, case_name, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from rendezvous import init_gloo

    init_gloo(rank, world, port)   # (`port`: a file:// rendezvous token, tests/rendezvous.py)
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from bevy_hikari_amd.distributed import BandRenderer
    from cases import make_case, random_case
    from oracle_lib import oracle_engine, set_threads

    set_threads(2)
    case = random_case(int(case_name[6:])) if case_name.startswith("random") else make_case(case_name)
    s = case.settings
    e = oracle_engine()
    e.upload_noise()
    e.upload_scene(case.scene)
    w, h = case.camera.width, case.camera.height
    e.resize(w, h, s.upscale.ratio())
    r = BandRenderer(e, rank, world, backend_device="cpu")
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    for n in case.frames:
        r.render(hk.frame_uniform(s, n), view, pview, case.lights, s, w, h, antialias=case.antialias)
    _, rh, _ = e.buffer_info(F.BUF_TONE_MAPPED)
    b0, b1 = r.band(rh)
    out = {"b0": b0, "b1": b1, "rh": rh}
    fsr = s.upscale.kind == F.UPSCALE_FSR1
    for b, name in ((F.BUF_TONE_MAPPED, "tone_mapped"), (F.BUF_UPSCALE_OUTPUT, "upscale_output"), (F.BUF_TAA_OUTPUT, "taa_output"),
                    (F.BUF_UPSCALE_SHARPENED, "upscale_sharpened")):
        _, bh, _ = e.buffer_info(b)
        if fsr and b in (F.BUF_UPSCALE_OUTPUT, F.BUF_UPSCALE_SHARPENED):   # FSR1: a band owns its share of the window rows
            y0, y1 = r.band(bh)
        else:
            scale = 2 if bh > rh else 1
            y0, y1 = min(bh, scale * b0), (bh if b1 == rh else min(bh, scale * b1))
        out[name] = e.read(b)[y0:y1]
        out[name + "_rows"] = np.array([y0, y1])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case_name", [(2, "cornell_aa_default"), (3, "yard_aa_smaa2x"), (2, "cornell_aa_fsr"), (3, "yard_aa_fsr_notaa"), (2, "random3"), (3, "random7"), (4, "random12"),
                                              (3, "random21"), (2, "random30"), (4, "random35"), (3, "random22"), (4, "random33"), (2, "random46"), (3, "random5"),
                                              (8, "cornell_aa_default"), (6, "cornell_aa_fsr"), (7, "yard_aa_smaa2x"), (8, "yard_aa_fsr_notaa")])
def test_antialias_bands_equal_single_rank(tmp_path, world, case_name):
    """All five stages on bands - for the named cases SMAA Tu4x + TAA (exchange D) or TAA + FSR1 (exchange E), for the random ones whatever the
    seeded settings say (aprons depend on them: emissive spatial reuse, denoise off, ratio != 1, 0 bounces ...): the
    union of the bands' rows equals the single-rank image bit for bit, for a static camera."""
    from cases import make_case, random_case, run_case, snapshot
    from oracle_lib import oracle_plugin

    mp.spawn(_aa_worker, args=(world, _free_port(), case_name, str(tmp_path)), nprocs=world, join=True)
    case = random_case(int(case_name[6:])) if case_name.startswith("random") else make_case(case_name)
    ref = oracle_plugin()
    run_case(ref, case)
    full = snapshot(ref)
    fsr = case.settings.upscale.kind == 0 and case.antialias
    for name in ("tone_mapped", "upscale_output", "taa_output") + (("upscale_sharpened",) if fsr else ()):
        covered = 0
        for rank in range(world):
            d = np.load(tmp_path / f"rank{rank}.npz")
            y0, y1 = (int(v) for v in d[name + "_rows"])
            covered += y1 - y0
            assert (d[name].view(np.uint8) == full[name][y0:y1].view(np.uint8)).all(), f"rank {rank} rows [{y0},{y1}) differ in {name}"
        assert covered == full[name].shape[0], name


@pytest.mark.parametrize("world,case_name,split", [(3, "cornell_b2", None), (2, "cornell_aa_default", None), (3, "cornell_aa_fsr", "uneven"), (4, "yard_aa_smaa2x", "balanced"),
                                                   (2, "yard_aa_fsr_notaa", None), (8, "cornell_b2", None), (8, "cornell_aa_default", "balanced")])
def test_rank_0_gathers_the_final_image(tmp_path, world, case_name, split):
    """SURVEY 8e step 7 over the host transport: after the last frame rank 0 collects every band's rows of the image the overlay
    presents (tone-mapped; with the anti-aliasing tail the TAA / SMAA Tu4x / sharpened FSR1 output, whose rows are cut where the
    boundaries fall at THEIR heights) - and holds the single-rank image, bit for bit."""
    from bevy_hikari_amd.distributed import _final_buffer
    from cases import make_case, run_case
    from oracle_lib import oracle_plugin

    port = _free_port()
    mp.spawn(_worker, args=(world, port, case_name, str(tmp_path), None, split, True), nprocs=world, join=True)
    case = make_case(case_name)
    ref = oracle_plugin()
    run_case(ref, case)
    want = ref.engine.read(_final_buffer(case.settings, case.antialias))
    got = np.load(tmp_path / "gathered.npy")
    assert got.shape == want.shape and (got.view(np.uint8) == want.view(np.uint8)).all()
