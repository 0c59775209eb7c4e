#!/usr/bin/env python3
"""CPU sweep of the band-sharded frame over the host transport (gloo, the oracle as the compute): seeds [first, last) of
cases.random_case, each with a random number of ranks (2..6), alternately the equal split, random boundaries and the split by
cost; all five stages (the anti-aliasing tail included where the case has one) and the gather on rank 0.  Checks, per seed: the
union of the bands == the single-rank frame for the tone-mapped image and the AA outputs, and rank 0's gathered final image ==
the single-rank one.  Prints one JSON line.   python tests/tools/fuzz_bands_gloo.py 0 40"""
import json
import os
import socket
import sys
import tempfile
import time

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _worker(rank, world, port, seed, mode, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from rendezvous import init_gloo

    init_gloo(rank, world, port)   # (`port`: a file:// rendezvous token, tests/rendezvous.py)
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from bevy_hikari_amd.distributed import BandRenderer, _final_buffer
    from cases import random_case
    from oracle_lib import oracle_engine, set_threads

    set_threads(1)
    case = random_case(seed)
    s = case.settings
    e = oracle_engine()
    e.upload_noise(); e.upload_scene(case.scene)
    w, h = case.camera.width, case.camera.height
    e.resize(w, h, s.upscale.ratio())
    r = BandRenderer(e, rank, world, backend_device="cpu")
    _, rh, _ = e.buffer_info(F.BUF_TONE_MAPPED)
    if mode == "uneven":
        cuts = np.random.default_rng(seed).choice(np.arange(1, rh), size=world - 1, replace=False)
        r.set_bounds([0] + sorted(int(c) for c in cuts) + [rh])
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    for k, n in enumerate(case.frames):
        r.render(hk.frame_uniform(s, n), view, pview, case.lights, s, w, h, antialias=case.antialias, balance=(mode == "balanced" and k == 0),
                 gather=(n == case.frames[-1]))
    final = _final_buffer(s, case.antialias)
    out = {}
    if rank == 0:
        out["gathered"] = e.read(final)
    b0, b1 = r.band(rh)
    out["tone_rows"] = np.array([b0, b1])
    out["tone"] = e.read(F.BUF_TONE_MAPPED)[b0:b1]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import bevy_hikari_amd as hk
    from bevy_hikari_amd import _ffi as F
    from bevy_hikari_amd.distributed import _final_buffer
    from cases import random_case, run_case
    from oracle_lib import oracle_plugin

    bad, kinds, t0 = {}, {}, time.time()
    for seed in range(first, last):
        case = random_case(seed)
        rng = np.random.default_rng(977 * seed + 3)
        probe = oracle_plugin().engine
        probe.resize(case.camera.width, case.camera.height, case.settings.upscale.ratio())
        _, rh, _ = probe.buffer_info(F.BUF_TONE_MAPPED)
        world = int(min(rng.integers(2, 7), rh))
        mode = ("equal", "uneven", "balanced")[seed % 3]
        kinds[f"{mode}x{world}"] = kinds.get(f"{mode}x{world}", 0) + 1
        from rendezvous import new_rendezvous

        port = new_rendezvous()
        with tempfile.TemporaryDirectory() as d:
            try:
                mp.spawn(_worker, args=(world, port, seed, mode, d), nprocs=world, join=True)
            except Exception as exc:   # a rank raised: record and go on
                bad[seed] = {"world": world, "mode": mode, "error": str(exc)[-600:]}
                continue
            ref = oracle_plugin()
            run_case(ref, case)
            want_final = ref.engine.read(_final_buffer(case.settings, case.antialias))
            want_tone = ref.engine.read(F.BUF_TONE_MAPPED)
            got = np.load(os.path.join(d, "rank0.npz"))["gathered"]
            problems = []
            if got.shape != want_final.shape or not (got.view(np.uint8) == want_final.view(np.uint8)).all():
                problems.append("gathered final image")
            for rank in range(world):
                z = np.load(os.path.join(d, f"rank{rank}.npz"))
                b0, b1 = (int(v) for v in z["tone_rows"])
                if rank != 0 and not (z["tone"].view(np.uint8) == want_tone[b0:b1].view(np.uint8)).all():
                    problems.append(f"tone-mapped rows of rank {rank}")
            if problems:
                bad[seed] = {"world": world, "mode": mode, "problems": problems}
    print(json.dumps({"seeds": [first, last], "cases": kinds, "mismatching_seeds": len(bad), "first": dict(list(bad.items())[:4]), "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
