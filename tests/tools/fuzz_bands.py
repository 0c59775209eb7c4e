#!/usr/bin/env python3
"""Wide sweep of tests/test_multi_gpu.py::test_multi_engine_bands_of_unequal_height: seeds [first, last) of cases.random_case
(random settings x scene x odd image size x anti-aliasing tail), each rendered by hk_multi_* with a random number of bands
(2..6, all on device 0), alternately random boundaries (one-row bands included) and the split by cost of the first frame
(HK_FRAME_BALANCE_BANDS), against a single context: every buffer the frame's consumers read, bit for bit.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from bevy_hikari_amd.distributed import MultiEngine
from cases import random_case, run_case

first, last = int(sys.argv[1]), int(sys.argv[2])
bad, kinds, t0 = {}, {"uneven": 0, "balanced": 0, "bands": {}}, time.time()
for seed in range(first, last):
    case = random_case(seed)
    s = case.settings
    rng = np.random.default_rng(31 * seed + 5)
    w, h = case.camera.width, case.camera.height
    probe = hk.Engine(device=0)
    probe.resize(w, h, s.upscale.ratio())
    _, rh, _ = probe.buffer_info(F.BUF_TONE_MAPPED)
    probe.close()
    bands = int(rng.integers(2, 7))
    if rh < bands * 8:
        bands = 2
    mode = "balanced" if seed % 2 else "uneven"
    if mode == "balanced" and rh < bands * 8:
        mode = "uneven"
    kinds[mode] += 1
    kinds["bands"][bands] = kinds["bands"].get(bands, 0) + 1
    m = MultiEngine([0] * bands)
    m.upload_noise(); m.upload_scene(case.scene)
    m.resize(w, h, s.upscale.ratio())
    if mode == "uneven":
        cuts = rng.choice(np.arange(1, rh), size=bands - 1, replace=False)
        m.set_band_bounds([0] + sorted(int(c) for c in cuts) + [rh])
    view, pview = case.camera.view_uniform(), case.camera.previous_view_uniform()
    for k, n in enumerate(case.frames):
        flags = (F.FRAME_ANTIALIAS if case.antialias else 0) | (F.FRAME_BALANCE_BANDS if mode == "balanced" and k == 0 else 0)
        m.frame_render(hk.frame_uniform(s, n), view, pview, case.lights, s.to_c(), flags)
    m.wait()
    ref = hk.HikariPlugin(device=0)
    run_case(ref, case)
    e = ref.engine
    prev = 1 - case.frames[-1] % 2
    want = [F.BUF_TONE_MAPPED, F.BUF_RENDER0, F.BUF_RENDER0 + 1, F.BUF_RENDER0 + 2, F.BUF_VARIANCE0 + 2, F.BUF_RESERVOIR0 + prev + 6, F.BUF_RESERVOIR0 + prev + 8,
            F.BUF_POSITION, F.BUF_ALBEDO]
    if s.denoise:
        want += [F.BUF_DENOISE_RENDER0, F.BUF_DENOISE_RENDER0 + 1] + ([F.BUF_DENOISE_RENDER0 + 2] if s.indirect_bounces else [])
    if case.antialias:
        want += [F.BUF_TAA_OUTPUT] if s.taa == hk.Taa.Jasmine else []
        want += [F.BUF_UPSCALE_OUTPUT] + ([F.BUF_UPSCALE_SHARPENED] if s.upscale.kind == F.UPSCALE_FSR1 else [])
    for b in want:
        a, r = m.read(b), e.read(b)
        if F.BUF_RESERVOIR0 <= b < F.BUF_RESERVOIR0 + 10:
            rw, rh2, _ = e.buffer_info(F.BUF_TONE_MAPPED)
            a, r = a.reshape(-1, 16)[:rw * rh2], r.reshape(-1, 16)[:rw * rh2]
        if a.shape != r.shape or not (a.view(np.uint8) == r.view(np.uint8)).all():
            bad.setdefault(seed, {"bands": bands, "mode": mode, "bounds": m.contexts[0].band_bounds(), "buffers": []})["buffers"].append(int(b))
    m.close()
print(json.dumps({"seeds": [first, last], "cases": kinds, "mismatching_seeds": len(bad), "first": dict(list(bad.items())[:4]), "seconds": round(time.time() - t0, 1)}))
