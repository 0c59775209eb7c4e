#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle.

These fixtures are the ORACLE's outputs: they pin the oracle against regressions and let the GPU tests compare against committed
data; they are not outputs of the reference.  The fixtures that ARE produced from the reference - by executing its WGSL - are
tests/golden/wgsl_*.npz (tests/tools/wgsl_pin.py --write)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from cases import CASE_NAMES, make_case, run_case, snapshot
from oracle_lib import oracle_plugin

out_dir = os.path.join(ROOT, "tests", "golden")
os.makedirs(out_dir, exist_ok=True)
only = sys.argv[1:]   # optional: regenerate just these cases (npz bytes are not reproducible, so do not rewrite the others)
for name in CASE_NAMES:
    if only and name not in only:
        continue
    case = make_case(name)
    p = oracle_plugin()
    run_case(p, case)
    snap = snapshot(p)
    data = {"sha256_" + k: np.frombuffer(hashlib.sha256(v.tobytes()).digest(), dtype=np.uint8) for k, v in snap.items()}
    for k in ("tone_mapped", "denoise_render0", "denoise_render1", "denoise_render2", "render2", "variance2"):
        data[k] = snap[k]
    if case.antialias:
        for k in ("upscale_output", "taa_output", "upscale_sharpened"):
            data[k] = snap[k]
    st = p.engine.stats()
    data["rays"] = np.array([st.rays_primary, st.rays_tlas, st.rays_blas], dtype=np.uint64)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **data)
    print(name, "rays", data["rays"], "tone_mapped max", float(snap["tone_mapped"].view(np.float16).astype(np.float32)[..., :3].max()))
