#!/usr/bin/env python3
"""Run the HIP path and the CPU oracle side by side and print per-buffer mismatch statistics.
Development aid (uses the oracle, so it lives in tools/, never in the product)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import bevy_hikari_amd as hk
from bevy_hikari_amd import _ffi as F
from oracle_lib import oracle_plugin

NAMES = {F.BUF_POSITION: "position", F.BUF_NORMAL: "normal", F.BUF_DEPTH_GRADIENT: "depth_gradient", F.BUF_INSTANCE_MATERIAL: "instance_material",
         F.BUF_VELOCITY_UV: "velocity_uv", F.BUF_ALBEDO: "albedo", F.BUF_DENOISE_INTERNAL_VARIANCE: "internal_variance", F.BUF_TONE_MAPPED: "tone_mapped"}
for i in range(3):
    NAMES[F.BUF_VARIANCE0 + i] = f"variance{i}"
    NAMES[F.BUF_RENDER0 + i] = f"render{i}"
    NAMES[F.BUF_DENOISE_RENDER0 + i] = f"denoise_render{i}"
for i in range(10):
    NAMES[F.BUF_RESERVOIR0 + i] = f"reservoir{i}"
for i in range(4):
    NAMES[F.BUF_DENOISE_INTERNAL0 + i] = f"internal{i}"


def compare(gpu, cpu, verbose=True):
    bad = {}
    for b, name in sorted(NAMES.items()):
        a, c = gpu.engine.read(b), cpu.engine.read(b)
        ne = (a.view(np.uint8).reshape(a.shape[0], a.shape[1], -1) != c.view(np.uint8).reshape(c.shape[0], c.shape[1], -1)).any(axis=2)
        n = int(ne.sum())
        if n:
            bad[name] = n
            if verbose:
                ys, xs = np.nonzero(ne)
                print(f"   {name}: {n} px differ, first at (x={xs[0]}, y={ys[0]}): gpu={a[ys[0], xs[0]]} cpu={c[ys[0], xs[0]]}")
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--bounces", type=int, default=2)
    ap.add_argument("--ratio", type=float, default=1.0)
    ap.add_argument("--emissive-spatial", action="store_true")
    args = ap.parse_args()
    settings = hk.HikariSettings(indirect_bounces=args.bounces, upscale=hk.Upscale.SmaaTu4x(args.ratio), emissive_spatial_reuse=args.emissive_spatial)
    scene = hk.load_cornell()
    cam = hk.cornell_camera(args.size, args.size)
    gpu, cpu = hk.HikariPlugin(device=0, flags=F.CTX_COUNT_RAYS), oracle_plugin()
    for p in (gpu, cpu):
        p.set_scene(scene)
    for n in range(1, args.frames + 1):
        t0 = time.time(); gpu.render(cam, settings, frame_number=n); gpu.engine.wait(); t1 = time.time()
        cpu.render(cam, settings, frame_number=n); t2 = time.time()
        bad = compare(gpu, cpu)
        a, b = gpu.output(settings), cpu.output(settings)
        rel = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
        print(f"frame {n}: gpu {1e3*(t1-t0):.1f} ms, cpu {1e3*(t2-t1):.0f} ms, rel L2 {rel:.3e}, mismatching buffers: {bad if bad else 'none'}")
    sg, sc = gpu.engine.stats(), cpu.engine.stats()
    print("rays gpu", sg.rays_primary, sg.rays_tlas, sg.rays_blas, " cpu", sc.rays_primary, sc.rays_tlas, sc.rays_blas)
