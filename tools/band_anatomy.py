#!/usr/bin/env python3
"""Where one band of an 8-way split spends its frame: the band rendered alone on one MI355X, three ways.

  1. wall clock per frame in the product mode (three streams) and on one stream;
  2. HIP events around every pass, each dispatch alone on the GPU (HK_CTX_SINGLE_STREAM + a full timing mask): the band's
     kernel sum, and beside it the same passes of the WHOLE frame scaled by the band's share of the rows - what the band would cost if
     time fell with work.  The excess is the band's fixed cost per stage: launch ramp, the tail a stage ends with, apron rows;
  3. the same band under `rocprofv3 --kernel-trace` (this script re-runs itself as the traced child): every dispatch of the last
     frames in order - kernels, memsets the library enqueues, their gaps.  (--kernel-trace serialises the streams: the trace's
     frame is the one-stream frame, which is what makes its gaps attributable.)

    python tools/band_anatomy.py --config 2 [--band 3] [--bands 8] [--balanced] > profiles/r06_band_anatomy_config2.json
"""
import argparse
import glob
import json
import os
import shutil
import signal
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def setup(config, flags):
    import bench
    import bevy_hikari_amd as hk

    scene, camera, settings, lights, description = bench.workload(hk, config, None, None, None)
    e = hk.Engine(device=0, flags=flags)
    e.upload_noise()
    e.upload_scene(scene)
    e.resize(camera.width, camera.height, 1.0)
    return hk, e, camera, settings, lights, description


class Runner:
    def __init__(self, config, flags):
        self.hk, self.e, self.camera, self.settings, self.lights, self.description = setup(config, flags)
        self.sc = self.settings.to_c()
        self.view, self.pview = self.camera.view_uniform(), self.camera.previous_view_uniform()
        self.n = 0

    def frames(self, count):
        for _ in range(count):
            self.n += 1
            self.e.frame_render(self.hk.frame_uniform(self.settings, self.n), self.view, self.pview, self.lights, self.sc)
        self.e.wait()

    def balanced_bounds(self, bands):
        self.e.set_band(0, bands)
        self.n += 1
        self.e.frame_begin(self.hk.frame_uniform(self.settings, self.n), self.view, self.pview, self.lights)
        b = self.e.balance_bands()
        self.e.set_band(0, 1)
        return b

    def to_band(self, band, bands, bounds):
        self.e.set_band(0, 1)
        self.frames(2)                       # whole-frame state current
        self.e.set_band(band, bands)
        self.e.set_band_bounds(bounds)
        self.frames(2)

    def wall(self, k):
        t0 = time.perf_counter()
        self.frames(k)
        return (time.perf_counter() - t0) / k * 1e3

    def passes(self, k):
        from bevy_hikari_amd import _ffi as F

        self.e.reset_stats()
        self.e.set_timing_mask((1 << F.PASS_COUNT) - 1)
        self.frames(k)
        st = self.e.stats()
        self.e.set_timing_mask(0)
        return {F.PASS_NAMES[i]: round(st.pass_ms_total[i] / k, 5) for i in range(F.PASS_COUNT) if st.pass_launches[i]}


def child(args):
    """the traced run: the band alone, one stream is what the tracer makes of it anyway"""
    r = Runner(args.config, 0)
    r.frames(8)
    bounds = json.loads(args.bounds) if args.bounds else None
    r.to_band(args.band, args.bands, bounds)
    r.frames(args.frames)


def trace(args, bounds):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--child", "--config", str(args.config), "--band", str(args.band), "--bands", str(args.bands), "--frames", "6"]
    if bounds:
        cmd += ["--bounds", json.dumps(bounds)]
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        p = subprocess.Popen([rocprof, "--kernel-trace", "-d", d, "--"] + cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            p.wait(timeout=300)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.wait()
            return None
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if not dbs:
            return None
        db = sqlite3.connect(dbs[0])
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        start = "start" if "start" in cols else "start_timestamp"
        end = "end" if "end" in cols else "end_timestamp"
        rows = db.execute(f"select name, {start}, {end}, grid_x, workgroup_x from kernels order by {start}").fetchall()
        db.close()
    firsts = [i for i, r in enumerate(rows) if "k_prepass" in r[0]]
    if len(firsts) < 4:
        return None
    # the last three complete frames of the band
    out = []
    for f, nxt in zip(firsts[-4:-1], firsts[-3:]):
        t0 = rows[f][1]
        seq, busy, prev_end = [], 0, None
        for name, s, e, gx, wx in rows[f:nxt]:
            short = name.replace("void hkd::", "").replace("hkd::", "").split("(")[0][:60]
            seq.append({"kernel": short, "start_us": round((s - t0) / 1e3, 1), "us": round((e - s) / 1e3, 1), "gap_before_us": round((s - prev_end) / 1e3, 1) if prev_end else 0.0,
                        "workgroups": int(gx // max(1, wx))})
            busy += e - s
            prev_end = e
        out.append({"frame_us_to_next_frames_first_kernel": round((rows[nxt][1] - t0) / 1e3, 1), "kernel_sum_us": round(busy / 1e3, 1), "dispatches": len(seq),
                    "gap_sum_us": round((rows[nxt][1] - t0 - busy) / 1e3, 1), "timeline": seq})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--band", type=int, default=3)
    ap.add_argument("--bands", type=int, default=8)
    ap.add_argument("--balanced", action="store_true", help="the split by cost (hk_balance_bands) instead of equal rows")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--bounds", default=None)
    ap.add_argument("--no-trace", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    from bevy_hikari_amd import _ffi as F

    K = args.frames or (48 if args.config in (2, 5) else 8)
    prod = Runner(args.config, 0)
    prod.frames(12)
    full_wall = prod.wall(K)
    bounds = prod.balanced_bounds(args.bands) if args.balanced else None
    H = prod.camera.height
    if bounds:
        b0, b1 = bounds[args.band], bounds[args.band + 1]
    else:
        base, rem = divmod(H, args.bands)
        b0 = args.band * base + min(args.band, rem)
        b1 = b0 + base + (1 if args.band < rem else 0)
    share = (b1 - b0) / H
    prod.to_band(args.band, args.bands, bounds)
    band_wall = prod.wall(K)
    del prod
    single = Runner(args.config, F.CTX_SINGLE_STREAM)
    single.frames(12)
    full_single_wall = single.wall(K)
    full_passes = single.passes(max(4, K // 4))
    single.to_band(args.band, args.bands, bounds)
    band_single_wall = single.wall(K)
    band_passes = single.passes(max(4, K // 4))
    description = single.description
    settings = single.settings
    del single
    ap_sp = 21 if settings.indirect_spatial_reuse else (11 if settings.emissive_spatial_reuse else 0)
    ap_dn = 16 if settings.denoise else 0
    clamp = lambda v: min(max(v, 0), H)
    rows = {"band": [b0, b1], "prepass": [clamp(b0 - ap_sp - ap_dn), clamp(b1 + ap_sp + ap_dn)], "demodulation": [clamp(b0 - 15), clamp(b1 + 15)], "denoise_l0": [clamp(b0 - 7), clamp(b1 + 7)],
            "denoise_l1": [clamp(b0 - 3), clamp(b1 + 3)], "denoise_l2": [clamp(b0 - 1), clamp(b1 + 1)]}
    per_pass = {}
    for name, ms in band_passes.items():
        full = full_passes.get(name, 0.0)
        r = rows.get(name, rows["band"])
        ideal_band = full * share
        ideal_rows = full * (r[1] - r[0]) / H
        per_pass[name] = {"band_ms": ms, "full_frame_ms": full, "full_frame_x_row_share_ms": round(ideal_band, 5), "apron_rows_cost_ms": round(ideal_rows - ideal_band, 5),
                          "fixed_cost_ms": round(ms - ideal_rows, 5), "rows_dispatched": r[1] - r[0]}
    ksum = sum(band_passes.values())
    out = {"config": args.config, "workload": description, "bands": args.bands, "band": args.band, "bounds": bounds, "band_rows": [b0, b1], "row_share": round(share, 5),
           "frames_per_measurement": K,
           "wall_ms": {"full_frame_product": round(full_wall, 4), "full_frame_one_stream": round(full_single_wall, 4), "band_product": round(band_wall, 4), "band_one_stream": round(band_single_wall, 4),
                       "full_frame_x_row_share": round(full_wall * share, 4)},
           "band_one_stream": {"kernel_sum_ms": round(ksum, 4), "gaps_and_host_ms": round(band_single_wall - ksum, 4),
                               "of_the_kernel_sum": {"work_at_the_full_frames_rate_ms": round(sum(v["full_frame_x_row_share_ms"] for v in per_pass.values()), 4),
                                                     "apron_rows_ms": round(sum(v["apron_rows_cost_ms"] for v in per_pass.values()), 4),
                                                     "fixed_cost_per_stage_ms (ramp + tail: the part that does not fall with the rows)": round(sum(v["fixed_cost_ms"] for v in per_pass.values()), 4)}},
           "per_pass": per_pass,
           "method": "HIP events around every pass of a HK_CTX_SINGLE_STREAM context (each dispatch alone on the GPU), wall clock over K frames; full_frame_x_row_share = the whole frame's pass "
                     "x the band's share of the rows; apron = the rows a pass dispatches beyond the band (primary rays +-37, demodulation +-15, a-trous +-7/3/1) at the full frame's rate; "
                     "fixed = what is left"}
    if not args.no_trace:
        out["kernel_trace_last_frames"] = trace(args, bounds)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
