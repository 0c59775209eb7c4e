#!/usr/bin/env python3
"""What each rank of an N-GPU band-sharded run spends, measured on ONE GPU: for N = 1, 2, 4, 8 every band of the frame is
rendered alone (hk_set_band, no exchanges) and timed by the wall clock over K frames - with equal-row bands, with the split by
geometry pixels of hk_balance_bands, and (round 6) with the split the MEASURED-TIME controller settles on: the per-band times go
through hk_rebalanced_band_bounds, the bands are measured again, a few times over (what BandRenderer.rebalance does every M frames of
a real run, with one all-gather of N floats).  The halo bytes each rank receives per frame come from hk_band_schedule.  From that: the
PREDICTED frame time and scaling curve of the real N-GPU run, to be compared with the driver's SCALE_rNN.json.  No N > 1 hardware run
exists: these are predictions under the stated link assumptions, not results.

    python tools/band_probe.py [--configs 2 4] [--frames 24] > profiles/r06_band_probe.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload definitions)
import bevy_hikari_amd as hk  # noqa: E402
from bevy_hikari_amd import _ffi as F  # noqa: E402
from bevy_hikari_amd.distributed import band_gather_schedule, band_schedule, rebalanced_band_bounds  # noqa: E402

# assumptions of the prediction (xGMI is point-to-point: a neighbour exchange uses ONE link per direction)
LINK_GBS = 50.0      # effective one-direction rate of one xGMI link for MB-sized ncclSend/Recv (7 links x ~153 GB/s bidirectional per GPU)
EXCHANGE_US = 30.0   # fixed cost of one grouped send/recv exchange on the stream


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="+", default=[2, 4])
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--single", action="store_true", help="(internal) only the whole frame of the one config, in this process: prints {single_ms}")
    args = ap.parse_args()
    if len(args.configs) > 1 and not args.single:
        # one process per config: the band context must be its process's first and only one (see below)
        import subprocess

        merged = None
        for cfg in args.configs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--configs", str(cfg), "--frames", str(args.frames)], check=True, capture_output=True, text=True)
            d = json.loads(r.stdout)
            if merged is None:
                merged = d
            else:
                merged["configs"].update(d["configs"])
        print(json.dumps(merged, indent=1))
        return
    out = {"assumptions": {"link_gbs_one_direction": LINK_GBS, "exchange_fixed_us": EXCHANGE_US,
                           "method": "every band rendered alone on one MI355X (hk_set_band), wall clock over K frames after a full-frame warm-up; "
                                     "predicted N-GPU frame = max over bands + sum over the frame's exchanges (two halo exchanges + the gather of the tone-mapped image on rank 0) of (fixed + largest transfer "
                                     "from ONE peer / link rate): xGMI is point-to-point with a link per peer, the rows of the upper and of the lower neighbour arrive side by side (round 3 priced their sum on one link)",
                           "critical_path_round_6": "a band's frames are enqueued back to back; what one frame adds to the band's main stream is its measured time (primary rays, the three "
                                                    "temporal dispatches, the spatial pass) + exchange A, which the spatial pass waits for.  Exchange B, demodulation, the a-trous levels, tone "
                                                    "mapping and the gather run on the post / communicator streams beside the NEXT frame's light passes (round 6: the render / variance planes "
                                                    "are double-buffered by frame parity) - they bound the frame only if their own chain (exchange B + post-processing + gather) is longer than the "
                                                    "main stream's: frame = max(band + exchange A, exchange B + post-processing alone + gather)"},
           "configs": {}}
    for config in args.configs:
        scene, camera, settings, lights, description = bench.workload(hk, config, None, None, None)
        W, H = camera.width, camera.height
        sc = settings.to_c()
        view, pview = camera.view_uniform(), camera.previous_view_uniform()
        # The single-GPU frame comes from a process of its own (`--single`), the bands from a context that sees a BAND at its first frame, as
        # a rank's context does - which stream priorities and whether the primary rays get their own stream is decided there (round 6) - and
        # that is the first and only context of THIS process: a process has few hardware queues per stream priority, and a context
        # created after another one died measured twice a band's time.
        def make(first_band):
            x = hk.Engine(device=0)
            x.upload_noise()
            x.upload_scene(scene)
            x.resize(W, H, 1.0)
            if first_band:
                x.set_band(*first_band)
            return x

        K = args.frames if config == 2 else max(10, args.frames // 2)
        if args.single:
            whole = make(None)
            single_ms, kf = [], 0
            for rep_ in range(3):
                t0 = time.perf_counter()
                for _ in range(12 if rep_ == 0 else K):
                    kf += 1
                    whole.frame_render(hk.frame_uniform(settings, kf), view, pview, lights, sc)
                whole.wait()
                if rep_:
                    single_ms.append((time.perf_counter() - t0) / K * 1e3)
            print(json.dumps({"single_ms": min(single_ms)}))
            return
        import subprocess

        single_ms = [json.loads(subprocess.run([sys.executable, os.path.abspath(__file__), "--single", "--configs", str(config), "--frames", str(args.frames)],
                                               check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1])["single_ms"]]
        e = make((0, 2))
        n = 0

        def frames(count):
            nonlocal n
            for _ in range(count):
                n += 1
                e.frame_render(hk.frame_uniform(settings, n), view, pview, lights, sc)
            e.wait()

        frames(12)
        rows = {}
        _post_cache = {}

        def post_chain_ms(bands):
            """demodulation + a-trous levels (+ tone mapping) of the middle band of an N-way equal split, each dispatch timed by HIP events"""
            if bands in _post_cache:
                return _post_cache[bands]
            e.set_band(0, 1)
            e.set_band_bounds(None)
            frames(2)
            e.set_band(bands // 2, bands)
            frames(2)
            e.reset_stats()
            mask = sum(1 << p for p in (F.PASS_DEMODULATION, F.PASS_DENOISE_L0, F.PASS_DENOISE_L1, F.PASS_DENOISE_L2, F.PASS_DENOISE_L3, F.PASS_TONE_MAPPING))
            e.set_timing_mask(mask)
            frames(4)
            st = e.stats()
            e.set_timing_mask(0)
            e.set_band(0, 1)
            _post_cache[bands] = sum(st.pass_ms_total[p] for p in range(F.PASS_COUNT)) / 4.0
            return _post_cache[bands]

        def measure_bands(bands, bounds):
            out_ms = []
            for b in range(bands):
                e.set_band(0, 1)
                frames(2)                      # whole-frame state stays current between the band measurements
                e.set_band(b, bands)
                e.set_band_bounds(bounds)
                frames(2)
                t0 = time.perf_counter()
                frames(K)
                out_ms.append((time.perf_counter() - t0) / K * 1e3)
            return out_ms

        for bands, balanced in ((1, False), (2, False), (4, False), (8, False), (2, True), (4, True), (8, True), (2, "measured"), (4, "measured"), (8, "measured")):
            per_band, recv = [], []
            bounds = None
            history = None
            if balanced:   # the split by cost (hk_balance_bands), derived once from this frame's primary rays
                e.set_band(0, bands)
                n += 1
                e.frame_begin(hk.frame_uniform(settings, n), view, pview, lights)
                bounds = e.balance_bands()
            if balanced == "measured":
                # round 6: from the split by geometry pixels, the controller on MEASURED band times, eight rounds; the split
                # with the smallest slowest band is kept (a real run keeps iterating: every M frames, one all-gather of N floats)
                history = []
                best = None
                for step in range(9):
                    ms = measure_bands(bands, bounds)
                    history.append({"bounds": list(bounds), "band_ms": [round(x, 4) for x in ms], "max_over_mean": round(max(ms) / (sum(ms) / bands), 4)})
                    if best is None or max(ms) < max(best[1]):
                        best = (list(bounds), ms)
                    if step < 8:   # (damping 0.6, then 0.35 once the boundaries move by a few rows only: the times are noisy to ~1 %)
                        bounds = rebalanced_band_bounds(bounds, ms, H, None, min_rows=8, max_shift=0, damping=0.6 if step < 4 else 0.35)
                bounds, per_band = best
            elif bands == 1:   # the single-GPU frame: the context that rendered whole frames from the start (measured above)
                per_band = [min(single_ms)]
            else:
                per_band = measure_bands(bands, bounds)
            for b in range(bands):
                # bytes this band receives per frame, per exchange (static view: no history rows)
                ex = []
                for stage in (F.STAGE_SPATIAL, F.STAGE_POST_PROCESS):
                    per_peer = {}
                    if bands > 1:
                        for t in band_schedule(W, H, 1.0, b, bands, stage, n, sc, bounds):
                            if t.is_recv:
                                per_peer[t.peer] = per_peer.get(t.peer, 0) + t.bytes
                    # (total received, most from ONE peer): every peer has its own xGMI link, the neighbours' rows arrive side by side
                    ex.append((sum(per_peer.values()), max(per_peer.values(), default=0)))
                recv.append(ex)
            exch_ms = 0.0
            exch_each = [0.0, 0.0]
            gather_ms = 0.0
            gather_bytes = 0
            if bands > 1:
                for k in range(2):
                    worst = max(r[k][1] for r in recv)
                    if worst:
                        exch_each[k] = EXCHANGE_US * 1e-3 + worst / (LINK_GBS * 1e9) * 1e3
                        exch_ms += exch_each[k]
                # rank 0 collects the tone-mapped rows of the others, one link per sender in parallel
                gather_bytes = max(t.bytes for b in range(1, bands) for t in band_gather_schedule(W, H, 1.0, settings.upscale.kind, b, bands, 0, F.BUF_TONE_MAPPED, bounds))
                gather_ms = EXCHANGE_US * 1e-3 + gather_bytes / (LINK_GBS * 1e9) * 1e3
                exch_ms += gather_ms
            # round 6: exchange B + the post-processing + the gather are a chain of their own (post / communicator streams)
            post_alone = post_chain_ms(bands)
            off_path_chain = exch_each[1] + post_alone + gather_ms
            frame_r6 = max(max(per_band) + exch_each[0], off_path_chain)
            key = f"{bands}_{balanced}" if balanced == "measured" else (f"{bands}_balanced" if balanced else bands)
            rows[key] = {"bounds": bounds, "mean_band_ms": round(sum(per_band) / bands, 4), "max_over_mean": round(max(per_band) / (sum(per_band) / bands), 4),
                           "exchange_a_ms_predicted": round(exch_each[0], 4), "exchange_b_ms_predicted": round(exch_each[1], 4), "post_processing_alone_ms": round(post_alone, 4),
                           "off_critical_path_chain_ms": round(off_path_chain, 4), "frame_ms_predicted_round_6": round(frame_r6, 4),
                           **({"controller_rounds": history} if history else {}), "band_ms": [round(x, 4) for x in per_band], "max_band_ms": round(max(per_band), 4), "halo_bytes_received_per_band": [[x[0] for x in r] for r in recv], "halo_bytes_from_one_peer_per_band": [[x[1] for x in r] for r in recv], "gather_bytes_largest_band": gather_bytes,
                           "exchange_ms_predicted": round(exch_ms, 4), "frame_ms_predicted": round(max(per_band) + exch_ms, 4),
                           # hk_frame_render(HK_FRAME_GATHER) since round 4: rank 0 collects frame n's rows on the communicator's stream while
                           # frame n + 1 renders; the gather is off the critical path as long as it is shorter than a band's frame
                           "gather_ms_predicted": round(gather_ms, 4),
                           "frame_ms_predicted_gather_overlapped": round(max(max(per_band) + exch_ms - gather_ms, gather_ms), 4)}
        t1 = rows[1]["frame_ms_predicted"]
        for key in rows:
            bands = int(str(key).split("_")[0])
            rows[key]["speedup_predicted_round_6"] = round(t1 / rows[key]["frame_ms_predicted_round_6"], 3)
            rows[key]["speedup_predicted"] = round(t1 / rows[key]["frame_ms_predicted"], 3)
            rows[key]["efficiency_predicted"] = round(t1 / rows[key]["frame_ms_predicted"] / bands, 3)
            rows[key]["speedup_predicted_gather_overlapped"] = round(t1 / rows[key]["frame_ms_predicted_gather_overlapped"], 3)
        out["configs"][str(config)] = {"workload": description, "frames_per_measurement": K, "bands": {str(k): v for k, v in rows.items()}}
        e.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
