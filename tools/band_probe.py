#!/usr/bin/env python3
"""What each rank of an N-GPU band-sharded run spends, measured on ONE GPU: for N = 1, 2, 4, 8 every band of the frame is
rendered alone (hk_set_band, no exchanges; once with equal-row bands, once with the split by cost of hk_balance_bands) and timed
by the wall clock over K frames; the halo bytes each rank receives per
frame come from hk_band_schedule.  From that: the PREDICTED frame time and scaling curve of the real N-GPU run
(max over bands + the exchanges priced with the stated link assumptions), to be compared with the driver's SCALE_rNN.json.

    python tools/band_probe.py [--configs 2 4] [--frames 24] > profiles/r03_band_probe.json
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
from bevy_hikari_amd.distributed import band_gather_schedule, band_schedule  # noqa: E402

# assumptions of the prediction (xGMI is point-to-point: a neighbour exchange uses ONE link per direction)
LINK_GBS = 50.0      # effective one-direction rate of one xGMI link for MB-sized ncclSend/Recv (7 links x ~153 GB/s bidirectional per GPU)
EXCHANGE_US = 30.0   # fixed cost of one grouped send/recv exchange on the stream


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="+", default=[2, 4])
    ap.add_argument("--frames", type=int, default=24)
    args = ap.parse_args()
    out = {"assumptions": {"link_gbs_one_direction": LINK_GBS, "exchange_fixed_us": EXCHANGE_US,
                           "method": "every band rendered alone on one MI355X (hk_set_band), wall clock over K frames after a full-frame warm-up; "
                                     "predicted N-GPU frame = max over bands + sum over the frame's exchanges (two halo exchanges + the gather of the tone-mapped image on rank 0) of (fixed + largest transfer "
                                     "from ONE peer / link rate): xGMI is point-to-point with a link per peer, the rows of the upper and of the lower neighbour arrive side by side (round 3 priced their sum on one link)"},
           "configs": {}}
    for config in args.configs:
        scene, camera, settings, lights, description = bench.workload(hk, config, None, None, None)
        W, H = camera.width, camera.height
        sc = settings.to_c()
        view, pview = camera.view_uniform(), camera.previous_view_uniform()
        e = hk.Engine(device=0)
        e.upload_noise()
        e.upload_scene(scene)
        e.resize(W, H, 1.0)
        K = args.frames if config == 2 else max(6, args.frames // 4)
        n = 0

        def frames(count):
            nonlocal n
            for _ in range(count):
                n += 1
                e.frame_render(hk.frame_uniform(settings, n), view, pview, lights, sc)
            e.wait()

        frames(12)
        rows = {}
        for bands, balanced in ((1, False), (2, False), (4, False), (8, False), (2, True), (4, True), (8, True)):
            per_band, recv = [], []
            bounds = None
            if balanced:   # the split by cost (hk_balance_bands), derived once from this frame's primary rays
                e.set_band(0, bands)
                n += 1
                e.frame_begin(hk.frame_uniform(settings, n), view, pview, lights)
                bounds = e.balance_bands()
            for b in range(bands):
                e.set_band(0, 1)
                frames(2)                      # whole-frame state stays current between the band measurements
                e.set_band(b, bands)
                e.set_band_bounds(bounds)
                frames(2)
                t0 = time.perf_counter()
                frames(K)
                per_band.append((time.perf_counter() - t0) / K * 1e3)
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
            gather_ms = 0.0
            gather_bytes = 0
            if bands > 1:
                for k in range(2):
                    worst = max(r[k][1] for r in recv)
                    if worst:
                        exch_ms += EXCHANGE_US * 1e-3 + worst / (LINK_GBS * 1e9) * 1e3
                # rank 0 collects the tone-mapped rows of the others, one link per sender in parallel
                gather_bytes = max(t.bytes for b in range(1, bands) for t in band_gather_schedule(W, H, 1.0, settings.upscale.kind, b, bands, 0, F.BUF_TONE_MAPPED, bounds))
                gather_ms = EXCHANGE_US * 1e-3 + gather_bytes / (LINK_GBS * 1e9) * 1e3
                exch_ms += gather_ms
            rows[f"{bands}_balanced" if balanced else bands] = {"bounds": bounds, "band_ms": [round(x, 4) for x in per_band], "max_band_ms": round(max(per_band), 4), "halo_bytes_received_per_band": [[x[0] for x in r] for r in recv], "halo_bytes_from_one_peer_per_band": [[x[1] for x in r] for r in recv], "gather_bytes_largest_band": gather_bytes,
                           "exchange_ms_predicted": round(exch_ms, 4), "frame_ms_predicted": round(max(per_band) + exch_ms, 4),
                           # hk_frame_render(HK_FRAME_GATHER) since round 4: rank 0 collects frame n's rows on the communicator's stream while
                           # frame n + 1 renders; the gather is off the critical path as long as it is shorter than a band's frame
                           "gather_ms_predicted": round(gather_ms, 4),
                           "frame_ms_predicted_gather_overlapped": round(max(max(per_band) + exch_ms - gather_ms, gather_ms), 4)}
        t1 = rows[1]["frame_ms_predicted"]
        for key in rows:
            bands = int(str(key).split("_")[0])
            rows[key]["speedup_predicted"] = round(t1 / rows[key]["frame_ms_predicted"], 3)
            rows[key]["efficiency_predicted"] = round(t1 / rows[key]["frame_ms_predicted"] / bands, 3)
            rows[key]["speedup_predicted_gather_overlapped"] = round(t1 / rows[key]["frame_ms_predicted_gather_overlapped"], 3)
        out["configs"][str(config)] = {"workload": description, "frames_per_measurement": K, "bands": {str(k): v for k, v in rows.items()}}
        e.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
