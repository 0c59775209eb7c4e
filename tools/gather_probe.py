#!/usr/bin/env python3
"""Sweep of hk_measure_gather (include/hikari_hip_debug.h): dependent, divergent gathers - what a BVH walk of a scene beyond the LDS
copy asks of the memory system, with everything else taken away.  VERDICT r03 next 2a: footprints of 32 MB (one ordering of config
3's trees), 393 MB (config 4's trees as 16-B nodes) and 1.2 GB (config 4's eight orderings of 32-B nodes), 16 / 32 / 64 bytes per
step, 1..8 waves per SIMD.  Prints one JSON object (profiles/r04_gather_probe.json).

    python tools/gather_probe.py            # the sweep
    python tools/gather_probe.py --calibrate  # three launches of known byte counts, for `rocprofv3 --pmc FETCH_SIZE` (tools/pmc_calibrate.sh)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_hikari_amd as hk  # noqa: E402


def main():
    e = hk.Engine(device=0)
    if "--calibrate" in sys.argv:
        # footprint 2 GiB (8x the Infinity Cache): practically every record is a miss to HBM; bytes a launch must fetch =
        # CUs x 4 SIMDs x waves x 64 lanes x steps x record bytes (k_gather_chase<1|2|4>, warm-up launch + timed launch: 2 dispatches each)
        out = {}
        for rec in (16, 32, 64):
            loads, gbs = e.measure_gather(2 << 30, rec, 4, 256)
            out[str(rec)] = {"requests_per_dispatch": 256 * 4 * 4 * 64 * 256, "bytes_per_dispatch": 256 * 4 * 4 * 64 * 256 * rec, "gloads_s": loads, "gbytes_s": gbs}
        print(json.dumps(out))
        return
    out = {"what": "hk_measure_gather: every lane chases its own chain of dependent loads through a random permutation cycle (64 unrelated addresses per wave-level "
                   "load, no reuse); rates in 1e9 wave-level load instructions / s and GB/s of loaded bytes", "steps": 512, "sweep": []}
    # what limits the rate?  (a) the level that serves the request: 16 KiB sits in every CU's L1, 2 MiB in every XCD's L2, 32 MiB in the
    # Infinity Cache; (b) the number of CUs asking: 1, 8, 32, 64, 128 workgroups of one wave per SIMD
    out["by_level_and_width"] = []
    for footprint in (16 << 10, 256 << 10, 2 << 20, 32 << 20, 1 << 30):
        for wgs in (1, 8, 32, 64, 128, 256, 1024):
            loads, gbs = e.measure_gather(footprint, 32, 1, 512, wgs)
            out["by_level_and_width"].append({"footprint_KiB": footprint >> 10, "workgroups": wgs, "g_lane_steps_s": round(loads / 2 * 64, 2),
                                              "ns_per_step": round(wgs * 4 / (loads / 2), 1) if loads else None,
                                              "lane_steps_per_us_per_workgroup": round(loads / 2 * 64 * 1e3 / wgs, 1)})
    for footprint in (32 << 20, 393 << 20, 1200 << 20, 4 << 30):
        for rec in (16, 32, 64):
            for waves in (1, 2, 4, 7, 8):
                loads, gbs = e.measure_gather(footprint, rec, waves, 512)
                out["sweep"].append({"footprint_MiB": footprint >> 20, "bytes_per_step": rec, "waves_per_simd": waves, "gloads_s": round(loads, 3), "gbytes_s": round(gbs, 1),
                                     "ns_per_step": round(waves * 256 * 4 / (loads / (rec // 16)) , 2) if loads else None})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
