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
    for footprint in (32 << 20, 393 << 20, 1200 << 20, 4 << 30):
        for rec in (16, 32, 64):
            for waves in (1, 2, 4, 7, 8):
                loads, gbs = e.measure_gather(footprint, rec, waves, 512)
                out["sweep"].append({"footprint_MiB": footprint >> 20, "bytes_per_step": rec, "waves_per_simd": waves, "gloads_s": round(loads, 3), "gbytes_s": round(gbs, 1),
                                     "ns_per_step": round(waves * 256 * 4 / (loads / (rec // 16)) , 2) if loads else None})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
