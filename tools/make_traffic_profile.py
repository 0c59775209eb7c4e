#!/usr/bin/env python3
"""profiles/<tag>_indirect_hbm_traffic.json from the PMC summaries of tools/gpu_round.sh (<tag>_pmc_{sq,fetch,write}.txt): the HBM bytes
per launch of the dominant kernel (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md HBM section), its VALU wave-instructions
per launch and its lane utilisation - what bench.py reads for roofline.traffic / roofline.valu_issue.
    python tools/make_traffic_profile.py gpurun_out/r03_final profiles/r03_indirect_hbm_traffic.json"""
import json
import re
import sys


def blocks(path):
    out = {}
    for b in re.split(r"\n(?=\S)", open(path).read()):
        lines = b.strip().splitlines()
        if not lines:
            continue
        vals = {}
        for l in lines[1:]:
            p = l.split()
            if len(p) >= 4 and p[1] == "avg":
                vals[p[0]] = (float(p[2]), int(p[3].split("=")[1]))
        out[lines[0]] = vals
    return out


def pick(d, needle):
    for k, v in d.items():
        if needle in k:
            return k, v
    raise SystemExit(f"no kernel matching {needle!r}")


def main():
    prefix, dst = sys.argv[1], sys.argv[2]
    needle = sys.argv[3] if len(sys.argv) > 3 else "k_indirect<true, false, 2>"
    sq, fetch, write = blocks(prefix + "_pmc_sq.txt"), blocks(prefix + "_pmc_fetch.txt"), blocks(prefix + "_pmc_write.txt")
    name, s = pick(sq, needle)
    _, f = pick(fetch, needle)
    _, w = pick(write, needle)
    fetch_kb, write_kb = f["FETCH_SIZE"][0], w["WRITE_SIZE"][0]
    hbm = int(2 * fetch_kb * 1024 + write_kb * 1024)
    algo = 188 * 1920 * 1080
    out = {
        "kernel": name.split("(")[0].replace("void hkd::", "") + " (indirect_lit_ambient, MULTIPLE_BOUNCES, LDS-staged scene, one-level walk, fused schedule)",
        "workload": "Cornell 1920x1080, 2 bounces (bench.py --steps 6 --warmup 4 --blocks 1)",
        "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes; averages over {f['FETCH_SIZE'][1]} dispatches ({prefix}_pmc_*.txt, committed under profiles/)",
        "FETCH_SIZE_KB_raw": round(fetch_kb, 1), "WRITE_SIZE_KB": round(write_kb, 1),
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B -> read bytes = 2 x FETCH_SIZE; WRITE_SIZE as reported",
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": algo, "ratio_to_algorithmic": round(hbm / algo, 3),
        "limiter": {
            "what": "VALU issue at the rate four waves per SIMD sustain (hk_measure_valu), not HBM",
            "valu_wave_instructions": round(s["SQ_INSTS_VALU"][0], 1),
            "lane_utilisation": round(s["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * s["SQ_ACTIVE_INST_VALU"][0]), 3),
            "waves": s["SQ_WAVES"][0],
            "source": "SQ_INSTS_VALU; SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)",
        },
    }
    # the second kernel of the frame (bench.py roofline.second_kernel): k_spatial_reuse<false, false>, 240 B/px algorithmic
    try:
        n2, s2 = pick(sq, "k_spatial_reuse<false, false>")
        _, f2 = pick(fetch, "k_spatial_reuse<false, false>")
        _, w2 = pick(write, "k_spatial_reuse<false, false>")
        hbm2, algo2 = int(2 * f2["FETCH_SIZE"][0] * 1024 + w2["WRITE_SIZE"][0] * 1024), 240 * 1920 * 1080
        out["second_kernel"] = {"kernel": "k_spatial_reuse<false, false> (spatial_reuse, light.wgsl:1503-1684; plain form)", "FETCH_SIZE_KB_raw": round(f2["FETCH_SIZE"][0], 1),
                                "WRITE_SIZE_KB": round(w2["WRITE_SIZE"][0], 1), "hbm_bytes_per_launch": hbm2, "algorithmic_bytes_per_launch": algo2,
                                "ratio_to_algorithmic": round(hbm2 / algo2, 3),
                                # profiles/r04_fetch_calibration.json: FETCH_SIZE tallies 64 B per memory-side read request; a coalesced stream issues 128-B
                                # requests (x 2 is right), a gather of 64-B records 64-B requests (x 1 is right).  This kernel does both: the truth lies between.
                                "hbm_bytes_per_launch_lower_bound": int(f2["FETCH_SIZE"][0] * 1024 + w2["WRITE_SIZE"][0] * 1024),
                                "ratio_to_algorithmic_lower_bound": round((f2["FETCH_SIZE"][0] * 1024 + w2["WRITE_SIZE"][0] * 1024) / algo2, 3),
                                "valu_wave_instructions": round(s2["SQ_INSTS_VALU"][0], 1),
                                "lane_utilisation": round(s2["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * s2["SQ_ACTIVE_INST_VALU"][0]), 3)}
    except SystemExit:
        pass
    # the whole frame from the same passes: every production kernel of a config-2 frame runs once per frame, so
    # the frame's counter traffic is the sum of their per-launch averages (COUNT-variant replays and the probes excluded)
    frame = {}
    for kname, f1 in fetch.items():
        short = kname.replace("void hkd::", "").split("(")[0]
        if not short.startswith("k_") or "FETCH_SIZE" not in f1:
            continue
        if re.search(r"<(true|false), true, \d>|k_prepass<true|k_stream|k_valu|k_gather|k_copy|k_resolve|k_join|k_count", short):
            continue
        w1 = next((v for k, v in write.items() if k == kname), None)
        if not w1 or "WRITE_SIZE" not in w1:
            continue
        frame[short] = int(2 * f1["FETCH_SIZE"][0] * 1024 + w1["WRITE_SIZE"][0] * 1024)
    out["frame"] = {"hbm_bytes_per_frame": sum(frame.values()), "per_kernel": frame, "algorithmic_bytes_per_frame": 1700 * 1920 * 1080,
                    "note": "sum over the production kernels of a config-2 frame of (2 x FETCH_SIZE + WRITE_SIZE) per launch; uniform-tile store elision "
                            "skips the background tiles' reservoir stores, which SURVEY 8d's 1.70 KB/px charges to every pixel"}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["limiter"]), out["ratio_to_algorithmic"])


if __name__ == "__main__":
    main()
