#!/usr/bin/env python3
"""profiles/<tag>_walk_hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_fetch.sh on configs 3 and 4
(gpurun_out/<tag>_config{3,4}_pmc_{FETCH,WRITE}_SIZE.txt): HBM-side bytes per launch of every kernel and, for the trace launches of the
queue-based indirect pass (k_wf_trace_wide), per pass - what bench.py's extra_configs[3|4].roofline.hbm_side reads.
    python tools/make_walk_traffic.py gpurun_out/r05_final profiles/r05_walk_hbm_traffic.json"""
import json
import re
import sys


def blocks(path, counter):
    out = {}
    for b in re.split(r"\n(?=\S)", open(path).read()):
        lines = b.strip().splitlines()
        for l in lines[1:]:
            p = l.split()
            if len(p) >= 4 and p[0] == counter and p[1] == "avg":
                out[lines[0]] = (float(p[2]), int(p[3].split("=")[1]))
    return out


def short(name):
    return re.sub(r"\(.*", "", name.replace("void hkd::", "").replace("hkd::", "")).strip()


def main():
    prefix, dst = sys.argv[1], sys.argv[2]
    out = {"source": "tools/pmc_fetch.sh 3 / 4 (rocprofv3 --kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE: one counter per pass) on `bench.py --config N` frames; per-launch averages",
           "unit_note": "counters in KB; profiles/r04_fetch_calibration.json: FETCH_SIZE tallies 64 B per memory-side read request - lower bound = x 1 (gathers: the walks), "
                        "upper bound = x 2 (coalesced 128-B streams)"}
    for cfg, stages in ((3, 4), (4, 3)):
        try:
            f = blocks(f"{prefix}_config{cfg}_pmc_FETCH_SIZE.txt", "FETCH_SIZE")
            w = blocks(f"{prefix}_config{cfg}_pmc_WRITE_SIZE.txt", "WRITE_SIZE")
        except OSError as e:
            print("skipping config", cfg, e, file=sys.stderr)
            continue
        per = {}
        for k, (fk, n) in f.items():
            if "hkd::" not in k or k not in w:
                continue
            wk = w[k][0]
            per[short(k)] = {"launches": n, "fetch_kb": round(fk, 1), "write_kb": round(wk, 1), "hbm_bytes_per_launch_lower": int((fk + wk) * 1024),
                             "hbm_bytes_per_launch_upper": int((2 * fk + wk) * 1024)}
        out[f"config{cfg}"] = per
        t = next((v for k, v in per.items() if k.startswith("k_wf_trace_wide<false, false>")), None)
        if t:
            out[f"config{cfg}_trace_stages"] = {"stages_per_pass": stages, "hbm_bytes_per_pass_lower": t["hbm_bytes_per_launch_lower"] * stages,
                                                "hbm_bytes_per_pass_upper": t["hbm_bytes_per_launch_upper"] * stages,
                                                "note": "what leaves the L2s towards Infinity Cache / HBM during the pass's trace launches (the records, triangles and instances the "
                                                        "caches did not hold, + the ray / hit planes)"}
    json.dump(out, open(dst, "w"), indent=1)
    print(dst, {k: v for k, v in out.items() if k.endswith("_trace_stages")})


if __name__ == "__main__":
    main()
