#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "pipelining" 2>&1 | tail -8 | tee gpurun_out/r05_call5_pytest.txt
bash tools/ab_variants.sh "3 4" base norank 2>&1 | tee gpurun_out/r05_rank_ab3.txt
for M in none lds; do
  HK_PREPASS_PIPELINE=$M timeout 300 python bench.py --config 2 --no-cpu-baseline --no-extra-configs --sustained-seconds 0 --no-hbm-probe 2> /dev/null | tail -1 > gpurun_out/r05_pipe2_$M.json
done
for C in 3 4; do
  timeout 400 python bench.py --config $C --no-cpu-baseline 2> gpurun_out/r05_bench_$C.err | tail -1 > gpurun_out/r05_bench_$C.json
done
python - <<PY | tee gpurun_out/r05_pipe_ab2.txt
import json
for t in ("none","lds"):
    d=json.load(open("gpurun_out/r05_pipe2_%s.json"%t)); print("config 2 HK_PREPASS_PIPELINE=%s"%t, d["ms_per_step"], d["blocks_ms_per_step"], d["replay_bit_identical"])
for c in (3,4):
    try:
        d=json.load(open("gpurun_out/r05_bench_%d.json"%c)); print(c, d["ms_per_step"], json.dumps(d["roofline"].get("bvh_walk"))[:3000])
    except Exception as e: print(c,"FAILED",e)
PY
