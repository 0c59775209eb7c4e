#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "pipelining or dynamic or instance_updates_in_flight" 2>&1 | tail -8 | tee gpurun_out/r05_call4_pytest.txt
timeout 900 python -m pytest tests/test_dynamic_scene.py tests/test_default_mode_sequence_gpu.py -q -m gpu -x 2>&1 | tail -8 | tee -a gpurun_out/r05_call4_pytest.txt
bash tools/ab_variants.sh "3 4" base norank noemit 2>&1 | tee gpurun_out/r05_rank_ab2.txt
for C in 2 3 4; do
  HK_NO_PREPASS_PIPELINE=1 timeout 300 python bench.py --config $C --no-cpu-baseline --no-extra-configs --sustained-seconds 0 --no-hbm-probe 2> /dev/null | tail -1 > gpurun_out/r05_nopipe_$C.json
  timeout 300 python bench.py --config $C --no-cpu-baseline --no-extra-configs --sustained-seconds 0 2> gpurun_out/r05_pipe_$C.err | tail -1 > gpurun_out/r05_pipe_$C.json
  python - <<PY
import json
for t in ("nopipe","pipe"):
    try:
        d=json.load(open("gpurun_out/r05_%s_$C.json"%t)); print("config $C", t, d["ms_per_step"], d["blocks_ms_per_step"], d["replay_bit_identical"])
    except Exception as e: print("config $C", t, "FAILED", e)
PY
done 2>&1 | tee gpurun_out/r05_pipe_ab.txt
