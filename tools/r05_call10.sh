#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_default_mode_sequence_gpu.py tests/test_flat_walk.py -q -m gpu -x -k "config2 or flat or one_level" 2>&1 | tail -8 | tee gpurun_out/r05_call10_pytest.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "bit_exact_vs_oracle_every_frame or golden or pipelining or config5" 2>&1 | tail -5 | tee -a gpurun_out/r05_call10_pytest.txt
for V in base flat4 flatkept base flat4 flatkept; do
  if [ $V = base ]; then unset HIKARI_HIP_LIB; else export HIKARI_HIP_LIB=$PWD/build_ab/$V.so; fi
  timeout 300 python bench.py --config 2 --no-cpu-baseline --no-extra-configs --sustained-seconds 0 --no-hbm-probe --passes 2> /dev/null | tail -1 > gpurun_out/r05_flat_$V.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r05_flat_$V.json")); print("config 2 $V", d["ms_per_step"], d["blocks_ms_per_step"], {k: round(v,4) for k,v in d["pass_ms"].items() if "direct" in k})
PY
done 2>&1 | tee gpurun_out/r05_flat_kept_ab.txt
