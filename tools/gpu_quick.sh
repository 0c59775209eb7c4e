#!/bin/bash
# Quick GPU iteration: selected tests + bench lines.  Usage: tools/gpu_quick.sh <tag> "<pytest -k expr or empty>" "<configs>"
TAG=${1:-q}; KEXPR=$2; CONFIGS=${3:-"2 3 4"}
OUT=$PWD/gpurun_out; mkdir -p $OUT
if [ -n "$KEXPR" ]; then
  if [ "$KEXPR" = "all" ]; then timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1
  else timeout 1500 python -m pytest tests -x -q -m gpu -k "$KEXPR" > $OUT/${TAG}_pytest.log 2>&1; fi
  tail -15 $OUT/${TAG}_pytest.log
fi
for C in $CONFIGS; do
  EXTRA="--no-cpu-baseline --blocks 3"; [ $C != 2 ] && EXTRA="$EXTRA --no-hbm-probe"
  timeout 900 python bench.py --config $C --passes $EXTRA > $OUT/${TAG}_bench_config$C.json 2> $OUT/${TAG}_bench_config$C.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_bench_config$C.json").read().strip().splitlines()[-1])
    print("config $C:", d["value"], "Mray/s", d["ms_per_step"], "ms", "indirect alone", d["roofline"]["alone"]["avg_launch_ms"], d["roofline"].get("hbm_ceiling_measured"), {k: v for k, v in d.get("pass_ms", {}).items()})
except Exception as e:
    print("config $C failed", e); print(open("$OUT/${TAG}_bench_config$C.err").read()[-1500:])
PY
done
