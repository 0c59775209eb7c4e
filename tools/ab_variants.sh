#!/bin/bash
# tools/ab_variants.sh "<configs>" <variant> ...   per-pass times alone + frame of bench.py --config N with build_ab/<variant>.so ("base" = the shipped library)
mkdir -p gpurun_out
CONFIGS=$1; shift
for cfg in $CONFIGS; do
  for v in "$@"; do
    if [ $v = base ]; then unset HIKARI_HIP_LIB; else export HIKARI_HIP_LIB=$PWD/build_ab/$v.so; fi
    timeout 300 python bench.py --config $cfg --passes --no-cpu-baseline --no-extra-configs --sustained-seconds 0 2> gpurun_out/ab_${cfg}_${v}.err | tail -1 > gpurun_out/ab_${cfg}_${v}.json
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_${cfg}_${v}.json"))
    print("config ${cfg} ${v}: frame %.3f ms, %.0f Mray/s, replay_ok=%s, passes=%s" % (d["ms_per_step"], d["value"], d["replay_bit_identical"], {k: round(x,3) for k,x in d.get("pass_ms",{}).items()}))
except Exception as e:
    print("config ${cfg} ${v}: FAILED", e)
PY
  done
done
