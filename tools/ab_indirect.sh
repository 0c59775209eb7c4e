#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on the indirect pass of configs 3 / 4: one line per variant and config - the
# pass alone (ms) and the frame (ms).   Usage: tools/ab_indirect.sh <variant ...>   ("default" = the library in the package)
OUT=$PWD/gpurun_out; mkdir -p $OUT
for V in "$@"; do
  for C in 3 4; do
    if [ "$V" = default ]; then LIB=""; else LIB="$PWD/build_ab/$V.so"; fi
    env ${LIB:+HIKARI_HIP_LIB=$LIB} timeout 300 python bench.py --config $C --no-cpu-baseline --no-hbm-probe --no-extra-configs --blocks 3 > $OUT/ab_$V_$C.json 2> $OUT/ab_$V_$C.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/ab_$V_$C.json").read().strip().splitlines()[-1])
    print("$V config $C: indirect alone", d["roofline"]["alone"]["avg_launch_ms"], "frame", d["ms_per_step"], d["replay_bit_identical"])
except Exception as e:
    print("$V config $C failed", e, open("$OUT/ab_$V_$C.err").read()[-400:])
PY
  done
done
