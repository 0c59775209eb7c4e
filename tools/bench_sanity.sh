#!/bin/bash
# the driver's invocations of bench.py with flag combinations it may pass: one JSON line each, wall time
mkdir -p gpurun_out
for a in "--gpus 1 --steps 10 --warmup 3" "--gpus 1 --steps 1 --warmup 0" "--steps 200 --warmup 10" ""; do
  t0=$SECONDS
  python bench.py $a 2> gpurun_out/sanity.err | tail -1 > gpurun_out/sanity.json
  python - "$a" $((SECONDS - t0)) <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/sanity.json"))
    print(repr(sys.argv[1]), "wall", sys.argv[2], "s:", d["steps"], d["warmup"], d["value"], d["ms_per_step"], d["replay_bit_identical"],
          sorted((d.get("extra_configs") or {}).keys()), (d.get("sustained") or {}).get("seconds"), "roofline" in d, "cpu_baseline" in d)
except Exception as e:
    print(repr(sys.argv[1]), "FAILED", e, open("gpurun_out/sanity.err").read()[-800:])
PY
done
