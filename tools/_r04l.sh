OUT=$PWD/gpurun_out; mkdir -p $OUT
run() { # tag, env
  for C in 3 4; do
    env $2 timeout 600 python bench.py --config $C --no-cpu-baseline --no-hbm-probe --blocks 3 > $OUT/r04l_$1_config$C.json 2> $OUT/r04l_$1_config$C.err
    python - <<PY
import json
d = json.loads(open("$OUT/r04l_$1_config$C.json").read().strip().splitlines()[-1])
print("$1 config $C:", d["ms_per_step"], "ms  indirect", d["roofline"]["avg_launch_ms"], "alone", d["roofline"]["alone"]["avg_launch_ms"], d["replay_bit_identical"])
PY
  done
}
run wg8 "HK_X=1"
run wg7 "HK_WF_TRACE_WG_PER_CU=7"
run wg6 "HK_WF_TRACE_WG_PER_CU=6"
run wg5 "HK_WF_TRACE_WG_PER_CU=5"
run wg4 "HK_WF_TRACE_WG_PER_CU=4"
run wg8b "HK_X=2"
