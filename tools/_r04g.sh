OUT=$PWD/gpurun_out; mkdir -p $OUT
run() { # tag, env
  for C in 3 4; do
    env $2 timeout 600 python bench.py --config $C --no-cpu-baseline --no-hbm-probe --blocks 3 > $OUT/r04g_$1_config$C.json 2> $OUT/r04g_$1_config$C.err
    python - <<PY
import json
d = json.loads(open("$OUT/r04g_$1_config$C.json").read().strip().splitlines()[-1])
print("$1 config $C:", d["ms_per_step"], "ms  indirect", d["roofline"]["avg_launch_ms"], "alone", d["roofline"]["alone"]["avg_launch_ms"], d["replay_bit_identical"])
PY
  done
}
run wg8 "HK_X=1"
run wg6 "HK_WF_TRACE_WG_PER_CU=6"
run wg4 "HK_WF_TRACE_WG_PER_CU=4"
run wg3 "HK_WF_TRACE_WG_PER_CU=3"
run wg2 "HK_WF_TRACE_WG_PER_CU=2"
run wg1 "HK_WF_TRACE_WG_PER_CU=1"
HK_WF_TRACE_WG_PER_CU=4 timeout 600 python tools/wf_timeline.py 3 4 > $OUT/r04g_wf_timeline_wg4.json 2> /dev/null
timeout 600 python tools/wf_timeline.py 3 4 > $OUT/r04g_wf_timeline_wg8.json 2> /dev/null
python - <<PY
import json
for t in ("wg8","wg4"):
    d=json.load(open("$OUT/r04g_wf_timeline_%s.json" % t))
    for cfg,v in d.items():
        for s in v["trace_stages"]:
            print(t, cfg, s["stage"], "launch", s["launch_us"], "dry", s["queue_dry_after_us"], "tail", s["tail_us"], "resid", s["mean_wave_residency"], s["long_walks_256_steps_up"])
PY
