OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python tools/wf_timeline.py 3 4 > $OUT/r04f_wf_timeline.json 2> $OUT/r04f_wf_timeline.err; tail -3 $OUT/r04f_wf_timeline.err
python - <<PY
import json
d=json.load(open("$OUT/r04f_wf_timeline.json"))
for cfg,v in d.items():
    print(v["wall_clock_khz"])
    for s in v["trace_stages"]:
        print(cfg, s["stage"], "launch", s["launch_us"], "dry", s["queue_dry_after_us"], "tail", s["tail_us"], "resid", s["mean_wave_residency"], "rays", s["rays"], "max", s["max_node_steps"], s["long_walks_256_steps_up"])
PY
