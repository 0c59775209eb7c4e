OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wavefront_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "wavefront or threaded or config3 or config4 or sponza" > $OUT/r04e_pytest.log 2>&1; tail -3 $OUT/r04e_pytest.log
timeout 600 python tools/wf_timeline.py 3 4 > $OUT/r04e_wf_timeline.json 2> $OUT/r04e_wf_timeline.err; tail -3 $OUT/r04e_wf_timeline.err
python - <<PY
import json
d=json.load(open("$OUT/r04e_wf_timeline.json"))
for cfg,v in d.items():
    for s in v["trace_stages"]:
        print(cfg, s["stage"], "launch", s["launch_us"], "dry", s["queue_dry_after_us"], "tail", s["tail_us"], "resid", s["mean_wave_residency"], "rays", s["rays"], "max", s["max_node_steps"], "us/step", round(s["tail_us"]/s["max_node_steps"],2))
PY
for C in 3 4; do
  timeout 600 python bench.py --config $C --passes --no-cpu-baseline --blocks 3 > $OUT/r04e_bench_config$C.json 2> $OUT/r04e_bench_config$C.err
  python - <<PY
import json
d = json.loads(open("$OUT/r04e_bench_config$C.json").read().strip().splitlines()[-1])
print("config $C:", d["value"], "Mray/s", d["ms_per_step"], "ms", d.get("pass_ms"), d["replay_bit_identical"])
PY
done
