OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 python tools/gather_probe.py > $OUT/r04_gather_probe.json 2> $OUT/r04_gather_probe.err; tail -3 $OUT/r04_gather_probe.err
python - <<PY
import json
d=json.load(open("$OUT/r04_gather_probe.json"))
for r in d["sweep"]:
    if r["waves_per_simd"] in (1,7): print(r)
PY
bash tools/pmc_calibrate.sh r04
cat $OUT/r04_fetch_calibration_known.json
timeout 600 python bench.py --no-cpu-baseline > $OUT/r04c_bench.json 2> $OUT/r04c_bench.err; tail -2 $OUT/r04c_bench.err
python - <<PY
import json
d=json.loads(open("$OUT/r04c_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k,v in d["extra_configs"].items(): print(k, v["ms_per_step"], json.dumps(v.get("roofline"))[:1500])
PY
