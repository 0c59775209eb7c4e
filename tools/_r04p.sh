OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_wavefront_gpu.py tests/test_device_refit.py -x -q -m gpu -k "sponza or config3 or config4 or threaded or flight_helmet or refit or default or wavefront" > $OUT/r04p_pytest.log 2>&1; tail -5 $OUT/r04p_pytest.log
for C in 3 4; do
  timeout 600 python bench.py --config $C --passes --no-cpu-baseline --no-hbm-probe --blocks 3 > $OUT/r04p_bench_config$C.json 2> $OUT/r04p_bench_config$C.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/r04p_bench_config$C.json").read().strip().splitlines()[-1])
    print("config $C:", d["value"], "Mray/s", d["ms_per_step"], "ms", d.get("pass_ms"), d["replay_bit_identical"])
except Exception as e:
    print("failed", e, open("$OUT/r04p_bench_config$C.err").read()[-800:])
PY
done
