OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_multi_gpu.py tests/test_parity_gpu.py tests/test_bench_ranks.py tests/test_cpp_host.py -x -q -m gpu -k "multi or band or bench or cpp or gather" > $OUT/r04j_pytest.log 2>&1; tail -3 $OUT/r04j_pytest.log
timeout 600 python tools/band_probe.py --configs 2 4 > $OUT/r04j_band_probe.json 2> $OUT/r04j_band_probe.err; tail -2 $OUT/r04j_band_probe.err
python - <<PY
import json
d=json.load(open("$OUT/r04j_band_probe.json"))
for cfg,v in d["configs"].items():
    for k,r in v["bands"].items():
        print(cfg,k,"max band",r["max_band_ms"],"exch",r["exchange_ms_predicted"],"frame",r["frame_ms_predicted"],"speedup",r["speedup_predicted"])
PY
