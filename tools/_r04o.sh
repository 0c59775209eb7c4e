OUT=$PWD/gpurun_out; mkdir -p $OUT
for C in 3 4; do
  timeout 600 python bench.py --config $C --no-cpu-baseline --blocks 2 --steps 6 > $OUT/r04o_bench_config$C.json 2> $OUT/r04o_bench_config$C.err
  python - <<PY
import json
d = json.loads(open("$OUT/r04o_bench_config$C.json").read().strip().splitlines()[-1])
print("config $C:", d["ms_per_step"], json.dumps(d["roofline"]["bvh_walk"]["per_ray"]), json.dumps(d["roofline"]["bvh_walk"]["walk_counts_per_launch"]))
PY
done
