OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 python tools/gather_probe.py > $OUT/r04_gather_probe.json 2> $OUT/r04_gather_probe.err; tail -3 $OUT/r04_gather_probe.err
python - <<PY
import json
d=json.load(open("$OUT/r04_gather_probe.json"))
for r in d["by_level_and_width"]: print(r)
PY
timeout 600 python tools/wf_timeline.py 3 4 > $OUT/r04_wf_timeline.json 2> $OUT/r04_wf_timeline.err; tail -3 $OUT/r04_wf_timeline.err
cat $OUT/r04_wf_timeline.json
