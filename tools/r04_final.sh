#!/bin/bash
# Round-4 evidence run: GPU suite, bench lines of configs 2..5 (+ kernel stats), PMC passes of config 2 (tools/gpu_round.sh), the traffic
# profile bench.py reads, the band probe, the device-refit probe, the gather probe + FETCH_SIZE calibration and the trace stages' bulk /
# tail split.   Usage: tools/r04_final.sh [tag] [notests]
TAG=${1:-r04_final}
bash tools/gpu_round.sh $TAG $2
OUT=$PWD/gpurun_out
python tools/make_traffic_profile.py $OUT/$TAG $OUT/${TAG}_indirect_hbm_traffic.json
timeout 400 python tools/band_probe.py > $OUT/${TAG}_band_probe.json 2> /dev/null
timeout 600 python tools/refit_probe.py 2000 20000 > $OUT/${TAG}_device_refit_probe.json 2> /dev/null
timeout 300 python tools/gather_probe.py > $OUT/${TAG}_gather_probe.json 2> /dev/null
timeout 300 python tools/wf_timeline.py 3 4 > $OUT/${TAG}_wf_timeline.json 2> /dev/null
timeout 300 python tools/wf_timeline.py --no-wide-walk 3 4 > $OUT/${TAG}_wf_timeline_skip_link.json 2> /dev/null
bash tools/pmc_calibrate.sh $TAG > /dev/null 2>&1
# the wide walk's A/B on the scenes it serves: the same bench lines with HK_CTX_NO_WIDE_WALK (threaded skip-link walk for every ray)
for C in 3 4; do
  timeout 600 python bench.py --config $C --no-wide-walk --passes --no-cpu-baseline --no-hbm-probe --blocks 3 > $OUT/${TAG}_bench_config${C}_no_wide_walk.json 2> /dev/null
done
