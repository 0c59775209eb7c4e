#!/bin/bash
# Round-3 evidence run: GPU suite, bench lines of configs 2..5 (+ kernel stats), PMC passes of config 2 (tools/gpu_round.sh), the
# band probe and the device-refit probe.   Usage: tools/r03_final.sh [tag] [notests]
TAG=${1:-r03_final}
bash tools/gpu_round.sh $TAG $2
OUT=$PWD/gpurun_out
python tools/make_traffic_profile.py $OUT/$TAG $OUT/${TAG}_indirect_hbm_traffic.json
timeout 400 python tools/band_probe.py > $OUT/${TAG}_band_probe.json 2> /dev/null
timeout 600 python tools/refit_probe.py 2000 20000 > $OUT/${TAG}_device_refit_probe.json 2> /dev/null
timeout 300 python tools/section_profile.py > $OUT/${TAG}_sections.json 2> /dev/null
