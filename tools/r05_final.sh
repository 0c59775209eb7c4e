#!/bin/bash
# Round-5 evidence run (one GPU call): the GPU suite, bench lines of configs 2..5 (+ rocprofv3 kernel stats per config, the PMC passes of
# config 2: tools/gpu_round.sh), the traffic profiles bench.py reads (config 2: make_traffic_profile.py; configs 3 / 4: pmc_fetch.sh +
# make_walk_traffic.py), the lane utilisation of configs 3 / 4, the band probe.   Usage: tools/r05_final.sh [tag] [notests]
TAG=${1:-r05_final}
bash tools/gpu_round.sh $TAG $2
OUT=$PWD/gpurun_out
python tools/make_traffic_profile.py $OUT/$TAG $OUT/${TAG}_indirect_hbm_traffic.json
for C in 3 4; do bash tools/pmc_fetch.sh $C $TAG > /dev/null 2>&1; done
python tools/make_walk_traffic.py $OUT/$TAG $OUT/${TAG}_walk_hbm_traffic.json
for C in 3 4; do bash tools/pmc_lanes.sh $C > $OUT/${TAG}_lanes_config$C.txt 2>&1; done
timeout 400 python tools/band_probe.py > $OUT/${TAG}_band_probe.json 2> /dev/null
timeout 120 python tools/coop_probe.py > $OUT/${TAG}_coop_probe.json 2> /dev/null
