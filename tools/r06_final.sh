#!/bin/bash
# Round-6 evidence run (one GPU call): the GPU suite, bench lines of configs 2..5 (+ rocprofv3 kernel stats per config, the PMC passes of
# config 2: tools/gpu_round.sh), the traffic profiles bench.py reads, the lane utilisation of configs 3 / 4, the band probe with the
# measured-time controller, the band anatomy, the motion bench.   Usage: tools/r06_final.sh [tag] [notests]
TAG=${1:-r06_final}
bash tools/gpu_round.sh $TAG $2
OUT=$PWD/gpurun_out
python tools/make_traffic_profile.py $OUT/$TAG $OUT/${TAG}_indirect_hbm_traffic.json
for C in 3 4; do bash tools/pmc_fetch.sh $C $TAG > /dev/null 2>&1; done
python tools/make_walk_traffic.py $OUT/$TAG $OUT/${TAG}_walk_hbm_traffic.json
for C in 3 4; do bash tools/pmc_lanes.sh $C > $OUT/${TAG}_lanes_config$C.txt 2>&1; done
timeout 600 python tools/band_probe.py > $OUT/${TAG}_band_probe.json 2> /dev/null
timeout 300 python tools/band_anatomy.py --config 2 > $OUT/${TAG}_band_anatomy_config2.json 2> /dev/null
timeout 500 python tools/band_anatomy.py --config 4 --balanced > $OUT/${TAG}_band_anatomy_config4.json 2> /dev/null
timeout 400 python bench.py --motion > $OUT/${TAG}_motion.json 2> /dev/null
