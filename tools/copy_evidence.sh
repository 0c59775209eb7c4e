#!/bin/bash
# Copy what tools/r06_final.sh left under gpurun_out/ into profiles/ (tracked).   Usage: tools/copy_evidence.sh [tag]
# bench lines: the LAST stdout line of each run, pretty-printed; the traffic profiles bench.py reads also under their fixed names.
TAG=${1:-r06_final}
G=gpurun_out
P=profiles
for C in 2 3 4 5; do
  tail -1 $G/${TAG}_bench_config$C.json | python -m json.tool > $P/${TAG}_bench_config$C.json || echo "config $C: no bench line"
done
for f in $G/${TAG}_*.txt $G/${TAG}_band_probe.json $G/${TAG}_coop_probe.json $G/${TAG}_indirect_hbm_traffic.json $G/${TAG}_walk_hbm_traffic.json; do
  [ -s $f ] && cp $f $P/
done
R=${TAG%%_*}
cp $G/${TAG}_indirect_hbm_traffic.json $P/${R}_indirect_hbm_traffic.json
cp $G/${TAG}_walk_hbm_traffic.json $P/${R}_walk_hbm_traffic.json
for f in default_mode_sequence_config3_1080p default_mode_sequence_config4_4k default_mode_config4_4k_row_ranges_vs_oracle; do
  [ -s $G/$f.json ] && cp $G/$f.json $P/${R}_$f.json
done
grep -n -E "passed|failed|error" $G/${TAG}_pytest.log | tail -3 > $P/${TAG}_pytest.txt
cat $P/${TAG}_pytest.txt
