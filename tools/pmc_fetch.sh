#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel of a config's frames (one counter per pass; the calibration of profiles/r04_fetch_calibration.json
# applies: 64 B per read request - x 1 for gathers of records up to 64 B, x 2 for coalesced 128-B streams).  Usage: tools/pmc_fetch.sh <config> [tag]
C=${1:-4}; TAG=${2:-fetch}
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $OLDPWD/bench.py --config $C --steps 3 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe --no-extra-configs --sustained-seconds 0"
for K in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $K -d $OUT/prof_${K}_$C -- $CMD > /dev/null 2>&1
  DB=$(find $OUT/prof_${K}_$C -name "*.db" | head -1)
  [ -n "$DB" ] && python $OLDPWD/tools/pmc_summary.py $DB > $OUT/${TAG}_config${C}_pmc_$K.txt
  rm -rf $OUT/prof_${K}_$C
done
cd $OLDPWD
grep -A1 "k_wf_trace_wide\|k_prepass<false, 4>\|k_direct_lit<false, false, 0>\|k_wf_shade" $OUT/${TAG}_config${C}_pmc_FETCH_SIZE.txt | cut -c1-120
