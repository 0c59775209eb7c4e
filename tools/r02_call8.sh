#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $OLDPWD/bench.py --config 3 --steps 3 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe"
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TA_[A-Za-z_]+|TCP_[A-Za-z_]+|TCC_(HIT|MISS|REQ|READ|EA0?_RDREQ)[A-Za-z_0-9]*)\b" | sort -u | tr '\n' ' ' | cut -c1-3000 > $OUT/c8_counters.txt; cat $OUT/c8_counters.txt | cut -c1-1500
for SET in "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $SET -d $OUT/prof_$N -- $CMD > $OUT/c8_$N.log 2>&1
  DB=$(find $OUT/prof_$N -name "*.db" | head -1); [ -n "$DB" ] && python $OLDPWD/tools/pmc_summary.py $DB > $OUT/c8_pmc_$N.txt || tail -3 $OUT/c8_$N.log
  grep -A5 "k_wf_trace\|k_indirect<true, true" $OUT/c8_pmc_$N.txt | head -16
  rm -rf $OUT/prof_$N
done
