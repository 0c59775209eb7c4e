#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $OLDPWD/bench.py --steps 6 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe"
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $SET -d $OUT/prof_$N -- $CMD > $OUT/c10_$N.log 2>&1
  DB=$(find $OUT/prof_$N -name "*.db" | head -1); [ -n "$DB" ] && python $OLDPWD/tools/pmc_summary.py $DB > $OUT/c10_pmc_$N.txt || tail -3 $OUT/c10_$N.log
  grep -A7 "k_indirect<true, false, true>\|k_prepass<false, true>\|k_spatial_reuse<false>" $OUT/c10_pmc_$N.txt | head -30
  rm -rf $OUT/prof_$N
done
