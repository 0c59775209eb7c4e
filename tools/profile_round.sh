#!/bin/bash
# One call on the GPU box: the round's evidence for profiles/ - bench line, kernel-trace stats, PMC passes.
# Usage: tools/profile_round.sh <tag>      (writes gpurun_out/<tag>_*)
TAG=${1:-r01_final}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python bench.py --passes > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -1 $OUT/${TAG}_bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
CMD="python $OLDPWD/bench.py --steps 6 --warmup 4 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/prof_sq -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -- $CMD > /dev/null 2>&1
cd $OLDPWD
for d in trace sq fetch write; do
  DB=$(find $OUT/prof_$d -name "*.db" | head -1)
  [ -z "$DB" ] && { echo "no db for $d"; continue; }
  if [ $d = trace ]; then python tools/rocpd_summary.py $DB > $OUT/${TAG}_kernel_stats.txt; else python tools/pmc_summary.py $DB > $OUT/${TAG}_pmc_$d.txt; fi
done
head -14 $OUT/${TAG}_kernel_stats.txt | cut -c1-200
grep -A9 "k_spatial_reuse<false, false>\|k_indirect<true" $OUT/${TAG}_pmc_sq.txt | head -40
grep -A2 "k_indirect<true" $OUT/${TAG}_pmc_fetch.txt $OUT/${TAG}_pmc_write.txt
