#!/bin/bash
# One call on the GPU box: GPU test suite, bench lines for BASELINE configs 2..5, rocprofv3 kernel stats per config and the
# PMC passes (SQ + HBM traffic) of config 2.   Usage: tools/gpu_round.sh <tag> [notests]   (writes gpurun_out/<tag>_*)
TAG=${1:-r02}
OUT=$PWD/gpurun_out
mkdir -p $OUT
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1
  tail -5 $OUT/${TAG}_pytest.log
fi
timeout 600 python bench.py --passes > $OUT/${TAG}_bench_config2.json 2> $OUT/${TAG}_bench_config2.err
tail -1 $OUT/${TAG}_bench_config2.json | cut -c1-600
for C in 3 4 5; do
  timeout 900 python bench.py --config $C --passes --no-cpu-baseline --blocks 3 > $OUT/${TAG}_bench_config$C.json 2> $OUT/${TAG}_bench_config$C.err
  tail -1 $OUT/${TAG}_bench_config$C.json | cut -c1-400
done
cd /tmp && export TMPDIR=/tmp
for C in 2 3 4 5; do
  STEPS=6; [ $C != 2 ] && STEPS=3
  CMD="python $OLDPWD/bench.py --config $C --steps $STEPS --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe --no-extra-configs --sustained-seconds 0"
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace_c$C -- $CMD > /dev/null 2>&1
  DB=$(find $OUT/prof_trace_c$C -name "*.db" | head -1)
  [ -n "$DB" ] && python $OLDPWD/tools/rocpd_summary.py $DB > $OUT/${TAG}_config${C}_kernel_stats.txt
  head -12 $OUT/${TAG}_config${C}_kernel_stats.txt | cut -c1-180
done
CMD="python $OLDPWD/bench.py --steps 6 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe --no-extra-configs --sustained-seconds 0"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/prof_sq -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -- $CMD > /dev/null 2>&1
cd $OLDPWD
for d in sq fetch write; do
  DB=$(find $OUT/prof_$d -name "*.db" | head -1)
  [ -z "$DB" ] && { echo "no db for $d"; continue; }
  python tools/pmc_summary.py $DB > $OUT/${TAG}_pmc_$d.txt
done
grep -A9 "k_spatial_reuse<false, false>\|k_indirect<true" $OUT/${TAG}_pmc_sq.txt | head -40
grep -A2 "k_indirect<true" $OUT/${TAG}_pmc_fetch.txt $OUT/${TAG}_pmc_write.txt
rm -rf $OUT/prof_trace_c* $OUT/prof_sq $OUT/prof_fetch $OUT/prof_write
