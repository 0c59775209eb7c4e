#!/bin/bash
# Round-2 evidence run: tools/gpu_round.sh (GPU suite, bench lines of configs 2..5, rocprofv3 kernel stats, PMC passes of config 2)
# plus the PMC passes of config 3 (wavefront schedule: k_wf_trace is the dominant kernel there).
TAG=${1:-r02_final}
bash tools/gpu_round.sh $TAG $2
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $OLDPWD/bench.py --config 3 --steps 3 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/prof_sq3 -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch3 -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write3 -- $CMD > /dev/null 2>&1
cd $OLDPWD
for d in sq3 fetch3 write3; do
  DB=$(find $OUT/prof_$d -name "*.db" | head -1)
  [ -z "$DB" ] && { echo "no db for $d"; continue; }
  python tools/pmc_summary.py $DB > $OUT/${TAG}_config3_pmc_$d.txt
done
grep -A9 "k_wf_trace\|k_wf_shade" $OUT/${TAG}_config3_pmc_sq3.txt | head -24
grep -A2 "k_wf_" $OUT/${TAG}_config3_pmc_fetch3.txt $OUT/${TAG}_config3_pmc_write3.txt | head -40
rm -rf $OUT/prof_sq3 $OUT/prof_fetch3 $OUT/prof_write3
# instance motion on the device: probe + kernel durations of the refit / rebuild kernels
timeout 600 python tools/refit_probe.py 2000 20000 > $OUT/${TAG}_device_refit_probe.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_refit -- python $OLDPWD/tools/refit_probe.py 20000 > /dev/null 2>&1
cd $OLDPWD
DB=$(find $OUT/prof_refit -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB | grep -E "^kernel|k_refit|k_sah|k_lbvh|k_copy_region|k_gather|radix|onesweep|histogram" | head -20 > $OUT/${TAG}_refit_kernel_stats.txt
cat $OUT/${TAG}_refit_kernel_stats.txt | cut -c1-170
rm -rf $OUT/prof_refit
