#!/bin/bash
# where do the waves of the large-scene kernels wait?  two SQ passes on bench config 3 (float nodes unless HK_COMPACT... is unset)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out; mkdir -p $OUT
CMD="python bench.py --config ${1:-3} --steps 4 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe --no-extra-configs --sustained-seconds 0"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $OUT/prof_w1 -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES -d $OUT/prof_w2 -- $CMD > /dev/null 2>&1
for d in w1 w2; do
  DB=$(find $OUT/prof_$d -name "*_results.db" | head -1)
  python tools/pmc_summary.py $DB > $OUT/pmc_wait_$d.txt
done
grep -A9 "k_wf_trace\|k_prepass<false\|k_direct_lit<false, false" $OUT/pmc_wait_w1.txt | head -60
grep -A9 "k_wf_trace\|k_prepass<false\|k_direct_lit<false, false" $OUT/pmc_wait_w2.txt | head -60
