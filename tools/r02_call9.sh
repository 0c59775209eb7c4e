#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
bench_line() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; p = d.get('pass_ms', {})
        print('$1', 'ms/frame', d['ms_per_step'], 'min', d['min_ms_per_step'], 'Mray/s', d['value'], 'sched', r.get('schedule'), 'indirect', r['avg_launch_ms'], 'alone', r['alone']['avg_launch_ms'], 'same', d['replay_bit_identical'], ' '.join(f'{k[:9]}={v:.3f}' for k, v in p.items()))
"; }
timeout 900 python -m pytest tests/test_wavefront_gpu.py -x -q > $OUT/c9_wf_pytest.log 2>&1; tail -4 $OUT/c9_wf_pytest.log | cut -c1-300
for C in 3 4; do
  timeout 600 python bench.py --config $C --no-cpu-baseline --no-hbm-probe --blocks 3 2>/dev/null | tee $OUT/c9_bench_c${C}.json | bench_line "c$C octant queues"
done
timeout 300 python bench.py --no-cpu-baseline --no-hbm-probe --blocks 3 --ctx-flags 64 2>/dev/null | bench_line "c2 wavefront"
cd /tmp && export TMPDIR=/tmp
CMD="python $OLDPWD/bench.py --config 3 --steps 3 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe"
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OUT/prof_tcc -- $CMD > /dev/null 2>&1
DB=$(find $OUT/prof_tcc -name "*.db" | head -1); [ -n "$DB" ] && python $OLDPWD/tools/pmc_summary.py $DB > $OUT/c9_pmc_tcc.txt
grep -A5 "k_wf_trace" $OUT/c9_pmc_tcc.txt | head -8
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_wf -- $CMD > /dev/null 2>&1
DB=$(find $OUT/prof_wf -name "*.db" | head -1); [ -n "$DB" ] && python $OLDPWD/tools/rocpd_summary.py $DB > $OUT/c9_config3_kernel_stats.txt
grep -E "k_wf" $OUT/c9_config3_kernel_stats.txt | head -5 | cut -c1-175
rm -rf $OUT/prof_tcc $OUT/prof_wf
