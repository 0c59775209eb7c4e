#!/bin/bash
# Round 2, GPU call 3: phase-parked trace kernel + device refit - parity, parameter A/B, PMC lane utilisation
OUT=$PWD/gpurun_out; mkdir -p $OUT
bench_line() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; p = d.get('pass_ms', {})
        print('$1', 'ms/frame', d['ms_per_step'], 'min', d['min_ms_per_step'], 'Mray/s', d['value'], 'sched', r.get('schedule'), 'indirect', r['avg_launch_ms'], 'alone', r['alone']['avg_launch_ms'], 'same', d['replay_bit_identical'], ' '.join(f'{k[:9]}={v:.3f}' for k, v in p.items()))
"; }
timeout 900 python -m pytest tests/test_wavefront_gpu.py tests/test_device_refit.py -q > $OUT/c3_wf_pytest.log 2>&1; tail -30 $OUT/c3_wf_pytest.log | cut -c1-300
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_dynamic_scene.py -x -q -m gpu -k "dynamic or instance or config3 or sponza or threaded" > $OUT/c3_pytest_subset.log 2>&1; tail -3 $OUT/c3_pytest_subset.log
for V in default wfF wfG wfH wfI; do
  L=$PWD/build_ab/$V.so; [ $V = default ] && L=$PWD/bevy-hikari_amd/libhikari_hip.so
  for C in 3 4; do
    HIKARI_HIP_LIB=$L timeout 600 python bench.py --config $C --no-cpu-baseline --no-hbm-probe --blocks 3 2>/dev/null | tee $OUT/c3_bench_c${C}_$V.json | bench_line "c$C $V"
  done
done
timeout 300 python bench.py --no-cpu-baseline --no-hbm-probe --blocks 3 --ctx-flags 64 2>/dev/null | tee $OUT/c3_bench_c2_wavefront.json | bench_line "c2 wavefront"
timeout 300 python bench.py --config 5 --no-cpu-baseline --no-hbm-probe --blocks 3 --ctx-flags 64 2>/dev/null | tee $OUT/c3_bench_c5_wavefront.json | bench_line "c5 wavefront"
cd /tmp && export TMPDIR=/tmp
CMD="python $OLDPWD/bench.py --config 3 --steps 3 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_wf -- $CMD > /dev/null 2>&1
DB=$(find $OUT/prof_wf -name "*.db" | head -1); [ -n "$DB" ] && python $OLDPWD/tools/rocpd_summary.py $DB > $OUT/c3_wf_config3_kernel_stats.txt
grep -E "k_wf|k_indirect|k_direct|k_prepass" $OUT/c3_wf_config3_kernel_stats.txt | head -12 | cut -c1-175
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/prof_sq -- $CMD > /dev/null 2>&1
DB=$(find $OUT/prof_sq -name "*.db" | head -1); [ -n "$DB" ] && python $OLDPWD/tools/pmc_summary.py $DB > $OUT/c3_wf_config3_pmc_sq.txt
grep -A9 "k_wf_trace" $OUT/c3_wf_config3_pmc_sq.txt | head -12
rm -rf $OUT/prof_wf $OUT/prof_sq
