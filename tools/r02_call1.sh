#!/bin/bash
# Round 2, GPU call: wavefront schedule parity + A/B of fused-kernel variants + bench lines of configs 2..5 for both schedules.
OUT=$PWD/gpurun_out; mkdir -p $OUT
bench_line() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; p = d.get('pass_ms', {})
        print('$1', 'ms/frame', d['ms_per_step'], 'min', d['min_ms_per_step'], 'Mray/s', d['value'], 'sched', r.get('schedule'), 'indirect', r['avg_launch_ms'], 'alone', r['alone']['avg_launch_ms'], 'same', d['replay_bit_identical'], ' '.join(f'{k[:9]}={v:.3f}' for k, v in p.items()))
"; }
timeout 900 python -m pytest tests/test_wavefront_gpu.py -x -q > $OUT/c1_wf_pytest.log 2>&1; tail -15 $OUT/c1_wf_pytest.log
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_wavefront_gpu.py > $OUT/c1_pytest.log 2>&1; tail -5 $OUT/c1_pytest.log
for V in both both_reload reload; do
  HIKARI_HIP_LIB=$PWD/build_ab/$V.so timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "test_bit_exact_vs_oracle_every_frame or test_random_settings_vs_oracle or test_full_size_1080p" > $OUT/c1_pytest_$V.log 2>&1; echo "variant $V: $(tail -1 $OUT/c1_pytest_$V.log)"
done
for rep in 1 2; do
  for V in default both both_reload reload; do
    L=$PWD/build_ab/$V.so; [ $V = default ] && L=$PWD/bevy-hikari_amd/libhikari_hip.so
    HIKARI_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-hbm-probe --blocks 3 --passes 2>/dev/null | tee $OUT/c1_bench_c2_${V}_$rep.json | bench_line "c2 $V"
  done
done
timeout 300 python bench.py --no-cpu-baseline --no-hbm-probe --blocks 3 --passes --ctx-flags 64 2>$OUT/c1_bench_c2_wf.err | tee $OUT/c1_bench_c2_wavefront.json | bench_line "c2 wavefront"
for C in 3 4 5; do
  timeout 900 python bench.py --config $C --passes --no-cpu-baseline --no-hbm-probe --blocks 3 --ctx-flags 128 2>$OUT/c1_bench_c${C}_fused.err | tee $OUT/c1_bench_c${C}_fused.json | bench_line "c$C fused"
  timeout 900 python bench.py --config $C --passes --no-cpu-baseline --no-hbm-probe --blocks 3 2>$OUT/c1_bench_c${C}_wf.err | tee $OUT/c1_bench_c${C}_wavefront.json | bench_line "c$C default"
done
cd /tmp && export TMPDIR=/tmp
for C in 2 3; do
  FL=""; [ $C = 2 ] && FL="--ctx-flags 64"
  CMD="python $OLDPWD/bench.py --config $C --steps 3 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe $FL"
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_wf_c$C -- $CMD > /dev/null 2>&1
  DB=$(find $OUT/prof_wf_c$C -name "*.db" | head -1)
  [ -n "$DB" ] && python $OLDPWD/tools/rocpd_summary.py $DB > $OUT/c1_wf_config${C}_kernel_stats.txt
  head -14 $OUT/c1_wf_config${C}_kernel_stats.txt | cut -c1-170
  rm -rf $OUT/prof_wf_c$C
done
