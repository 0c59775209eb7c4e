#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
bench_line() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; p = d.get('pass_ms', {})
        print('$1', 'ms/frame', d['ms_per_step'], 'min', d['min_ms_per_step'], 'Mray/s', d['value'], 'sched', r.get('schedule'), 'indirect', r['avg_launch_ms'], 'alone', r['alone']['avg_launch_ms'], 'same', d['replay_bit_identical'], ' '.join(f'{k[:9]}={v:.3f}' for k, v in p.items()))
"; }
HIKARI_HIP_LIB=$PWD/build_ab/pk.so timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_wavefront_gpu.py -x -q -m gpu -k "test_bit_exact_vs_oracle_every_frame or test_random_settings_vs_oracle or test_full_size_1080p or sponza or wavefront_bit_exact" > $OUT/c6_pytest_pk.log 2>&1; echo "variant pk: $(tail -1 $OUT/c6_pytest_pk.log)"
for rep in 1 2 3; do
  for V in default pk; do
    L=$PWD/build_ab/$V.so; [ $V = default ] && L=$PWD/bevy-hikari_amd/libhikari_hip.so
    HIKARI_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-hbm-probe --blocks 3 --passes 2>/dev/null | bench_line "c2 $V"
  done
done
for V in default pk; do
  L=$PWD/build_ab/$V.so; [ $V = default ] && L=$PWD/bevy-hikari_amd/libhikari_hip.so
  for C in 3 5; do
    HIKARI_HIP_LIB=$L timeout 600 python bench.py --config $C --no-cpu-baseline --no-hbm-probe --blocks 3 --passes 2>/dev/null | bench_line "c$C $V"
  done
done
