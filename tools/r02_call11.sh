#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
bench_line() { python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; p = d.get('pass_ms', {})
        print('$1', 'ms/frame', d['ms_per_step'], 'min', d['min_ms_per_step'], 'Mray/s', d['value'], 'indirect', r['avg_launch_ms'], 'alone', r['alone']['avg_launch_ms'], 'same', d['replay_bit_identical'], ' '.join(f'{k[:9]}={v:.3f}' for k, v in p.items()))
"; }
for V in hoist hoist_both_reload; do
  HIKARI_HIP_LIB=$PWD/build_ab/$V.so timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "test_bit_exact_vs_oracle_every_frame or test_random_settings_vs_oracle or test_full_size_1080p or sponza" > $OUT/c11_pytest_$V.log 2>&1; echo "variant $V: $(tail -1 $OUT/c11_pytest_$V.log)"
done
for rep in 1 2 3; do
  for V in default hoist hoist_both_reload hoist_reload; do
    L=$PWD/build_ab/$V.so; [ $V = default ] && L=$PWD/bevy-hikari_amd/libhikari_hip.so
    HIKARI_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-hbm-probe --blocks 3 --passes 2>/dev/null | bench_line "c2 $V"
  done
done
for V in default hoist_both_reload; do
  L=$PWD/build_ab/$V.so; [ $V = default ] && L=$PWD/bevy-hikari_amd/libhikari_hip.so
  HIKARI_HIP_LIB=$L timeout 300 python bench.py --config 5 --no-cpu-baseline --no-hbm-probe --blocks 3 --passes 2>/dev/null | bench_line "c5 $V"
done
