#!/bin/bash
# One-level walk A/B on one GPU box: queue depth (HK_FLAT_CAP builds), direction orderings (HK_FLAT_ORDERINGS), vs the two-level walk
# (HK_FLAT_DISABLE).  Prints ms/frame + per-pass times of config 2 (and config 5 for the main build).
run() { # label lib env...
  local label=$1 lib=$2; shift 2
  env "$@" HIKARI_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-hbm-probe --no-extra-configs --sustained-seconds 0 --passes --steps 48 --warmup 8 --blocks 3 ${CFG} 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); p = d.get('pass_ms', {})
        print('$label', d['config']['traversal'], 'ms/frame', d['ms_per_step'], 'indirect alone', d['roofline']['alone']['avg_launch_ms'], ' '.join(f'{k[:9]}={v:.3f}' for k, v in p.items()))
"
}
MAIN=$PWD/bevy-hikari_amd/libhikari_hip.so
for rep in 1 2; do
  CFG=""
  run two_level $MAIN HK_FLAT_DISABLE=1
  run cap2_ord8 $MAIN X=1
  run cap2_ord4 $MAIN HK_FLAT_ORDERINGS=4
  run cap2_ord2 $MAIN HK_FLAT_ORDERINGS=2
  run cap2_ord1 $MAIN HK_FLAT_ORDERINGS=1
  run cap1_ord8 $PWD/build_ab/flat_cap1.so X=1
  run cap3_ord8 $PWD/build_ab/flat_cap3.so X=1
done
CFG="--config 5 --steps 12"
run c5_two_level $MAIN HK_FLAT_DISABLE=1
run c5_cap2_ord8 $MAIN X=1
run c5_cap3_ord8 $PWD/build_ab/flat_cap3.so X=1
