#!/bin/bash
# A/B two builds of libhikari_hip.so on the same GPU box, interleaved: tools/ab.sh build_ab/A.so build_ab/B.so [reps]
A=$1; B=$2; REPS=${3:-3}
for i in $(seq $REPS); do
  for L in $A $B; do
    HIKARI_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --passes --steps 48 --warmup 8 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); p = d.get('pass_ms', {})
        print('$L', 'ms/frame', d['ms_per_step'], 'sum_passes', round(sum(p.values()), 4), ' '.join(f'{k[:9]}={v:.3f}' for k, v in p.items()))
"
  done
done
