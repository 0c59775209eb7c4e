#!/bin/bash
# interleaved comparison of N builds: tools/abn.sh reps lib1 lib2 ...
REPS=$1; shift
for i in $(seq $REPS); do
  for L in "$@"; do
    HIKARI_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --passes --steps 48 --warmup 8 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); p = d.get('pass_ms', {})
        print('$L', 'ms/frame', d['ms_per_step'], ' '.join(f'{k[:9]}={v:.3f}' for k, v in p.items() if k[:3] in ('pre','dir','ind')))
"
  done
done
