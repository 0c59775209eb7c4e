#!/bin/bash
# dev loop on the GPU box: parity (fast subset) + per-pass timing
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "every_frame or 1080p" 2>&1 | tail -3
python bench.py --no-cpu-baseline --passes --steps 24 --warmup 8 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('Mray/s', d['value'], 'ms/frame', d['ms_per_step'], 'indirect ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])
        print({k: v for k, v in d.get('pass_ms', {}).items()})
    else:
        print(line, end='')
"
