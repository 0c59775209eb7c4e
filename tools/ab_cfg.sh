#!/bin/bash
# A/B on the non-headline configs: tools/ab_cfg.sh "<configs>" lib1 lib2 ...
CFG=$1; shift
for L in "$@"; do
  echo "== $L"; HIKARI_HIP_LIB=$PWD/$L python tools/config_probe.py $CFG 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(d['config'][:30], 'ms', d['ms_per_frame'], 'Mray/s', d['mray_per_s'], 'indirect', d['pass_ms'].get('indirect_lit_ambient'))
"
done
