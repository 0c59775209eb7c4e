#!/bin/bash
# Lane utilisation (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)) and VALU instructions of every kernel of a config's frames.
# Usage: tools/pmc_lanes.sh <config> [tag]
C=${1:-4}; TAG=${2:-lanes}
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $OLDPWD/bench.py --config $C --steps 3 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe --no-extra-configs --sustained-seconds 0"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/prof_lanes_$C -- $CMD > /dev/null 2>&1
cd $OLDPWD
DB=$(find $OUT/prof_lanes_$C -name "*.db" | head -1)
[ -n "$DB" ] && python tools/pmc_summary.py $DB > $OUT/${TAG}_config${C}_pmc_sq.txt
rm -rf $OUT/prof_lanes_$C
python - <<PY
import re
k=None; d={}
for ln in open("$OUT/${TAG}_config${C}_pmc_sq.txt"):
    if not ln.startswith("    "): k=ln.strip()[:70]; d[k]={}
    else:
        p=ln.split(); d[k][p[0]]=float(p[2])
for k,v in d.items():
    if "SQ_ACTIVE_INST_VALU" in v and v["SQ_ACTIVE_INST_VALU"]>0:
        print(f"{k:70s} util {v['SQ_THREAD_CYCLES_VALU']/(64*v['SQ_ACTIVE_INST_VALU']):.3f} valu {v['SQ_INSTS_VALU']/1e6:8.1f} M  waves {v['SQ_WAVES']:.0f}")
PY
