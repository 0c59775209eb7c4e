OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $OLDPWD/bench.py --steps 6 --warmup 4 --blocks 1 --no-cpu-baseline --no-hbm-probe --no-extra-configs --sustained-seconds 0"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/prof_sq -- $CMD > /dev/null 2>&1
cd $OLDPWD
DB=$(find $OUT/prof_sq -name "*.db" | head -1)
python tools/pmc_summary.py $DB > $OUT/r03_mid_pmc_sq.txt
rm -rf $OUT/prof_sq
python - <<'PY'
import re
txt=open('gpurun_out/r03_mid_pmc_sq.txt').read()
blocks=re.split(r'\n(?=\S)', txt)
for b in blocks:
    lines=b.strip().splitlines()
    if not lines or 'k_' not in lines[0]: continue
    vals={}
    for l in lines[1:]:
        p=l.split()
        vals[p[0]]=(float(p[2]), int(p[3].split('=')[1]))
    if 'SQ_INSTS_VALU' in vals:
        v=vals['SQ_INSTS_VALU'][0]; n=vals['SQ_INSTS_VALU'][1]
        lu=vals['SQ_THREAD_CYCLES_VALU'][0]/(64*vals['SQ_ACTIVE_INST_VALU'][0]) if vals.get('SQ_ACTIVE_INST_VALU',(0,))[0] else 0
        print(f"{lines[0][:70]:70s} n={n:3d} VALU={v/1e6:8.1f}M  waves={vals['SQ_WAVES'][0]:9.0f} per_wave={v/max(vals['SQ_WAVES'][0],1):8.0f} lane_util={lu:.3f}")
PY
