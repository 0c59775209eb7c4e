OUT=$PWD/gpurun_out; mkdir -p $OUT
HERE=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python $HERE/bench.py --steps 6 --warmup 40 --blocks 1 --no-cpu-baseline --no-hbm-probe --no-extra-configs --sustained-seconds 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_tl -- $CMD > /dev/null 2>&1
DB=$(find $OUT/prof_tl -name "*.db" | head -1)
cd $HERE
python tools/timeline.py $DB 3 > $OUT/r04k_timeline.txt
tail -60 $OUT/r04k_timeline.txt
rm -rf $OUT/prof_tl
