OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python tools/cohort_probe.py 3 4 > $OUT/r04_cohort_probe.json 2> $OUT/r04_cohort_probe.err; tail -3 $OUT/r04_cohort_probe.err; cat $OUT/r04_cohort_probe.json
