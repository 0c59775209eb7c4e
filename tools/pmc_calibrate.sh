#!/bin/bash
# FETCH_SIZE on gathers of KNOWN byte counts (MI355X_MICROARCH.md: "other access widths uncalibrated"):
# tools/gather_probe.py --calibrate launches k_gather_chase<1|2|4> (16 / 32 / 64-B records, one miss per lane and step over 2 GiB);
# this collects FETCH_SIZE per dispatch in its own pass.  Usage (on the GPU box): tools/pmc_calibrate.sh <tag>
TAG=${1:-r04}
OUT=$PWD/gpurun_out; mkdir -p $OUT
HERE=$PWD
python tools/gather_probe.py --calibrate > $OUT/${TAG}_fetch_calibration_known.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_cal -- python $HERE/tools/gather_probe.py --calibrate > /dev/null 2>&1
cd $HERE
DB=$(find $OUT/prof_cal -name "*.db" | head -1)
[ -n "$DB" ] && python tools/pmc_summary.py $DB > $OUT/${TAG}_fetch_calibration_pmc.txt
grep -A3 "k_gather_chase" $OUT/${TAG}_fetch_calibration_pmc.txt
rm -rf $OUT/prof_cal
