#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_refit.py -q -x > $OUT/c7_refit_pytest.log 2>&1; tail -30 $OUT/c7_refit_pytest.log | cut -c1-300
timeout 600 python tools/refit_probe.py 2000 20000 > $OUT/c7_refit_probe.json 2> $OUT/c7_refit_probe.err; cat $OUT/c7_refit_probe.json; tail -3 $OUT/c7_refit_probe.err
