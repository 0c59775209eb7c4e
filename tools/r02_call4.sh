#!/bin/bash
# Round 2, GPU call 4: device refit parity + probe; full GPU suite
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_refit.py -q > $OUT/c4_refit_pytest.log 2>&1; tail -25 $OUT/c4_refit_pytest.log | cut -c1-250
timeout 600 python tools/refit_probe.py 2000 20000 > $OUT/c4_refit_probe.json 2> $OUT/c4_refit_probe.err; cat $OUT/c4_refit_probe.json; tail -3 $OUT/c4_refit_probe.err
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/c4_pytest.log 2>&1; tail -4 $OUT/c4_pytest.log
