#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_refit.py -q > $OUT/c5_refit_pytest.log 2>&1; tail -5 $OUT/c5_refit_pytest.log | cut -c1-250
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/c5_pytest.log 2>&1; tail -4 $OUT/c5_pytest.log
