#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_device_refit.py -q > $OUT/c12_refit_pytest.log 2>&1; tail -25 $OUT/c12_refit_pytest.log | cut -c1-300
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/c12_pytest.log 2>&1; tail -4 $OUT/c12_pytest.log
python -c "
import __graft_entry__ as g
g.smoke()
"
