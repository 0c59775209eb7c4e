#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_default_mode_sequence_gpu.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_call2_pytest.txt
cat gpurun_out/r05_call2_pytest.txt
bash tools/ab_variants.sh "3 4" base 2>&1 | tee gpurun_out/r05_reforder_ab.txt
