#!/bin/bash
# round 5, first GPU call: the sequence / full-size default-mode parity tests + this round's starting per-pass lines
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_default_mode_sequence_gpu.py tests/test_parity_gpu.py -k "sequence or default_mode or threaded_traversal_config" -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_call1_pytest.txt
cat gpurun_out/r05_call1_pytest.txt
bash tools/ab_variants.sh "2 3 4" base 2>&1 | tee gpurun_out/r05_start_ab.txt
