#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_wavefront_gpu.py tests/test_default_mode_sequence_gpu.py tests/test_device_refit.py -q -m gpu -x -k "wide or wavefront or config3 or config4 or sequence or default_mode or refit or sponza" 2>&1 | tail -8 | tee gpurun_out/r05_call9_pytest.txt
bash tools/ab_variants.sh "3 4" base q0 2>&1 | tee gpurun_out/r05_quantised_ab.txt
