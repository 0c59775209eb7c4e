#!/bin/bash
# Parity subset + per-pass A/B of k_spatial_reuse variants (build_ab/<name>.so).   Usage: tools/ab_spatial.sh "<configs>" <variant> ...
CONFIGS=$1; shift
for v in "$@"; do
  HIKARI_HIP_LIB=$PWD/build_ab/$v.so timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "cornell or yard or spatial or tiny or background" 2>&1 | tail -1
done
bash tools/ab_variants.sh "2" base "$@" base "$@"
[ -n "$CONFIGS" ] && bash tools/ab_variants.sh "$CONFIGS" base "$@"
