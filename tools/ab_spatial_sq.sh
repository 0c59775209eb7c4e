#!/bin/bash
# SQ counters of k_spatial_reuse per variant (base = shipped library, others = build_ab/<name>.so): lane utilisation and VALU wave-instructions
# per launch, configs 2 and 5.   Usage: tools/ab_spatial_sq.sh base compact ...
for cfg in 2 5; do
  for v in "$@"; do
    if [ $v = base ]; then unset HIKARI_HIP_LIB; else export HIKARI_HIP_LIB=$PWD/build_ab/$v.so; fi
    echo "config $cfg $v"
    bash tools/pmc_lanes.sh $cfg sq_$v | grep -i "spatial"
  done
done
