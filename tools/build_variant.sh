#!/bin/bash
# Build a variant of libhikari_hip.so with extra -D flags into build_ab/<name>.so:  tools/build_variant.sh <name> [-DFLAG ...]
NAME=$1; shift
cd "$(dirname "$0")/../bevy-hikari_amd/csrc" || exit 1
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function "$@" -o ../../build_ab/$NAME.so \
  kernels.hip kernels_denoise.hip kernels_aa.hip kernels_wavefront.hip kernels_scene.hip context.hip scene_layout.hip scene_refit.hip probes.hip host_logic.cpp scene_builder.cpp comm.cpp
