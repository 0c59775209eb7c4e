#!/bin/bash
# Build a variant of libhikari_hip.so with extra -D flags into build_ab/<name>.so:  tools/build_variant.sh <name> [-DFLAG ...]
NAME=$1; shift
cd "$(dirname "$0")/.." || exit 1
mkdir -p build_ab
python tools/build_lib.py -o build_ab/$NAME.so --objdir build/obj_$NAME "$@"
