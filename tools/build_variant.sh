#!/bin/bash
# builds the library of the current tree into variants/lib_<name>.so (A/B timing of two variants on ONE gpurun box:
# VIEKF_LIB=variants/lib_<name>.so python bench.py ...); extra compiler flags as further arguments, e.g.
#   tools/build_variant.sh stamps -DVIEKF_STAMPS        tools/build_variant.sh ablate -DVIEKF_ABLATE
set -e
name=$1; shift
cd "$(dirname "$0")/../vi_ekf_amd/csrc"
mkdir -p ../../variants
make -s -j8 OUT=../../variants/lib_$name.so OBJDIR=../../build/obj_$name EXTRA="$*"
ls -la ../../variants/lib_$name.so
