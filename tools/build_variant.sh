#!/bin/bash
# builds the library of the current tree into variants/lib_<name>.so (A/B timing of two variants on ONE gpurun box:
# VIEKF_LIB=variants/lib_<name>.so python bench.py ...)
set -e
cd "$(dirname "$0")/../vi_ekf_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-but-set-variable -ffp-contract=fast -shared \
  -o ../../variants/lib_$1.so viekf_capi.hip viekf_yaml.cpp viekf_seq.cpp 2>&1 | grep -E "error" || true
ls -la ../../variants/lib_$1.so
