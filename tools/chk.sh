#!/bin/bash
# build the library + device asm and print register / spill statistics (developer helper)
set -e
cd /root/repo
make -C vi_ekf_amd/csrc 2>&1 | grep -E "error" || true
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function -S --cuda-device-only -o /tmp/asm/capi.s vi_ekf_amd/csrc/viekf_capi.hip 2>&1 | grep error || true
python tools/asm_stats.py /tmp/asm/capi.s | grep -E "tile|resident"
python tools/asm_regions.py /tmp/asm/capi.s _ZN5viekf15k_step_resident | awk '{ if ($6>0) print }'
if [ "$1" == "stamps" ]; then
  (cd vi_ekf_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -DVIEKF_STAMPS -shared -o ../../scratch_dbg/lib_stamps.so viekf_capi.hip viekf_yaml.cpp 2>&1 | grep -E "error" || true)
fi
