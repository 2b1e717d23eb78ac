#!/bin/bash
# extra rocprofv3 kernel statistics (see tools/profile_run.sh): the two-per-CU instance (N = 12) and the fused cadence
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profx
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n12 -- python3 bench.py --feat 12 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/n12.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cad -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/cad.log 2>&1
find $OUT -name "*kernel_stats.csv"
