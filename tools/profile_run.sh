#!/bin/bash
# Runs on the GPU box (gpurun): the three rocprofv3 passes behind profiles/<round>/resident_*.  Counters are collected in
# their own passes (never together with a trace domain other than the kernel trace).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq3 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/sq3.log 2>&1 || echo "sq3 pass failed (counter names)"
find $OUT -name "*.csv" | head -20
