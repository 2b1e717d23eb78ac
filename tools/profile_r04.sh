#!/bin/bash
# Runs on the GPU box (gpurun): the rocprofv3 passes behind profiles/r04/*.  Counters are collected in their own passes (never together
# with a trace domain other than the kernel trace).  Three single-GPU BASELINE configs + the sequencer route:
#   resident_B1024_N50 (headline), resident_B256_N25 (configs[1]), stream_B1024_N150 (configs[4]), seq_cadence_B1024_N50
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof4
rm -rf $OUT; mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
run() {   # tag, bench args, steps, warmup
  local tag=$1 args=$2 k=$3 w=$4
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_stats -- python3 bench.py $args --steps $k --warmup $w --no-cpu-baseline --no-secondary > $OUT/${tag}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_fetch -- python3 bench.py $args --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-secondary > $OUT/${tag}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_write -- python3 bench.py $args --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-secondary > $OUT/${tag}_write.log 2>&1
  echo "$tag done"
}
sq() {    # tag, bench args
  local tag=$1 args=$2
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $OUT/${tag}_sq1 -- python3 bench.py $args --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-secondary > $OUT/${tag}_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${tag}_sq2 -- python3 bench.py $args --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-secondary > $OUT/${tag}_sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/${tag}_sq3 -- python3 bench.py $args --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-secondary > $OUT/${tag}_sq3.log 2>&1 || echo "$tag sq3 pass failed (counter names)"
  echo "$tag sq done"
}
run resident_B1024_N50 "" 50 5
sq resident_B1024_N50 ""
run resident_B256_N25 "--batch 256 --feat 25" 200 20
sq resident_B256_N25 "--batch 256 --feat 25"
run stream_B1024_N150 "--feat 150" 20 5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/seq_stats -- python3 tools/seq_bench.py 1024 50 0 40 > $OUT/seq_shared.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/seqi_stats -- python3 tools/seq_bench.py 1024 50 1 20 > $OUT/seq_indep.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -20
