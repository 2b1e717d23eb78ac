#!/bin/bash
# rocprofv3 passes of the wide-P configuration (B=1024, N=150), see tools/profile_run.sh
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/profw
rm -rf $OUT; mkdir -p $OUT
ARGS="--feat 150 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary"
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head
