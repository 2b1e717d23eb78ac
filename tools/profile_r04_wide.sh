#!/bin/bash
# the wide-P part of tools/profile_r04.sh alone (after the pass of k_update_feat_panelsvc changed) + the default bench line
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof4
mkdir -p $OUT
rm -rf $OUT/stream_B1024_N150_*
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tag=stream_B1024_N150; args="--feat 150"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_stats -- python3 bench.py $args --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_fetch -- python3 bench.py $args --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-secondary > $OUT/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_write -- python3 bench.py $args --steps 10 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-secondary > $OUT/${tag}_write.log 2>&1
echo "$tag done"
