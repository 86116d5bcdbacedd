#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; copies the per-kernel summary to profiles/<tag>_*.csv
set -u
TAG=${1:-r1}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="${@:---steps 16 --warmup 2 --no-cpu-baseline}"
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o $TAG -- python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$?"; ls $OUT
cat $OUT/bench.json
head -5 $OUT/*kernel_stats.csv
