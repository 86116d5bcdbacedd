#!/bin/bash
# round 2, first GPU call: the new full-size parity tests, the new bench line (C2 + C3 + C4 + CPU baseline), kernel trace
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2a; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
( time timeout 600 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -c 600 $OUT/bench_default.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$OUT/prof_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --also none --no-cpu-baseline ) > $OUT/prof_c2.out 2> $OUT/prof_c2.err
echo "prof rc=$?"
find $OUT/prof_c2 -name "*kernel_stats.csv" | head
