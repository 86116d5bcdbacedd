#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2e; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
