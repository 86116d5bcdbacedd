#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_tex2; mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 600 python tools/tex_probe.py > $OUT/probe.log 2> $OUT/probe.err; cat $OUT/probe.log | head -12; tail -2 $OUT/probe.err
