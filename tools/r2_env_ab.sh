#!/bin/bash
# same-box A/B of environment switches on bench.py --config $CFG (default c3): args are "NAME=VALUE" settings ("-" = none), $REPS rounds
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_env_ab; mkdir -p $OUT
for R in $(seq 1 ${REPS:-2}); do
for V in "$@"; do
  ( [ "$V" != "-" ] && export "$V"
    timeout 400 python bench.py --config ${CFG:-c3} --steps 2 --warmup 1 --also none --no-cpu-baseline > $OUT/run.json 2> $OUT/run.err
    echo "round $R $V ${CFG:-c3} $(python -c "import json;d=json.load(open('$OUT/run.json'));print(round(d['value'],1))" 2>&1)" )
done
done
