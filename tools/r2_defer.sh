#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2l; mkdir -p $OUT
export TMPDIR=/tmp
for Q in 1 3 2 7; do
  ( AKR_PT_DEFER_METAL=$Q timeout 400 python bench.py --config c3 --steps 2 --warmup 1 --also none --no-cpu-baseline ) > $OUT/c3_d$Q.json 2> $OUT/c3_d$Q.err
  echo "defer=$Q rc=$? $(python -c "import json;d=json.load(open('$OUT/c3_d$Q.json'));print(round(d['value'],1),'Msamples/s')" 2>&1)"
done
