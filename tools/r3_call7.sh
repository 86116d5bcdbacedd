#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=tools/r3_batch.sh
O=gpurun_out/r3
$T tests product
for R in 1 2; do
for S in 1 0; do
  for CFG in c3 c4; do
    AKR_PT_SIMPLE=$S timeout 600 python bench.py --config $CFG --steps 2 --warmup 1 --also none --no-cpu-baseline > $O/bench_${CFG}_simple$S.json 2> $O/bench_${CFG}_simple$S.err
    echo "bench $CFG round $R simple=$S $(python -c "import json;d=json.load(open('$O/bench_${CFG}_simple$S.json'));print(round(d['value'],1))" 2>&1 | tail -1)"
  done
done
done
