#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2g; mkdir -p $OUT
for V in product fullw3; do
  if [ $V = product ]; then unset AKR_HIP_LIB; else export AKR_HIP_LIB=$GRAFT_REPO_ROOT/akari_render_amd/variants/libakari_hip_$V.so; fi
  ( timeout 400 python bench.py --config c3 --steps 2 --warmup 1 --also none --no-cpu-baseline ) > $OUT/c3_$V.json 2> $OUT/c3_$V.err
  echo "$V rc=$? $(python -c "import json;d=json.load(open('$OUT/c3_$V.json'));print(round(d['value'],1),'Msamples/s')" 2>&1)"
done
