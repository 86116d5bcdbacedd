#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2i; mkdir -p $OUT
export AKR_HIP_LIB=$GRAFT_REPO_ROOT/akari_render_amd/variants/libakari_hip_cheapfr.so
for Q in 1 0; do
  ( AKR_PT_LOBE_QUEUE=$Q timeout 400 python bench.py --config c3 --steps 2 --warmup 1 --also none --no-cpu-baseline ) > $OUT/c3_cheap_q$Q.json 2> $OUT/c3_cheap_q$Q.err
  echo "cheap fresnel queue=$Q rc=$? $(python -c "import json;d=json.load(open('$OUT/c3_cheap_q$Q.json'));print(round(d['value'],1),'Msamples/s')" 2>&1)"
done
