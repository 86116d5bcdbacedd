#!/bin/bash
# textured-room throughput for A/B builds under akari_render_amd/variants (args: variant names; "product" = the shipped library)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_tex; mkdir -p $OUT
for NF in 1 8; do
for V in "$@"; do
  if [ $V = product ]; then unset AKR_HIP_LIB; else export AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_$V.so; fi
  TEXBENCH_NFLOOR=$NF TEXBENCH_ONLY=${TEXBENCH_ONLY:-textured} timeout 300 python tools/textured_bench.py 3 > $OUT/${V}_$NF.json 2> $OUT/${V}_$NF.err
  python -c "
import json
for k,v in json.load(open('$OUT/${V}_$NF.json')).items(): print('$V n_floor=$NF', round(v['msamples_per_s'],1), round(v['shaded_per_sample'],2), k)"
done
done
