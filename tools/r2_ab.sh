#!/bin/bash
# same-box A/B of library builds: bench.py --config $CFG (default c3), variants alternating, $REPS rounds (args: variant names under
# akari_render_amd/variants, "product" = the shipped library). Different boxes differ by +-2 %: only same-run numbers compare.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_ab; mkdir -p $OUT
for R in $(seq 1 ${REPS:-3}); do
for V in "$@"; do
  if [ $V = product ]; then unset AKR_HIP_LIB; else export AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_$V.so; fi
  timeout 400 python bench.py --config ${CFG:-c3} --steps 2 --warmup 1 --also none --no-cpu-baseline > $OUT/$V.json 2> $OUT/$V.err
  echo "round $R $V ${CFG:-c3} $(python -c "import json;d=json.load(open('$OUT/$V.json'));print(round(d['value'],1))" 2>&1)"
done
done
