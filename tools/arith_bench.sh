#!/bin/bash
# The two arithmetic tiers of the pt megakernel side by side on one box: C2 / C3 / C4 with option arith = 0 (the AKR-F32 contract) and 1 (relaxed).
# bash tools/arith_bench.sh <out_dir> [configs]      (needs a GPU)
out=${1:-gpurun_out/arith}; mkdir -p "$out"
for c in ${2:-c2 c3 c4}; do
  for a in 0 1; do
    AKR_ARITH=$a python bench.py --config $c --steps 4 --warmup 1 --also none --no-cpu-baseline > "$out/${c}_arith$a.json" 2> "$out/${c}_arith$a.err"
    python - "$out/${c}_arith$a.json" $c $a <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]} arith={sys.argv[3]}: {d['value']:.1f} {d['unit']}  ({d['ms_per_step']:.1f} ms / step)", flush=True)
PY
  done
done
