#!/bin/bash
# Every PMC summary the bench line quotes, on the sources as they are: tools/pmc_bench.sh over the three timed configurations and the
# secondary legs, summaries collected under gpurun_out/pmc_all/r6_pmc_<name>.json (copy them to profiles/).  bash tools/pmc_all.sh [names...]
names=${@:-c2 c3 c4 reference_default forest_100k_kept forest_10k_kept forest_10k_flattened textured_exhaustive_interpreter textured_exhaustive_per_scene textured_bvh_interpreter textured_bvh_per_scene}
mkdir -p gpurun_out/pmc_all
for n in $names; do
  echo "== $n"
  bash tools/pmc_bench.sh $n > gpurun_out/pmc_all/$n.log 2>&1
  grep -E "^set[0-9]+ rc=" gpurun_out/pmc_all/$n.log | tr '\n' ' '; echo
  if [ -f gpurun_out/pmc_bench_$n/summary.json ]; then
    cp gpurun_out/pmc_bench_$n/summary.json gpurun_out/pmc_all/r6_pmc_$n.json
    python -c "import json; d=json.load(open('gpurun_out/pmc_all/r6_pmc_$n.json')); print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k in ('kernel', 'hbm_bytes_per_sample', 'valu_busy', 'valu_lane_utilisation', 'wait_share', 'l2_hit', 'value_under_profiler_msamples_s')})"
  else
    tail -5 gpurun_out/pmc_all/$n.log
  fi
  rm -rf gpurun_out/pmc_bench_$n/*/  # (the raw counter CSVs: tens of MB)
done
