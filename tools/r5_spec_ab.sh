#!/bin/bash
# A/B of device-side switches of the per-scene kernels on the textured room (exhaustive NF=1, BVH NF=8). Output gpurun_out/r5b/
O=gpurun_out/r5b; mkdir -p $O
export AKR_KERNEL_CACHE=/tmp/akr_cache_r5b
timeout 300 python -m pytest tests/test_gpu_specialise.py -x -q -k "cache" > $O/pytest_cache.txt 2>&1; tail -30 $O/pytest_cache.txt | grep -E "^E|^>|passed|failed"
run() {  # label, extra flags, defer_on, waves-variant-name
  for NF in 1 8; do
    AKR_SPEC_EXTRA_FLAGS="$2" TEXBENCH_DEFER_ON="$3" TEXBENCH_ONLY="$4" timeout 300 python tools/textured_bench.py 4 $NF > $O/t.json 2>> $O/err.txt
    python -c "
import json; d = json.load(open('$O/t.json'))
for k, v in d.items(): print('$1 | nfloor=$NF |', round(v['msamples_per_s'], 1), '| vgprs', v['kernel']['vgprs'], 'scratch', v['kernel']['scratch_bytes'])"
  done
}
P="textured, per-scene kernel"
run "baseline"            ""                        "" "$P"
run "baseline again"      ""                        "" "$P"
run "lean"                "-DAKR_TEX_LEAN=1"        "" "$P"
run "no park"             "-DAKR_PT_PARK_TEX=0"     "" "$P"
run "lean + no park"      "-DAKR_TEX_LEAN=1 -DAKR_PT_PARK_TEX=0" "" "$P"
run "defer_on 2"          ""                        "2" "$P"
run "defer_on 3"          ""                        "3" "$P"
run "4 waves lean"        "-DAKR_TEX_LEAN=1"        "" "textured, per-scene kernel, 4 waves"
run "no unroll walk"      "-DAKR_WALK_FULL_UNROLL=0" "" "$P"
