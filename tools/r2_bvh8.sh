#!/bin/bash
# round 2: the 8-wide compressed BVH -- parity suite, then C4 with the product build and the A/B variants
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2b; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for V in product node128 w3 w3n128; do
  if [ $V = product ]; then unset AKR_HIP_LIB; else export AKR_HIP_LIB=$GRAFT_REPO_ROOT/akari_render_amd/variants/libakari_hip_$V.so; fi
  ( timeout 400 python bench.py --config c4 --steps 1 --warmup 0 --also none --no-cpu-baseline ) > $OUT/c4_$V.json 2> $OUT/c4_$V.err
  echo "$V rc=$? $(python -c "import json;d=json.load(open('$OUT/c4_$V.json'));c=d['counters'];r=c['n_closest']+c['n_shadow'];print(round(d['value'],1),'Msamples/s', round(c['n_node_visits']/r,2),'nodes/ray',round(c['n_tri_tests']/r,2),'tris/ray', d['config'].get('compile_upload_s'))" 2>&1)"
done
unset AKR_HIP_LIB
( timeout 600 python bench.py --also c3 --no-cpu-baseline --steps 3 ) > $OUT/bench_c2c3.json 2> $OUT/bench_c2c3.err
echo "bench rc=$?"
