#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2c; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_probes.py tests/test_gpu_textures.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for V in product unmerged; do
  if [ $V = product ]; then unset AKR_HIP_LIB; else export AKR_HIP_LIB=$GRAFT_REPO_ROOT/akari_render_amd/variants/libakari_hip_$V.so; fi
  ( timeout 400 python bench.py --config c4 --steps 1 --warmup 0 --also none --no-cpu-baseline ) > $OUT/c4_$V.json 2> $OUT/c4_$V.err
  echo "$V rc=$? $(python -c "import json;d=json.load(open('$OUT/c4_$V.json'));c=d['counters'];r=c['n_closest']+c['n_shadow'];print(round(d['value'],1),'Msamples/s', round(c['n_node_visits']/r,2),'nodes/ray',round(c['n_tri_tests']/r,2),'tris/ray')" 2>&1)"
done
