#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2d; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_multigpu.py tests/test_gpu_parity.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for V in product dual w5 dualw5; do
  if [ $V = product ]; then unset AKR_HIP_LIB; else export AKR_HIP_LIB=$GRAFT_REPO_ROOT/akari_render_amd/variants/libakari_hip_$V.so; fi
  ( timeout 400 python bench.py --config c4 --steps 1 --warmup 0 --also none --no-cpu-baseline ) > $OUT/c4_$V.json 2> $OUT/c4_$V.err
  echo "$V rc=$? $(python -c "import json;d=json.load(open('$OUT/c4_$V.json'));c=d['counters'];r=c['n_closest']+c['n_shadow'];print(round(d['value'],1),'Msamples/s', round(c['n_node_visits']/r,2),'nodes/ray',round(c['n_tri_tests']/r,2),'tris/ray')" 2>&1)"
done
unset AKR_HIP_LIB
# the N = 2 code path of bench.py on one GPU (gloo, both ranks on device 0): strong scaling, tile shards, film reduce
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --backend gloo --also none ) > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo "2-rank gloo rc=$?"; tail -c 700 $OUT/bench_2rank_gloo.json
