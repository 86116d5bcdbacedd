#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2h; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_colorspace.py tests/test_gpu_sobol.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not hall" ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for Q in 1 0; do
  ( AKR_PT_LOBE_QUEUE=$Q timeout 400 python bench.py --config c3 --steps 2 --warmup 1 --also none --no-cpu-baseline ) > $OUT/c3_q$Q.json 2> $OUT/c3_q$Q.err
  echo "queue=$Q rc=$? $(python -c "import json;d=json.load(open('$OUT/c3_q$Q.json'));print(round(d['value'],1),'Msamples/s')" 2>&1)"
done
( timeout 400 python bench.py --config c2 --steps 3 --warmup 1 --also none --no-cpu-baseline ) > $OUT/c2.json 2> $OUT/c2.err
echo "c2 rc=$? $(python -c "import json;d=json.load(open('$OUT/c2.json'));print(round(d['value'],1),'Msamples/s')" 2>&1)"
