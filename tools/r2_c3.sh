#!/bin/bash
# parity subset + C3 / textured probes under environment switches (args: "NAME=VALUE" or "-" for none)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_c3; mkdir -p $OUT
( timeout 1200 python -m pytest ${PYTEST_FILES:-tests/test_gpu_parity.py tests/test_gpu_textures.py tests/test_gpu_aov.py tests/test_gpu_probes.py} -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for V in "$@"; do
  ( [ "$V" != "-" ] && export "$V"; timeout 400 python bench.py --config c3 --steps 2 --warmup 1 --also none --no-cpu-baseline > $OUT/c3.json 2> $OUT/c3.err
    echo "$V c3 $(python -c "import json;d=json.load(open('$OUT/c3.json'));print(round(d['value'],1))" 2>&1)"
    TEXBENCH_ONLY=textured timeout 300 python tools/textured_bench.py 3 2>/dev/null | python -c "
import json,sys
for k,v in json.load(sys.stdin).items(): print('   ', round(v['msamples_per_s'],1),k)" )
done
