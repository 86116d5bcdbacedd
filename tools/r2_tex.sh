#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2k; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_textures.py tests/test_gpu_colorspace.py tests/test_gpu_aov.py tests/test_gpt.py tests/test_mcmc.py tests/test_gpu_sobol.py tests/test_gpu_parity.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 300 python tools/textured_bench.py 4 > $OUT/textured.json 2> $OUT/textured.err; echo "textured rc=$?"; cat $OUT/textured.json
timeout 300 python tools/textured_bench.py 4 8 > $OUT/textured_bvh.json 2> $OUT/textured_bvh.err; echo "textured bvh rc=$?"; cat $OUT/textured_bvh.json
