#!/bin/bash
# Round-4 gpurun payload (rewritten per call; the reusable pieces are tools/pc_sample.sh, gather_calib.sh, r3_batch.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NFLOOR=1 bash tools/r3_batch.sh tex product texu 2>&1 | grep "\[textured\]"
NFLOOR=8 bash tools/r3_batch.sh tex product texu 2>&1 | grep "\[textured\]"
export AKR_DATA_DIR=$PWD/akari_render_amd/data AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_texu.so
timeout 600 python -m pytest tests/test_gpu_textures.py -x -q -m gpu 2>&1 | tail -3
