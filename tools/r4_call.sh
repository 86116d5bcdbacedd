#!/bin/bash
# Round-4 gpurun payload (rewritten per call; the reusable pieces are tools/pc_sample.sh, gather_calib.sh, r3_batch.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEPS=8 REPS=2 bash tools/r3_batch.sh bench c2 product fuse32 fuse64
STEPS=8 bash tools/r3_batch.sh bench c3 product fuse32 fuse64
STEPS=4 bash tools/r3_batch.sh bench c4 product fuse64
