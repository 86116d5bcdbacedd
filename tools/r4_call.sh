#!/bin/bash
# Round-4 gpurun payload (rewritten per call; the reusable pieces are tools/pc_sample.sh, gather_calib.sh, r3_batch.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
REPS=2 bash tools/r3_batch.sh bench c3 product os o2
bash tools/r3_batch.sh bench c2 product os o2
