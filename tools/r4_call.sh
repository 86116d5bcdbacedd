#!/bin/bash
# Round-4 gpurun payload (rewritten per call; the reusable pieces are tools/pc_sample.sh, gather_calib.sh, r3_batch.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
REPS=2 bash tools/r3_batch.sh bench c2 product walk4
REPS=2 bash tools/r3_batch.sh bench c3 product walk4
NFLOOR=1 bash tools/r3_batch.sh tex product walk4
PYTEST_ARGS="--deselect tests/test_gpu_multigpu.py::test_bench_control_flow_with_eight_ranks_on_one_gpu" bash tools/r3_batch.sh tests walk4
tail -3 gpurun_out/r3/gputest_walk4.log
