#!/bin/bash
cd "$GRAFT_REPO_ROOT"
T=tools/r3_batch.sh
PYTEST_ARGS="--deselect tests/test_gpu_fullsize.py" $T tests pflfd pflfull
REPS=2 $T bench c2 product pflfd
REPS=2 $T bench c3 product pflfull
REPS=1 $T bench c4 product pflfull
STEPS=8 $T shard fd product pflfd
STEPS=8 $T shard full product pflfull
