#!/bin/bash
cd "$GRAFT_REPO_ROOT"
T=tools/r3_batch.sh
$T tests product
$T tests comboA comboB
tools/micro/valu_rate > gpurun_out/r3/valu_rate.txt 2>&1; tail -4 gpurun_out/r3/valu_rate.txt
REPS=2 $T bench c2 product walk1 walk2 comboA comboB
REPS=2 $T bench c3 product walk1 walk2 lean leanpark comboA comboB
REPS=1 $T bench c4 product lean leanpark strag4 strag8 strag16 comboA comboB
$T shard fd product walk1 walk2
$T shard full product leanpark comboA comboB
