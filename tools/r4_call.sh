#!/bin/bash
# Round-4 gpurun payload (rewritten per call; the reusable pieces are tools/pc_sample.sh, gather_calib.sh, r3_batch.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
for MODE in "8000 2100000" "4000 2200000 big" "6000 2300000 tex" "2500 2400000 wavefront" "4000 2500000 shard" "1500 2600000 gpt" "2000 2700000 aov" "800 2800000 mcmc"; do
  timeout 500 python tools/soak.py $MODE 2>&1 | grep -E "MISMATCH|cases from seed|rror" | tail -3 | sed "s/^/soak [$MODE] /"
done 2>&1 | tee $O/soak_final.txt
