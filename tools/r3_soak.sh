#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_soak; mkdir -p $O
for MODE in "8000 1000000" "3000 2000000 big" "3000 3000000 tex" "1500 4000000 wavefront" "1500 5000000 shard" "1500 6000000 gpt" "1000 7000000 aov" "500 8000000 mcmc" "1500 9000000 big tex"; do
  timeout 900 python tools/soak.py $MODE 2>&1 | grep -E "MISMATCH|cases from seed|rror" | tail -5 | sed "s/^/soak [$MODE] /"
done 2>&1 | tee $O/soak_big.txt
