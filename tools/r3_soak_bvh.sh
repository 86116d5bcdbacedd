#!/bin/bash
# randomised parity soak aimed at the BVH kernels (after the node format changed to 64 bytes / six children)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_soak; mkdir -p $O
for MODE in "4000 11000000 big" "2000 12000000 big tex" "1500 13000000 wavefront" "1000 14000000 shard big" "800 15000000 gpt big" "600 16000000 aov big" "300 17000000 mcmc big"; do
  timeout 600 python tools/soak.py $MODE 2>&1 | grep -E "MISMATCH|cases from seed|rror" | tail -5 | sed "s/^/soak [$MODE] /"
done 2>&1 | tee $O/soak_bvh64.txt
