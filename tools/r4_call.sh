#!/bin/bash
# Round-4 gpurun payload (rewritten per call; the reusable pieces are tools/pc_sample.sh, gather_calib.sh, r3_batch.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
export AKR_DATA_DIR=$PWD/akari_render_amd/data
for NF in 1 8; do for V in product noeval nograph; do
  if [ $V = product ]; then unset AKR_HIP_LIB; else export AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_$V.so; fi
  TEXBENCH_ONLY="textured" timeout 600 python tools/textured_bench.py 4 $NF > $O/texdiag_${NF}_$V.json 2>/dev/null
  echo "textured room nfloor=$NF $V: $(python -c "import json;d=json.load(open('$O/texdiag_${NF}_$V.json'));print({k:round(v['msamples_per_s'],1) for k,v in d.items()})" 2>&1 | tail -1)"
done; done
unset AKR_HIP_LIB
for MODE in "3000 1100000" "1500 1200000 big" "2500 1300000 tex" "1000 1400000 wavefront" "1500 1500000 shard" "600 1600000 gpt" "800 1700000 aov" "300 1800000 mcmc"; do
  timeout 400 python tools/soak.py $MODE 2>&1 | grep -E "MISMATCH|cases from seed|rror" | tail -3 | sed "s/^/soak [$MODE] /"
done 2>&1 | tee $O/soak_big.txt
