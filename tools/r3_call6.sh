#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3/shardpmc; rm -rf $O; mkdir -p $O
i=0
for W in 8 1; do
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -f csv -d $O -o w${W}_set$i -- python tools/shard_one.py 0 $W 8 > $O/w${W}_set$i.out 2> $O/w${W}_set$i.err
  echo "world $W set$i rc=$?"
done
done
python - <<'PY'
import csv, glob, collections
for W in (8, 1):
    res = collections.defaultdict(float)
    for f in sorted(glob.glob(f"gpurun_out/r3/shardpmc/**/w{W}_set*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if "k_pt_pass" in row["Kernel_Name"]:
                res[row["Counter_Name"]] += float(row["Counter_Value"])
    print("world", W, {k: f"{v:.4g}" for k, v in sorted(res.items())})
PY
