#!/bin/bash
# PMC counters of the k_pt_pass / k_wf_* kernels for an arbitrary command (separate rocprofv3 passes, kernel-trace only).
# Usage: tools/pmc_cmd.sh <tag> <command...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT -o set$i -- "$@" > $OUT/set$i.out 2> $OUT/set$i.err
  echo "set$i rc=$? : $SET"
done
python - <<PY
import csv, glob, collections, json
out = "$OUT"
res = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_pt_pass" in k or "k_wf_" in k:
            res[k][row["Counter_Name"]] += float(row["Counter_Value"])
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
