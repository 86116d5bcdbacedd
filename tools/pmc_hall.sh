#!/bin/bash
# PMC passes for the 10M-triangle hall (BASELINE configs[3]); one counter set per rocprofv3 run
set -u
TAG=${1:-hall}; N=${2:-1e7}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" \
  "GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT -o set$i -- python tools/hall_bench.py $N 8 > $OUT/set$i.out 2> $OUT/set$i.err
  echo "set$i rc=$? : $SET"
done
python - <<PY
import csv, glob, collections, json
out = "$OUT"
res = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
summ = {k: {c: {"sum": v, "dispatches": cnt[(k, c)]} for c, v in res[k].items()} for k in res if "k_pt_pass" in k}
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
for k, v in summ.items():
    print(k); [print("  ", c, "%.4g" % x["sum"], x["dispatches"]) for c, x in sorted(v.items())]
print(open(out + "/set1.out").read()[:600])
PY
