#!/bin/bash
# Collects PMC counters for the bench kernel in separate rocprofv3 passes (one --pmc set per run, no tracing domains
# other than kernel-trace). Usage: tools/pmc_run.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="${@:---steps 16 --warmup 2 --no-cpu-baseline}"
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT -o set$i -- python bench.py $ARGS > $OUT/set$i.out 2> $OUT/set$i.err
  echo "set$i rc=$? : $SET"
done
python - <<PY
import csv, glob, collections, json, os
out = "$OUT"
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
last = {}  # counter -> (dispatch id, value summed over the rows of that dispatch) of the LAST k_pt_pass dispatch
per_disp = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
        if "k_pt_pass" in k:
            per_disp[(f, row["Counter_Name"], k)][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
summ = {}
for k in res:
    if "k_pt_pass" not in k: continue
    summ[k] = {c: {"sum": v, "dispatches": cnt[(k, c)]} for c, v in res[k].items()}
# HBM traffic of the timed (last) launch, MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB and come from
# separate passes; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 B, so the read side is doubled.
traffic = {}
for (f, c, k), d in per_disp.items():
    if c in ("FETCH_SIZE", "WRITE_SIZE"):
        did = max(d)
        traffic.setdefault(k, {})[c + "_KiB_last_launch"] = d[did]
for k, t in traffic.items():
    if "FETCH_SIZE_KiB_last_launch" in t and "WRITE_SIZE_KiB_last_launch" in t:
        t["hbm_bytes_last_launch"] = (2.0 * t["FETCH_SIZE_KiB_last_launch"] + t["WRITE_SIZE_KiB_last_launch"]) * 1024.0
        t["bench_args"] = "$ARGS"
summ["_traffic"] = traffic
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summ, indent=1))
PY
