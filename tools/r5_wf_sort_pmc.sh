#!/bin/bash
# L2 hits / misses and fabric reads of k_wf_trace with and without sorted ray queues (one --pmc set per pass, kernel trace only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i
for S in 0 1; do
  i=0
  for SET in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $SET -f csv -d $O -o pmc${S}_$i -- python $O/run.py $S > $O/pmc${S}_$i.out 2> $O/pmc${S}_$i.err
  done
  python - <<PY
import csv, glob, collections
res = collections.defaultdict(float)
for f in glob.glob("$O/**/pmc${S}_*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_wf_trace" in row["Kernel_Name"]:
            res[row["Counter_Name"]] += float(row["Counter_Value"])
r = dict(res)
print("sort=$S", {k: "%.4g" % v for k, v in r.items()})
print("   L2 hit %.3f  misses %.3g  fetch GB %.1f  wait share %.3f  lane utilisation %.3f" % (r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"]), r["TCC_MISS_sum"],
      r["FETCH_SIZE"] * 1024 / 1e9, r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"], r["SQ_THREAD_CYCLES_VALU"] / (64 * r["SQ_ACTIVE_INST_VALU"])))
PY
done
