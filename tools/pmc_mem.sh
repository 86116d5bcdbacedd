#!/bin/bash
# memory-pipeline counters (TA / TCP / TCC) for an arbitrary command: tools/pmc_mem.sh <tag> <command...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcmem_$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for SET in \
  "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
  "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
  "MemUnitStalled" ; do
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
