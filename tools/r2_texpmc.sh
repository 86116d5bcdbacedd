#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/texpmc; rm -rf $OUT; mkdir -p $OUT
for V in "textured" "same room, constant materials with the lobes the graphs select"; do
  tag=$(echo "$V" | cut -c1-4)
  i=0
  for SET in \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
    "GRBM_GUI_ACTIVE" "SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" ; do
    i=$((i+1))
    TEXBENCH_ONLY="$V" timeout 300 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT -o ${tag}_set$i -- python tools/textured_bench.py 1 > $OUT/${tag}_set$i.out 2> $OUT/${tag}_set$i.err
    echo "$tag set$i rc=$?"
  done
done
python - <<PY
import csv, glob, collections, json
for tag in ("text", "same"):
    res = collections.defaultdict(float)
    for f in sorted(glob.glob("$OUT/**/%s_set*counter_collection.csv" % tag, recursive=True)):
        for row in csv.DictReader(open(f)):
            if "k_pt_pass" in row["Kernel_Name"]: res[row["Counter_Name"]] += float(row["Counter_Value"])
    print(tag, json.dumps(dict(res)))
PY
