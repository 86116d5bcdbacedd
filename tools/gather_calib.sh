#!/bin/bash
# Calibration of rocprofv3's HBM-side read counters on a KNOWN byte count, per access pattern (tools/micro/gather_bw.hip):
# a coalesced stream (the guide's calibrated case), random 64-byte records read with four / two / one 16-byte loads per lane
# (= the BVH traversal's node and triangle fetches), random 128-byte pairs. One --pmc set per rocprofv3 pass, kernel-trace only.
# Output: gpurun_out/r4_gather_calib/summary.json (copy to profiles/r4_fetch_size_calibration.json).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_gather_calib
rm -rf $OUT; mkdir -p $OUT
BIN=$OUT/gather_bw
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/gather_bw.hip -o $BIN || exit 1
MIB=${MIB:-4096}
for PAT in stream gather64 gather128 gather32 gather16; do
  $BIN $PAT $MIB > $OUT/$PAT.plain.json 2> $OUT/$PAT.plain.err
  echo "plain $PAT: $(cat $OUT/$PAT.plain.json)"
  i=0
  for SET in \
    "FETCH_SIZE" \
    "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" \
    "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
    "TCC_EA0_RDREQ_DRAM_sum TCC_READ_SECTORS_sum" \
    "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT -o ${PAT}_set$i -- $BIN $PAT $MIB > $OUT/${PAT}_set$i.out 2> $OUT/${PAT}_set$i.err
    echo "  $PAT set$i rc=$? : $SET"
  done
done
python - <<PY
import csv, glob, collections, json, os
out = "$OUT"
summary = {"source": "tools/gather_calib.sh: tools/micro/gather_bw.hip under rocprofv3 --kernel-trace --pmc <one set per pass>; counters are summed over the "
                     "two dispatches of a run and divided by 2; requested = bytes the lanes' loads name, line64 / line128 = distinct-record bytes if a miss fetches a 64- / 128-byte line",
           "patterns": {}}
for pat in ("stream", "gather64", "gather128", "gather32", "gather16"):
    try:
        plain = json.loads(open(f"{out}/{pat}.plain.json").read().strip().splitlines()[-1])
    except Exception as e:
        summary["patterns"][pat] = {"error": repr(e)}; continue
    c = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in sorted(glob.glob(f"{out}/**/{pat}_set*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            if "k_gather" not in row["Kernel_Name"] and "k_stream" not in row["Kernel_Name"]: continue
            c[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
    per = {k: v / max(1, len(n[k])) for k, v in c.items()}
    req = plain["requested_bytes_per_launch"]
    recs = plain["lanes"] * plain["records_per_lane"]
    s = {"plain_run": plain, "counters_per_launch": per, "requested_bytes": req, "records": recs}
    if "FETCH_SIZE" in per:
        s["fetch_size_bytes"] = per["FETCH_SIZE"] * 1024.0
        s["requested_over_fetch_size"] = req / (per["FETCH_SIZE"] * 1024.0)
        s["fetch_size_bytes_per_record"] = per["FETCH_SIZE"] * 1024.0 / recs
    for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_BUBBLE_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_READ_SECTORS_sum", "TCC_READ_sum", "TCP_TCC_READ_REQ_sum"):
        if k in per: s[k.replace("_sum", "") + "_per_record"] = per[k] / recs
    summary["patterns"][pat] = s
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
for pat, s in summary["patterns"].items():
    print(pat, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items() if k not in ("plain_run", "counters_per_launch")}, "GB/s", s.get("plain_run", {}).get("requested_gbs"))
PY
