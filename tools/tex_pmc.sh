#!/bin/bash
# PMC passes over tools/textured_bench.py's room (textured, and the same room with constant materials that select the same
# lobes): one --pmc set per rocprofv3 run, kernel-trace only.  usage: tools/tex_pmc.sh [nfloor=1]   (1: exhaustive kernel, 8: tessellated
# floor, BVH kernel); summary -> gpurun_out/texpmc_<nfloor>/summary.json (-> profiles/r5_pmc_textured_room_{exhaustive,bvh}.json)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NF=${1:-1}
OUT=gpurun_out/texpmc_$NF; rm -rf $OUT; mkdir -p $OUT
export AKR_KERNEL_CACHE=${AKR_KERNEL_CACHE:-/tmp/akr_cache_pmc}
for V in "textured" "textured, per-scene kernel" "same room, constant materials with the lobes the graphs select"; do
  tag=$(echo "$V" | cut -c1-4); [ "$V" = "textured, per-scene kernel" ] && tag=spec
  i=0
  for SET in \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
    "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
    i=$((i+1))
    TEXBENCH_ONLY="$V" timeout 300 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT -o ${tag}_set$i -- python tools/textured_bench.py 3 $NF > $OUT/${tag}_set$i.out 2> $OUT/${tag}_set$i.err
    echo "$tag set$i rc=$?"
  done
done
python - <<PY
import csv, glob, collections, json
out = {}
for tag, name in (("text", "textured"), ("spec", "textured, per-scene kernel"), ("same", "same room, constant materials with the lobes the graphs select")):
    res = collections.defaultdict(float); kernel = None
    for f in sorted(glob.glob("$OUT/**/%s_set*counter_collection.csv" % tag, recursive=True)):
        for row in csv.DictReader(open(f)):
            if "k_pt_pass" in row["Kernel_Name"] or "akr_pt_pass_spec" in row["Kernel_Name"]:
                res[row["Counter_Name"]] += float(row["Counter_Value"]); kernel = row["Kernel_Name"].split("(")[0]
    b = json.loads(open("$OUT/%s_set1.out" % tag).read().strip().splitlines()[-1])[name]
    c = dict(res)
    samples = 4 * 64 * 1920 * 1080   # the warm-up pass + 3 timed passes of 64 spp, both launches counted
    xcd = c["GRBM_GUI_ACTIVE"] / 8.0
    # FETCH_SIZE tallies every L2 miss at 64 B (profiles/r4_fetch_size_calibration.json); the textured room's traffic is texel gathers
    # (4 - 16 bytes asked of a line): taken as it is (L2 misses x 64 B), not doubled as for a coalesced stream
    hbm = (1.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    out[name] = {"kernel": kernel, "msamples_per_s_under_profiler": b["msamples_per_s"], "shaded_per_sample": b["shaded_per_sample"],
                 "wait_share": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "valu_busy": c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * xcd),
                 "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
                 "valu_insts_per_sample": c["SQ_INSTS_VALU"] * 64 / samples / 64, "vmem_insts_per_sample_x64": c["SQ_INSTS_VMEM"] * 64 / samples,
                 "hbm_bytes_per_sample": hbm / samples, "l2_hit": c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
                 "counters": c}
out["n_floor"] = $NF
out["source"] = "tools/tex_pmc.sh $NF: rocprofv3 --kernel-trace --pmc <one set per pass> -- python tools/textured_bench.py 3 $NF (TEXBENCH_ONLY=<variant>); 1920x1080, 4 passes of 64 spp; hbm = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; valu_busy = SQ_INSTS_VALU x 2 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)"
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps({k: ({kk: vv for kk, vv in v.items() if kk != "counters"} if isinstance(v, dict) else v) for k, v in out.items()}, indent=1))
PY
