#!/bin/bash
# PMC passes over one bench.py configuration or secondary leg (one --pmc set per rocprofv3 run, kernel-trace only), summarised into the
# JSON that bench.py's roofline blocks read: profiles/r6_pmc_<name>.json (copy gpurun_out/pmc_bench_<name>/summary.json there). The summary
# carries the hash of the library sources (bench.csrc_hash): bench.py quotes it only for the kernels it was measured on.
# usage: tools/pmc_bench.sh c2|c3|c4|reference_default|forest_100k_kept|forest_10k_kept|forest_10k_flattened|textured_{exhaustive,bvh}_{interpreter,per_scene}
set -u
CFG=${1:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_bench_$CFG
rm -rf $OUT; mkdir -p $OUT
case $CFG in
  c2|c3|c4) ARGS="--config $CFG --steps 1 --warmup 0 --also none --no-cpu-baseline" ;;
  *) ARGS="--leg $CFG" ;;
esac
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
  "GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" \
  "TA_BUSY_avr TA_TA_BUSY_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT -o set$i -- python bench.py $ARGS > $OUT/set$i.out 2> $OUT/set$i.err
  echo "set$i rc=$? : $SET"
done
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, ".")
import bench as _bench
out, cfg = "$OUT", "$CFG"
res = collections.defaultdict(float); disp = collections.defaultdict(set)
kernel = None
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_pt_pass" not in k and "akr_pt_pass_spec" not in k and "k_wf_" not in k: continue  # (k_pt_pass, k_pt_pass_inst, a per-scene kernel's wrapper; the wavefront schedule's kernels together)
        kernel = k if kernel is None or kernel == k else " + ".join(sorted(set(kernel.split(" + ")) | {k}))
        res[row["Counter_Name"]] += float(row["Counter_Value"]); disp[row["Counter_Name"]].add(row["Dispatch_Id"])
bench = json.loads(open(out + "/set1.out").read().strip().splitlines()[-1])
n_launch = max(len(v) for v in disp.values())
samples = bench["counters"]["n_samples"] if "counters" in bench else bench["total_samples"]  # (a leg: every launch of its session, warm-up included -- the counters sum over them all)
c = dict(res)
xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0   # summed over the 8 XCDs
# FETCH_SIZE (KiB) tallies every L2 miss at 64 B. Calibrated on known byte counts (profiles/r4_fetch_size_calibration.json,
# tools/gather_calib.sh): a coalesced stream reads 2.000 x FETCH_SIZE (the guide's gfx950 correction) -- the cbox launches' traffic is
# sampler states and film, streamed; a random 64-byte record read with four 16-byte loads (the BVH traversal's node / triangle
# fetch, what C4's traffic is made of) reads 1.0008 x FETCH_SIZE.
gathers = cfg == "c4" or cfg.startswith("forest") or cfg.startswith("textured")  # traffic made of random 64-byte records / texel gathers
fetch_factor = 1.0 if gathers else 2.0
hbm = (fetch_factor * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
valu_busy = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * xcd_cycles)
lane = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
wait = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
ta = c.get("TA_BUSY_avr", 0.0) / xcd_cycles
s = {"config": cfg, "kernel": kernel, "csrc_hash": _bench.csrc_hash(), "bench_args": "$ARGS", "launches": n_launch, "samples_per_launch": samples / n_launch,
     "hbm_bytes_per_launch": hbm / n_launch, "hbm_bytes_per_sample": hbm / samples,
     "valu_busy": valu_busy, "valu_lane_utilisation": lane, "wait_share": wait, "ta_busy": ta,
     "l2_hit": c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
     "value_under_profiler_msamples_s": bench["value"],
     "counters": c,
     "fetch_size_factor": fetch_factor,
     "source": "tools/pmc_bench.sh " + cfg + ": rocprofv3 --kernel-trace --pmc <one set per pass> -- python bench.py $ARGS; hbm = (factor x FETCH_SIZE + WRITE_SIZE) x 1024, factor = bytes read / FETCH_SIZE calibrated per access pattern on a known byte count (profiles/r4_fetch_size_calibration.json: 2.000 coalesced stream, 1.0008 random 64-byte records); valu_busy = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
if gathers:
    del s["hbm_bytes_per_launch"]  # traffic scales with the rays traced: bench.py multiplies bytes per sample by its own launch size
s["binding_limiter"] = ("VALU issue (the 36-triangle walk and the shading run from LDS / registers; HBM sees only sampler states + film)" if not gathers
                        else ("lane utilisation of the two-level traversal (divergent stages, on-the-fly records)" if cfg.startswith("forest") and "kept" in cfg else "texel / record gathers and the divergent graph code") if cfg != "c4" else "the memory system's rate for random 64-byte records (tools/micro/gather_bw.hip: 25.8 G records/s from HBM, 56 G/s from the Infinity Cache, whatever "
                             "the occupancy or the loads in flight per lane): this kernel misses the L2 " + str(round(c["TCC_MISS_sum"] / samples, 1)) + " times per sample")
s["fabric_read_bytes_per_sample"] = fetch_factor * c["FETCH_SIZE"] * 1024.0 / samples
s["l2_misses_per_sample"] = c["TCC_MISS_sum"] / samples
s["wave_cycle_shares"] = {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "waitcnt": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stall": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]}
json.dump(s, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in s.items() if k != "counters"}, indent=1))
PY
