#!/bin/bash
# rocprofv3 kernel trace of one 64-spp pass group of the wavefront schedule on the hall, with and without sorted queues: where the time goes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; rm -rf $O; mkdir -p $O
cat > $O/run.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from akari_render_amd import abi, capi, procedural
sort = int(sys.argv[1])
ctx = capi.Context(0)
sd = procedural.sponza_like(10_000_000, seed=1234, width=1920, height=1080)
hall = capi.Scene(ctx, sd)
with capi.options(wavefront=1, wf_sort=sort):
    film = capi.Film(ctx, 1920, 1080)
    cfg = abi.PtConfig.default(); cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth = 64, 64, 12, 5
    se = capi.PtSession(ctx, hall, cfg, film)
t0 = time.perf_counter(); se.passes(1, blocking=True); dt = time.perf_counter() - t0
s = se.end()
print("sort", sort, "wall_s", dt, "Msamples/s", s["n_samples"] / dt / 1e6, "kernel_ms", s["kernel_ms"])
PY
for S in 0 1; do
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O -o sort$S -- python $O/run.py $S > $O/sort$S.out 2> $O/sort$S.err
  tail -1 $O/sort$S.out
  python - <<PY
import csv, glob
f = glob.glob("$O/**/sort${S}_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:8]:
    print("  %-70s calls %6s total_ms %10.1f avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
done
