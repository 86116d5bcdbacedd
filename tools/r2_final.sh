#!/bin/bash
# round 2 end state: the GPU test suite, the textured-room bench (exhaustive + BVH variant), optionally C3's PMC passes (the
# summary bench.py's roofline block reads), the driver's bench command, kernel-trace stats of the same command
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_final; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for NF in 1 8; do TEXBENCH_NFLOOR=$NF timeout 300 python tools/textured_bench.py 3 > $OUT/textured_$NF.json 2> $OUT/textured_$NF.err; echo "textured n_floor=$NF rc=$?"; done
if [ "${1:-}" = "pmc3" ]; then bash tools/pmc_bench.sh c3 > $OUT/pmc3.log 2>&1; echo "pmc3 rc=$?"; cp gpurun_out/pmc_bench_c3/summary.json profiles/r2_pmc_c3.json; fi
if [ "${1:-}" = "pmc" ]; then  # all the counter summaries bench.py's roofline blocks read, and the textured room's
  for C in c2 c3 c4; do bash tools/pmc_bench.sh $C > $OUT/pmc_$C.log 2>&1; echo "pmc $C rc=$?"; cp gpurun_out/pmc_bench_$C/summary.json profiles/r2_pmc_$C.json; done
  bash tools/r2_texpmc.sh > $OUT/pmc_tex.log 2>&1; echo "pmc textured rc=$?"
fi
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
echo "prof rc=$?"
