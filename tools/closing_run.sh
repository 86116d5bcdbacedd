#!/bin/bash
# The round's closing measurements on the sources as they are (one gpurun call): GPU tests, smoke, soak, extreme-transform check, the PMC
# summaries of every bench leg, the default bench line (alone, and under rocprofv3 --kernel-trace --stats). Everything lands in
# gpurun_out/close/; copy what is to be kept into profiles/.   bash tools/closing_run.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/close
rm -rf $OUT; mkdir -p $OUT
HASH=$(python -c "import bench; print(bench.csrc_hash())")
{ echo "# python -m pytest tests -m gpu -q on MI355X, round 6 closing sources (csrc hash $HASH)"
  ( time python -m pytest tests -m gpu -q 2>&1 | grep -E " passed| failed" ) 2>&1 | grep -E "passed|failed|real"
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; } > $OUT/gputest_summary.txt
{ python tools/soak.py 1500 0; python tools/soak.py 1500 20000 inst; python tools/soak.py 300 30000 wavefront; python tools/soak.py 600 40000 wavefront carry; python tools/soak.py 600 50000 wavefront carry inst; python tools/soak.py 300 60000 wavefront carry inst tex; } 2>&1 | grep "cases from" > $OUT/soak.txt
for s in 800x600 1024x768 1024x1024 1600x900 1920x1080 3840x2160; do KS_SIZE=$s KS_GROUPS=1,2 python tools/kept_schedules.py 10000 100000; done > $OUT/kept_schedules.txt 2>&1
python tools/inst_extreme_check.py 400 0 oracle 2>&1 | tail -1 > $OUT/inst_extreme_400.txt
bash tools/pmc_all.sh > $OUT/pmc_all.txt 2>&1
cp gpurun_out/pmc_all/r6_pmc_*.json $OUT/
mkdir -p profiles_tmp && cp $OUT/r6_pmc_*.json profiles/   # (on the box: bench.py reads the summaries from profiles/)
python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o r6 -- python bench.py --steps 20 --warmup 5 --also none --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/r6_bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof profiles_tmp
cat $OUT/gputest_summary.txt $OUT/soak.txt $OUT/inst_extreme_400.txt; grep -c msamples $OUT/kept_schedules.txt
head -c 1500 $OUT/bench.json; echo; head -5 $OUT/r6_bench_kernel_stats.csv
