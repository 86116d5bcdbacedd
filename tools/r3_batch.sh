#!/bin/bash
# Round-3 same-box A/B of library variants (akari_render_amd/variants/libakari_hip_<name>.so, built with
# `python akari_render_amd/build.py --variant <name> ...`; "product" = the shipped library).
#   tools/r3_batch.sh bench <cfg> <variant>...      bench.py --config <cfg>, 2 steps, every variant, $REPS rounds (default 2)
#   tools/r3_batch.sh shard <cfg-flag> <variant>... tools/shard_balance.py 8 (1080p) per variant; cfg-flag: fd | full
#   tools/r3_batch.sh tests <variant>...            pytest -m gpu with the variant as the library under test
# Different boxes differ by +-2 %: only numbers of one call compare.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3; mkdir -p $OUT
MODE=$1; shift
export AKR_DATA_DIR=$PWD/akari_render_amd/data   # a variant library sits in another directory than the tables of the pmj02bn sampler
lib() { if [ "$1" = product ]; then unset AKR_HIP_LIB; else export AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_$1.so; fi; }
case $MODE in
bench)
  CFG=$1; shift
  for R in $(seq 1 ${REPS:-2}); do
    for V in "$@"; do
      lib $V
      timeout 600 python bench.py --config $CFG --steps ${STEPS:-2} --warmup 1 --also none --no-cpu-baseline > $OUT/bench_${CFG}_$V.json 2> $OUT/bench_${CFG}_$V.err
      echo "bench $CFG round $R $V $(python -c "import json;d=json.load(open('$OUT/bench_${CFG}_$V.json'));c=d['counters'];r=c['n_closest']+c['n_shadow'];print(round(d['value'],1),'Msamples/s  nodes/ray',round(c['n_node_visits']/r,2),'tris/ray',round(c['n_tri_tests']/r,2))" 2>&1 | tail -1)"
    done
  done ;;
shard)
  KIND=$1; shift
  for V in "$@"; do
    lib $V
    timeout 900 python tools/shard_balance.py 8 ${STEPS:-8} --$KIND ${EXTRA:-} > $OUT/shard_${KIND}_$V.json 2> $OUT/shard_${KIND}_$V.err
    echo "shard $KIND $V $(python -c "import json;d=json.load(open('$OUT/shard_${KIND}_$V.json'));print('T1',round(d['T1_ms'],1),'max rank',max(d['per_rank_ms']),'eff',round(d['kernel_scaling_efficiency'],3))" 2>&1 | tail -1)"
  done ;;
tex)   # tools/textured_bench.py: the textured room (NFLOOR=1: exhaustive kernel; 8: BVH kernel), variant "textured" and its constant twin
  for V in "$@"; do
    lib $V
    for ONLY in "textured" "same room, constant materials with the lobes the graphs select"; do
      TEXBENCH_ONLY="$ONLY" timeout 600 python tools/textured_bench.py 4 ${NFLOOR:-8} > $OUT/tex_${NFLOOR:-8}_$V.json 2> $OUT/tex_${NFLOOR:-8}_$V.err
      echo "tex nfloor=${NFLOOR:-8} $V [$ONLY] $(python -c "import json;d=json.load(open('$OUT/tex_${NFLOOR:-8}_$V.json'));print({k:round(v['msamples_per_s'],1) for k,v in d.items()})" 2>&1 | tail -1)"
    done
  done ;;
tests)
  for V in "$@"; do
    lib $V
    timeout 1200 python -m pytest tests -x -q -m gpu ${PYTEST_ARGS:-} > $OUT/gputest_$V.log 2>&1
    echo "tests $V rc=$? $(tail -1 $OUT/gputest_$V.log)"
  done ;;
esac
