#!/bin/bash
# Per-instruction stall attribution of k_pt_pass: rocprofv3 PC sampling (beta) over one bench.py step, aggregated by instruction and
# source line (the "lines" variant of the library = the product's code with -gline-tables-only; build it first:
#   python akari_render_amd/build.py --variant lines --only pt_kernels.hip -gline-tables-only).
# usage: tools/pc_sample.sh <c2|c3|c4> ; output gpurun_out/r4_pcs_<cfg>/ (summary_*.json / .txt; copy to profiles/).
# If the lease has no PC sampling the script says so with the tool's own error text (kept in <method>_<interval>.err).
set -u
CFG=${1:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_pcs_$CFG; rm -rf $OUT; mkdir -p $OUT
export AKR_DATA_DIR=$PWD/akari_render_amd/data
[ -f akari_render_amd/variants/libakari_hip_lines.so ] && export AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_lines.so
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
(rocprofv3-avail info --pc-sampling || rocprofv3-avail list --pc-sampling || rocprofv3 -L) > $OUT/avail.txt 2>&1
grep -i -A6 "pc.sampl\|host_trap\|stochastic" $OUT/avail.txt | head -40
ARGS="--config $CFG --steps 1 --warmup 0 --also none --no-cpu-baseline"
for SPEC in ${SPECS:-"host_trap:time:10000" "stochastic:cycles:1048576" "host_trap:time:1000" "stochastic:cycles:131072"}; do
  M=${SPEC%%:*}; R=${SPEC#*:}; U=${R%%:*}; I=${R#*:}
  timeout 900 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I --kernel-trace \
     -f csv -d $OUT -o ${M}_$I -- python bench.py $ARGS > $OUT/${M}_$I.out 2> $OUT/${M}_$I.err
  rc=$?
  F=$(find $OUT -name "${M}_${I}_pc_sampling*.csv" | head -1)
  if [ -n "$F" ] && [ $(wc -l < "$F") -gt 1 ]; then
    echo "pc sampling $M/$U/$I rc=$rc: $(wc -l < $F) samples"
    python tools/pc_aggregate.py "$F" "$(find $OUT -name "${M}_${I}_kernel_trace.csv" | head -1)" $OUT/summary_${M}_$I > $OUT/summary_${M}_$I.log 2>&1
    tail -30 $OUT/summary_${M}_$I.log
    find $OUT -name "${M}_${I}_pc_sampling*.csv" -size +40M -exec sh -c 'head -2000000 "$1" > "$1.head" && rm "$1"' _ {} \;
  else
    echo "pc sampling $M/$U/$I rc=$rc: NO SAMPLES; tool said: $(grep -i -m3 "error\|not supported\|unavailable\|fail\|denied" $OUT/${M}_$I.err | tr '\n' '|')"
    tail -5 $OUT/${M}_$I.err
  fi
done
