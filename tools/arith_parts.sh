#!/bin/bash
# What each part of the relaxed arithmetic tier buys, and what it costs in comparisons that flip against the contract tier: variant builds of
# pt_kernels_relaxed.hip with one part taken back each, then tools/rx_diag.py (relaxed vs exact films) and C2 / C3 throughput per variant.
# Build here (no GPU needed):  bash tools/arith_parts.sh build      On the GPU box:  bash tools/arith_parts.sh run <out_dir>
set -e
cd "$(dirname "$0")/.."
declare -A V=( [nocontract]="-ffp-contract=off" [ieeediv]="-fhip-fp32-correctly-rounded-divide-sqrt" [notrans]="-DAKR_RX_TRANS=0" [norcp]="-DAKR_RX_RCP=0" [nodaz]="-fno-gpu-flush-denormals-to-zero" )
if [ "$1" = build ]; then
  for v in "${!V[@]}"; do python -m akari_render_amd.build --variant rx_$v --only pt_kernels_relaxed.hip ${V[$v]} & done; wait
else
  set +e; export AKR_DATA_DIR="$PWD/akari_render_amd/data"
  out=${2:-gpurun_out/arith_parts}; mkdir -p "$out"
  for v in full "${!V[@]}"; do
    lib=""; [ $v != full ] && lib="$PWD/akari_render_amd/variants/libakari_hip_rx_$v.so"
    echo "== $v"
    AKR_HIP_LIB=$lib python tools/rx_diag.py c1 > "$out/diag_$v.txt" 2>&1; cut -c1-260 "$out/diag_$v.txt"
    for c in c2 c3; do
      AKR_HIP_LIB=$lib AKR_ARITH=1 python bench.py --config $c --steps 3 --warmup 1 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', round(d['value'],1))"
    done
  done 2>&1 | tee "$out/summary.txt"
fi
