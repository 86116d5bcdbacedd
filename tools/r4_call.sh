#!/bin/bash
# Round-4 gpurun payload (rewritten per call; the reusable pieces are tools/pc_sample.sh, gather_calib.sh, r3_batch.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/vgpr_bank.hip -o $O/vgpr_bank 2>/dev/null && timeout 300 $O/vgpr_bank | tee $O/vgpr_bank.txt
MIB=4096 bash tools/gather_calib.sh 2>&1 | tee $O/gather_calib.log | grep -v "^  "
B=gpurun_out/r4_gather_calib/gather_bw
for SPEC in "gather64 4096 64 8192 40" "gather64x2 4096 64 8192 0" "gather64x2 4096 64 8192 40" "gather64 4096 64 8192 20" "gather64 512 64 8192 0" "gather64 128 64 8192 0"; do
  echo "occupancy/MLP: $SPEC -> $($B $SPEC)"
done | tee $O/gather_occupancy.txt
bash tools/pc_sample.sh c2 2>&1 | tee $O/pcs_c2.log | tail -80
if ls gpurun_out/r4_pcs_c2/summary_*.json >/dev/null 2>&1; then
  SPECS="host_trap:time:10000 stochastic:cycles:1048576" bash tools/pc_sample.sh c3 2>&1 | tee $O/pcs_c3.log | tail -60
  SPECS="host_trap:time:10000 stochastic:cycles:1048576" bash tools/pc_sample.sh c4 2>&1 | tee $O/pcs_c4.log | tail -60
fi
REPS=2 bash tools/r3_batch.sh bench c2 product asmmin
bash tools/r3_batch.sh bench c3 product asmmin
bash tools/r3_batch.sh bench c4 product
