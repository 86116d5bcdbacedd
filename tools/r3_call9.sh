#!/bin/bash
cd "$GRAFT_REPO_ROOT"
T=tools/r3_batch.sh
$T tests bvh6
REPS=2 $T bench c4 product bvh6
NFLOOR=8 $T tex product bvh6
export AKR_DATA_DIR=$PWD/akari_render_amd/data
timeout 300 python tools/wf_profile.py cbox 2
AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_bvh6.so timeout 300 python tools/wf_profile.py cbox 2
AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_bvh6.so timeout 300 python tools/wf_profile.py hall 1
