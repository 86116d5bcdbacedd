#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=tools/r3_batch.sh
O=gpurun_out/r3
$T tests product
rm -rf $O/wfprof2; timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/wfprof2 -o wf -- python tools/wf_profile.py hall 1 > $O/wf_profile2.json 2> $O/wf_profile2.err; cat $O/wf_profile2.json; find $O/wfprof2 -name "*kernel_stats.csv" | head -1 | xargs head -4
AKR_DATA_DIR=$PWD/akari_render_amd/data AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_wfs4.so timeout 600 python tools/wf_profile.py hall 1
timeout 600 python tools/wf_profile.py cbox 2
