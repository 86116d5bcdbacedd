#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=tools/r3_batch.sh
O=gpurun_out/r3
$T tests product
NFLOOR=8 $T tex product
NFLOOR=1 $T tex product
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 1500 $O/bench_gloo2.json; tail -3 $O/bench_gloo2.err
rm -rf $O/wfprof; timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $O/wfprof -o wf -- python tools/wf_profile.py hall 1 > $O/wf_profile.json 2> $O/wf_profile.err; cat $O/wf_profile.json; find $O/wfprof -name "*kernel_stats.csv" | head -1 | xargs head -8
