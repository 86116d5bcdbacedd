#!/bin/bash
cd "$GRAFT_REPO_ROOT"
T=tools/r3_batch.sh
$T tests product
REPS=2 $T bench c2 product
REPS=2 $T bench c3 product c3w3
REPS=2 $T bench c4 product
NFLOOR=8 $T tex product tstrag tpark tboth
NFLOOR=1 $T tex product tpark
STEPS=8 $T shard fd product; cp gpurun_out/r3/shard_fd_product.json gpurun_out/r3/r3_shard_balance_c2_1080p.json
STEPS=8 $T shard full product; cp gpurun_out/r3/shard_full_product.json gpurun_out/r3/r3_shard_balance_c3_1080p.json
STEPS=4 EXTRA=--4k $T shard fd product; cp gpurun_out/r3/shard_fd_product.json gpurun_out/r3/r3_shard_balance_c2_4k.json
STEPS=4 EXTRA=--4k $T shard full product; cp gpurun_out/r3/shard_full_product.json gpurun_out/r3/r3_shard_balance_c3_4k.json
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_default_product.json 2> gpurun_out/r3/bench_default_product.err; python -c "
import json;d=json.load(open('gpurun_out/r3/bench_default_product.json'));print('bench value',d['value']);e=d.get('extra_configs',{});print({k:(round(v['value'],1) if 'value' in v else v) for k,v in e.items() if k!='schedules'});print({k:(round(v['value'],1) if 'value' in v else v) for k,v in e.get('schedules',{}).items()})"
