#!/bin/bash
cd "$GRAFT_REPO_ROOT"
T=tools/r3_batch.sh
$T tests product
PYTEST_ARGS="--deselect tests/test_gpu_fullsize.py" $T tests v1 v3 v5
REPS=2 $T bench c4 product v1 v4 v5
REPS=2 $T bench c3 product v1 v2 v3
REPS=1 $T bench c2 product v1
NFLOOR=8 $T tex product v1 v4
NFLOOR=1 $T tex product v1
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_default_product.json 2> gpurun_out/r3/bench_default_product.err; python -c "
import json;d=json.load(open('gpurun_out/r3/bench_default_product.json'));print('bench value',d['value'],'roofline',{k:d['roofline'].get(k) for k in ('frac','frac_measured','measured_counters')});print(json.dumps(d.get('extra_configs',{}).get('schedules'),indent=0))"
