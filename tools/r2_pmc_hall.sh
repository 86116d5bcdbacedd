#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_hall.sh r2_hall_bvh8 1e7 > gpurun_out/pmc_r2_hall.log 2>&1
bash tools/pmc_mem.sh r2_hall_bvh8 python tools/hall_bench.py 1e7 8 > gpurun_out/pmcmem_r2_hall.log 2>&1
tail -50 gpurun_out/pmc_r2_hall.log
