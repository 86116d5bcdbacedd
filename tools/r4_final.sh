#!/bin/bash
# Round-4 closing measurements on one box: tests, randomised soak, PMC summaries of the three bench configurations (stamped with the
# hash of the library sources), the rocprofv3 kernel statistics of the driver's bench command, the bench line itself, shard balance
# (tiles and sample ranges), textured-room numbers. Everything lands in gpurun_out/r4_final/; the summaries are then copied to profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "tests rc=$? $(grep -E "passed|failed" $O/gputest.log | tail -1)"
for MODE in "500 100000" "150 200000 big" "250 300000 tex" "150 400000 wavefront" "150 500000 shard" "100 600000 gpt" "100 700000 aov"; do
  timeout 300 python tools/soak.py $MODE 2>&1 | grep -E "MISMATCH|cases from seed|rror" | tail -3 | sed "s/^/soak [$MODE] /"
done 2>&1 | tee $O/soak.txt
for CFG in c2 c3 c4; do
  bash tools/pmc_bench.sh $CFG > $O/pmc_$CFG.log 2>&1
  cp gpurun_out/r4_pmc_bench_$CFG/summary.json profiles/r4_pmc_$CFG.json && cp profiles/r4_pmc_$CFG.json $O/
  python -c "import json;d=json.load(open('profiles/r4_pmc_$CFG.json'));print('pmc $CFG', {k:d.get(k) for k in ('valu_busy','valu_lane_utilisation','wait_share','l2_hit','ta_busy','hbm_bytes_per_sample','fabric_read_bytes_per_sample','l2_misses_per_sample','value_under_profiler_msamples_s','csrc_hash')})"
done
bash tools/profile_bench.sh r4 --gpus 1 --steps 20 --warmup 5 > $O/profile_bench.log 2>&1; cp gpurun_out/prof_r4/*kernel_stats.csv $O/r4_bench_kernel_stats.csv; cp gpurun_out/prof_r4/bench.json $O/r4_bench_under_rocprof.json; head -4 $O/r4_bench_kernel_stats.csv
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r4_bench.json 2> $O/r4_bench.err; python -c "
import json;d=json.load(open('$O/r4_bench.json'));r=d['roofline'];print('BENCH',round(d['value'],1),'ms/step',round(d['ms_per_step'],2),'frac',round(r['frac'],3),'frac_measured',r.get('frac_measured'),'cpu',d.get('cpu_baseline',{}).get('value'));e=d['extra_configs'];print({k:(round(v['value'],1), round(v.get('roofline',{}).get('frac',0),3), v.get('roofline',{}).get('frac_measured')) for k,v in e.items() if 'value' in v});print({k:round(v['value'],1) for k,v in e.get('schedules',{}).items() if 'value' in v})"
for K in fd full; do for R in "" "--4k"; do for S in "" "--split samples"; do
  STEPS=8
  timeout 900 python tools/shard_balance.py 8 $STEPS --$K $R $S > "$O/shard_${K}${R}$(echo $S | tr -d ' -').json" 2>> $O/shard.err
  python -c "import json;d=json.load(open('$O/shard_${K}${R}$(echo $S | tr -d ' -').json'));print('shard $K $R $S eff',round(d['kernel_scaling_efficiency'],3),'T1',round(d['T1_ms'],1),'max rank',max(d['per_rank_ms']))"
done; done; done
for NF in 1 8; do timeout 900 python tools/textured_bench.py 4 $NF > $O/textured_nfloor$NF.json 2>> $O/tex.err; python -c "import json;d=json.load(open('$O/textured_nfloor$NF.json'));print('tex nfloor=$NF',{k[:28]:round(v['msamples_per_s'],1) for k,v in d.items()})"; done
for NF in 1 8; do bash tools/tex_pmc.sh $NF > $O/tex_pmc_$NF.log 2>&1; cp gpurun_out/texpmc_$NF/summary.json $O/r4_pmc_textured_room_nfloor$NF.json; python -c "import json;d=json.load(open('$O/r4_pmc_textured_room_nfloor$NF.json'));print('texpmc nfloor=$NF',{k[:10]:{kk:round(vv,3) for kk,vv in v.items() if kk in ('wait_share','valu_busy','valu_lane_utilisation','hbm_bytes_per_sample','msamples_per_s_under_profiler')} for k,v in d.items() if isinstance(v,dict)})"; done
