#!/bin/bash
# Round-5 closing measurements, in two calls (gpurun's budget is per call): tools/r5_final.sh a | b. Everything lands in
# gpurun_out/r5_final/; the summaries are then copied to profiles/r5_*.
#   a: GPU tests, randomised soak (incl. kept scenes), rocprofv3 kernel statistics of the driver's bench command, the bench line
#   b: PMC summaries of the three bench configurations (stamped with the hash of the library sources), textured room (bench + PMC),
#      sampler bench, forest (kept vs flattened), the conservative-reject check build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_final; mkdir -p $O
if [ "${1:-a}" = "c" ]; then  # after a late change: tests, the PMC stamps of the three configurations, the bench line
  timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "tests rc=$? $(grep -E "passed|failed" $O/gputest.log | tail -1)"
  for CFG in c2 c3 c4; do
    bash tools/pmc_bench.sh $CFG > $O/pmc_$CFG.log 2>&1
    cp gpurun_out/r4_pmc_bench_$CFG/summary.json $O/r5_pmc_$CFG.json; rm -rf gpurun_out/r4_pmc_bench_$CFG
    python -c "import json;d=json.load(open('$O/r5_pmc_$CFG.json'));print('pmc $CFG', {k:d.get(k) for k in ('valu_lane_utilisation','hbm_bytes_per_sample','value_under_profiler_msamples_s','csrc_hash')})"
  done
  for f in c2 c3 c4; do cp $O/r5_pmc_$f.json profiles/r5_pmc_$f.json; done   # (bench.py reads them from profiles/)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5_bench.json 2> $O/r5_bench.err; python -c "
import json;d=json.load(open('$O/r5_bench.json'));r=d['roofline'];print('BENCH',round(d['value'],1),'frac',round(r['frac'],3),'frac_measured',r.get('frac_measured'));e=d['extra_configs'];print({k:(round(v['value'],1), v.get('roofline',{}).get('frac_measured')) for k,v in e.items() if isinstance(v,dict) and 'value' in v})"
elif [ "${1:-a}" = "a" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "tests rc=$? $(grep -E "passed|failed" $O/gputest.log | tail -1)"
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
  for MODE in "400 1100000" "120 1200000 big" "200 1300000 tex" "120 1400000 wavefront" "120 1500000 shard" "80 1600000 gpt" "80 1700000 aov" "60 1750000 mcmc" "400 1800000 inst" "120 1810000 inst big" "200 1820000 inst tex" "100 1830000 inst shard"; do
    timeout 300 python tools/soak.py $MODE 2>&1 | grep -E "MISMATCH|cases from seed|rror" | tail -3 | sed "s/^/soak [$MODE] /"
  done 2>&1 | tee $O/r5_soak.txt
  AKR_SPECIALISE=1 timeout 300 python tools/soak.py 150 1900000 tex 2>&1 | grep -E "MISMATCH|cases from seed|rror" | tail -3 | sed "s/^/soak [150 1900000 tex, per-scene kernels] /" | tee -a $O/r5_soak.txt
  bash tools/profile_bench.sh r5 --gpus 1 --steps 20 --warmup 5 > $O/profile_bench.log 2>&1; cp gpurun_out/prof_r5/*kernel_stats.csv $O/r5_bench_kernel_stats.csv; cp gpurun_out/prof_r5/bench.json $O/r5_bench_under_rocprof.json; head -4 $O/r5_bench_kernel_stats.csv; rm -rf gpurun_out/prof_r5
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5_bench.json 2> $O/r5_bench.err; python -c "
import json;d=json.load(open('$O/r5_bench.json'));r=d['roofline'];print('BENCH',round(d['value'],1),'ms/step',round(d['ms_per_step'],2),'frac',round(r['frac'],3),'frac_measured',r.get('frac_measured'),'cpu',d.get('cpu_baseline',{}).get('value'), 'v1024', d.get('value_1024spp_launch'));e=d['extra_configs'];print({k:(round(v['value'],1), round(v.get('roofline',{}).get('frac',0),3), v.get('roofline',{}).get('frac_measured')) for k,v in e.items() if isinstance(v,dict) and 'value' in v});print({k:round(v['value'],1) for k,v in e.get('schedules',{}).items() if 'value' in v})"
else
  for CFG in c2 c3 c4; do
    bash tools/pmc_bench.sh $CFG > $O/pmc_$CFG.log 2>&1
    cp gpurun_out/r4_pmc_bench_$CFG/summary.json $O/r5_pmc_$CFG.json; rm -rf gpurun_out/r4_pmc_bench_$CFG
    python -c "import json;d=json.load(open('$O/r5_pmc_$CFG.json'));print('pmc $CFG', {k:d.get(k) for k in ('valu_busy','valu_lane_utilisation','wait_share','l2_hit','hbm_bytes_per_sample','l2_misses_per_sample','value_under_profiler_msamples_s','csrc_hash')})"
  done
  timeout 600 python tools/textured_bench.py 4 1 > $O/r5_textured_bench_nfloor1.json 2>> $O/tex.err; timeout 600 python tools/textured_bench.py 4 8 > $O/r5_textured_bench_nfloor8.json 2>> $O/tex.err
  for NF in 1 8; do python -c "import json;d=json.load(open('$O/r5_textured_bench_nfloor$NF.json'));print('tex nfloor=$NF',{k[:34]:round(v['msamples_per_s'],1) for k,v in d.items() if isinstance(v,dict) and 'msamples_per_s' in v})"; done
  for NF in 1 8; do bash tools/tex_pmc.sh $NF > $O/tex_pmc_$NF.log 2>&1; cp gpurun_out/texpmc_$NF/summary.json $O/r5_pmc_textured_room_nfloor$NF.json; rm -rf gpurun_out/texpmc_$NF; python -c "import json;d=json.load(open('$O/r5_pmc_textured_room_nfloor$NF.json'));print('texpmc nfloor=$NF',{k[:24]:{kk:round(vv,3) for kk,vv in v.items() if kk in ('wait_share','valu_busy','valu_lane_utilisation','valu_insts_per_sample','msamples_per_s_under_profiler')} for k,v in d.items() if isinstance(v,dict)})"; done
  timeout 600 python tools/sampler_bench.py both > $O/r5_sampler_bench.txt 2>&1; tail -12 $O/r5_sampler_bench.txt
  timeout 200 python tools/forest_bench.py 1000 100000 8 kept > $O/r5_forest_1000x100k.json 2>&1; timeout 200 python tools/forest_bench.py 1000 10000 8 kept,flat > $O/r5_forest_1000x10k_kept_vs_flat.json 2>&1; cat $O/r5_forest_1000x100k.json $O/r5_forest_1000x10k_kept_vs_flat.json | cut -c1-330
  AKR_HIP_LIB=$GRAFT_REPO_ROOT/akari_render_amd/variants/libakari_hip_instcheck.so timeout 400 python tools/inst_pretest_check.py > $O/r5_inst_pretest_check.txt 2>&1; tail -2 $O/r5_inst_pretest_check.txt
fi
