# A/B of library variants on the kept forest (tools/kept_schedules.py): bash tools/ab_inst_exact.sh [variant ...]   ("" = the product)
cd $GRAFT_REPO_ROOT
for v in "${@:-product}"; do
  echo "== variant $v"
  if [ "$v" != product ]; then export AKR_HIP_LIB=$PWD/akari_render_amd/variants/libakari_hip_$v.so; else unset AKR_HIP_LIB; fi
  KS_GROUPS=${KS_GROUPS:-2} python tools/kept_schedules.py ${KS_TRIS:-10000 100000} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.rstrip()); continue
    if 'msamples_per_s' in d: print(d['tris_per_mesh'], d['schedule'], d['wf_groups'], round(d['msamples_per_s'], 1), round(d['candidates_per_ray'], 2))
    else: print(d)
"
done
