#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  echo "=== variant: $V"
  AKR_EXTRA_HIPCC_FLAGS="$V" python akari_render_amd/build.py --force > /dev/null 2> gpurun_out/build_variant.err || { tail -5 gpurun_out/build_variant.err; continue; }
  python tools/hall_bench.py 1e7 8 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('msamples_per_s','rays_per_s_G','nodes_per_ray')})"
done
