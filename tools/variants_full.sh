#!/bin/bash
# builds the library with different flags on the GPU box and runs perf_quick --full for each
cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  echo "=== variant: $V"
  AKR_EXTRA_HIPCC_FLAGS="$V" python akari_render_amd/build.py --force > /dev/null 2> gpurun_out/build_variant.err || { tail -5 gpurun_out/build_variant.err; continue; }
  python tools/perf_quick.py 4 --full | grep -v "^golden"
done
