#!/bin/bash
# builds the library with different flags on the GPU box and runs tools/shard_balance.py for each
cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  echo "=== variant: $V"
  AKR_EXTRA_HIPCC_FLAGS="$V" python akari_render_amd/build.py --force > /dev/null 2> gpurun_out/build_variant.err || { tail -5 gpurun_out/build_variant.err; continue; }
  python tools/shard_balance.py 8 16 | cut -c1-60,230-
done
