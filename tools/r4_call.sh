#!/bin/bash
# Round-4 gpurun payload (rewritten per call; the reusable pieces are tools/pc_sample.sh, gather_calib.sh, r3_batch.sh).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_textures.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
export AKR_DATA_DIR=$PWD/akari_render_amd/data
for ON in "" 1 2 3; do
  for MASK in 1; do
    TEXBENCH_DEFER_ON=$ON TEXBENCH_DEFER_MASK=$MASK TEXBENCH_ONLY="textured" timeout 600 python tools/textured_bench.py 4 8 > $O/tex_defer_$ON.json 2> $O/tex_defer_$ON.err
    echo "textured BVH room, defer_on='$ON' mask=$MASK: $(python -c "import json;d=json.load(open('$O/tex_defer_$ON.json'));print({k:round(v['msamples_per_s'],1) for k,v in d.items()})" 2>&1 | tail -1)"
  done
done
TEXBENCH_DEFER_ON=3 TEXBENCH_DEFER_MASK=3 TEXBENCH_ONLY="textured" timeout 600 python tools/textured_bench.py 4 8 > $O/tex_defer_3_3.json 2>/dev/null; echo "defer_on=3 mask=3: $(cat $O/tex_defer_3_3.json | cut -c1-120)"
TEXBENCH_ONLY="textured, conductor deferral off" timeout 600 python tools/textured_bench.py 4 8 > $O/tex_defer_off.json 2>/dev/null; echo "deferral off: $(cat $O/tex_defer_off.json | cut -c1-140)"
TEXBENCH_ONLY="same room, constant materials with the lobes the graphs select" timeout 600 python tools/textured_bench.py 4 8 > $O/tex_twin.json 2>/dev/null; echo "twin: $(cat $O/tex_twin.json | cut -c1-160)"
