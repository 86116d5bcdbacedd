#!/bin/bash
# Round 5, per-scene kernels: GPU parity of the new tests, the existing texture tests under AKR_SPECIALISE=1, and the textured
# room's throughput with the interpreter and the per-scene kernels (3 / 4 waves). Output: gpurun_out/r5a/
O=gpurun_out/r5a; mkdir -p $O
export AKR_KERNEL_CACHE=/tmp/akr_cache_r5
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_specialise.py -x -q > $O/pytest_spec.txt 2>&1; tail -4 $O/pytest_spec.txt
AKR_SPECIALISE=1 timeout 900 python -m pytest tests/test_gpu_textures.py tests/test_gpu_colorspace.py -x -q > $O/pytest_tex_under_spec.txt 2>&1; tail -3 $O/pytest_tex_under_spec.txt
for NF in 1 8; do
  timeout 900 python tools/textured_bench.py 4 $NF > $O/textured_nfloor$NF.json 2>> $O/tex.err
  python - <<PY
import json
d = json.load(open("$O/textured_nfloor$NF.json"))
print("tex nfloor=$NF", {k[:40]: round(v["msamples_per_s"], 1) for k, v in d.items()})
for k, v in d.items():
    if v["kernel"]["specialised"]: print("   ", k, v["kernel"])
PY
done
