#!/bin/bash
# Builds tools/ray_sort_sim.cpp, dumps the triangles of the bench's hall (procedural.sponza_like(N, seed 1234)) and runs the model.
#   tools/ray_sort_sim.sh [n_tris=10000000] [width=1920] [height=1080] [max_depth=6] [out=profiles/r5_ray_sort_sim.json]
set -e
cd "$(dirname "$0")/.."
N=${1:-10000000}; W=${2:-1920}; H=${3:-1080}; D=${4:-6}; OUT=${5:-profiles/r5_ray_sort_sim.json}
g++ -O2 -std=c++17 -pthread -I akari_render_amd/csrc tools/ray_sort_sim.cpp akari_render_amd/csrc/host/bvh.cpp -o /tmp/ray_sort_sim
python - <<PY
import numpy as np
from akari_render_amd import procedural
sd = procedural.sponza_like($N, seed=1234)
out = []
for inst in sd.instances:
    m = sd.meshes[inst.mesh]
    T = np.asarray(inst.transform, np.float64).reshape(4, 4).T
    v = np.asarray(m.vertices, np.float64)
    vw = (v @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    out.append(vw[np.asarray(m.indices).reshape(-1, 3)].reshape(-1, 9))
np.concatenate(out).astype(np.float32).tofile("/tmp/hall_$N.f32")
print("dumped", sum(o.shape[0] for o in out), "triangles")
PY
/tmp/ray_sort_sim /tmp/hall_$N.f32 $W $H $D $OUT
