#!/bin/bash
# Wavefront schedule with and without sorted ray queues (option wf_sort) on the hall and the cbox with a forced BVH, next to the megakernel.
O=gpurun_out/r5h; mkdir -p $O
python - <<'PY' 2>&1 | tee $O/wf_sort.txt
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from akari_render_amd import abi, capi, procedural
ctx = capi.Context(0)
def run(scene, w, h, spp, **opts):
    with capi.options(**opts):
        film = capi.Film(ctx, w, h)
        cfg = abi.PtConfig.default(); cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth = 2 * spp, 64, 12, 5
        cfg.filter_type, cfg.filter_radius = abi.FILTER_GAUSSIAN, 1.5
        se = capi.PtSession(ctx, scene, cfg, film)
    se.passes(spp // 64, blocking=True); s0 = se.stats()
    t0 = time.perf_counter(); se.passes(spp // 64, blocking=True); dt = time.perf_counter() - t0
    s1 = se.end()
    return (s1["n_samples"] - s0["n_samples"]) / dt / 1e6, film.read()
sd = procedural.sponza_like(10_000_000, seed=1234, width=1920, height=1080)
hall = capi.Scene(ctx, sd)
out = {}
films = {}
for name, opts in (("megakernel", dict(wavefront=0)), ("wavefront", dict(wavefront=1, wf_sort=0)), ("wavefront sorted", dict(wavefront=1, wf_sort=1))):
    v, f = run(hall, 1920, 1080, 128, **opts)
    out["hall " + name] = v; films[name] = f
    print("hall", name, round(v, 1), flush=True)
print("hall films identical:", all(np.array_equal(films["megakernel"].view(np.uint32), f.view(np.uint32)) for f in films.values()))
with capi.options(force_bvh=1):
    cbox = capi.Scene(ctx, "scenes/cbox/scene.json", 1920, 1080)
films = {}
for name, opts in (("megakernel", dict(wavefront=0)), ("wavefront", dict(wavefront=1, wf_sort=0)), ("wavefront sorted", dict(wavefront=1, wf_sort=1))):
    v, f = run(cbox, 1920, 1080, 256, **opts)
    out["cbox forced bvh " + name] = v; films[name] = f
    print("cbox forced bvh", name, round(v, 1), flush=True)
print("cbox films identical:", all(np.array_equal(films["megakernel"].view(np.uint32), f.view(np.uint32)) for f in films.values()))
print(json.dumps(out))
PY
