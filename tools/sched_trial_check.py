"""The timed schedule trial of a long render on a large flattened scene (api_pt.cpp schedule_trial_eligible): which schedule the session kept and what the
render then runs at, against the same render with the trial switched off.  python tools/sched_trial_check.py   (needs a GPU)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from akari_render_amd import abi, capi, procedural
ctx = capi.Context(0)
for name, sd in (("forest 1000 x 10 k, flattened", procedural.instanced_forest(1000, 10_000, width=1920, height=1080)), ("hall 10 M", procedural.sponza_like(10_000_000, 1234, 1920, 1080))):
    with capi.options(instancing=0):
        scene = capi.Scene(ctx, sd)
    films = {}
    for trial in (0, -1):
        with capi.options(sched_trial=trial):
            film = capi.Film(ctx, 1920, 1080)
            cfg = abi.PtConfig.default(); cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth = 512, 16, 12, 5
            se = capi.PtSession(ctx, scene, cfg, film)
            t = time.perf_counter(); se.passes(16, blocking=True); t1 = time.perf_counter() - t
            t = time.perf_counter(); se.passes(16, blocking=True); t2 = time.perf_counter() - t
            status = se.kernel_info()["status"]; st = se.end()
        films[trial] = film.read()
        print(json.dumps({"scene": name, "sched_trial": trial, "first_256_spp_msamples_s": 256 * 1920 * 1080 / t1 / 1e6, "next_256_spp_msamples_s": 256 * 1920 * 1080 / t2 / 1e6, "status": status}), flush=True)
    print(json.dumps({"scene": name, "films_identical": bool(np.array_equal(films[0].view(np.uint32), films[-1].view(np.uint32)))}), flush=True)
    del scene
