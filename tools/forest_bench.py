"""procedural.instanced_forest at 1080p: kept as meshes + instances vs flattened (where the flattened records fit), throughput and
traversal counters. python tools/forest_bench.py [n_instances tris_per_mesh spp [modes]]   modes: "kept", "flat" or "kept,flat" """
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from akari_render_amd import abi, capi, procedural

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
tris = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
modes = (sys.argv[4] if len(sys.argv) > 4 else "kept").split(",")
ctx = capi.Context(0)
t0 = time.time(); sd = procedural.instanced_forest(n_inst, tris, width=1920, height=1080); t1 = time.time()
films = {}
for mode in modes:
    with capi.options(instancing=1 if mode == "kept" else 0):
        t1 = time.time(); scene = capi.Scene(ctx, sd); t2 = time.time()
        info = scene.info()
        film = capi.Film(ctx, 1920, 1080)
        cfg = abi.PtConfig.default(); cfg.spp = spp * 2; cfg.spp_per_pass = spp; cfg.max_depth = 12; cfg.rr_depth = 5
        se = capi.PtSession(ctx, scene, cfg, film)
        se.passes(1, blocking=True); s0 = se.stats()
        ta = time.perf_counter(); se.passes(1, blocking=True); tb = time.perf_counter()
        s1 = se.end()
    d = {k: s1[k] - s0[k] for k in s1 if k not in ("n_launches",)}
    rays = d["n_closest"] + d["n_shadow"]
    films[mode] = film.read()
    img = film.resolve()
    print(json.dumps({"mode": mode, "n_instances": n_inst, "tris_per_mesh": tris, "n_tris": info.n_triangles, "uses_bvh": info.uses_bvh, "bvh_nodes": info.n_bvh_nodes,
                      "bvh_depth": info.bvh_depth, "device_MB": round(info.device_bytes / 1e6, 2), "gen_s": round(t1 - t0, 2), "compile_upload_s": round(t2 - t1, 2),
                      "msamples_per_s": round(d["n_samples"] / (tb - ta) / 1e6, 2), "rays_per_s_G": round(rays / (tb - ta) / 1e9, 3),
                      "nodes_per_ray": round(d["n_node_visits"] / rays, 2), "tris_per_ray": round(d["n_tri_tests"] / rays, 2),
                      "closest_per_sample": round(d["n_closest"] / d["n_samples"], 3),
                      "mean_rgb": [round(float(x), 5) for x in img.mean(axis=(0, 1))], "finite": bool(np.isfinite(img).all())}), flush=True)
    del se, film, scene
if len(films) == 2:
    a, b = films.values()
    print(json.dumps({"films_differ_in": int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32)))}))
