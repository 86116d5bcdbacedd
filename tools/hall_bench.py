"""BASELINE configs[3]-style run: procedural hall with N triangles at 1080p; prints throughput + traversal counters."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from akari_render_amd import abi, capi, procedural
n_tris = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = capi.Context(0)
t0 = time.time(); sd = procedural.sponza_like(n_tris, 1234, 1920, 1080); t1 = time.time()
scene = capi.Scene(ctx, sd); t2 = time.time()
info = scene.info()
film = capi.Film(ctx, 1920, 1080)
cfg = abi.PtConfig.default(); cfg.spp = spp * 2; cfg.spp_per_pass = spp; cfg.max_depth = int(os.environ.get("HB_DEPTH", "12")); cfg.rr_depth = 5
se = capi.PtSession(ctx, scene, cfg, film)
se.passes(1, blocking=True); s0 = se.stats()
ta = time.perf_counter(); se.passes(1, blocking=True); tb = time.perf_counter()
s1 = se.end()
d = {k: s1[k] - s0[k] for k in s1 if k not in ("n_launches",)}
nb = 56 * d["n_closest"] + 292 * d["n_shaded"] + 64 * d["n_shadow"] + 156 * d["n_samples"] + 64 * d["n_node_visits"] + 48 * d["n_tri_tests"]
img = film.resolve()
print(json.dumps({"n_tris": info.n_triangles, "bvh_nodes": info.n_bvh_nodes, "device_MB": info.device_bytes / 1e6, "gen_s": t1 - t0, "compile_upload_s": t2 - t1,
                  "msamples_per_s": d["n_samples"] / (tb - ta) / 1e6, "rays_per_s_G": (d["n_closest"] + d["n_shadow"]) / (tb - ta) / 1e9,
                  "nodes_per_ray": d["n_node_visits"] / (d["n_closest"] + d["n_shadow"]), "tris_per_ray": d["n_tri_tests"] / (d["n_closest"] + d["n_shadow"]),
                  "closest_per_sample": d["n_closest"] / d["n_samples"], "model_bytes_per_sample": nb / d["n_samples"], "model_GBs": nb / (tb - ta) / 1e9,
                  "mean_rgb": [float(x) for x in img.mean(axis=(0, 1))], "finite": bool(np.isfinite(img).all())}))
