"""Throughput of the mcmc_opt integrator: cbox at 1920x1080, max_depth 7, no direct pass, by number of chains (one lane per
chain: the reference's default of 512 chains fills 8 waves of a 1024-SIMD chip). python tools/mcmc_bench.py [spp]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from akari_render_amd import abi, capi
from oracle import scene_json

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = 1920, 1080
sd = scene_json.load_scene(os.path.join(ROOT, "scenes/cbox/scene.json"), W, H)
sd.ggx_table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)
ctx = capi.Context(0)
scene = capi.Scene(ctx, sd)
out = {}
for n_chains in (512, 16384, 262144, 1048576):
    film = capi.Film(ctx, W, H)
    cfg = abi.McmcConfig.default()
    cfg.n_chains, cfg.direct_spp, cfg.n_bootstrap = n_chains, -1, 100000
    cfg.spp = spp if n_chains > 512 else 1
    if n_chains == 512:  # bound the sequential depth: 1/16 spp worth of mutations
        cfg.spp_per_pass = 1
    t0 = time.perf_counter(); st, res, _ = capi.mcmc_render(ctx, scene, cfg, film); t1 = time.perf_counter()
    out[n_chains] = {"wall_s": t1 - t0, "kernel_ms": st["kernel_ms"], "mmutations_per_s": res["n_mutations"] / st["kernel_ms"] / 1e3,
                     "acceptance": res["acceptance_rate"], "b": res["normalization"], "image_mean": float(film.resolve().mean())}
print(json.dumps(out))
