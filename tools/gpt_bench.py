"""Throughput of the gpt integrator: cbox at 1920x1080, max_depth 7 (gpt::Config default). One gpt sample = 1 base path +
4 offset paths. Prints gpt samples/s, paths/s, rays/s next to the plain path tracer at the same depth.
python tools/gpt_bench.py [spp]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from akari_render_amd import abi, capi
from oracle import scene_json

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1920, 1080
sd = scene_json.load_scene(os.path.join(ROOT, "scenes/cbox/scene.json"), W, H)
sd.ggx_table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)
ctx = capi.Context(0)
scene = capi.Scene(ctx, sd)
out = {}
for name, recon in (("none", abi.GPT_RECON_NONE), ("weighted", abi.GPT_RECON_WEIGHTED)):
    film = capi.Film(ctx, W, H)
    cfg = abi.GptConfig.default(); cfg.spp, cfg.reconstruction = 2, recon
    capi.gpt_render(ctx, scene, cfg, film)  # warm-up
    film.clear(); cfg.spp = spp
    t0 = time.perf_counter(); st = capi.gpt_render(ctx, scene, cfg, film); t1 = time.perf_counter()
    out[name] = {"wall_s": t1 - t0, "kernel_ms": st["kernel_ms"], "gpt_msamples_per_s": W * H * spp / st["kernel_ms"] / 1e3,
                 "mpaths_per_s": st["n_samples"] / st["kernel_ms"] / 1e3, "mrays_per_s": st["n_closest"] / st["kernel_ms"] / 1e3,
                 "rays_per_path": st["n_closest"] / st["n_samples"]}
film = capi.Film(ctx, W, H)
pc = abi.PtConfig.default(); pc.spp, pc.spp_per_pass, pc.max_depth = 64, 64, 7
capi.pt_render(ctx, scene, pc, film); film.clear()
st = capi.pt_render(ctx, scene, pc, film)
out["pt_same_depth"] = {"msamples_per_s": st["n_samples"] / st["kernel_ms"] / 1e3, "mrays_per_s": (st["n_closest"] + st["n_shadow"]) / st["kernel_ms"] / 1e3}
print(json.dumps(out))
