"""The relaxed arithmetic tier (option arith = 1) against the contract tier on the same seeds: relRMSE, the per-pixel error's median and
99th percentile, and how many pixels are off by more than 1e-3 of their value -- pixels in which a comparison flipped (and, with the
independent sampler, the rest of the pass's stream moved). python tools/rx_diag.py [c1]   (needs a GPU)"""
import sys, json, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from akari_render_amd import capi, distributed, abi
from oracle import pyoracle, scene_json
from tests.helpers import make_config, resolve_np, rel_rmse
ctx = capi.Context(0)
cbox = os.path.join(ROOT, 'scenes', 'cbox', 'scene.json')
def run(label, sd, cfg, osc=None):
    w, h = sd.camera.width, sd.camera.height
    n = w*h
    scene = capi.Scene(ctx, sd)
    films = {}
    for a in (0, 1):
        with capi.options(arith=a):
            f = capi.Film(ctx, w, h); capi.pt_render(ctx, scene, cfg, f); films[a] = f.read()
    owned = distributed.owned_pixel_mask(w, h, cfg.shard_rank, cfg.shard_count, cfg.tile_w, cfg.tile_h).ravel() if cfg.shard_count > 1 else np.ones(n, bool)
    e = resolve_np(films[0], w, h).reshape(-1,3)[owned].astype(np.float64); r = resolve_np(films[1], w, h).reshape(-1,3)[owned].astype(np.float64)
    lum = e @ np.array([0.2126,0.7152,0.0722])
    d = np.sqrt(((r-e)**2).sum(1))
    rel = d/np.maximum(lum.mean(),1e-30)
    pr = d/np.maximum(lum,1e-3*lum.mean())
    out = dict(label=label, relRMSE=float(np.sqrt((d**2).mean())/lum.mean()), median_pix=float(np.median(pr)), p99=float(np.percentile(pr,99)), frac_gt_1e3=float((pr>1e-3).mean()), n_gt_1e3=int((pr>1e-3).sum()), n_pix=int(owned.sum()),
               relRMSE_without_outliers=float(np.sqrt((d[pr<=1e-3]**2).mean())/lum.mean()), mean_ratio=float(r.mean()/e.mean()-1))
    print(json.dumps(out), flush=True)
from tests.helpers import cbox_variant
if 'variants' in sys.argv:
    for which in ("glass_coat", "kinds"):
        run(f"cbox {which}", cbox_variant(scene_json.load_scene(cbox, 96, 96), which), make_config(spp=64, spp_per_pass=32, max_depth=10))
    sys.exit(0)
sd = scene_json.load_scene(cbox, 256, 256)
for smp in (0, 2, 1):
    run(f"C1 full sampler {smp}", sd, make_config(spp=64, spp_per_pass=64, max_depth=12, rr_depth=5, sampler_type=smp))
    run(f"C1 FD sampler {smp}", sd, make_config(spp=64, spp_per_pass=64, max_depth=12, rr_depth=5, force_diffuse=1, sampler_type=smp))
if 'c1' in sys.argv: sys.exit(0)
sd = scene_json.load_scene(cbox, 1920, 1080)
for smp in (0, 2):
    run(f"C2 shard sampler {smp}", sd, distributed.shard_config(make_config(spp=1024, spp_per_pass=64, max_depth=12, rr_depth=5, force_diffuse=1, sampler_type=smp), 7, 255, 32, 32))
    run(f"C3 shard sampler {smp}", sd, distributed.shard_config(make_config(spp=4096, spp_per_pass=64, max_depth=12, rr_depth=5, sampler_type=smp), 100, 510, 32, 32))
