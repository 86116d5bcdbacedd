"""One rank's share of the 1080p cbox frame, alone on the GPU (for PMC passes): python tools/shard_one.py [rank=0] [world=8] [steps=8] [--full]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from akari_render_amd import abi, capi, distributed
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
rank, world, steps = (nums + [0, 8, 8][len(nums):])[:3]
ctx = capi.Context(0)
scene = capi.Scene(ctx, os.path.join(ROOT, "scenes/cbox/scene.json"), 1920, 1080)
film = capi.Film(ctx, 1920, 1080)
cfg = abi.PtConfig.default()
cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.force_diffuse = 64 * steps, 64, 12, 0 if "--full" in sys.argv else 1
cfg = distributed.shard_config(cfg, rank, world)
se = capi.PtSession(ctx, scene, cfg, film)
se.passes(steps, blocking=True)
st = se.end()
print(json.dumps({"rank": rank, "world": world, "kernel_ms": st["kernel_ms"], "n_samples": st["n_samples"]}))
