"""Quick perf + bit-parity check on the GPU box: python tools/perf_quick.py [steps] [--full]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from akari_render_amd import abi, capi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
ctx = capi.Context(0)
gold = np.load(os.path.join(ROOT, "tests/golden/cbox_64x64_16spp.npz"))
from oracle import scene_json
sd = scene_json.load_scene(os.path.join(ROOT, "scenes/cbox/scene.json"), 64, 64)
ok = True
for key, fd in (("full", 0), ("force_diffuse", 1)):
    cfg = abi.PtConfig.default(); cfg.spp = cfg.spp_per_pass = 16; cfg.max_depth = 12; cfg.force_diffuse = fd
    scene = capi.Scene(ctx, sd); film = capi.Film(ctx, 64, 64)
    capi.pt_render(ctx, scene, cfg, film)
    nd = int(np.count_nonzero(film.read().view(np.uint32) != gold[key].view(np.uint32)))
    ok &= nd == 0
    print("golden", key, "diff floats:", nd, flush=True)
for fd in ((1, 0) if "--full" in sys.argv else (1,)):
    scene = capi.Scene(ctx, os.path.join(ROOT, "scenes/cbox/scene.json"), 1920, 1080)
    film = capi.Film(ctx, 1920, 1080)
    cfg = abi.PtConfig.default(); cfg.spp = 64 * (steps + 1); cfg.spp_per_pass = 64; cfg.max_depth = 12; cfg.force_diffuse = fd
    se = capi.PtSession(ctx, scene, cfg, film)
    se.passes(1, blocking=True); s0 = se.stats()
    t0 = time.perf_counter(); se.passes(steps, blocking=True); t1 = time.perf_counter()
    s1 = se.end()
    ns = s1["n_samples"] - s0["n_samples"]
    print(json.dumps({"force_diffuse": fd, "msamples_per_s": ns / (t1 - t0) / 1e6, "kernel_ms": s1["kernel_ms"] - s0["kernel_ms"], "bit_exact_vs_golden": ok}), flush=True)
