"""The three samplers on the same frame, same box: cbox 1920x1080, force_diffuse (C2) and full graph (C3), one launch of 16 passes
x 64 spp timed after a warm-up launch. spp of the render = 2048 (a power of two: the index permutation's cycle walk never repeats)
and 25600 (what bench.py's 25 steps give). python tools/sampler_bench.py [c2|c3|both]   (needs a GPU)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from akari_render_amd import abi, capi

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
ctx = capi.Context(0)
scene = capi.Scene(ctx, os.path.join(ROOT, "scenes/cbox/scene.json"), 1920, 1080)
out = {}
for fd in ([1] if which == "c2" else [0] if which == "c3" else [1, 0]):
    for spp in (2048, 25600):
        for name, smp in (("independent", abi.SAMPLER_INDEPENDENT), ("sobol", abi.SAMPLER_SOBOL), ("pmj02bn", abi.SAMPLER_PMJ02BN)):
            film = capi.Film(ctx, 1920, 1080)
            cfg = abi.PtConfig.default()
            cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth, cfg.force_diffuse, cfg.sampler_type = spp, 64, 12, 5, fd, smp
            cfg.filter_type, cfg.filter_radius = abi.FILTER_GAUSSIAN, 1.5
            with capi.options(max_fused_passes=16):
                se = capi.PtSession(ctx, scene, cfg, film)
            se.passes(16, blocking=True); s0 = se.stats()
            t0 = time.perf_counter(); se.passes(16, blocking=True); t1 = time.perf_counter()
            s1 = se.end()
            key = f"{'c2' if fd else 'c3'} spp={spp} {name}"
            out[key] = {"msamples_per_s": (s1["n_samples"] - s0["n_samples"]) / (t1 - t0) / 1e6, "kernel_ms": s1["kernel_ms"] - s0["kernel_ms"]}
            print(key, round(out[key]["msamples_per_s"], 1), flush=True)
for k in list(out):
    if "independent" in k:
        base = out[k]["msamples_per_s"]
        for other in ("sobol", "pmj02bn"):
            k2 = k.replace("independent", other)
            out[k2]["vs_independent"] = out[k2]["msamples_per_s"] / base
print(json.dumps(out))
