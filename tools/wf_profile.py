"""The wavefront schedule on the 10 M-triangle hall (or the forced-BVH cbox) for a kernel trace: python tools/wf_profile.py [hall|cbox] [passes]
rocprofv3 --kernel-trace --stats -- python tools/wf_profile.py hall 2   shows how a pass splits into k_wf_trace / k_wf_shade time."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from akari_render_amd import abi, capi, procedural
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "hall"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = capi.Context(0)
capi.set_option("force_bvh", 1)
scene = capi.Scene(ctx, procedural.sponza_like(10_000_000, 1234, 1920, 1080)) if which == "hall" else capi.Scene(ctx, os.path.join(ROOT, "scenes/cbox/scene.json"), 1920, 1080)
out = {}
for mode in ("megakernel", "wavefront"):
    capi.set_option("wavefront", 1 if mode == "wavefront" else 0)
    film = capi.Film(ctx, 1920, 1080)
    cfg = abi.PtConfig.default(); cfg.spp = 64 * passes; cfg.spp_per_pass = 64; cfg.max_depth = 12; cfg.rr_depth = 5
    cfg.force_diffuse = 1 if which == "cbox" else 0
    se = capi.PtSession(ctx, scene, cfg, film)
    t0 = time.perf_counter(); se.passes(passes, blocking=True); t1 = time.perf_counter()
    st = se.end()
    out[mode] = {"msamples_per_s": st["n_samples"] / (t1 - t0) / 1e6, "launches": st["n_launches"], "kernel_ms": st["kernel_ms"],
                 "rays_per_sample": (st["n_closest"] + st["n_shadow"]) / st["n_samples"]}
print(json.dumps(out))
