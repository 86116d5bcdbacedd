"""Where does the full-graph kernel's time go? The 1080p Cornell box with (a) the stock materials, (b) the tall box made
diffuse (no metal lobe anywhere: the same kernel, no material divergence), (c) force_diffuse (the specialised kernel)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from akari_render_amd import abi, capi
from oracle import scene_json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ctx = capi.Context(0)
out = {}
for name in ("stock", "tallbox_diffuse", "all_metal", "force_diffuse"):
    sd = scene_json.load_scene(os.path.join(ROOT, "scenes", "cbox", "scene.json"), 1920, 1080)
    if name == "tallbox_diffuse":
        sd.materials[sd.material_names.index("tallBox_001")].metallic = 0.0
    if name == "all_metal":
        for m in sd.materials:
            if m.emission_strength == 0: m.metallic, m.roughness = 1.0, 0.3
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, 1920, 1080)
    cfg = abi.PtConfig.default()
    cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth = 2 * 256, 64, 12, 5
    cfg.force_diffuse = 1 if name == "force_diffuse" else 0
    se = capi.PtSession(ctx, scene, cfg, film)
    se.passes(4, blocking=True); s0 = se.stats()
    se.passes(4, blocking=True); s1 = se.end()
    ms = s1["kernel_ms"] - s0["kernel_ms"]; n = s1["n_samples"] - s0["n_samples"]
    out[name] = {"msamples_per_s": n / ms / 1e3, "closest_per_sample": (s1["n_closest"] - s0["n_closest"]) / n, "shaded_per_sample": (s1["n_shaded"] - s0["n_shaded"]) / n}
    print(name, out[name], flush=True)
print(json.dumps(out))
