"""Throughput of variants of the textured room (tests/helpers.py) at 1920x1080: which part of the texture machinery costs what.
python tools/textured_variants.py <variant> [steps]   variant: none | light | checker | image_nearest | image_srgb | full"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from akari_render_amd import abi, capi
from tests.helpers import textured_room

variant = sys.argv[1] if len(sys.argv) > 1 else "full"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = abi.NodeData
v = textured_room(1920, 1080, n_floor=1)
v.ggx_table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)
if variant != "full":
    keep = {"none": [], "light": [6], "checker": [0], "image_nearest": [0], "image_srgb": [0]}[variant]
    for i, m in enumerate(v.materials):
        if i not in keep:
            m.graph = None
    if 6 not in keep:
        v.materials[6].emission_color = (6.0, 6.0, 6.0)
    if variant == "checker":
        v.materials[0].graph.nodes = v.materials[0].graph.nodes[:6]
        v.materials[0].graph.inputs = {"base_color": 5}
    if variant == "image_nearest":
        v.materials[0].graph = abi.GraphData([N(abi.NODE_IMAGE, (3, abi.NODE_NONE, 0)), N(abi.NODE_SPECTRAL_UPLIFT, (0,))], {"base_color": 1})
    if variant == "image_srgb":
        v.materials[0].graph = abi.GraphData([N(abi.NODE_IMAGE, (0, abi.NODE_NONE, 1)), N(abi.NODE_SPECTRAL_UPLIFT, (0,))], {"base_color": 1})
    if variant == "none":
        v.images = []
ctx = capi.Context(0)
scene = capi.Scene(ctx, v)
film = capi.Film(ctx, 1920, 1080)
cfg = abi.PtConfig.default(); cfg.spp = 64 * (steps + 1); cfg.spp_per_pass = 64; cfg.max_depth = 12
se = capi.PtSession(ctx, scene, cfg, film)
se.passes(1, blocking=True); s0 = se.stats()
t0 = time.perf_counter(); se.passes(steps, blocking=True); t1 = time.perf_counter()
s1 = se.end()
ns = s1["n_samples"] - s0["n_samples"]
print(json.dumps({"variant": variant, "msamples_per_s": ns / (t1 - t0) / 1e6, "shaded_per_sample": (s1["n_shaded"] - s0["n_shaded"]) / ns,
                  "shadow_per_sample": (s1["n_shadow"] - s0["n_shadow"]) / ns}))
