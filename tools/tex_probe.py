"""Where the textured room's time goes: scene-level variants of tools/textured_bench.py's room (measurement only)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from akari_render_amd import abi, capi
from tests.helpers import textured_room

rng = np.random.default_rng(0)
big8 = rng.integers(0, 256, size=(4096, 4096, 4), dtype=np.uint8); big8[:, :, 3] = 255
bigf = rng.random((2048, 2048, 4)).astype(np.float32); bigf[:, :, 2] = 0.5 + 0.5 * bigf[:, :, 2]; bigf[:, :, 3] = 1.0
table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)

def room(**kw):
    sd = textured_room(1920, 1080, **kw)
    sd.images[0] = abi.ImageData(big8, abi.TEX_FILTER_LINEAR, abi.TEX_REPEAT)
    sd.images[1] = abi.ImageData(bigf, abi.TEX_FILTER_LINEAR, abi.TEX_MIRROR)
    sd.ggx_table = table
    return sd

def v_no_srgb():
    sd = room()
    for nd in sd.materials[1].graph.nodes:
        if nd.op == abi.NODE_IMAGE: nd.args = (nd.args[0], nd.args[1], 0)
    return sd
def v_nearest():
    sd = room()
    sd.images[0] = abi.ImageData(big8, abi.TEX_FILTER_NEAREST, abi.TEX_REPEAT)
    sd.images[1] = abi.ImageData(bigf, abi.TEX_FILTER_NEAREST, abi.TEX_MIRROR)
    return sd
def v_only(keep):
    sd = room(textured_light=(6 in keep))
    for i, m in enumerate(sd.materials):
        if i not in keep and i != 6: m.graph = None
    if 0 not in keep: sd.materials[0].metallic = 0.25
    return sd

variants = [("textured", room), ("constant light", lambda: room(textured_light=False)), ("no srgb decode", v_no_srgb), ("nearest filters", v_nearest),
            ("only the floor textured", lambda: v_only({0})), ("only the back wall textured", lambda: v_only({1})),
            ("only the left wall textured", lambda: v_only({2})), ("only the light textured", lambda: v_only({6}))]
ctx = capi.Context(0)
out = {}
for name, make in variants:
    sd = make()
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, 1920, 1080)
    cfg = abi.PtConfig.default(); cfg.spp = 64 * 4; cfg.spp_per_pass = 64; cfg.max_depth = 12
    se = capi.PtSession(ctx, scene, cfg, film)
    se.passes(1, blocking=True); s0 = se.stats()
    t0 = time.perf_counter(); se.passes(3, blocking=True); t1 = time.perf_counter()
    s1 = se.end()
    out[name] = round((s1["n_samples"] - s0["n_samples"]) / (t1 - t0) / 1e6, 1)
    print(out[name], name, flush=True)
print(json.dumps(out))
