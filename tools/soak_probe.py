"""Device material evaluation vs oracle for every material of a soak seed, dense uv sampling. python tools/soak_probe.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
import numpy as np
from akari_render_amd import abi, capi
from oracle import pyoracle
spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py")); soak = importlib.util.module_from_spec(spec); spec.loader.exec_module(soak)
seed = int(sys.argv[1])
sd, cfg = soak.rand_scene(seed)
sd.ggx_table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)
ctx = capi.Context(0)
scene = capi.Scene(ctx, sd)
osc = pyoracle.OracleScene(sd)
rng = np.random.default_rng(3)
uv = np.concatenate([rng.uniform(-7, 7, size=(100000, 2)), rng.uniform(0, 1, size=(100000, 2))]).astype(np.float32)
for mi, m in enumerate(sd.materials):
    d = capi.probe_material_inputs(ctx, scene, mi, uv)
    o = osc.material_inputs(mi, uv, 0)
    h = capi.probe_material_inputs_host(scene, mi, uv, 0)
    bad = np.nonzero((d.view(np.uint32) != o.view(np.uint32)).any(axis=1))[0]
    badh = np.nonzero((h.view(np.uint32) != o.view(np.uint32)).any(axis=1))[0]
    print("material", mi, "graph" if m.graph else "const", "device != oracle at", len(bad), "uvs; host != oracle at", len(badh))
    for i in bad[:4]:
        cols = np.nonzero(d[i].view(np.uint32) != o[i].view(np.uint32))[0]
        print("   uv", uv[i], "words", cols, "device", d[i][cols], "oracle", o[i][cols])
w, h = sd.camera.width, sd.camera.height
film = capi.Film(ctx, w, h); capi.pt_render(ctx, scene, cfg, film)
g = film.read(); of, _ = osc.render(cfg)
bad = np.nonzero(g.view(np.uint32) != of.view(np.uint32))[0]
for b in bad[:10]:
    print("film word", int(b), "pixel", (int(b) // 3) % w if b < 3 * w * h else int(b) % (w * h) % w, "gpu", g[b], "oracle", of[b], "ulp", int(g[b:b+1].view(np.int32)[0]) - int(of[b:b+1].view(np.int32)[0]))
