"""Finer bisect of one material's graph feeds. python tools/soak_bisect2.py <seed> <material>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
import numpy as np
from akari_render_amd import abi, capi
from oracle import pyoracle
spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py")); soak = importlib.util.module_from_spec(spec); spec.loader.exec_module(soak)
seed, mi = int(sys.argv[1]), int(sys.argv[2])
table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)
ctx = capi.Context(0)
pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
def diff(sd, cfg, show=False):
    sd.ggx_table = table
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, sd.camera.width, sd.camera.height)
    capi.pt_render(ctx, scene, cfg, film)
    o, _ = pyoracle.OracleScene(sd).render(cfg)
    g = film.read()
    bad = np.nonzero(g.view(np.uint32) != o.view(np.uint32))[0]
    if show and len(bad):
        print("     e.g. word", int(bad[0]), "gpu", g[bad[0]], "oracle", o[bad[0]])
    return len(bad)
sd, cfg = soak.rand_scene(seed)
names = list(sd.materials[mi].graph.inputs)
print("feeds", sd.materials[mi].graph.inputs)
for name in names:
    sd, cfg = soak.rand_scene(seed)
    del sd.materials[mi].graph.inputs[name]
    print("  without feed", name, "->", diff(sd, cfg))
for name in names:
    sd, cfg = soak.rand_scene(seed)
    sd.materials[mi].graph.inputs = {name: sd.materials[mi].graph.inputs[name]}
    print("  only feed", name, "->", diff(sd, cfg, True))
sd, cfg = soak.rand_scene(seed)
osc = pyoracle.OracleScene(sd)
uv = np.stack(np.meshgrid(np.linspace(-3, 3, 25), np.linspace(-3, 3, 25)), axis=-1).reshape(-1, 2).astype(np.float32)
v = osc.material_inputs(mi, uv, cfg.color)
for k, nm in ((7, "ior"), (8, "spec_level"), (6, "roughness"), (5, "metallic"), (13, "coat_w"), (14, "coat_rough"), (15, "coat_ior")):
    print("  ", nm, "range", float(np.nanmin(v[:, k])), float(np.nanmax(v[:, k])), "nan", int(np.isnan(v[:, k]).sum()))
print("   spec tint range", v[:, 9:12].min(), v[:, 9:12].max())
for c in (0, 1, 2, 3):
    sd, cfg = soak.rand_scene(seed); cfg.color = c
    print("  colour pipeline", c, "->", diff(sd, cfg))
