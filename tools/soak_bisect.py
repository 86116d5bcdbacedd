"""Which ingredient of a soak seed makes it differ: variants of the scene, mismatch count each. python tools/soak_bisect.py <seed>"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
import numpy as np
from akari_render_amd import abi, capi
from oracle import pyoracle
spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py")); soak = importlib.util.module_from_spec(spec); spec.loader.exec_module(soak)
seed = int(sys.argv[1])
table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)
ctx = capi.Context(0)
pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())

def diff(sd, cfg):
    sd.ggx_table = table
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, sd.camera.width, sd.camera.height)
    capi.pt_render(ctx, scene, cfg, film)
    o, _ = pyoracle.OracleScene(sd).render(cfg)
    return int(np.count_nonzero(film.read().view(np.uint32) != o.view(np.uint32)))

def fresh():
    return soak.rand_scene(seed)

sd, cfg = fresh(); print("as generated", diff(sd, cfg))
sd, cfg = fresh()
seen, keep = set(), []
for i in sd.instances:
    key = (i.mesh, i.transform.tobytes())
    if key not in seen: keep.append(i); seen.add(key)
print("instances", len(sd.instances), "->", len(keep)); sd.instances = keep; print("  without duplicate instances", diff(sd, cfg))
sd, cfg = fresh()
for m in sd.materials: m.base_alpha = 1.0
print("  constant alphas = 1", diff(sd, cfg))
sd, cfg = fresh()
for im in sd.images:
    if im.texels.dtype == np.uint8: im.texels[:, :, 3] = 255
    else: im.texels[:, :, 3] = 1.0
print("  opaque images", diff(sd, cfg))
for k in range(len(soak.rand_scene(seed)[0].materials)):
    sd, cfg = fresh()
    if sd.materials[k].graph is None: continue
    print("  material", k, "graph", [(n.op, n.args) for n in sd.materials[k].graph.nodes], sd.materials[k].graph.inputs)
    sd.materials[k].graph = None
    print("    without it:", diff(sd, cfg))
sd, cfg = fresh()
for m in sd.materials: m.normal = (0.0, 0.0, 0.0)
print("  no constant normal maps", diff(sd, cfg))
sd, cfg = fresh()
for m in sd.materials: m.metallic = 0.0
print("  no metal", diff(sd, cfg))
sd, cfg = fresh()
for m in sd.materials: m.coat_weight = 0.0
print("  no coat", diff(sd, cfg))
sd, cfg = fresh()
for m in sd.materials: m.transmission_weight = 0.0
print("  no transmission", diff(sd, cfg))
sd, cfg = fresh(); cfg.sampler_type = abi.SAMPLER_INDEPENDENT; print("  independent sampler", diff(sd, cfg))
sd, cfg = fresh(); cfg.color = 0; print("  default colour pipeline", diff(sd, cfg))
sd, cfg = fresh()
print("  materials:", [(m.kind, m.metallic, m.roughness, m.transmission_weight, m.coat_weight, m.emission_strength, m.normal) for m in sd.materials])
