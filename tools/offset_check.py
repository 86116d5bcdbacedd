"""Scenes modelled far from the origin: the forced-BVH Cornell box, the heightfield grid and an instanced scene (flattened and kept)
shifted, camera included, by (d, 2 d, -d / 2) for d = 0, 1e2, 1e3, 1e4, each against the oracle's exhaustive loop, bit for bit.
What found the padding defect of HISTORY R5.7 (before the fix: cbox at d = 1e3 differed in 12 film floats, at 1e4 in 3 621).
python tools/offset_check.py   (needs a GPU; uses oracle/)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from akari_render_amd import capi
from oracle import pyoracle, scene_json
from tests.helpers import instanced_scene, grid_scene, make_config
ctx = capi.Context(0)
pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
table = np.fromfile(os.path.join(ROOT, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
def shift(sd, off):
    for inst in sd.instances:
        t = np.asarray(inst.transform, dtype=np.float32).reshape(4, 4).copy(); t[3, :3] += np.float32(off); inst.transform = t.reshape(16)
    c = np.asarray(sd.camera.c2w, dtype=np.float32).reshape(4, 4).copy(); c[3, :3] += np.float32(off); sd.camera.c2w = c.reshape(16)
    return sd
def run(name, sd, force_bvh=0):
    sd.ggx_table = table
    w, h = sd.camera.width, sd.camera.height
    cfg = make_config(spp=8, spp_per_pass=8, max_depth=8)
    res = {}
    for mode in (0, 1):
        with capi.options(instancing=mode, force_bvh=force_bvh):
            sc = capi.Scene(ctx, sd); film = capi.Film(ctx, w, h); capi.pt_render(ctx, sc, cfg, film); res[mode] = (film.read(), sc.info().uses_bvh)
    o, _ = pyoracle.OracleScene(sd).render(cfg)           # exhaustive loop: the definition
    d = lambda a, b: int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32)))
    print(f"{name}: flat(uses_bvh {res[0][1]}) vs oracle {d(res[0][0], o)}, kept(uses_bvh {res[1][1]}) vs oracle {d(res[1][0], o)}, kept vs flat {d(res[0][0], res[1][0])}", flush=True)
for off in (0.0, 100.0, 1000.0, 10000.0):
    run(f"cbox forced bvh offset {off}", shift(scene_json.load_scene(os.path.join(ROOT, "scenes", "cbox", "scene.json"), 64, 64), (off, 2 * off, -0.5 * off)), force_bvh=1)
    run(f"grid offset {off}", shift(grid_scene(n=16, width=48, height=48), (off, 2 * off, -0.5 * off)))
    run(f"instanced offset {off}", shift(instanced_scene(width=48, height=40), (off, 2 * off, -0.5 * off)))
