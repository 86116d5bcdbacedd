"""Where a soak seed differs: closest-hit probes (camera-like and random rays) GPU vs oracle, then per-pixel film differences.
python tools/soak_debug.py <seed> [tex]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
import numpy as np
from akari_render_amd import abi, capi
from oracle import pyoracle
spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py")); soak = importlib.util.module_from_spec(spec); spec.loader.exec_module(soak)
seed = int(sys.argv[1])
sd, cfg = soak.rand_scene(seed, True if "tex" in sys.argv[2:] else None)
sd.ggx_table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)
ctx = capi.Context(0)
pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
scene = capi.Scene(ctx, sd)
osc = pyoracle.OracleScene(sd)
rng = np.random.default_rng(1)
n = 200000
o = rng.uniform(-1.5, 1.5, size=(n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d, np.zeros((n, 1)), np.full((n, 1), 1e20)], axis=1).astype(np.float32)
gh, gb = capi.probe_intersect(ctx, scene, rays)
oh, ot = osc.intersect_many(rays)
diff = np.nonzero((gh != oh).any(axis=1) | (gb.view(np.uint32) != ot[:, 1:].view(np.uint32)).any(axis=1))[0]
print("probe rays differing:", len(diff), "of", n)
for i in diff[:8]:
    print("  ray", i, "gpu", gh[i], gb[i], "oracle", oh[i], ot[i])
w, h = sd.camera.width, sd.camera.height
film = capi.Film(ctx, w, h)
st = capi.pt_render(ctx, scene, cfg, film)
g = film.read(); of, ost = osc.render(cfg)
bad = np.nonzero(g.view(np.uint32) != of.view(np.uint32))[0]
print("film floats differing:", len(bad), {k: (int(st[k]), int(ost[k])) for k in ("n_closest", "n_shadow", "n_shaded")})
px = sorted(set(int(b) // 3 if b < 3 * w * h else int(b) % (w * h) for b in bad))
print("  pixels", [(p % w, p // w) for p in px][:12])
for depth in range(1, cfg.max_depth + 1):  # the first depth at which the images part
    c = abi.PtConfig.from_buffer_copy(bytes(cfg)); c.max_depth = depth
    f2 = capi.Film(ctx, w, h); capi.pt_render(ctx, scene, c, f2)
    o2, _ = osc.render(c)
    print("  max_depth", depth, "floats differing", int(np.count_nonzero(f2.read().view(np.uint32) != o2.view(np.uint32))))
for nee in (0, 1):
    c = abi.PtConfig.from_buffer_copy(bytes(cfg)); c.use_nee = nee
    f2 = capi.Film(ctx, w, h); capi.pt_render(ctx, scene, c, f2); o2, _ = osc.render(c)
    print("  use_nee", nee, "floats differing", int(np.count_nonzero(f2.read().view(np.uint32) != o2.view(np.uint32))))
for fd in (0, 1):
    c = abi.PtConfig.from_buffer_copy(bytes(cfg)); c.force_diffuse = fd
    f2 = capi.Film(ctx, w, h); capi.pt_render(ctx, scene, c, f2); o2, _ = osc.render(c)
    print("  force_diffuse", fd, "floats differing", int(np.count_nonzero(f2.read().view(np.uint32) != o2.view(np.uint32))))
os.environ["AKR_FORCE_BVH"] = "1"
scene2 = capi.Scene(ctx, sd)
f2 = capi.Film(ctx, w, h); capi.pt_render(ctx, scene2, cfg, f2)
osc2 = pyoracle.OracleScene(sd)
print("  forced BVH: floats differing", int(np.count_nonzero(f2.read().view(np.uint32) != of.view(np.uint32))))
