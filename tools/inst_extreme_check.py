"""Kept scenes under transforms far outside the soak's range: log-uniform scales 1e-4 .. 1e4 (non-uniform, mirrored, sheared), offsets up
to 1e4, instances of meshes squashed into slivers; kept vs flattened films on the GPU, bit for bit (the flattened scene is pinned to
the oracle elsewhere). What this stresses is the CULLING of the two-level structure -- the object-space padding of the per-mesh trees
(host/scene_inst.cpp) and the conservative reject (dinst.h) -- where a wrong decision silently drops a hit.
python tools/inst_extreme_check.py [n_scenes first_seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from akari_render_amd import capi
from tests.helpers import extreme_instanced_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = capi.Context(0)
TABLE = np.fromfile(os.path.join(ROOT, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
bad = 0
t0 = time.time()
for seed in range(first, first + n):
    sd, cfg = extreme_instanced_scene(seed)
    sd.ggx_table = TABLE  # (both sides read the committed table)
    films, kinds = [], []
    try:
        for mode in (1, 0):
            with capi.options(instancing=mode):
                scene = capi.Scene(ctx, sd)
                kinds.append(scene.info().uses_bvh)
                film = capi.Film(ctx, 40, 32)
                capi.pt_render(ctx, scene, cfg, film)
                films.append(film.read())
    except capi.AkariError as e:
        print("seed", seed, "refused:", str(e)[:100], flush=True)
        continue
    nd = int(np.count_nonzero(films[0].view(np.uint32) != films[1].view(np.uint32)))
    if "oracle" in sys.argv:  # which of the two agrees with the oracle's exhaustive loop
        from oracle import pyoracle
        pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
        st = None
        if cfg.sampler_type != 0:
            st = np.zeros(2 * 40 * 32, dtype=np.uint64); st[0::2] = 0xFFFFFFFF
            st[1::2] = (np.arange(1280, dtype=np.uint64) % np.uint64(40)) | ((np.arange(1280, dtype=np.uint64) // np.uint64(40)) << np.uint64(32))
        o, _ = pyoracle.OracleScene(sd).render(cfg, states=st)
        print("seed", seed, "kept vs oracle", int(np.count_nonzero(films[0].view(np.uint32) != o.view(np.uint32))), "flattened vs oracle",
              int(np.count_nonzero(films[1].view(np.uint32) != o.view(np.uint32))), flush=True)
    if nd or kinds != [2, kinds[1]] or not np.isfinite(films[0]).all():
        bad += 1
        print("MISMATCH seed", seed, "floats", nd, "kinds", kinds, flush=True)
print(f"{n} scenes from seed {first}: {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
