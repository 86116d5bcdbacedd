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
from tests.helpers import instanced_scene, make_config

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = capi.Context(0)
TABLE = np.fromfile(os.path.join(ROOT, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
bad = 0
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    sd = instanced_scene(n_inst=int(rng.integers(3, 20)), n=int(rng.integers(3, 10)), width=40, height=32, seed=seed, emissive_instances=int(rng.integers(0, 3)),
                         with_normals=bool(rng.random() < 0.5), alpha=bool(rng.random() < 0.3), textured=bool(rng.random() < 0.3))
    if rng.random() < 0.3:  # slivers
        v = sd.meshes[0].vertices.copy()
        v[:, int(rng.integers(0, 3))] *= np.float32(10.0 ** rng.uniform(-5, -1))
        sd.meshes[0].vertices = v
        sd.meshes[0].normals = None
    world = 10.0 ** rng.uniform(-3, 3)      # the whole scene's unit
    offset = rng.uniform(-1, 1, size=3) * 10.0 ** rng.uniform(0, 4) * world * float(rng.random() < 0.6)
    for k, inst in enumerate(sd.instances):
        t = np.asarray(inst.transform, dtype=np.float64).reshape(4, 4).copy()  # transposed: rows are columns
        if k >= 2 and rng.random() < 0.5:     # a blob: its own extreme, non-uniform scale (the camera still looks at the cluster)
            s3 = 10.0 ** rng.uniform(-2, 2, size=3) * rng.choice([1.0, 1.0, -1.0], size=3)
            t[:3, :3] = t[:3, :3] * s3[:, None]
            if rng.random() < 0.3:
                t[0, :3] += rng.uniform(-2, 2) * t[1, :3]
        t[:3, :3] *= world
        t[3, :3] = t[3, :3] * world + offset
        inst.transform = t.astype(np.float32).reshape(16)
    c = np.asarray(sd.camera.c2w, dtype=np.float64).reshape(4, 4).copy()
    c[3, :3] = c[3, :3] * world + offset
    sd.camera.c2w = c.astype(np.float32).reshape(16)
    sd.ggx_table = TABLE  # (both sides read the committed table)
    cfg = make_config(spp=4, spp_per_pass=4, max_depth=int(rng.integers(2, 10)), force_diffuse=int(rng.random() < 0.3), sampler_type=int(rng.integers(0, 3)))
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
        print("MISMATCH seed", seed, "floats", nd, "kinds", kinds, "world", world, "offset", offset, flush=True)
print(f"{n} scenes from seed {first}: {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
