"""First light of the two-level traversal: instanced scenes rendered (a) kept as meshes + instances, (b) flattened, (c) by the oracle.
Prints the number of film floats that differ pairwise. python tools/inst_first_light.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from akari_render_amd import capi
from oracle import pyoracle
from tests.helpers import instanced_scene, make_config, n_bit_diff

ctx = capi.Context(0)
pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
cases = [("plain", dict(n_inst=6, n=4, with_normals=False, with_uvs=False, mirror=False, emissive_instances=0)),
         ("mirror", dict(n_inst=6, n=4, with_normals=False, with_uvs=False, mirror=True, emissive_instances=0)),
         ("normals_uvs", dict(n_inst=9, n=5, emissive_instances=0)),
         ("lights", dict(n_inst=12, n=6, emissive_instances=2)),
         ("alpha", dict(n_inst=12, n=6, emissive_instances=1, alpha=True))]
for name, kw in cases:
    sd = instanced_scene(width=32, height=32, **kw)
    for fd in (1, 0):
        for sampler in (0, 1):
            cfg = make_config(spp=8, spp_per_pass=8, force_diffuse=fd, sampler_type=sampler)
            films = {}
            for mode in (1, 0):
                with capi.options(instancing=mode):
                    scene = capi.Scene(ctx, sd)
                    assert scene.info().uses_bvh == (2 if mode else scene.info().uses_bvh)
                    film = capi.Film(ctx, 32, 32)
                    t0 = time.time()
                    st = capi.pt_render(ctx, scene, cfg, film)
                    films[mode] = (film.read(), st, time.time() - t0)
            ostates = None
            if sampler:
                ostates = np.zeros(2 * 32 * 32, dtype=np.uint64)
                ostates[0::2] = 0xFFFFFFFF
                ostates[1::2] = (np.arange(1024, dtype=np.uint64) % np.uint64(32)) | ((np.arange(1024, dtype=np.uint64) // np.uint64(32)) << np.uint64(32))
            o, ost = pyoracle.OracleScene(sd).render(cfg, states=ostates)
            a, b = films[1][0], films[0][0]
            print(f"{name:12s} fd={fd} sampler={sampler}: inst vs flat {n_bit_diff(a, b):6d}  flat vs oracle {n_bit_diff(b, o):6d}  inst vs oracle {n_bit_diff(a, o):6d} of {a.size}"
                  f"  closest {films[1][1]['n_closest']} / {films[0][1]['n_closest']} / {ost['n_closest']}", flush=True)
