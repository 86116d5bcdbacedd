"""tests/golden/forest_1000x100k_shard.npz: the ORACLE's film of one tile shard of procedural.instanced_forest(1000, 100_000) --
99.9 M instance-triangles, which the oracle flattens (12 GB, minutes of tree building on the CPU). Generated once, here, so that the
GPU test of the two-level acceleration structure has the oracle's answer at the full size without rebuilding it on every run.
Stored sparsely (indices + bit patterns of the non-zero film floats). python tools/make_forest_golden.py [n_instances tris_per_mesh]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from akari_render_amd import capi, distributed, procedural
from oracle import pyoracle
from tests.helpers import make_config

FOREST = dict(width=512, height=288)
SHARD = (5, 64, 8, 8)                      # shard 5 of 64, 8x8 tiles: 36 tiles spread over the frame
CONFIGS = {"independent": dict(spp=8, spp_per_pass=8, max_depth=8, rr_depth=5),
           "fd_sobol": dict(spp=4, spp_per_pass=4, max_depth=6, rr_depth=5, force_diffuse=1, sampler_type=2, sampler_seed=9)}


def main():
    n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    tris = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    sd = procedural.instanced_forest(n_inst, tris, **FOREST)
    sd.ggx_table = np.fromfile(os.path.join(ROOT, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    t = time.time()
    osc = pyoracle.OracleScene(sd, bvh=True)
    print("oracle scene + tree: %.1f s" % (time.time() - t), flush=True)
    pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
    out = {}
    w, h = FOREST["width"], FOREST["height"]
    for name, kw in CONFIGS.items():
        cfg = distributed.shard_config(make_config(**kw), *SHARD)
        states = None
        if cfg.sampler_type != 0:
            states = np.zeros(2 * w * h, dtype=np.uint64)
            states[0::2] = 0xFFFFFFFF
            states[1::2] = (np.arange(w * h, dtype=np.uint64) % np.uint64(w)) | ((np.arange(w * h, dtype=np.uint64) // np.uint64(w)) << np.uint64(32))
        t = time.time()
        film, st = osc.render(cfg, states=states)
        print(name, "render: %.1f s" % (time.time() - t), st, flush=True)
        bits = film.view(np.uint32)
        nz = np.flatnonzero(bits).astype(np.uint32)
        out[name + "_idx"], out[name + "_bits"] = nz, bits[nz]
        out[name + "_stats"] = np.array([st[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded")], dtype=np.uint64)
    out["n_triangles"] = np.array([sd.n_triangles()], dtype=np.uint64)
    path = os.path.join(ROOT, "tests", "golden", "forest_%dx%dk_shard.npz" % (n_inst, tris // 1000))
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
