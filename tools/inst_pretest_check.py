"""Run with AKR_HIP_LIB=akari_render_amd/variants/libakari_hip_instcheck.so (build.build_variant("instcheck", ["-DAKR_INST_PRETEST_CHECK=1"],
only=["pt_inst_kernels.hip"])): every candidate of a kept scene takes the exact test and a candidate the conservative reject
(dinst.h tri_may_hit) would have dropped although the exact test accepts it fails the render. Scenes chosen to stress the bound:
instances far from the origin, tiny and huge scales, sliver triangles, mirrored and sheared transforms, grazing views."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from akari_render_amd import abi, capi, procedural
from tests.helpers import instanced_scene, make_config

ctx = capi.Context(0)
n_cand = 0


def run(name, sd, **kw):
    global n_cand
    with capi.options(instancing=1):
        scene = capi.Scene(ctx, sd)
        assert scene.info().uses_bvh == 2
        film = capi.Film(ctx, sd.camera.width, sd.camera.height)
        try:
            st = capi.pt_render(ctx, scene, make_config(**{**dict(spp=16, spp_per_pass=16, max_depth=8), **kw}), film)
        except capi.AkariError as e:
            print(f"{name}: VIOLATION ({e})", flush=True)
            return False
    n_cand += st["n_tri_tests"]
    print(f"{name}: ok, {st['n_tri_tests']} candidates", flush=True)
    return True


def stress(seed, offset, scale, sliver):
    rng = np.random.default_rng(seed)
    sd = instanced_scene(n_inst=24, n=8, width=96, height=64, seed=seed, emissive_instances=2)
    if sliver:  # squash the blob's vertices so that most triangles become slivers
        v = sd.meshes[0].vertices.copy()
        v[:, 1] *= np.float32(sliver)
        sd.meshes[0].vertices = v
        sd.meshes[0].normals = None
    for inst in sd.instances:
        t = np.asarray(inst.transform, dtype=np.float32).reshape(4, 4).copy()  # transposed: rows are columns
        t[:3, :3] *= np.float32(scale)
        t[3, :3] = t[3, :3] * np.float32(scale) + np.float32(offset)
        if rng.random() < 0.3:  # a shear
            t[0, :3] += np.float32(0.7) * t[1, :3]
        inst.transform = t.reshape(16)
    c = np.asarray(sd.camera.c2w, dtype=np.float32).reshape(4, 4).copy()
    c[3, :3] = c[3, :3] * np.float32(scale) + np.float32(offset)
    sd.camera.c2w = c.reshape(16)
    return sd


ok = True
ok &= run("forest 1000 x 10k", procedural.instanced_forest(1000, 10_000, width=480, height=270), spp=4, spp_per_pass=4)
ok &= run("forest 200 x 200k", procedural.instanced_forest(200, 200_000, width=480, height=270), spp=4, spp_per_pass=4)
for seed, (offset, scale, sliver) in enumerate([(0, 1, 0), (1000, 1, 0), (-5000, 1, 0), (0, 1e-3, 0), (0, 1e3, 0), (300, 0.01, 0), (0, 1, 1e-3), (0, 1, 1e-5), (100, 1, 1e-4),
                                                (2e4, 10, 0), (0, 1e-6, 0), (0, 1e5, 1e-2)]):
    for fd in (0, 1):
        ok &= run(f"stress offset={offset} scale={scale} sliver={sliver} fd={fd}", stress(seed + 20, offset, scale, sliver), force_diffuse=fd)
for kw in (dict(textured=True, alpha=True), dict(with_normals=False, with_uvs=False)):
    ok &= run(f"small {kw}", instanced_scene(n_inst=40, n=12, width=96, height=64, emissive_instances=3, **kw))
print("candidates checked:", n_cand, "ALL OK" if ok else "VIOLATIONS")
sys.exit(0 if ok else 1)
