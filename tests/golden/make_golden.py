"""Regenerates the committed fixtures under tests/golden/. Run in the build container:

    python tests/golden/make_golden.py [--table]

Inputs are DATA files of the reference (/root/reference/scenes/cbox/{scene.json,Scene.bin}; the copies under
scenes/cbox/ are byte-identical) read by the independent Python reader, and outputs of the CPU oracle
(oracle/akr_oracle.c). Nothing here imports or executes reference code: the reference cannot be built or run in
this environment (no Rust toolchain, LuisaCompute absent), so these fixtures pin the ORACLE, not the reference.

  cbox_flat.npz                 flattened scenes/cbox (corner vertices, transforms, folded materials, camera)
  cbox_64x64_16spp.npz          oracle film (reference layout) of cbox, full materials and force_diffuse
  ggx_dielectric_s.f32          (--table, ~2 CPU-minutes on 8 cores) the 16^3 albedo table, 2^20 samples / entry
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from akari_render_amd import abi  # noqa: E402
from oracle import pyoracle, scene_json  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = "/root/reference/scenes/cbox/scene.json"
    if not os.path.exists(src):
        src = os.path.join(ROOT, "scenes", "cbox", "scene.json")
    sd = scene_json.load_scene(src)
    verts = np.concatenate([sd.meshes[i.mesh].vertices[sd.meshes[i.mesh].indices].reshape(-1, 3) for i in sd.instances])
    np.savez(os.path.join(HERE, "cbox_flat.npz"), corner_vertices=verts, transforms=np.stack([i.transform for i in sd.instances]),
             materials=np.array([bytes(m.to_struct()) for m in sd.materials]), c2w=sd.camera.c2w, fov=np.float32(sd.camera.fov))
    sd = scene_json.load_scene(src, 64, 64)
    sc = pyoracle.OracleScene(sd)
    films = {}
    for fd in (0, 1):
        cfg = abi.PtConfig.default()
        cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth, cfg.force_diffuse = 16, 16, 12, 5, fd
        film, _ = sc.render(cfg)
        films["force_diffuse" if fd else "full"] = film
    np.savez_compressed(os.path.join(HERE, "cbox_64x64_16spp.npz"), **films)
    if "--table" in sys.argv:
        from concurrent.futures import ThreadPoolExecutor
        L = pyoracle.lib()
        def entry(g):
            return L.or_ggx_dielectric_table_entry(g % 16, (g // 16) % 16, g // 256, 1 << 20)
        with ThreadPoolExecutor(os.cpu_count()) as ex:
            tab = np.array(list(ex.map(entry, range(4096))), dtype=np.float32)
        tab.tofile(os.path.join(HERE, "ggx_dielectric_s.f32"))


if __name__ == "__main__":
    main()
