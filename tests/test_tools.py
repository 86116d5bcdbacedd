"""The measurement tools that DESIGN.md's numbers lean on still build and run (no GPU)."""
import json
import os
import subprocess

import numpy as np


def test_bvh_simulator_builds_and_agrees_with_itself(tmp_path, root):
    """tools/bvh_sim.cpp links the library's own builder (host/bvh.cpp) and mirrors the device traversal: on a small random grid the
    device's octant order and an exact distance order must find the same hits (same number of path rays), the exact order never
    visiting more nodes."""
    exe = tmp_path / "bvh_sim"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "akari_render_amd", "csrc"), os.path.join(root, "tools", "bvh_sim.cpp"),
                    os.path.join(root, "akari_render_amd", "csrc", "host", "bvh.cpp"), "-o", str(exe)], check=True)
    # a closed 30 x 12 x 15 box (the hall's shell) of displaced grids, ~12 k triangles, as 9 floats per triangle
    rng = np.random.default_rng(3)
    tris = []
    L, H, W, n = 30.0, 12.0, 15.0, 32
    def grid(fn):
        u, v = np.meshgrid(np.linspace(0, 1, n + 1), np.linspace(0, 1, n + 1), indexing="xy")
        p = fn(u, v) + rng.normal(0, 0.01, size=(n + 1, n + 1, 3))
        a, b, c, d = p[:-1, :-1], p[:-1, 1:], p[1:, 1:], p[1:, :-1]
        tris.append(np.stack([a, b, c], 2).reshape(-1, 9))
        tris.append(np.stack([a, c, d], 2).reshape(-1, 9))
    grid(lambda u, v: np.stack([L * (u - 0.5), 0 * u, W * (v - 0.5)], -1))
    grid(lambda u, v: np.stack([L * (u - 0.5), H + 0 * u, W * (v - 0.5)], -1))
    grid(lambda u, v: np.stack([L * (u - 0.5), H * v, -W / 2 + 0 * u], -1))
    grid(lambda u, v: np.stack([L * (u - 0.5), H * v, W / 2 + 0 * u], -1))
    grid(lambda u, v: np.stack([-L / 2 + 0 * u, H * v, W * (u - 0.5)], -1))
    grid(lambda u, v: np.stack([L / 2 + 0 * u, H * v, W * (u - 0.5)], -1))
    path = tmp_path / "tris.f32"
    np.concatenate(tris).astype(np.float32).tofile(path)
    a = json.loads(subprocess.run([str(exe), str(path), "1500"], check=True, stdout=subprocess.PIPE, text=True).stdout)
    b = json.loads(subprocess.run([str(exe), str(path), "1500"], check=True, stdout=subprocess.PIPE, text=True, env=dict(os.environ, SIM_SORT="1")).stdout)
    assert a["n_tris"] == 12 * n * n and a["depth"] >= 3 and a["node_slots"] > 100
    assert a["rays"] == b["rays"] > 5000                     # the same paths: both orders find the same hits
    assert b["closest"]["nodes"] <= a["closest"]["nodes"] and 3.0 < a["nodes_per_ray"] < 40.0 and 0.5 < a["tris_per_ray"] < 12.0
