"""The measurement tools that DESIGN.md's numbers lean on still build and run (no GPU)."""
import json
import os
import subprocess
import sys

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


def test_valu_cost_report_prices_the_measured_cases(tmp_path, root):
    """tools/valu_cost_report.py on a hand-written listing: the parity rule of the register file (three source VGPRs of one parity
    double an fma), compares, selects, the canonicalising v_max x, x, transcendentals and packed ops get the measured costs."""
    listing = tmp_path / "k.s"
    listing.write_text("\n".join([
        "\tv_fma_f32 v10, v0, v1, v2",          # even, odd, even: free (2.3)
        "\tv_fma_f32 v10, v0, v2, v4",          # all even: 4.4
        "\tv_fmac_f32_e32 v5, v1, v3",          # sources v1, v3, v5: all odd: 4.4
        "\tv_fmac_f32_e32 v4, v1, v3",          # mixed: 2.3
        "\tv_fma_f32 v10, v0, s4, v1",          # SGPR source: 4.4
        "\tv_mul_f32_e32 v6, v0, v2",           # 2-source: 2.2 whatever the registers
        "\tv_cmp_lt_f32_e32 vcc, v0, v1",       # 4.4
        "\tv_cndmask_b32_e32 v7, v0, v1, vcc",  # 3.7
        "\tv_max_f32_e32 v8, v3, v3",           # canonicalise: 4.4
        "\tv_rcp_f32_e32 v9, v3",               # 8.2
        "\tv_pk_fma_f32 v[12:13], v[0:1], v[2:3], v[4:5]",  # 4.4
        "\ts_add_u32 s0, s0, 1",                # no VALU cost
        "\tds_read_b128 v[20:23], v30",
    ]) + "\n")
    out = tmp_path / "r.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "valu_cost_report.py"), str(listing), "1", "13", "--json", str(out)], check=True, stdout=subprocess.PIPE)
    r = json.load(open(out))
    assert r["valu_instructions"] == 11
    want = 2.3 + 4.4 + 4.4 + 2.3 + 4.4 + 2.2 + 4.4 + 3.7 + 4.4 + 8.2 + 4.4
    assert abs(r["modelled_cycles"] - want) < 1e-9
    assert r["classes"]["3-VGPR-source op, all sources in one register bank (same parity)"]["instructions"] == 2
