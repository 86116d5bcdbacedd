"""The oracle's own checker-side additions (oracle/or_accel.h), pinned on the CPU:

* the oracle-only BVH returns exactly what the exhaustive loop returns (hit, triangle, bits of t / u / v; closest and any hit;
  whole films) -- it is used for the 1 M / 10 M-triangle parity cases where the exhaustive loop would take hours;
* the build's triangle test (Woop form, f32) agrees with an independent f64 Moeller-Trumbore intersector written from the
  vertices on 10^6 rays, random and aimed at edges / vertices. The reference's intersector (Embree through LuisaCompute,
  crates/akari_render/src/scene.rs:88-110) is not in the tree; this is the cross-check that stands in for it.
"""
import numpy as np
import pytest

from oracle import pyoracle, scene_json
from tests.helpers import cbox_variant, check_against_mt_f64, grid_scene, make_config, n_bit_diff, probe_rays


def _scenes(cbox_path):
    from akari_render_amd import procedural

    return {
        "cbox": scene_json.load_scene(cbox_path, 32, 32),
        "cbox_alpha": cbox_variant(scene_json.load_scene(cbox_path, 32, 32), "alpha"),
        "grid": grid_scene(n=24, width=32, height=32, with_normals=True),
        "hall20k": procedural.sponza_like(20_000, seed=1234, width=48, height=27),
    }


@pytest.mark.parametrize("name", ["cbox", "cbox_alpha", "grid", "hall20k"])
def test_oracle_bvh_returns_the_exhaustive_hits(cbox_path, name):
    sd = _scenes(cbox_path)[name]
    ex = pyoracle.OracleScene(sd)
    acc = pyoracle.OracleScene(sd, bvh=True)
    assert acc.n_bvh_nodes > 0
    n = 60_000 if name == "hall20k" else 200_000
    rays = probe_rays(ex.world_vertices(), n // 2, n // 2, seed=11)
    # a tenth of the rays with a finite range (shadow-ray like)
    rays[::10, 7] = np.random.default_rng(5).random(rays[::10].shape[0]).astype(np.float32) * 3.0
    for any_hit in (False, True):
        a, ta = ex.intersect_many(rays, any_hit)
        b, tb = acc.intersect_many(rays, any_hit)
        if any_hit:  # which occluder is found first depends on the order; whether one exists does not
            assert np.array_equal(a[:, 0], b[:, 0])
        else:
            assert np.array_equal(a, b)
            assert n_bit_diff(ta, tb) == 0
    assert 0.05 < a[:, 0].mean() < 1.0


@pytest.mark.parametrize("name", ["grid", "hall20k"])
def test_oracle_bvh_film_is_the_exhaustive_film(cbox_path, name):
    sd = _scenes(cbox_path)[name]
    cfg = make_config(spp=4, spp_per_pass=2, max_depth=6)
    f0, s0 = pyoracle.OracleScene(sd).render(cfg)
    f1, s1 = pyoracle.OracleScene(sd, bvh=True).render(cfg)
    assert n_bit_diff(f0, f1) == 0
    for k in ("n_samples", "n_closest", "n_shadow", "n_shaded"):
        assert s0[k] == s1[k]
    assert s1["n_tri_tests"] < s0["n_tri_tests"] / 10


@pytest.mark.parametrize("name,n", [("cbox", 1_000_000), ("grid", 300_000)])
def test_triangle_test_agrees_with_f64_moeller_trumbore(cbox_path, name, n):
    """or_tri_test (the arithmetic contract the HIP kernels share) against textbook f64 Moeller-Trumbore."""
    sd = _scenes(cbox_path)[name]
    osc = pyoracle.OracleScene(sd)
    rays = probe_rays(osc.world_vertices(), n // 2, n // 2, seed=3)
    out, tuv = osc.intersect_many(rays)
    off = osc.tri_offsets()
    gid = np.where(out[:, 0] == 1, off[out[:, 1]] + out[:, 2], 0xFFFFFFFF).astype(np.uint32)
    mt_gid, mt = osc.mt_f64(rays)
    _, mt_own = osc.mt_f64(rays, gids=gid)
    r = check_against_mt_f64(osc.world_vertices(), rays, out[:, 0], gid, tuv, mt_gid, mt, mt_own)
    print(r)
    assert r["unexplained"] == 0, r
    assert r["max_dt_scaled"] < 1e-6 and r["max_du_scaled"] < 1e-6 and r["max_dv_scaled"] < 1e-6, r
    # half of the rays are AIMED at edges and vertices: only there do the two disagree (which of two abutting triangles
    # owns an edge point, or a hit / miss by the last bit at a silhouette edge)
    assert r["same_triangle"] + r["both_miss"] > 0.85 * n, r
    assert r["only_f32"] + r["only_f64"] < 0.03 * n, r


def test_random_scenes_bvh_equals_the_exhaustive_loop_and_host_graphs_equal_the_oracle(hip_lib):
    """tools/soak.py's generator on the CPU: (1) the oracle's own BVH (test infrastructure for the full-size cases) renders what its
    exhaustive loop renders, bit for bit, on scenes with degenerate / tiny / duplicated geometry; (2) the library's host-side shader
    graph evaluation and folding -- the code the kernels run -- equals the oracle's for random graph DAGs in every colour pipeline."""
    import importlib.util
    import os

    from akari_render_amd import capi

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("soak", os.path.join(root, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
    for seed in range(700000, 700120):
        sd, cfg = soak.rand_scene(seed)
        a, sa = pyoracle.OracleScene(sd, bvh=False).render(cfg)
        b, sb = pyoracle.OracleScene(sd, bvh=True).render(cfg)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), seed
        assert all(sa[k] == sb[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded")), seed
    n = 0
    for seed in range(710000, 710250):
        sd, cfg = soak.rand_scene(seed, True)
        sc = capi.Scene(None, sd)
        osc = pyoracle.OracleScene(sd)
        uv = np.random.default_rng(seed).uniform(-3, 4, size=(32, 2)).astype(np.float32)
        for mi in range(len(sd.materials)):
            x, y = capi.probe_material_inputs_host(sc, mi, uv, cfg.color), osc.material_inputs(mi, uv, cfg.color)
            nx, ny = np.isnan(x), np.isnan(y)  # (NaN payloads are not part of the contract)
            assert np.array_equal(nx, ny) and np.array_equal(x[~nx].view(np.uint32), y[~ny].view(np.uint32)), (seed, mi)
            n += 1
    assert n > 500
