"""Edge cases of the hot path on the GPU against the oracle: tiny and ragged frames, empty shards, scenes without lights,
degenerate triangles, zero-sample renders, the 4K film."""
import numpy as np
import pytest

from akari_render_amd import abi, capi, distributed
from oracle import pyoracle, scene_json
from tests.helpers import box_scene, grid_scene, make_config, n_bit_diff
from tests.test_gpu_parity import assert_parity, render_both

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (33, 9), (70, 41)])
def test_tiny_and_ragged_frames(ctx, cbox_path, w, h):
    """Frames smaller than a wave, not a multiple of the 8x8 lane blocks or of the 32x32 tiles."""
    sd = scene_json.load_scene(cbox_path, w, h)
    g, o, gst, ost, gs, os_ = render_both(ctx, sd, make_config(spp=9, spp_per_pass=4, force_diffuse=1), want_states=True)
    assert_parity(g, o, w, h, gst, ost)
    assert np.array_equal(gs, os_)


def test_more_shards_than_tiles(ctx, cbox_path):
    """40x40 = 4 tiles over 7 ranks: three ranks own nothing, the films still add up to the unsharded one."""
    sd = scene_json.load_scene(cbox_path, 40, 40)
    scene = capi.Scene(ctx, sd)
    cfg = make_config(spp=8, spp_per_pass=8)
    full = capi.Film(ctx, 40, 40)
    capi.pt_render(ctx, scene, cfg, full)
    total = np.zeros(7 * 40 * 40, dtype=np.float32)
    owned = []
    for r in range(7):
        f = capi.Film(ctx, 40, 40)
        st = capi.pt_render(ctx, scene, distributed.shard_config(cfg, r, 7), f)
        owned.append(st["n_samples"])
        total += f.read()
    assert owned.count(0) == 3 and sum(owned) == 40 * 40 * 8
    assert n_bit_diff(total, full.read()) == 0


def test_scene_without_lights(ctx):
    sd = box_scene(albedo=0.6, emission=0.0, width=24, height=24)
    scene = capi.Scene(ctx, sd)
    assert scene.info().n_lights == 0
    g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=4, max_depth=6))
    assert_parity(g, o, 24, 24, gst, ost)
    assert gst["n_shadow"] == 0 and float(np.abs(g[: 3 * 24 * 24]).max()) == 0.0


def test_degenerate_triangles_are_never_hit(ctx):
    sd = grid_scene(n=6, width=40, height=32)  # 72 + 2 triangles -> BVH path
    m = sd.meshes[0]
    idx = m.indices.copy()
    idx[5] = [idx[5][0], idx[5][0], idx[5][1]]   # zero-area triangle (repeated vertex)
    idx[11] = [idx[11][0], idx[11][1], idx[11][1]]
    m.indices = idx
    scene = capi.Scene(ctx, sd)
    assert scene.info().uses_bvh == 1
    g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=8))
    assert_parity(g, o, 40, 32, gst, ost)


def test_zero_samples_and_config_validation(ctx, cbox_path):
    scene = capi.Scene(ctx, cbox_path, 16, 16)
    film = capi.Film(ctx, 16, 16)
    st = capi.pt_render(ctx, scene, make_config(spp=0, spp_per_pass=4), film)
    assert st["n_samples"] == 0 and not film.read().any()
    for bad in (dict(spp_per_pass=0), dict(filter_type=7), dict(sampler_type=3), dict(tile_w=12)):
        cfg = make_config(spp=4)
        for k, v in bad.items():
            setattr(cfg, k, v)
        with pytest.raises(capi.AkariError):
            capi.pt_render(ctx, scene, cfg, film)
    with pytest.raises(capi.AkariError):  # film / scene resolution mismatch
        capi.pt_render(ctx, scene, make_config(spp=4), capi.Film(ctx, 8, 16))


def test_4k_film_one_pass(ctx, cbox_path):
    """BASELINE configs[4] frame size (3840x2160, 232 MB film): every pixel takes its samples and the values are finite (the
    size-independent properties; bit parity is asserted at sizes the oracle renders in seconds)."""
    W, H = 3840, 2160
    scene = capi.Scene(ctx, cbox_path, W, H)
    film = capi.Film(ctx, W, H)
    cfg = make_config(spp=2, spp_per_pass=2, force_diffuse=1)
    st = capi.pt_render(ctx, scene, cfg, film)
    f = film.read()
    assert st["n_samples"] == W * H * 2 and np.all(f[6 * W * H :] == 2.0) and np.all(np.isfinite(f))
    rgb = f[: 3 * W * H].reshape(H, W, 3)
    assert rgb[H // 2 - 100 : H // 2 + 100, W // 2 - 100 : W // 2 + 100].mean() > 0.05  # the box is in the middle of the frame


@pytest.mark.parametrize("n_extra", [0, 80], ids=["exhaustive", "bvh"])
def test_extreme_triangle_scales_hit_like_the_oracle(ctx, n_extra):
    """Triangles from 1e-18 to 1e+12 units across, some far from the origin: the coefficients of the precomputed transform
    span 1e-24 .. 1e+36 and the hit point's coordinates overflow for some rays. Hit / miss, triangle and barycentrics must
    still be the oracle's (the comparison chain of the oracle and the min-margin of the kernels treat inf / NaN alike)."""
    rng = np.random.default_rng(5)
    verts, idx = [], []
    sizes = [1e-18, 1e-12, 1e-6, 1e-3, 1.0, 1e3, 1e6, 1e12] + [1.0] * n_extra
    for s in sizes:
        c = rng.normal(size=3) * (1.0 if s < 1e3 else s) + (1e8 if s == 1e-6 else 0.0)
        a, b = rng.normal(size=3) * s, rng.normal(size=3) * s
        k = len(verts)
        verts += [c, c + a, c + b]
        idx.append([k, k + 1, k + 2])
    sd = box_scene()
    sd.meshes[0] = abi.MeshData(vertices=np.array(verts, dtype=np.float32), indices=np.array(idx, dtype=np.uint32))
    scene = capi.Scene(ctx, sd)
    assert scene.info().uses_bvh == (1 if n_extra else 0)
    osc = pyoracle.OracleScene(sd)
    v32 = np.array(verts, dtype=np.float32).reshape(-1, 3, 3)
    n = 1536
    rays = np.zeros((n, 8), dtype=np.float32)
    for i in range(n):  # aim at a point of a random triangle from a random origin, so that small triangles are hit at all
        t = v32[rng.integers(0, len(sizes))]
        w = rng.dirichlet((1, 1, 1)) if i % 3 else np.array([1.0, 0.0, 0.0])  # every third ray through a vertex
        target = (w[:, None] * t.astype(np.float64)).sum(0)
        o = target + rng.normal(size=3) * max(1.0, float(np.abs(t).max()) * 1e-3)
        d = target - o
        rays[i, :3], rays[i, 3:6], rays[i, 6], rays[i, 7] = o, d / np.linalg.norm(d), 0.0, 1e20
    hit, bary = capi.probe_intersect(ctx, scene, rays)
    n_hit = 0
    for i in range(n):
        h, inst, prim, b = osc.intersect(rays[i, :3], rays[i, 3:6], 0.0, 1e20)
        assert bool(hit[i, 0]) == h, (i, rays[i])
        if h:
            n_hit += 1
            assert int(hit[i, 2]) == prim and np.array_equal(bary[i].view(np.uint32), b.view(np.uint32)), (i, rays[i])
    assert n_hit > n // 4


def test_small_mesh_with_a_long_material_list_takes_the_bvh_path(ctx):
    """The exhaustive kernels stage a small scene's tables in LDS; a 12-triangle box with 300 materials (75 KB of records) does
    not fit and is rendered through the BVH instead -- same film."""
    import copy

    sd = box_scene(albedo=0.5, emission=0.8, width=24, height=24)
    assert capi.Scene(ctx, sd).info().uses_bvh == 0
    for k in range(299):
        m = copy.deepcopy(sd.materials[0])
        m.base_color = (0.2 + 0.002 * k, 0.5, 0.7 - 0.002 * k)
        sd.materials.append(m)
    mesh = sd.meshes[0]
    mesh.material_slots = (np.arange(12, dtype=np.uint32) * 23) % 300
    sd.instances[0].materials = list(range(300))
    scene = capi.Scene(ctx, sd)
    assert scene.info().uses_bvh == 1
    g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=8, max_depth=6))
    assert_parity(g, o, 24, 24, gst, ost)


def test_random_scene_soak(ctx):
    """tools/soak.py: random triangle soups and grid patches (degenerate, tiny, huge, coplanar neighbours), materials drawn from
    edge values of every input, random emitters / instances / cameras / samplers / configs / colour pipelines, exhaustive and BVH
    paths -- films and counters bit for bit against the oracle. 150 seeds here (10 400 were run in round 2, none differed)."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("soak", os.path.join(root, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    table = np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
    n_bvh = 0
    # the seeds that found something in round 2 first: 22901 / 22603 (forced graphs; constant alpha next to an opaque texture-fed base
    # colour), 201578 / 201749 (a graph feeding inf / NaN into specular_tint of a layer of weight 0)
    cases = [(22901, True), (22603, True), (201578, None), (201749, None)] + [(seed, None) for seed in range(100000, 100150)]
    for seed, textures in cases:
        sd, cfg = soak.rand_scene(seed, textures)
        sd.ggx_table = table
        scene = capi.Scene(ctx, sd)
        n_bvh += int(scene.info().uses_bvh != 0)
        film = capi.Film(ctx, sd.camera.width, sd.camera.height)
        st = capi.pt_render(ctx, scene, cfg, film)
        o, ost = pyoracle.OracleScene(sd).render(cfg)
        assert n_bit_diff(film.read(), o) == 0, f"seed {seed}"
        assert all(int(st[k]) == int(ost[k]) for k in ("n_samples", "n_closest", "n_shadow", "n_shaded")), f"seed {seed}"
    assert 10 < n_bvh < 144
