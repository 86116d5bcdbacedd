"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerance contract: relRMSE < 1e-3 on the resolved film at fixed seed. The build's arithmetic is
designed to be reproducible bit for bit (DESIGN.md "AKR-F32"), so these tests additionally require the raw film
accumulators to be identical -- any drift shows up as a count of differing floats before it shows up in relRMSE.
"""
import os

import numpy as np
import pytest

from akari_render_amd import abi, capi, distributed
from oracle import pyoracle, scene_json
from tests.helpers import (box_scene, cbox_variant, extreme_instanced_scene, far_modelled_mesh_scene, grid_scene, instanced_scene, make_config, n_bit_diff,
                           rel_rmse, resolve_np, shift_scene)

pytestmark = pytest.mark.gpu

REL_RMSE_TOL = 1e-3  # BASELINE.md / BASELINE.json north_star: "L2 error vs CPU reference < 1e-3"


def render_both(ctx, sd, cfg, want_states=False):
    scene = capi.Scene(ctx, sd)
    w, h = sd.camera.width, sd.camera.height
    film = capi.Film(ctx, w, h)
    if want_states:
        se = capi.PtSession(ctx, scene, cfg, film)
        se.passes((cfg.spp + cfg.spp_per_pass - 1) // cfg.spp_per_pass, blocking=True)
        gstates = se.sampler_states(w * h)
        gst = se.end()
    else:
        gst = capi.pt_render(ctx, scene, cfg, film)
        gstates = None
    g = film.read()
    osc = pyoracle.OracleScene(sd)
    ostates = pyoracle.init_pcg32_states(w * h, cfg.sampler_seed) if want_states else None
    o, ost = osc.render(cfg, states=ostates)
    return g, o, gst, ost, gstates, ostates


def assert_parity(g, o, w, h, gst=None, ost=None):
    gi, oi = resolve_np(g, w, h), resolve_np(o, w, h)
    assert np.all(np.isfinite(gi))
    err = rel_rmse(gi, oi)
    assert err < REL_RMSE_TOL, f"relRMSE {err}"
    nd = n_bit_diff(g, o)
    assert nd == 0, f"{nd} of {g.size} film floats differ (relRMSE {err:.3e}, max abs {np.max(np.abs(gi - oi)):.3e})"
    if gst is not None:
        for k in ("n_samples", "n_closest", "n_shadow", "n_shaded"):
            assert gst[k] == ost[k], k


def test_c1_cbox_256x256_64spp_full_graph(ctx, cbox_path):
    """BASELINE.json configs[0]: scenes/cbox 256x256, 64 spp, fixed seed, full shader graph."""
    sd = scene_json.load_scene(cbox_path, 256, 256)
    cfg = make_config(spp=64, spp_per_pass=64, max_depth=12, rr_depth=5)
    g, o, gst, ost, gs, os_ = render_both(ctx, sd, cfg, want_states=True)
    assert_parity(g, o, 256, 256, gst, ost)
    assert np.array_equal(gs, os_)  # per-pixel sampler states after the pass (sampler/mod.rs:168-177)
    assert gst["n_samples"] == 256 * 256 * 64


def test_c1_force_diffuse(ctx, cbox_path):
    sd = scene_json.load_scene(cbox_path, 256, 256)
    g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=64, force_diffuse=1))
    assert_parity(g, o, 256, 256, gst, ost)


def test_golden_fixture(ctx, cbox_path, root):
    gold = np.load(os.path.join(root, "tests", "golden", "cbox_64x64_16spp.npz"))
    sd = scene_json.load_scene(cbox_path, 64, 64)
    for key, fd in (("full", 0), ("force_diffuse", 1)):
        scene = capi.Scene(ctx, sd)
        film = capi.Film(ctx, 64, 64)
        capi.pt_render(ctx, scene, make_config(spp=16, spp_per_pass=16, force_diffuse=fd), film)
        assert n_bit_diff(film.read(), gold[key]) == 0


def test_scene_loaded_by_the_library(ctx, cbox_path):
    """End to end through akr_scene_load (C++ JSON reader) instead of the flat description."""
    scene = capi.Scene(ctx, cbox_path, 96, 64)
    film = capi.Film(ctx, 96, 64)
    cfg = make_config(spp=8, spp_per_pass=8)
    capi.pt_render(ctx, scene, cfg, film)
    o, _ = pyoracle.OracleScene(scene_json.load_scene(cbox_path, 96, 64)).render(cfg)
    # two independent readers of the same file (C++ in the library, Python beside the oracle): both round sin / cos of a rotation
    # correctly to f32, so the camera matrices -- and the films -- are the same bit for bit
    assert n_bit_diff(film.read(), o) == 0


def test_the_frame_bench_py_times_is_the_oracles(ctx, cbox_path):
    """bench.py builds its scene with akr_scene_load at 1920x1080 (bench.py: build_scene). One tile shard of exactly that scene
    and configuration (C2: force_diffuse, gaussian filter, depth 12) against the oracle on the Python reader's scene."""
    from akari_render_amd import distributed
    cfg = make_config(spp=16, spp_per_pass=8, max_depth=12, rr_depth=5, force_diffuse=1)
    cfg = distributed.shard_config(cfg, 3, 64)
    scene = capi.Scene(ctx, cbox_path, 1920, 1080)
    film = capi.Film(ctx, 1920, 1080)
    capi.pt_render(ctx, scene, cfg, film)
    o, _ = pyoracle.OracleScene(scene_json.load_scene(cbox_path, 1920, 1080)).render(cfg)
    assert n_bit_diff(film.read(), o) == 0


CONFIG_CASES = {
    "multi_pass_ragged": dict(spp=11, spp_per_pass=4),
    "no_nee": dict(spp=8, use_nee=0),
    "indirect_only": dict(spp=8, indirect_only=1),
    "depth0": dict(spp=4, max_depth=0),
    "depth1": dict(spp=8, max_depth=1),
    "rr_from_start": dict(spp=8, rr_depth=0, max_depth=20),
    "box_filter": dict(spp=8, filter_type=abi.FILTER_BOX, filter_radius=0.5),
    "pixel_offset": dict(spp=4, pixel_offset=(3, -2)),
    "debug_depth": dict(spp=8, debug_depth=2),
    "seed": dict(spp=8, sampler_seed=12345678901234567),
}


@pytest.mark.parametrize("name", list(CONFIG_CASES))
def test_config_variants(ctx, cbox_path, name):
    sd = scene_json.load_scene(cbox_path, 80, 56)
    kw = dict(CONFIG_CASES[name])
    cfg = make_config(**kw)
    g, o, gst, ost, gs, os_ = render_both(ctx, sd, cfg, want_states=True)
    assert_parity(g, o, 80, 56, gst, ost)
    assert np.array_equal(gs, os_)


@pytest.mark.parametrize("which", ["glass_coat", "kinds", "alpha"])
def test_material_variants(ctx, cbox_path, root, which):
    """Principled branches the stock cbox folds away (transmission, coat, specular layer, partial metal, normal
    socket), the other shader kinds (glass, diffuse, emission) and the stochastic alpha test."""
    sd = cbox_variant(scene_json.load_scene(cbox_path, 96, 96), which)
    tpath = os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32")
    sd.ggx_table = np.fromfile(tpath, dtype=np.float32) if os.path.exists(tpath) else (np.random.default_rng(0).random(4096) * 0.8).astype(np.float32)
    g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=32, spp_per_pass=16))
    assert_parity(g, o, 96, 96, gst, ost)


@pytest.mark.parametrize("normals", [False, True])
def test_bvh_scene(ctx, normals):
    """> 64 triangles: traversal of the compressed 6-wide BVH on the GPU against the oracle's exhaustive loop; instance transform with
    rotation + scale, per-triangle material slots, optional shading normals."""
    sd = grid_scene(n=24, width=96, height=64, with_normals=normals)
    scene = capi.Scene(ctx, sd)
    assert scene.info().uses_bvh == 1
    g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=16, spp_per_pass=8, max_depth=6))
    assert_parity(g, o, 96, 64, gst, ost)
    assert gst["n_node_visits"] > 0


def test_white_furnace_on_gpu(ctx):
    sd = box_scene(albedo=0.5, emission=1.0, width=16, height=16)
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, 16, 16)
    capi.pt_render(ctx, scene, make_config(spp=256, max_depth=12), film)
    img = film.resolve()
    expect = (1 - 0.5**13) / (1 - 0.5)
    assert abs(img.mean() - expect) < 0.01 * expect
    assert np.array_equal(img, resolve_np(film.read(), 16, 16))  # device resolve == film.rs:128-143


def test_sharded_films_sum_to_the_full_frame(ctx, cbox_path):
    sd = scene_json.load_scene(cbox_path, 200, 120)
    scene = capi.Scene(ctx, sd)
    cfg = make_config(spp=8, spp_per_pass=4)
    full = capi.Film(ctx, 200, 120)
    capi.pt_render(ctx, scene, cfg, full)
    acc = np.zeros(7 * 200 * 120, dtype=np.float32)
    for r in range(3):
        f = capi.Film(ctx, 200, 120)
        capi.pt_render(ctx, scene, distributed.shard_config(cfg, r, 3, 32, 16), f)
        part = f.read()
        assert np.array_equal(part[6 * 200 * 120 :] > 0, distributed.owned_pixel_mask(200, 120, r, 3, 32, 16).ravel())
        acc += part
    assert n_bit_diff(acc, full.read()) == 0


def test_progressive_passes_equal_one_shot(ctx, cbox_path):
    sd = scene_json.load_scene(cbox_path, 64, 64)
    scene = capi.Scene(ctx, sd)
    cfg = make_config(spp=24, spp_per_pass=8)
    a = capi.Film(ctx, 64, 64)
    capi.pt_render(ctx, scene, cfg, a)
    b = capi.Film(ctx, 64, 64)
    se = capi.PtSession(ctx, scene, cfg, b)
    assert se.passes(1, blocking=True) == 8
    assert se.passes(1) == 16
    assert se.passes(5, blocking=True) == 24
    st = se.end()
    assert st["n_samples"] == 64 * 64 * 24 and st["n_launches"] >= 1
    assert n_bit_diff(a.read(), b.read()) == 0


def test_full_size_properties_1080p(ctx, cbox_path):
    """BASELINE.json configs[1] geometry (1920x1080, force_diffuse) at one pass: size-independent properties."""
    sd = scene_json.load_scene(cbox_path, 1920, 1080)
    scene = capi.Scene(ctx, sd)
    cfg = make_config(spp=64, spp_per_pass=64, force_diffuse=1)
    film = capi.Film(ctx, 1920, 1080)
    st = capi.pt_render(ctx, scene, cfg, film)
    f = film.read()
    n = 1920 * 1080
    assert st["n_samples"] == n * 64
    assert np.all(f[6 * n :] == 64.0)            # weight == spp for every pixel
    assert np.all(f[3 * n : 6 * n] == 0.0)       # splat plane untouched
    assert np.all(np.isfinite(f)) and np.all(f[: 3 * n] >= 0)
    film2 = capi.Film(ctx, 1920, 1080)
    capi.pt_render(ctx, scene, cfg, film2)
    assert n_bit_diff(f, film2.read()) == 0      # deterministic
    # (exact comparison with the oracle at this resolution: tests/test_gpu_fullsize.py)


def test_procedural_hall_small(ctx):
    """The configs[3] generator at 20 k triangles: BVH traversal + six materials by slot, against the oracle."""
    from akari_render_amd import procedural

    sd = procedural.sponza_like(20_000, seed=1234, width=96, height=54)
    assert abs(sd.n_triangles() - 20_000) < 0.03 * 20_000
    g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=4, spp_per_pass=4, max_depth=5))
    assert_parity(g, o, 96, 54, gst, ost)


# ---- the wavefront schedule (wf_kernels.hip): same arithmetic, different kernels -> same bits ----
@pytest.fixture
def wavefront_mode():
    with capi.options(wavefront=1, force_bvh=1):  # akr_option_set: the library reads its environment hooks only once
        yield


@pytest.mark.parametrize("case", ["cbox_full", "cbox_diffuse", "glass_coat", "kinds", "alpha", "grid_normals", "hall", "ragged_passes", "no_nee"])
def test_wavefront_schedule_matches_oracle(ctx, cbox_path, root, wavefront_mode, case):
    cfg = make_config(spp=16, spp_per_pass=8, max_depth=8)
    if case in ("cbox_full", "cbox_diffuse", "ragged_passes", "no_nee"):
        sd = scene_json.load_scene(cbox_path, 96, 72)
        if case == "cbox_diffuse":
            cfg = make_config(spp=16, spp_per_pass=8, force_diffuse=1)
        if case == "ragged_passes":
            cfg = make_config(spp=11, spp_per_pass=4)
        if case == "no_nee":
            cfg = make_config(spp=8, use_nee=0)
    elif case in ("glass_coat", "kinds", "alpha"):
        sd = cbox_variant(scene_json.load_scene(cbox_path, 64, 64), case)
        sd.ggx_table = np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    elif case == "grid_normals":
        sd = grid_scene(n=24, width=80, height=48, with_normals=True)
    else:
        from akari_render_amd import procedural
        sd = procedural.sponza_like(20_000, seed=1234, width=96, height=54)
        cfg = make_config(spp=4, spp_per_pass=4, max_depth=5)
    g, o, gst, ost, gs, os_ = render_both(ctx, sd, cfg, want_states=True)
    assert gst["n_node_visits"] > 0  # went through the BVH / wavefront kernels
    assert_parity(g, o, sd.camera.width, sd.camera.height, gst, ost)
    assert np.array_equal(gs, os_)


@pytest.mark.parametrize("case", ["cbox_full", "hall", "hall_sobol"])
def test_a_session_that_times_both_schedules_renders_the_same_film(ctx, cbox_path, case):
    """Option sched_trial (api_pt.cpp: a long render of a large flattened scene starts with two passes under each schedule and keeps the faster one;
    1 = every session on a scene with a tree): megakernel launches and wavefront launch groups in ONE session, each starting from the sampler
    states and the film the other left -- film, sampler states and counters are the oracle's, whichever schedule the trial kept."""
    if case == "cbox_full":
        sd, cfg = scene_json.load_scene(cbox_path, 96, 72), make_config(spp=26, spp_per_pass=4, max_depth=8)
    else:
        from akari_render_amd import procedural
        sd = procedural.sponza_like(20_000, seed=1234, width=96, height=54)
        cfg = make_config(spp=12, spp_per_pass=2, max_depth=5, **({"sampler_type": abi.SAMPLER_SOBOL, "sampler_seed": 9} if case == "hall_sobol" else {}))
    w, h = sd.camera.width, sd.camera.height
    with capi.options(force_bvh=1, sched_trial=1):
        scene = capi.Scene(ctx, sd)
        film = capi.Film(ctx, w, h)
        se = capi.PtSession(ctx, scene, cfg, film)
        assert "timed trial" not in se.kernel_info()["status"]
        se.passes(1, blocking=False)                          # (a non-blocking call never runs the trial)
        se.passes(5, blocking=True)                           # the trial (4 passes) + one more
        status = se.kernel_info()["status"]
        assert "timed trial" in status and status.split()[0] in ("megakernel", "wavefront"), status
        se.passes((cfg.spp + cfg.spp_per_pass - 1) // cfg.spp_per_pass, blocking=True)  # the rest
        gs = se.sampler_states(w * h) if cfg.sampler_type == 0 else None
        gst = se.end()
    osc = pyoracle.OracleScene(sd)
    if cfg.sampler_type == 0:
        os_ = pyoracle.init_pcg32_states(w * h, cfg.sampler_seed)
        o, ost = osc.render(cfg, states=os_)
        assert np.array_equal(gs, os_)
    else:
        pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
        from tests.test_gpu_instancing import _index_states
        o, ost = osc.render(cfg, states=_index_states(w, h))
    assert_parity(film.read(), o, w, h, gst, ost)


@pytest.mark.parametrize("case", ["cbox_full", "hall"])
def test_wavefront_with_sorted_ray_queues_changes_no_bit(ctx, cbox_path, wavefront_mode, case):
    """option wf_sort (wf_sort.hip): the trace kernel reads its ray queues sorted by origin cell + direction octant. A ray's result is
    written to its own slot whatever the order it was traced in: film, sampler states and counters are the oracle's."""
    if case == "cbox_full":
        sd, cfg = scene_json.load_scene(cbox_path, 96, 72), make_config(spp=11, spp_per_pass=4, max_depth=8)
    else:
        from akari_render_amd import procedural
        sd, cfg = procedural.sponza_like(20_000, seed=1234, width=96, height=54), make_config(spp=4, spp_per_pass=4, max_depth=5)
    with capi.options(wf_sort=1):
        g, o, gst, ost, gs, os_ = render_both(ctx, sd, cfg, want_states=True)
    assert gst["n_node_visits"] > 0
    assert_parity(g, o, sd.camera.width, sd.camera.height, gst, ost)
    assert np.array_equal(gs, os_)


def test_wavefront_equals_megakernel_sharded(ctx, cbox_path, wavefront_mode):
    sd = scene_json.load_scene(cbox_path, 120, 80)
    scene = capi.Scene(ctx, sd)
    cfg = distributed.shard_config(make_config(spp=8, spp_per_pass=4), 1, 3, 32, 16)
    a = capi.Film(ctx, 120, 80)
    capi.pt_render(ctx, scene, cfg, a)
    o, _ = pyoracle.OracleScene(sd).render(cfg)
    assert n_bit_diff(a.read(), o) == 0


def test_convergence_follows_one_over_sqrt_spp(ctx, cbox_path):
    """SURVEY 8d's convergence sanity: against a 16384-spp image of the same scene the error of N-spp images falls like
    1/sqrt(N) (independent sampler: no correlation between samples), and images rendered with different seeds agree within it."""
    w = h = 64
    scene = capi.Scene(ctx, cbox_path, w, h)

    def render(spp, seed):
        film = capi.Film(ctx, w, h)
        cfg = make_config(spp=spp, spp_per_pass=min(spp, 256), max_depth=12, sampler_seed=seed)
        capi.pt_render(ctx, scene, cfg, film)
        return film.resolve().astype(np.float64)

    ref = render(16384, 1)
    errs = {n: np.sqrt(np.mean((render(n, 2) - ref) ** 2)) for n in (16, 64, 256, 1024)}
    for a, b in ((16, 64), (64, 256), (256, 1024)):
        ratio = errs[a] / errs[b]
        assert 1.6 < ratio < 2.5, (errs, ratio)  # x4 samples -> error / 2
    assert abs(render(1024, 3).mean() - ref.mean()) < 0.01 * ref.mean()


def test_bvh_balanced_fallback_builder(ctx):
    """The median-split builder that replaces an SAH tree deeper than the traversal stack (host/scene_build.cpp): same image."""
    sd = grid_scene(n=24, width=96, height=64, with_normals=True)
    with capi.options(bvh_balanced=1):
        g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=8, spp_per_pass=8, max_depth=6))
    assert_parity(g, o, 96, 64, gst, ost)


@pytest.mark.parametrize("mask", [0, 1, 3])
def test_conductor_hits_shaded_on_even_iterations_only(ctx, cbox_path, mask):
    """pt_kernels.hip (DEFER): in a scene with one metal among diffuse surfaces a hit on the metal is kept for one iteration
    when it arrives on an odd one. Per lane only the iteration changes, so the film and the sampler states stay bit-identical
    to the oracle whatever the period (mask 0 = the plain kernel, 1 = the shipped period, 3 = three iterations in four)."""
    sd = scene_json.load_scene(cbox_path, 96, 96)
    with capi.options(defer_metal=mask):
        g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=8, spp_per_pass=4, max_depth=7))
    assert_parity(g, o, 96, 96, gst, ost)


@pytest.mark.parametrize("offset", [1000.0, 10000.0])
def test_bvh_scenes_far_from_the_origin(ctx, cbox_path, root, offset):
    """A scene modelled far from the origin: the round-off of the slab test and of the triangle test grows with the coordinates, and
    the padding of the tree's boxes has to grow with it (host/scene_build.h bvh_box_padding). Before round 5 it looked at the
    scene's diagonal only and the forced-BVH Cornell box at offset 1000 lost hits (12 film floats off the oracle, 3 621 at 10 000)."""
    table = np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    cfg = make_config(spp=8, spp_per_pass=8, max_depth=8)
    for name, sd, opts in (("cbox", scene_json.load_scene(cbox_path, 64, 64), dict(force_bvh=1)), ("grid", grid_scene(n=16, width=48, height=48), dict()),
                           ("instanced flattened", instanced_scene(width=48, height=40), dict(instancing=0)),
                           ("instanced kept", instanced_scene(width=48, height=40), dict(instancing=1))):
        sd = shift_scene(sd, (offset, 2 * offset, -0.5 * offset))
        sd.ggx_table = table
        with capi.options(**opts):
            scene = capi.Scene(ctx, sd)
            assert scene.info().uses_bvh >= 1
            film = capi.Film(ctx, sd.camera.width, sd.camera.height)
            capi.pt_render(ctx, scene, cfg, film)
        o, _ = pyoracle.OracleScene(sd).render(cfg)  # the exhaustive loop: the definition of a hit
        assert n_bit_diff(film.read(), o) == 0, name


def _oracle_film(sd, cfg):
    w, h = sd.camera.width, sd.camera.height
    st = None
    if cfg.sampler_type != 0:  # index-based samplers: state = (sample index - 1, pixel coordinates)
        pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
        st = np.zeros(2 * w * h, dtype=np.uint64)
        st[0::2] = 0xFFFFFFFF
        st[1::2] = (np.arange(w * h, dtype=np.uint64) % np.uint64(w)) | ((np.arange(w * h, dtype=np.uint64) // np.uint64(w)) << np.uint64(32))
    o, _ = pyoracle.OracleScene(sd).render(cfg, states=st)
    return o


# 93 = the scene of HISTORY R5.7 (flattened: one film pixel off the oracle in round 5); the others: scenes in which tests/bvh_model.py
# found a round-5 tree culling a pair the triangle test accepts (tests/test_bvh_conservative.py)
@pytest.mark.parametrize("seed", [93, 50, 52, 63, 88, 107])
def test_needles_under_extreme_transforms(ctx, root, seed):
    """Instances under scales of 1e-4 .. 1e4, shears, offsets of 1e4 x the scene's unit, meshes squashed into slivers -- flattened and
    kept as meshes + instances, both against the oracle's exhaustive loop, bit for bit (the documented exception
    to "bit-exact" is gone; boxes follow each triangle's conditioning, scene_build.h tri_conditioning)."""
    sd, cfg = extreme_instanced_scene(seed)
    sd.ggx_table = np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    o = _oracle_film(sd, cfg)
    for mode, kind in ((0, 1), (1, 2)):
        with capi.options(instancing=mode, force_bvh=1):
            scene = capi.Scene(ctx, sd)
            assert scene.info().uses_bvh == kind
            film = capi.Film(ctx, sd.camera.width, sd.camera.height)
            capi.pt_render(ctx, scene, cfg, film)
        g = film.read()
        assert np.isfinite(g).all()
        assert n_bit_diff(g, o) == 0, ("kept" if mode else "flattened")


@pytest.mark.parametrize("offset", [1e3, 1e4])
def test_mesh_modelled_far_from_its_own_origin(ctx, root, offset):
    """a shared mesh whose vertices sit ~offset from its own origin, moved back by the instances' translations. The ray
    taken through an instance's inverse is then uncertain by ulp(offset); the per-mesh trees' padding has to follow (scene_inst.cpp)."""
    sd = far_modelled_mesh_scene(offset, width=48, height=40)
    sd.ggx_table = np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    cfg = make_config(spp=8, spp_per_pass=8, max_depth=8)
    o = _oracle_film(sd, cfg)
    for mode in (0, 1):
        with capi.options(instancing=mode):
            scene = capi.Scene(ctx, sd)
            film = capi.Film(ctx, 48, 40)
            capi.pt_render(ctx, scene, cfg, film)
        assert n_bit_diff(film.read(), o) == 0, ("kept" if mode else "flattened")
