"""Per-scene kernels on the GPU (csrc/host/specialise.cpp; the reference's JIT, svm/compiler.rs:16-76 + svm/eval.rs:428-467): a
session whose kernel was compiled from the scene's shader graphs renders the oracle's film, sampler states and counters bit for
bit -- i.e. exactly what the interpreter kernels render -- for both intersectors, both register budgets, every sampler and
colour pipeline; the code object cache works; the fall-backs fall back."""
import os
import sys

import numpy as np
import pytest

from akari_render_amd import abi, capi
from oracle import pyoracle
from tests.helpers import make_config, n_bit_diff, textured_room
from tests.test_gpu_parity import assert_parity
from tests.test_gpu_textures import with_table

pytestmark = pytest.mark.gpu


def render_specialised(ctx, sd, cfg, waves=0, expect=True):
    """film, stats, sampler states, kernel info of a session under option specialise = 1"""
    w, h = sd.camera.width, sd.camera.height
    with capi.options(specialise=1, specialise_waves=waves):
        scene = capi.Scene(ctx, sd)
        film = capi.Film(ctx, w, h)
        se = capi.PtSession(ctx, scene, cfg, film)
    info = se.kernel_info()
    assert bool(info["specialised"]) == expect, info
    total = cfg.sample_count if cfg.sample_count else cfg.spp
    se.passes((total + cfg.spp_per_pass - 1) // cfg.spp_per_pass, blocking=True)
    states = se.sampler_states(w * h)
    st = se.end()
    return film.read(), st, states, info


@pytest.mark.parametrize("waves", [3, 4])
@pytest.mark.parametrize("variant", ["exhaustive", "bvh", "cutout", "cutout_bvh", "constant_light"])
def test_per_scene_kernel_renders_the_oracles_film(ctx, root, variant, waves):
    sd = with_table(textured_room(48, 40, n_floor=8 if "bvh" in variant else 1, alpha_cutout="cutout" in variant, textured_light=variant != "constant_light"), root)
    cfg = make_config(spp=12, spp_per_pass=4, max_depth=8)
    g, gst, gs, info = render_specialised(ctx, sd, cfg, waves)
    assert info["min_waves"] == waves and info["n_shader_kinds"] >= 3 and info["status"] == "ok"
    ostates = pyoracle.init_pcg32_states(48 * 40, cfg.sampler_seed)
    o, ost = pyoracle.OracleScene(sd).render(cfg, states=ostates)
    assert_parity(g, o, 48, 40, gst, ost)
    assert np.array_equal(gs, ostates)


def test_interpreter_and_per_scene_kernel_agree_and_the_cache_is_used(ctx, root, tmp_path, monkeypatch):
    monkeypatch.setenv("AKR_KERNEL_CACHE", str(tmp_path / "cache"))
    sd = with_table(textured_room(40, 40, n_floor=6, alpha_cutout=True), root)
    cfg = make_config(spp=8, spp_per_pass=8, max_depth=7)
    with capi.options(specialise=0):
        scene = capi.Scene(ctx, sd)
        film = capi.Film(ctx, 40, 40)
        se = capi.PtSession(ctx, scene, cfg, film)
        assert se.kernel_info()["specialised"] == 0 and "specialise = 0" in se.kernel_info()["status"]
        se.passes(1, blocking=True)
        se.end()
        ref = film.read()
    ctx1 = capi.Context(0)  # (the session-wide context may hold this kernel already: modules are cached per context)
    g1, _, _, i1 = render_specialised(ctx1, sd, cfg)
    assert n_bit_diff(g1, ref) == 0
    assert i1["cache_hit"] == 0 and i1["compile_ms"] > 0.0 and i1["vgprs"] > 0  # (hiprtc has an in-process cache of its own: not necessarily a second)
    files = os.listdir(tmp_path / "cache")
    assert len(files) == 1 and files[0].startswith("akr_") and files[0].endswith(".co")
    g2, _, _, i2 = render_specialised(ctx1, sd, cfg)  # the same context: the loaded module
    assert n_bit_diff(g2, ref) == 0 and i2["cache_hit"] == 1 and i2["compile_ms"] == 0.0
    ctx2 = capi.Context(0)  # another context of this process: the code object comes from the disk
    g3, _, _, i3 = render_specialised(ctx2, sd, cfg)
    assert n_bit_diff(g3, ref) == 0 and i3["cache_hit"] == 1 and i3["compile_ms"] == 0.0 and i3["load_ms"] > 0.0
    # a damaged file is compiled again, not trusted
    path = tmp_path / "cache" / files[0]
    path.write_bytes(b"not a code object")
    ctx3 = capi.Context(0)
    g4, _, _, i4 = render_specialised(ctx3, sd, cfg)
    assert n_bit_diff(g4, ref) == 0 and i4["cache_hit"] == 0 and i4["compile_ms"] > 0.0  # (hiprtc keeps its own in-process cache: the second compile is quick)


@pytest.mark.parametrize("sampler", [abi.SAMPLER_PMJ02BN, abi.SAMPLER_SOBOL])
def test_index_based_samplers(ctx, root, sampler):
    pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
    sd = with_table(textured_room(40, 32, n_floor=6), root)
    cfg = make_config(spp=8, spp_per_pass=4, max_depth=6, sampler_type=sampler)
    g, gst, _, _ = render_specialised(ctx, sd, cfg)
    o, ost = pyoracle.OracleScene(sd).render(cfg)
    assert_parity(g, o, 40, 32, gst, ost)


@pytest.mark.parametrize("color", [1, 2, 3])
def test_colour_pipelines_share_one_kernel(ctx, root, color):
    """The node constants are the same in every pipeline (Rgb / uplift nodes convert at run time from TexScene.color), so one module
    serves them all; the folded constants come from the pipeline's own tables."""
    sd = with_table(textured_room(36, 36), root)
    cfg = make_config(spp=8, spp_per_pass=8, max_depth=6, color=color)
    g, gst, _, _ = render_specialised(ctx, sd, cfg)
    o, ost = pyoracle.OracleScene(sd).render(cfg)
    assert_parity(g, o, 36, 36, gst, ost)


def test_long_graph_and_shared_kinds(ctx, root):
    sd = with_table(textured_room(40, 32), root)
    N = abi.NodeData
    nodes, inputs = [], {}
    for k, name in enumerate(("base_color", "roughness", "metallic", "specular_tint", "coat_weight")):
        b = len(nodes)
        nodes += [N(abi.NODE_TEXCOORDS), N(abi.NODE_EXTRACT, (b, abi.FIELD_UV)), N(abi.NODE_CONST, (), (0.1 * k, 0.05 * k, 0.0)),
                  N(abi.NODE_CONST, (), (1.0 + 0.5 * k, 2.0 - 0.2 * k, 1.0)), N(abi.NODE_MAPPING, (b + 1, b + 2, b + 3, k % 2)),
                  N(abi.NODE_IMAGE, (k % 2, b + 4, 1 if k == 0 else 0))]
        if name in ("base_color", "specular_tint"):
            nodes += [N(abi.NODE_SPECTRAL_UPLIFT, (b + 5,)), N(abi.NODE_SEPARATE_COLOR, (b + 6,))]
        else:
            nodes += [N(abi.NODE_SEPARATE_COLOR, (b + 5,)), N(abi.NODE_EXTRACT, (b + 6, k % 3))]
        inputs[name] = b + 7
    sd.materials[0].graph = abi.GraphData(nodes, inputs)
    # two walls with one graph shape and different constants: one shader kind, constants from the node records
    wall = lambda s: abi.GraphData([N(abi.NODE_TEXCOORDS), N(abi.NODE_CONST, (), (0.1, 0.2, 0.0)), N(abi.NODE_CONST, (), s), N(abi.NODE_MAPPING, (0, 1, 2, abi.MAPPING_TEXTURE)),  # noqa: E731
                                    N(abi.NODE_IMAGE, (1, 3, 0)), N(abi.NODE_SPECTRAL_UPLIFT, (4,))], {"base_color": 5})
    sd.materials[3].graph, sd.materials[4].graph = wall((0.5, 0.25, 1.0)), wall((0.3, 0.7, 1.0))
    cfg = make_config(spp=6, spp_per_pass=3, max_depth=6)
    g, gst, _, info = render_specialised(ctx, sd, cfg)
    assert info["absent_mask"] & 1 == 0  # a graph feeds coat_weight: the coat stays in the kernel
    o, ost = pyoracle.OracleScene(sd).render(cfg)
    assert_parity(g, o, 40, 32, gst, ost)


def test_fall_backs(ctx, root, tmp_path, monkeypatch):
    sd = with_table(textured_room(32, 32), root)
    # force_diffuse kernels evaluate no surface graph: the precompiled kernel, whatever the option says
    _, _, _, info = render_specialised(ctx, sd, make_config(spp=4, spp_per_pass=4, force_diffuse=1), expect=False)
    assert "force_diffuse" in info["status"]
    # an unwritable cache directory costs the cache, not the render
    monkeypatch.setenv("AKR_KERNEL_CACHE", "/proc/akari_hip_cannot_be_created")
    cfg = make_config(spp=4, spp_per_pass=4, max_depth=5)
    ctx2 = capi.Context(0)
    g, gst, _, info = render_specialised(ctx2, sd, cfg)
    o, ost = pyoracle.OracleScene(sd).render(cfg)
    assert_parity(g, o, 32, 32, gst, ost)
    # automatic mode: a render this small is not worth a compile -- but a kernel that is already cached is used
    monkeypatch.setenv("AKR_KERNEL_CACHE", str(tmp_path / "auto"))
    ctx3 = capi.Context(0)
    with capi.options(specialise=-1):
        scene = capi.Scene(ctx3, sd)
        se = capi.PtSession(ctx3, scene, cfg, capi.Film(ctx3, 32, 32))
        assert se.kernel_info()["specialised"] == 0 and "threshold" in se.kernel_info()["status"]
        se.end()
    render_specialised(ctx3, sd, cfg)
    with capi.options(specialise=-1):
        se = capi.PtSession(ctx3, scene, cfg, capi.Film(ctx3, 32, 32))
        assert se.kernel_info()["specialised"] == 1 and se.kernel_info()["cache_hit"] == 1
        se.end()
    with pytest.raises(capi.AkariError):
        capi.set_option("specialise_waves", 7)


@pytest.mark.parametrize("first", [0, 100])
def test_random_textured_scenes(ctx, root, first):
    """tools/soak.py's random scenes with random shader-graph DAGs, per-scene kernels against the oracle."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak

    pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
    table = np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    done = 0
    for seed in range(910000 + first, 910000 + first + 60):
        sd, cfg = soak.rand_scene(seed, textures=True)
        if cfg.force_diffuse:
            continue
        sd.ggx_table = table
        try:
            osc = pyoracle.OracleScene(sd)
        except Exception:
            continue
        with capi.options(specialise=1):
            try:
                scene = capi.Scene(ctx, sd)
            except capi.AkariError as e:
                assert e.code == capi.ERR_UNSUPPORTED, str(e)
                continue
            w, h = sd.camera.width, sd.camera.height
            film = capi.Film(ctx, w, h)
            se = capi.PtSession(ctx, scene, cfg, film)
        info = se.kernel_info()
        if not info["specialised"]:
            se.end()
            continue
        total = cfg.sample_count if cfg.sample_count else cfg.spp
        se.passes((total + cfg.spp_per_pass - 1) // cfg.spp_per_pass, blocking=True)
        gst = se.end()
        o, ost = osc.render(cfg)
        assert n_bit_diff(film.read(), o) == 0, f"seed {seed}"
        for k in ("n_samples", "n_closest", "n_shadow", "n_shaded"):
            assert gst[k] == ost[k], (seed, k)
        done += 1
        if done >= 12:
            break
    assert done >= 8


def test_other_integrators_keep_the_interpreter(ctx, root):
    """aov, gpt and mcmc_opt sessions share akr_pt_begin's plumbing but launch their own kernels, which interpret shader graphs: with
    per-scene kernels forced on (and one for this very scene cached by the pt render below) they must still get the interpreter's
    parameter block -- value slots in LDS and all. (A cached kernel once reached a later mcmc_opt render of the same scene.)"""
    from tests import test_gpu_aov, test_mcmc

    sd = with_table(textured_room(40, 32, alpha_cutout=True), root)
    render_specialised(ctx, sd, make_config(spp=4, spp_per_pass=4, max_depth=5))
    with capi.options(specialise=1):
        cfg = abi.AovConfig.default()
        cfg.spp, cfg.aov = 3, abi.AOV_ROUGHNESS
        test_gpu_aov.both(ctx, sd, cfg)
        test_mcmc.both(ctx, sd, test_mcmc.mcmc_config(n_chains=64, spp=6))
    test_mcmc.both(ctx, sd, test_mcmc.mcmc_config(n_chains=64, spp=6))  # automatic mode: the cached kernel is not for them either
