"""The `gpt` integrator (akari_integrator/src/gpt.rs + the reconnection shift mapping inside pt.rs:329-900): the oracle's
estimator against the plain path tracer (CPU), and the HIP kernels against the oracle, film bit for bit (GPU)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from akari_render_amd import abi, capi
from oracle import pyoracle, scene_json
from tests.helpers import cbox_variant, grid_scene, n_bit_diff, rel_rmse, textured_room


def table(root):
    return np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)


def gpt_config(**kw) -> abi.GptConfig:
    c = abi.GptConfig.default()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def oracle_image(osc, cfg):
    film, aux = osc.gpt_render(cfg)
    w, h = osc.width, osc.height
    return film, aux, pyoracle.resolve(film, w, h, 1.0 / cfg.spp if cfg.reconstruction == abi.GPT_RECON_NONE else 1.0)


def test_gpt_config_default_and_method_json(hip_lib):
    """gpt::Config::default (gpt.rs:48-65); {"type": "gpt"} goes to akr_render_task, not to the pt-config entry point."""
    c = abi.GptConfig()
    assert hip_lib.akr_gpt_config_default(C.byref(c)) == 0
    assert bytes(c) == bytes(abi.GptConfig.default()) and C.sizeof(abi.GptConfig) == 80 == pyoracle.lib().or_sizeof_gpt_config()
    assert (c.spp, c.max_depth, c.rr_depth, c.spp_per_pass, c.reconnect, c.stride, c.reconstruction, c.reconstruction_iter) == (256, 7, 5, 64, 1, 1, 0, 30)
    cfg = abi.PtConfig()
    assert hip_lib.akr_pt_config_from_json(b'{"method": {"type": "gpt"}}', C.byref(cfg), None, 0) == capi.ERR_UNSUPPORTED


def test_oracle_gpt_estimates_the_same_image_as_pt(cbox_path, root):
    """Every variant of the estimator (MIS-combined splats, uniform / weighted reconstruction, separate weights, stride 2,
    no reconnection) converges to the path tracer's image; with the same number of base paths its error is lower."""
    w = h = 40
    sd = scene_json.load_scene(cbox_path, w, h)
    sd.ggx_table = table(root)
    osc = pyoracle.OracleScene(sd)
    pc = abi.PtConfig.default()
    pc.spp, pc.spp_per_pass, pc.max_depth = 2048, 64, 7
    ref = pyoracle.resolve(osc.render(pc)[0], w, h)
    pc.spp = 48
    pt_img = pyoracle.resolve(osc.render(pc)[0], w, h)
    err_pt = rel_rmse(pt_img, ref)
    variants = [dict(), dict(separate_weights=1), dict(reconstruction=abi.GPT_RECON_UNIFORM), dict(reconstruction=abi.GPT_RECON_WEIGHTED),
                dict(reconstruction=abi.GPT_RECON_WEIGHTED, separate_weights=1), dict(stride=2),
                dict(reconstruction=abi.GPT_RECON_UNIFORM, reconnect=0)]
    for kw in variants:
        cfg = gpt_config(spp=48, **kw)
        film, aux, img = oracle_image(osc, cfg)
        assert np.all(np.isfinite(img))
        # With a reconstruction the gradient images' first row / column hold one half of the MIS pair only (the other half
        # would come from a pixel outside the image, gpt.rs:336-346 + 424-461), so the reference's reconstruction is off
        # along those two borders; the comparison stays away from them.
        a, b = (img, ref) if cfg.reconstruction == abi.GPT_RECON_NONE else (img[12:, 12:], ref[12:, 12:])
        assert abs(a.mean() - b.mean()) < 0.04 * b.mean(), (kw, a.mean(), b.mean())
        assert rel_rmse(a, b) < (err_pt if cfg.reconstruction == abi.GPT_RECON_NONE else rel_rmse(pt_img[12:, 12:], b)), (kw, rel_rmse(a, b), err_pt)
        n = w * h
        assert not film[: 3 * n].any() and not film[6 * n :].any()  # gpt writes the splat channels only
        if cfg.reconstruction != abi.GPT_RECON_NONE:
            primal, gx, gy = aux
            assert abs(primal.mean() / cfg.spp - ref.mean()) < 0.04 * ref.mean()
            assert not gx[:, w].any() and not gx[h].any() and not gy[:, w].any() and not gy[h].any()  # never accumulated (gpt.rs:424-461)
            # the accumulated x-gradients are the finite differences of the image, up to noise
            fd = ref[:, 1:] - ref[:, :-1]
            assert np.abs(gx[:h, 1:w] / cfg.spp - fd).mean() < 0.5 * np.abs(fd).mean() + 0.02
    # configurations the reference cannot run
    for bad in (dict(reconnect=0), dict(stride=0), dict(stride=w), dict(sampler_type=abi.SAMPLER_PMJ02BN)):
        with pytest.raises(AssertionError):
            osc.gpt_render(gpt_config(spp=1, **bad))


def both(ctx, sd, cfg):
    scene = capi.Scene(ctx, sd)
    w, h = sd.camera.width, sd.camera.height
    film = capi.Film(ctx, w, h)
    recon = cfg.reconstruction != abi.GPT_RECON_NONE
    res = capi.gpt_render(ctx, scene, cfg, film, want_aux=recon)
    o_film, o_aux = pyoracle.OracleScene(sd).gpt_render(cfg)
    g = film.read()
    if recon:
        for name, a, b in zip(("primal", "gx", "gy"), res[1], o_aux):
            assert n_bit_diff(a, b) == 0, f"{name} sums: {n_bit_diff(a, b)} floats differ"
        assert film.splat_scale == 1.0
    else:
        assert film.splat_scale == np.float32(1.0) / np.float32(cfg.spp)
    assert n_bit_diff(g, o_film) == 0, f"{n_bit_diff(g, o_film)} of {g.size} film floats differ"
    assert np.array_equal(film.resolve(), pyoracle.resolve(o_film, w, h, film.splat_scale))
    return g


SCENES = ["cbox", "glass_coat", "kinds", "grid_bvh", "textured"]


def make_scene(name, cbox_path, root, w=48, h=36):
    if name == "cbox":
        sd = scene_json.load_scene(cbox_path, w, h)
    elif name in ("glass_coat", "kinds"):
        sd = cbox_variant(scene_json.load_scene(cbox_path, w, h), name)
    elif name == "grid_bvh":
        sd = grid_scene(n=12, width=w, height=h, with_normals=True)
    else:
        sd = textured_room(w, h, alpha_cutout=True)
    sd.ggx_table = table(root)
    return sd


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", SCENES)
@pytest.mark.parametrize("recon", range(3), ids=abi.GPT_RECON_NAMES)
def test_gpt_parity(ctx, cbox_path, root, scene_name, recon):
    sd = make_scene(scene_name, cbox_path, root)
    cfg = gpt_config(spp=6, max_depth=6, rr_depth=2, reconstruction=recon, reconstruction_iter=7)
    g = both(ctx, sd, cfg)
    assert g[3 * 48 * 36 : 6 * 48 * 36].any()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["separate_none", "separate_weighted", "stride3", "no_reconnect", "no_nee", "indirect_only", "box_filter_seed", "deep"])
def test_gpt_parity_options(ctx, cbox_path, root, case):
    kw = {"separate_none": dict(separate_weights=1), "separate_weighted": dict(separate_weights=1, reconstruction=abi.GPT_RECON_WEIGHTED),
          "stride3": dict(stride=3, reconstruction=abi.GPT_RECON_UNIFORM), "no_reconnect": dict(reconnect=0, reconstruction=abi.GPT_RECON_UNIFORM),
          "no_nee": dict(use_nee=0), "indirect_only": dict(indirect_only=1),
          "box_filter_seed": dict(filter_type=abi.FILTER_BOX, filter_radius=0.5, sampler_seed=(1 << 33) + 5),
          "deep": dict(max_depth=14, rr_depth=9, spp=3)}[case]
    sd = make_scene("kinds" if case in ("deep", "stride3") else "cbox", cbox_path, root, 40, 40)
    base = dict(spp=5, max_depth=7, rr_depth=3, reconstruction_iter=4)
    base.update(kw)
    both(ctx, sd, gpt_config(**base))


@pytest.mark.gpu
def test_gpt_rejections_and_render_task(ctx, cbox_path, tmp_path, monkeypatch):
    sd = scene_json.load_scene(cbox_path, 32, 32)
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, 32, 32)
    for bad, code in ((dict(reconnect=0), capi.ERR_INVALID_ARGUMENT), (dict(stride=0), capi.ERR_INVALID_ARGUMENT), (dict(stride=32), capi.ERR_INVALID_ARGUMENT),
                      (dict(reconstruction=3), capi.ERR_INVALID_ARGUMENT), (dict(sampler_type=abi.SAMPLER_PMJ02BN), capi.ERR_UNSUPPORTED)):
        with pytest.raises(capi.AkariError) as e:
            capi.gpt_render(ctx, scene, gpt_config(spp=1, **bad), film)
        assert e.value.code == code, bad
    # the method-file route, with the debug images the reference writes next to a reconstruction (gpt.rs:609-636)
    monkeypatch.chdir(tmp_path)
    method = {"method": {"type": "gpt", "spp": 4, "max_depth": 5, "reconstruction": "weighted", "reconstruction_iter": 5},
              "film": {"out": "gpt.exr", "filter": {"type": "gaussian", "radius": 1.5}}}
    capi.render_task(ctx, scene, json.dumps(method))
    from tests.test_output_stage import read_exr_rgb

    img = read_exr_rgb(str(tmp_path / "gpt.exr"))
    cfg = gpt_config(spp=4, max_depth=5, reconstruction=abi.GPT_RECON_WEIGHTED, reconstruction_iter=5)
    film.clear()
    _, (primal, gx, gy) = capi.gpt_render(ctx, scene, cfg, film, want_aux=True)
    assert np.array_equal(img, film.resolve())
    assert np.array_equal(read_exr_rgb(str(tmp_path / "output" / "gpt_primal.exr")), primal * np.float32(0.25))
    assert np.array_equal(read_exr_rgb(str(tmp_path / "output" / "gpt_gx.exr")), gx * np.float32(0.25))
    assert np.array_equal(read_exr_rgb(str(tmp_path / "output" / "gpt_gy.exr")), gy * np.float32(0.25))
    # pmj02bn in the method file: refused (the reference's sampler cannot be cloned), accepted with the CLI override
    method["sampler"] = {"type": "pmj02bn", "seed": 0}
    with pytest.raises(capi.AkariError):
        capi.render_task(ctx, scene, json.dumps(method))


@pytest.mark.gpu
def test_gpt_and_mcmc_at_1080p_agree_with_the_path_tracer(ctx, cbox_path, root):
    """Full frame size (BASELINE configs' 1920x1080), through a size-independent property: the three estimators agree on the
    image's mean radiance per channel, and gpt's splat bookkeeping conserves it (sum over the frame of the MIS-combined
    contributions = sum of the base paths' radiance in expectation)."""
    from tests.test_mcmc import mcmc_config

    w, h = 1920, 1080
    sd = scene_json.load_scene(cbox_path, w, h)
    sd.ggx_table = table(root)
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, w, h)
    pc = abi.PtConfig.default()
    pc.spp, pc.spp_per_pass, pc.max_depth = 16, 16, 7
    capi.pt_render(ctx, scene, pc, film)
    ref = film.resolve().reshape(-1, 3).astype(np.float64).mean(0)
    film.clear()
    capi.gpt_render(ctx, scene, gpt_config(spp=4), film)
    img = film.resolve()
    assert np.all(np.isfinite(img)) and film.splat_scale == 0.25
    g = img.reshape(-1, 3).astype(np.float64).mean(0)
    assert np.all(np.abs(g - ref) < 0.02 * ref), (g, ref)
    film.clear()
    film.splat_scale = 1.0
    st, res, chains = capi.mcmc_render(ctx, scene, mcmc_config(spp=8, n_chains=262144, n_bootstrap=100000, direct_spp=8), film)
    m = film.resolve().reshape(-1, 3).astype(np.float64).mean(0)
    assert res["n_mutations"] >= w * h * 8 - 262144 and 0.5 < res["acceptance_rate"] < 0.99
    assert np.all(np.abs(m - ref) < 0.04 * ref), (m, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("color", [abi.COLOR_REPR_ACESCG, abi.COLOR_RGB_ACESCG | abi.COLOR_REPR_ACESCG])
@pytest.mark.parametrize("recon", [abi.GPT_RECON_NONE, abi.GPT_RECON_WEIGHTED], ids=["none", "weighted"])
def test_gpt_in_a_non_default_colour_pipeline(ctx, cbox_path, root, recon, color):
    """akr_gpt_config.color: materials folded for the pipeline, every splat converted to the film's sRGB primaries (film.rs:167-194)."""
    sd = make_scene("textured" if recon == abi.GPT_RECON_NONE else "cbox", cbox_path, root, 40, 32)
    g = both(ctx, sd, gpt_config(spp=4, max_depth=5, rr_depth=2, reconstruction=recon, reconstruction_iter=3, color=color))
    g0 = both(ctx, sd, gpt_config(spp=4, max_depth=5, rr_depth=2, reconstruction=recon, reconstruction_iter=3))
    assert n_bit_diff(g, g0) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("recon", range(3), ids=abi.GPT_RECON_NAMES)
@pytest.mark.parametrize("case", ["cbox_s1", "kinds_s3", "grid_bvh_s2"])
def test_gpt_sharded_across_ranks_is_the_single_gpu_render(ctx, cbox_path, root, recon, case):
    """akr_gpt_begin(.., akr_shard, ..): three ranks (here: three sessions on one GPU) each sample their 16x8 tiles plus the halo
    whose offset paths land in them and fold their own pixels; the sum of the three films (reconstruction none) or of the three
    sets of primal / gradient sums followed by ONE reconstruction (uniform, weighted) is the unsharded render bit for bit: every
    film / sum entry is written by exactly one rank, and a halo pixel draws the same numbers on every rank that samples it."""
    name, stride = {"cbox_s1": ("cbox", 1), "kinds_s3": ("kinds", 3), "grid_bvh_s2": ("grid_bvh", 2)}[case]
    w, h = 56, 40   # ragged: 3.5 x 5 tiles
    sd = make_scene(name, cbox_path, root, w, h)
    cfg = gpt_config(spp=4, max_depth=6, rr_depth=2, reconstruction=recon, reconstruction_iter=5, stride=stride)
    scene = capi.Scene(ctx, sd)
    full = capi.Film(ctx, w, h)
    se = capi.GptSession(ctx, scene, cfg, full)
    se.sample()
    sums_full = se.read_sums()
    se.finish()
    ref = capi.Film(ctx, w, h)
    capi.gpt_render(ctx, scene, cfg, ref)       # the one-call entry point is the same thing
    assert n_bit_diff(full.read(), ref.read()) == 0 and full.splat_scale == ref.splat_scale
    world = 3
    films, sums, sessions = [], [], []
    for r in range(world):
        f = capi.Film(ctx, w, h)
        s = capi.GptSession(ctx, scene, cfg, f, rank=r, world=world, tile_w=16, tile_h=8)
        s.sample(2)
        s.sample()       # in two steps: the session keeps its place
        films.append(f)
        sums.append(s.read_sums())
        sessions.append(s)
    if recon == abi.GPT_RECON_NONE:
        for s in sessions:
            s.finish()
        total = np.zeros_like(films[0].read())
        for f in films:
            total = total + f.read()
        assert n_bit_diff(total, full.read()) == 0
        from akari_render_amd import distributed
        for r, f in enumerate(films):   # and nothing outside a rank's own tiles
            own = distributed.owned_pixel_mask(w, h, r, world, 16, 8)
            splat = f.read()[3 * w * h:6 * w * h].reshape(h, w, 3)
            assert not splat[~own].any() and splat[own].any()
    else:
        acc = np.zeros_like(sums[0])
        for a in sums:
            acc = acc + a
        assert n_bit_diff(acc, sums_full) == 0
        sessions[0].write_sums(acc)   # what akr_gpt_reduce leaves on the root
        for s in sessions:
            s.finish()
        assert n_bit_diff(films[0].read(), full.read()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("recon", range(3), ids=abi.GPT_RECON_NAMES)
def test_gpt_reduce_through_rccl_with_a_world_of_one(ctx, cbox_path, root, recon):
    """akr_gpt_reduce really runs its ncclReduce (film for `none`, the primal / gradient sums otherwise) -- with the one rank a
    one-GPU box allows: the reduced session finishes into the film of a session that was never reduced, and a session ended with
    akr_gpt_abort leaves its film as the samples left it (no splat scale, no reconstruction)."""
    w, h = 48, 32
    sd = make_scene("cbox", cbox_path, root, w, h)
    cfg = gpt_config(spp=3, max_depth=5, rr_depth=2, reconstruction=recon, reconstruction_iter=4)
    scene = capi.Scene(ctx, sd)
    plain = capi.Film(ctx, w, h)
    capi.gpt_render(ctx, scene, cfg, plain)
    comm = capi.Comm(ctx, capi.comm_unique_id(), 0, 1)
    film = capi.Film(ctx, w, h)
    se = capi.GptSession(ctx, scene, cfg, film)
    se.sample()
    se.reduce(comm, root=0)
    se.finish()
    assert n_bit_diff(film.read(), plain.read()) == 0 and film.splat_scale == plain.splat_scale
    raw = capi.Film(ctx, w, h)
    se = capi.GptSession(ctx, scene, cfg, raw)
    se.sample()
    before = raw.read()
    st = se.abort()
    assert st["n_samples"] > 0 and n_bit_diff(raw.read(), before) == 0 and raw.splat_scale == 1.0
    comm.close()
