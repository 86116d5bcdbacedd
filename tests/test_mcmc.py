"""The `mcmc_opt` integrator (akari_integrator/src/mcmc_opt.rs): the oracle's estimator against the path tracer (CPU); the HIP
kernels against the oracle (GPU) -- Markov-chain states and normalisation bit for bit, the film up to the summation order of
the float atomics the reference splats with too."""
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


def mcmc_config(**kw) -> abi.McmcConfig:
    c = abi.McmcConfig.default()
    c.n_chains, c.n_bootstrap, c.spp, c.direct_spp = 96, 3000, 16, 4
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_mcmc_config_default_and_method_json(hip_lib):
    c = abi.McmcConfig()
    assert hip_lib.akr_mcmc_config_default(C.byref(c)) == 0
    assert bytes(c) == bytes(abi.McmcConfig.default()) and C.sizeof(abi.McmcConfig) == 96 == pyoracle.lib().or_sizeof_mcmc_config()
    assert (c.spp, c.max_depth, c.rr_depth, c.spp_per_pass, c.n_chains, c.n_bootstrap, c.direct_spp, c.mcmc_depth) == (256, 7, 5, 64, 512, 100000, 64, 0xFFFFFFFF)
    assert (c.exponential_mutation, round(c.small_sigma, 6), round(c.large_step_prob, 6), c.image_mutation_prob, c.image_mutation_size) == (1, 0.01, 0.1, 0.0, 0.0)
    cfg = abi.PtConfig()
    assert hip_lib.akr_pt_config_from_json(b'{"method": {"type": "mcmc_opt"}}', C.byref(cfg), None, 0) == capi.ERR_UNSUPPORTED


def test_oracle_mcmc_estimates_the_same_image_as_pt(cbox_path, root):
    """Metropolis chains with and without the separate direct pass converge to the path tracer's image; a chain's counters add up."""
    w = h = 32
    sd = scene_json.load_scene(cbox_path, w, h)
    sd.ggx_table = table(root)
    osc = pyoracle.OracleScene(sd)
    pc = abi.PtConfig.default()
    pc.spp, pc.spp_per_pass, pc.max_depth = 2048, 64, 7
    ref = pyoracle.resolve(osc.render(pc)[0], w, h)
    for kw in (dict(direct_spp=-1), dict(direct_spp=64), dict(direct_spp=-1, exponential_mutation=0), dict(direct_spp=-1, image_mutation_prob=0.2, image_mutation_size=4.0)):
        cfg = mcmc_config(spp=192, n_chains=64, n_bootstrap=4000, **kw)
        film, res, chains = osc.mcmc_render(cfg)
        img = pyoracle.resolve(film, w, h, float(res["splat_scale"]))
        assert np.all(np.isfinite(img)) and abs(img.mean() - ref.mean()) < 0.08 * ref.mean(), (kw, img.mean(), ref.mean())
        assert rel_rmse(img, ref) < 3.0, (kw, rel_rmse(img, ref))
        per_chain = w * h * cfg.spp // cfg.n_chains
        assert res["contribution"] == np.float32(w * h * cfg.spp / (per_chain * cfg.n_chains))
        large = per_chain - chains["n_mutations"]  # every mutation is either a small step (n_mutations) or a large step
        assert np.all(large == chains["b_cnt"]) and 0.04 < large.sum() / (per_chain * cfg.n_chains) < 0.16
        assert np.all(chains["n_accepted"] <= chains["n_mutations"]) and 0.5 < res["acceptance_rate"] < 0.98
        assert np.all(chains["chain_id"] == np.arange(cfg.n_chains)) and np.all(chains["cur_iter"] <= per_chain)
    # a scene without light: "Bootstrap failed" (mcmc_opt.rs:352)
    for m in sd.materials:
        m.emission_strength = 0.0
    with pytest.raises(AssertionError):
        pyoracle.OracleScene(sd).mcmc_render(mcmc_config())


def both(ctx, sd, cfg):
    scene = capi.Scene(ctx, sd)
    w, h = sd.camera.width, sd.camera.height
    n = w * h
    film = capi.Film(ctx, w, h)
    st, res, chains = capi.mcmc_render(ctx, scene, cfg, film)
    o_film, o_res, o_chains = pyoracle.OracleScene(sd).mcmc_render(cfg)
    # every chain walked the same states ...
    for name in chains.dtype.names:
        assert np.array_equal(chains[name].view(np.uint32), o_chains[name].view(np.uint32)), f"MarkovState.{name} differs"
    assert res["normalization"] == o_res["normalization"] and res["acceptance_rate"] == o_res["acceptance_rate"]
    assert np.float32(res["splat_scale"]) == o_res["splat_scale"] == film.splat_scale and np.float32(res["contribution"]) == o_res["contribution"]
    per_chain = max(n * min(cfg.spp, cfg.spp_per_pass) // cfg.n_chains, 1)
    assert st["n_samples"] == res["n_mutations"] >= per_chain * cfg.n_chains
    # ... the direct pass is the path tracer's, bit for bit ...
    g = film.read()
    assert n_bit_diff(g[: 3 * n], o_film[: 3 * n]) == 0 and n_bit_diff(g[6 * n :], o_film[6 * n :]) == 0
    # ... and the splats agree up to the order in which float atomics summed them
    gs, os_ = g[3 * n : 6 * n], o_film[3 * n : 6 * n]
    assert np.allclose(gs, os_, rtol=2e-4, atol=2e-5 * float(np.abs(os_).max())), float(np.abs(gs - os_).max() / np.abs(os_).max())
    assert rel_rmse(film.resolve(), pyoracle.resolve(o_film, w, h, float(o_res["splat_scale"]))) < 1e-4
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["cbox", "glass_coat", "grid_bvh", "textured"])
def test_mcmc_parity(ctx, cbox_path, root, scene_name):
    from tests.test_gpt import make_scene

    sd = make_scene(scene_name, cbox_path, root, 40, 32)
    both(ctx, sd, mcmc_config())


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["no_direct", "indirect_chains_only", "gaussian", "image_mutation", "shallow_mcmc_depth", "two_passes", "many_chains", "pmj_direct"])
def test_mcmc_parity_options(ctx, cbox_path, root, case):
    kw = {"no_direct": dict(direct_spp=-1), "indirect_chains_only": dict(direct_spp=0), "gaussian": dict(exponential_mutation=0, small_sigma=0.02),
          "image_mutation": dict(image_mutation_prob=0.25, image_mutation_size=6.0, large_step_prob=0.3),
          "shallow_mcmc_depth": dict(mcmc_depth=1, max_depth=6), "two_passes": dict(spp=10, spp_per_pass=4, seed=77),
          "many_chains": dict(n_chains=5000, n_bootstrap=2000, spp=3),  # more chains than bootstrap paths, fewer than one mutation per pixel and chain
          "pmj_direct": dict(sampler_type=abi.SAMPLER_PMJ02BN, sampler_seed=3)}[case]
    from tests.test_gpt import make_scene

    if case == "pmj_direct":
        pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
    sd = make_scene("kinds" if case in ("gaussian", "shallow_mcmc_depth") else "cbox", cbox_path, root, 36, 36)
    both(ctx, sd, mcmc_config(**kw))


@pytest.mark.gpu
def test_mcmc_rejections_and_render_task(ctx, cbox_path, tmp_path, monkeypatch):
    sd = scene_json.load_scene(cbox_path, 32, 32)
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, 32, 32)
    for bad in (dict(n_chains=0), dict(n_bootstrap=0), dict(spp_per_pass=0)):
        with pytest.raises(capi.AkariError) as e:
            capi.mcmc_render(ctx, scene, mcmc_config(**bad), film)
        assert e.value.code == capi.ERR_INVALID_ARGUMENT
    dark = scene_json.load_scene(cbox_path, 32, 32)
    for m in dark.materials:
        m.emission_strength = 0.0
    with pytest.raises(capi.AkariError, match="Bootstrap failed"):
        capi.mcmc_render(ctx, capi.Scene(ctx, dark), mcmc_config(direct_spp=-1), film)
    monkeypatch.chdir(tmp_path)
    method = {"method": {"type": "mcmc_opt", "spp": 8, "n_chains": 128, "n_bootstrap": 2000, "direct_spp": 4, "mcmc_depth": None,
                         "method": {"type": "kelemen", "exponential_mutation": True, "small_sigma": 0.01, "large_step_prob": 0.2,
                                    "image_mutation_prob": 0.0, "adaptive": False}},
              "film": {"out": "mcmc.exr", "filter": {"type": "gaussian", "radius": 1.5}}}
    capi.render_task(ctx, scene, json.dumps(method))
    from tests.test_output_stage import read_exr_rgb

    img = read_exr_rgb(str(tmp_path / "mcmc.exr"))
    film.clear()
    capi.mcmc_render(ctx, scene, mcmc_config(spp=8, n_chains=128, n_bootstrap=2000, direct_spp=4, large_step_prob=0.2), film)
    assert rel_rmse(img, film.resolve()) < 1e-4 and img.mean() > 0.02
    # --save-intermediate / --save-stats (mcmc_opt.rs:640-676): one image per pass, resolved with b / spp-so-far
    method["method"]["spp_per_pass"] = 4
    capi.render_task(ctx, scene, json.dumps(method), name="run", save_intermediate=True, save_stats=True)
    stats = json.load(open(tmp_path / "run.json"))
    assert [e["spp"] for e in stats["intermediate"]] == [4, 8] and [e["path"] for e in stats["intermediate"]] == ["run-4.exr", "run-8.exr"]
    assert stats["intermediate"][1]["time"] >= stats["intermediate"][0]["time"] > 0
    last, final, half = read_exr_rgb(str(tmp_path / "run-8.exr")), read_exr_rgb(str(tmp_path / "mcmc.exr")), read_exr_rgb(str(tmp_path / "run-4.exr"))
    assert np.array_equal(last, final) and not np.array_equal(half, final) and abs(half.mean() - final.mean()) < 0.2 * final.mean()



@pytest.mark.gpu
@pytest.mark.parametrize("color", [abi.COLOR_REPR_ACESCG, abi.COLOR_RGB_ACESCG | abi.COLOR_REPR_ACESCG])
def test_mcmc_in_a_non_default_colour_pipeline(ctx, cbox_path, root, color):
    """akr_mcmc_config.color: the direct pass, the chains' materials and the splats all run in the pipeline."""
    from tests.test_gpt import make_scene

    sd = make_scene("cbox", cbox_path, root, 36, 28)
    a = both(ctx, sd, mcmc_config(color=color))
    b = both(ctx, sd, mcmc_config())
    assert a["normalization"] != b["normalization"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [3, 2])
def test_mcmc_sharded_chain_sets_are_the_one_gpu_chain_set(ctx, cbox_path, root, world):
    """akr_mcmc_render_shard: rank r of `world` runs chains [r c / world, (r + 1) c / world) and its tiles of the direct pass. Rendered one
    after the other on this GPU: every chain's final state is the oracle's (= the one-GPU run's) bit for bit, the direct pass's planes sum
    to the oracle's exactly (disjoint tiles), the normalisation from the ranks' sums is the oracle's up to the order of a double sum, and
    the splats agree up to the order of the float atomics (as on one GPU)."""
    from tests.test_gpt import make_scene

    sd = make_scene("cbox", cbox_path, root, 40, 32)
    cfg = mcmc_config(n_chains=100, spp=10, spp_per_pass=4)  # 100 chains over 3 ranks: uneven shares
    w, h = 40, 32
    n = w * h
    scene = capi.Scene(ctx, sd)
    o_film, o_res, o_chains = pyoracle.OracleScene(sd).mcmc_render(cfg)
    total = np.zeros(7 * n, dtype=np.float32)
    chains = np.zeros(cfg.n_chains, dtype=abi.MARKOV_STATE_DTYPE)
    partials, executed = [], 0
    for rank in range(world):
        film = capi.Film(ctx, w, h)
        st, part, ch = capi.mcmc_render_shard(ctx, scene, cfg, film, rank, world)
        lo, hi = rank * cfg.n_chains // world, (rank + 1) * cfg.n_chains // world
        assert not ch[:lo].view(np.uint8).any() and not ch[hi:].view(np.uint8).any()  # the other ranks' records stay zero
        chains[lo:hi] = ch[lo:hi]
        g = film.read()
        assert not np.any((total[: 3 * n] != 0) & (g[: 3 * n] != 0))  # direct pass: disjoint tiles
        total += g
        partials.append(part)
        executed += part.n_executed
        assert part.bootstrap_sum == partials[0].bootstrap_sum and part.spp == cfg.spp
    for name in chains.dtype.names:
        assert np.array_equal(chains[name].view(np.uint32), o_chains[name].view(np.uint32)), f"MarkovState.{name} differs"
    res = capi.mcmc_combine_host(None, partials)
    assert abs(res["normalization"] - o_res["normalization"]) <= 1e-12 * o_res["normalization"]
    assert res["acceptance_rate"] == o_res["acceptance_rate"] and np.float32(res["contribution"]) == o_res["contribution"]
    assert abs(np.float32(res["splat_scale"]) - o_res["splat_scale"]) <= np.spacing(np.float32(o_res["splat_scale"]))
    assert res["n_mutations"] == executed == max(n * 4 // cfg.n_chains, 1) * cfg.n_chains * 2 + max(n * 2 // cfg.n_chains, 1) * cfg.n_chains
    assert np.array_equal(total[: 3 * n], o_film[: 3 * n]) and np.array_equal(total[6 * n:], o_film[6 * n:])
    gs, os_ = total[3 * n: 6 * n], o_film[3 * n: 6 * n]
    assert np.allclose(gs, os_, rtol=2e-4, atol=2e-5 * float(np.abs(os_).max()))
    with pytest.raises(capi.AkariError):
        capi.mcmc_render_shard(ctx, scene, cfg, capi.Film(ctx, w, h), 3, 3)


@pytest.mark.gpu
def test_mcmc_combine_over_rccl_world_of_one(ctx, cbox_path, root):
    """akr_mcmc_combine (film reduce + all-reduce of the normalisation sums over RCCL) on a world of one = akr_mcmc_render."""
    from tests.test_gpt import make_scene

    sd = make_scene("cbox", cbox_path, root, 36, 36)
    cfg = mcmc_config(spp=6)
    scene = capi.Scene(ctx, sd)
    f1, f2 = capi.Film(ctx, 36, 36), capi.Film(ctx, 36, 36)
    _, res, chains = capi.mcmc_render(ctx, scene, cfg, f1)
    _, part, ch = capi.mcmc_render_shard(ctx, scene, cfg, f2, 0, 1)
    comm = capi.Comm(ctx, capi.comm_unique_id(), 0, 1)
    got = comm.mcmc_combine(f2, part, root=0)
    comm.close()
    assert np.array_equal(ch.view(np.uint8), chains.view(np.uint8))
    assert got["normalization"] == res["normalization"] and got["acceptance_rate"] == res["acceptance_rate"] and got["n_mutations"] == res["n_mutations"]
    assert f2.splat_scale == f1.splat_scale == np.float32(res["splat_scale"])
    n = 36 * 36
    a, b = f1.read(), f2.read()
    assert np.array_equal(a[: 3 * n], b[: 3 * n]) and np.allclose(a[3 * n: 6 * n], b[3 * n: 6 * n], rtol=2e-4, atol=2e-5 * float(np.abs(a).max()))


def _combine_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    from akari_render_amd import distributed

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    distributed.init_process_group("gloo")
    # a rank's partial sums (synthetic: the arithmetic of the exchange is what runs here; the chains themselves need a GPU)
    part = abi.McmcPartial(bootstrap_sum=12.5, b_sum=1.25 * (rank + 1), n_bootstrap=1000, b_cnt=10 + rank, n_accepted=700 + rank, n_mutations=900 + 2 * rank,
                           n_executed=1000 + rank, spp=16, contribution=1.5)
    t = torch.tensor([part.b_sum, float(part.b_cnt), float(part.n_accepted), float(part.n_mutations), float(part.n_executed)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)  # what akr_mcmc_combine does over RCCL
    everyone = abi.McmcPartial(bootstrap_sum=part.bootstrap_sum, b_sum=float(t[0]), n_bootstrap=part.n_bootstrap, b_cnt=int(t[1]), n_accepted=int(t[2]),
                               n_mutations=int(t[3]), n_executed=int(t[4]), spp=part.spp, contribution=part.contribution)
    res = capi.mcmc_combine_host(None, [everyone])
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_mcmc_normalisation_from_two_ranks_over_gloo(tmp_path):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "res.json")
    mp.spawn(_combine_worker, args=(2, port, out), nprocs=2, join=True)
    res = json.load(open(out))
    b = (12.5 + 1.25 + 2.5) / (1000 + 10 + 11)
    assert res["normalization"] == b and res["acceptance_rate"] == (700 + 701) / (900 + 902) and res["n_mutations"] == 2001
    assert np.float32(res["splat_scale"]) == np.float32(b) / np.float32(16) and res["contribution"] == 1.5
    # gathered partials give the same numbers; partials of different renders are refused
    parts = [abi.McmcPartial(bootstrap_sum=12.5, b_sum=1.25 * (r + 1), n_bootstrap=1000, b_cnt=10 + r, n_accepted=700 + r, n_mutations=900 + 2 * r, n_executed=1000 + r,
                             spp=16, contribution=1.5) for r in range(2)]
    assert capi.mcmc_combine_host(None, parts) == res
    parts[1].spp = 8
    with pytest.raises(capi.AkariError):
        capi.mcmc_combine_host(None, parts)
