"""N > 1 path on CPU: tile ownership + film reduce over torch.distributed (gloo, world_size 2). The render on
each rank is done by the CPU oracle (no GPU here); the sharding rule and the reduce are the product's."""
import os
import socket

import numpy as np
import pytest

from akari_render_amd import abi, distributed
from oracle import pyoracle, scene_json
from tests.helpers import make_config


def test_ownership_mask_partitions_the_frame():
    for (w, h, world) in [(64, 64, 2), (100, 70, 3), (1920, 1080, 8), (33, 9, 4)]:
        masks = [distributed.owned_pixel_mask(w, h, r, world) for r in range(world)]
        assert np.array_equal(np.sum(masks, axis=0), np.ones((h, w), dtype=int))
        assert sum(distributed.owned_pixel_count(w, h, r, world) for r in range(world)) == w * h
    # tiles interleave: at 1080p over 8 ranks every rank owns 12.0-12.9 % of the pixels
    counts = [distributed.owned_pixel_count(1920, 1080, r, 8) for r in range(8)]
    assert max(counts) / min(counts) < 1.08


def test_tiles_are_dealt_along_the_morton_curve():
    """"tile t owned by GPU t mod G in Morton order". Over 8 ranks every aligned 4 x 2 block of tiles holds one tile of each
    rank, so at 1080p (60 x 34 tiles of 32 x 32: 15 x 17 whole blocks) every rank owns exactly 255 tiles, whatever the row length."""
    m = [distributed.owned_pixel_mask(1920, 1080, r, 8) for r in range(8)]
    for r in range(8):
        tiles = m[r][::32, ::32]  # one sample per tile (the last tile row is 24 pixels high: still sampled at its first row)
        assert tiles.shape == (34, 60) and int(tiles.sum()) == 255
        blocks = tiles.reshape(17, 2, 15, 4).transpose(0, 2, 1, 3).reshape(17 * 15, 8)
        assert np.all(blocks.sum(axis=1) == 1)
    # the curve itself: x in the even bits, y in the odd ones
    assert [int(distributed.tile_morton(x, y)) for x, y in [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (3, 5), (255, 255)]] == [0, 1, 2, 3, 4, 0b100111, 0xffff]
    # rank = code mod world for any world size
    ty, tx = np.meshgrid(np.arange(9), np.arange(13), indexing="ij")
    for world in (2, 3, 5, 7):
        owner = (distributed.tile_morton(tx, ty) % np.uint64(world)).astype(int)
        for r in range(world):
            assert np.array_equal(distributed.owned_pixel_mask(13 * 16, 9 * 8, r, world, 16, 8)[::8, ::16], owner == r)


def test_oracle_shards_follow_the_mask(oracle_lib, cbox_path):
    sd = scene_json.load_scene(cbox_path, 72, 40)
    sc = pyoracle.OracleScene(sd)
    cfg = make_config(spp=2, spp_per_pass=2, max_depth=3)
    full, _ = sc.render(cfg)
    acc = np.zeros_like(full)
    for r in range(3):
        part, _ = sc.render(distributed.shard_config(cfg, r, 3, 16, 8))
        mask = distributed.owned_pixel_mask(72, 40, r, 3, 16, 8).ravel()
        assert np.array_equal(part[6 * 72 * 40 :] > 0, mask)
        acc += part
    assert np.array_equal(acc, full)  # disjoint tiles: the sum is exact


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cbox_path, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    torch.set_num_threads(1)
    distributed.init_process_group("gloo")
    sd = scene_json.load_scene(cbox_path, 48, 48)
    sc = pyoracle.OracleScene(sd)
    cfg = distributed.shard_config(make_config(spp=4, spp_per_pass=2, max_depth=4), rank, world, 16, 16)
    film, _ = sc.render(cfg, n_threads=2)
    assert not film[3 * 48 * 48: 6 * 48 * 48].any()  # a pt film never touches its splat plane
    t = torch.from_numpy(film)
    distributed.reduce_film(t, dst=0, planes=5)  # rgb + weight: the 4 N floats of SURVEY.md 8(e) (akr_film_reduce_planes over RCCL)
    if rank == 0:
        np.save(out_path, t.numpy())
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_film_reduce(oracle_lib, cbox_path, tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(2, _free_port(), cbox_path, out), nprocs=2, join=True)
    sd = scene_json.load_scene(cbox_path, 48, 48)
    full, _ = pyoracle.OracleScene(sd).render(make_config(spp=4, spp_per_pass=2, max_depth=4))
    assert np.array_equal(np.load(out), full)


def sample_range_config(cfg: abi.PtConfig, rank: int, world: int) -> abi.PtConfig:
    """Rank r of `world` renders samples [r spp / world, (r + 1) spp / world) of every pixel (akr_pt_config.sample_begin / _count)."""
    c = cfg.copy()
    per = cfg.spp // world
    c.sample_begin, c.sample_count = rank * per, (cfg.spp - rank * per) if rank == world - 1 else per
    return c


def _range_worker(rank, world, port, cbox_path, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    torch.set_num_threads(1)
    distributed.init_process_group("gloo")
    sd = scene_json.load_scene(cbox_path, 40, 32)
    sc = pyoracle.OracleScene(sd)
    cfg = sample_range_config(make_config(spp=10, spp_per_pass=4, max_depth=4, sampler_type=abi.SAMPLER_SOBOL, sampler_seed=2), rank, world)
    film, _ = sc.render(cfg, n_threads=2)
    t = torch.from_numpy(film)
    distributed.reduce_film(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_sample_ranges_reduce_to_the_one_shot_render(oracle_lib, cbox_path, tmp_path):
    """The other way to shard (SURVEY 8e "sample-range split"): two ranks, each ALL pixels and half of the samples of an index-based
    sampler, films summed over gloo: the weight plane is the one-shot render's exactly, the radiance within the re-association of
    its f32 additions."""
    import torch.multiprocessing as mp

    from tests.helpers import rel_rmse, resolve_np

    out = str(tmp_path / "film.npy")
    mp.spawn(_range_worker, args=(2, _free_port(), cbox_path, out), nprocs=2, join=True)
    sd = scene_json.load_scene(cbox_path, 40, 32)
    full, _ = pyoracle.OracleScene(sd).render(make_config(spp=10, spp_per_pass=4, max_depth=4, sampler_type=abi.SAMPLER_SOBOL, sampler_seed=2))
    got = np.load(out)
    n = 40 * 32
    assert np.array_equal(got[6 * n:], full[6 * n:]) and np.all(got[6 * n:] == 10)
    assert rel_rmse(resolve_np(got, 40, 32), resolve_np(full, 40, 32)) < 1e-6
