"""Multi-GPU plumbing: one process per GPU, pixel tiles sharded across ranks, one sum-reduce of the film.

The reference has no multi-device code. Every pixel-sample of the path tracer is independent and
the per-pixel sampler stream does not depend on which rank renders the pixel, so rank r simply renders the tiles
(tx, ty) with morton(tx, ty) % world == r (akr_pt_config.shard_*) into a film that stays zero elsewhere; torch.distributed (backend
"nccl" == RCCL over xGMI on ROCm, "gloo" on CPU) then sums the films onto rank 0. There is no collective inside
the render itself.
"""
from __future__ import annotations

import os

import numpy as np

from . import abi


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: str):
    """Initialises torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import torch.distributed as dist

    rank, world, _ = env_rank_world()
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_config(cfg: abi.PtConfig, rank: int, world: int, tile_w: int = 32, tile_h: int = 32) -> abi.PtConfig:
    c = cfg.copy()
    c.shard_rank, c.shard_count, c.tile_w, c.tile_h = rank, world, tile_w, tile_h
    return c


def tile_morton(tx, ty):
    """Position of tile (tx, ty) on the Z-order curve (csrc/kernels.h tile_morton): x in the even bits, y in the odd ones."""
    def spread(v):
        v = np.asarray(v, dtype=np.uint64) & np.uint64(0xffff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00ff00ff)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0f0f0f0f)
        v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
        return v
    return spread(tx) | (spread(ty) << np.uint64(1))


def owned_pixel_mask(width: int, height: int, rank: int, world: int, tile_w: int = 32, tile_h: int = 32) -> np.ndarray:
    """Host mirror of the kernels' tile ownership rule (csrc/kernels.h tile_owner: a tile's Morton code modulo the ranks): bool[H, W]."""
    if world <= 1:
        return np.ones((height, width), dtype=bool)
    ty, tx = np.meshgrid(np.arange(height) // tile_h, np.arange(width) // tile_w, indexing="ij")
    return (tile_morton(tx, ty) % world) == rank


def owned_pixel_count(width: int, height: int, rank: int, world: int, tile_w: int = 32, tile_h: int = 32) -> int:
    return int(owned_pixel_mask(width, height, rank, world, tile_w, tile_h).sum())


def reduce_film(film_tensor, dst: int = 0, planes: int = 7):
    """Sum-reduces the per-rank films (f32[7*W*H], reference layout) onto rank `dst`. Disjoint tiles make the sum
    exact: every element receives one non-zero contribution plus zeros."""
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        if planes == 7:
            dist.reduce(film_tensor, dst=dst, op=dist.ReduceOp.SUM)
        else:  # the planes a pt / aov film holds: rgb [0, 3N) and weight [6N, 7N) (akr_film_reduce_planes does the same over RCCL)
            n = film_tensor.numel() // 7
            runs = [(0, 3 * n)] * bool(planes & 1) + [(3 * n, 6 * n)] * bool(planes & 2) + [(6 * n, 7 * n)] * bool(planes & 4)
            for a, b in runs:
                dist.reduce(film_tensor[a:b], dst=dst, op=dist.ReduceOp.SUM)
    return film_tensor
