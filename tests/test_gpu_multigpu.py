"""The native exchange step (akr_comm_* / akr_film_reduce: RCCL through the C ABI, csrc/host/comm.cpp).

On a one-GPU box RCCL runs with a world of one rank (communicator bootstrap, ncclReduce and ncclAllReduce really execute);
with two or more GPUs visible, two processes render complementary tile shards of one frame and the reduced film must be the
single-GPU film bit for bit (disjoint tiles: every element receives one non-zero addend)."""
import multiprocessing as mp
import os
import tempfile

import numpy as np
import pytest

from akari_render_amd import capi, distributed
from oracle import scene_json
from tests.helpers import make_config, n_bit_diff

pytestmark = pytest.mark.gpu


def test_native_reduce_world_of_one(ctx, cbox_path):
    sd = scene_json.load_scene(cbox_path, 96, 64)
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, 96, 64)
    capi.pt_render(ctx, scene, make_config(spp=4, spp_per_pass=4), film)
    before = film.read()
    comm = capi.Comm(ctx, capi.comm_unique_id(), 0, 1)
    comm.reduce_film(film, root=0)
    assert n_bit_diff(film.read(), before) == 0
    comm.reduce_film(film, root=-1, blocking=False)   # all-reduce, asynchronous on the context's stream
    ctx.synchronize()
    assert n_bit_diff(film.read(), before) == 0
    # the planes a pt film holds (rgb + weight = 4 N floats, two collectives in one RCCL group), and every other subset
    assert not before[3 * 96 * 64: 6 * 96 * 64].any()
    for planes in (capi.FILM_PLANES_PT, 1, 2, 3, 4, 6, capi.FILM_PLANES_ALL):
        comm.reduce_film(film, root=0, planes=planes)
        assert n_bit_diff(film.read(), before) == 0
    for bad in (0, 8):
        with pytest.raises(capi.AkariError):
            comm.reduce_film(film, root=0, planes=bad)
    comm.close()
    with pytest.raises(capi.AkariError):
        capi.Comm(ctx, capi.comm_unique_id(), 3, 2)   # rank >= world


def _rank_main(rank, world, id_path, cbox_path, out_path):
    """One RCCL rank. Whatever goes wrong lands in <out_path>.rank<r>.err with the library's own message (akr_last_error, which
    carries ncclGetErrorString / ncclGetLastError for RCCL failures), so that a run nobody is watching can be diagnosed."""
    try:
        _rank_body(rank, world, id_path, cbox_path, out_path)
    except BaseException as ex:  # noqa: BLE001
        import traceback

        with open(f"{out_path}.rank{rank}.err", "w") as f:
            f.write(f"rank {rank} of {world}: {type(ex).__name__}: {ex}\nakr_last_error: {capi.last_error()}\n"
                    f"HIP devices visible: {capi.device_count()}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, "
                    f"NCCL_DEBUG={os.environ.get('NCCL_DEBUG')}\n{traceback.format_exc()}")
        raise


def _rank_body(rank, world, id_path, cbox_path, out_path):
    import time

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = capi.Context(rank)
    if rank == 0:
        uid = capi.comm_unique_id()
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_path + ".tmp", id_path)
    else:
        for _ in range(600):
            if os.path.exists(id_path):
                break
            time.sleep(0.05)
        uid = open(id_path, "rb").read()
    comm = capi.Comm(ctx, uid, rank, world)
    sd = scene_json.load_scene(cbox_path, 200, 120)
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, 200, 120)
    capi.pt_render(ctx, scene, distributed.shard_config(make_config(spp=8, spp_per_pass=4), rank, world, 32, 16), film)
    comm.reduce_film(film, root=0)
    if rank == 0:
        np.save(out_path, film.read())
    comm.close()


def test_native_reduce_two_gpus(ctx, cbox_path):
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU run covers it; one-GPU boxes run the world-of-one test)")
    sd = scene_json.load_scene(cbox_path, 200, 120)
    scene = capi.Scene(ctx, sd)
    full = capi.Film(ctx, 200, 120)
    capi.pt_render(ctx, scene, make_config(spp=8, spp_per_pass=4), full)
    with tempfile.TemporaryDirectory() as d:
        id_path, out_path = os.path.join(d, "id"), os.path.join(d, "film.npy")
        sp = mp.get_context("spawn")
        procs = [sp.Process(target=_rank_main, args=(r, 2, id_path, cbox_path, out_path)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
        errs = []
        for r, p in enumerate(procs):
            if p.exitcode is None:
                p.kill()
                errs.append(f"rank {r}: still running after 300 s (killed) -- a hung RCCL bootstrap or reduce")
            elif p.exitcode != 0:
                ef = f"{out_path}.rank{r}.err"
                errs.append(open(ef).read() if os.path.exists(ef) else f"rank {r}: exit code {p.exitcode}, no error file (died in native code?)")
        assert not errs, "akr_film_reduce over two GPUs failed:\n" + "\n".join(errs)
        assert n_bit_diff(np.load(out_path), full.read()) == 0


def _gloo_rank_main(rank, world, port, cbox_path, out_path):
    """One rank of the torch.distributed path of akari_render_amd/distributed.py: HIP render of this rank's tiles (all ranks
    share GPU 0 here), films summed onto rank 0 with gloo on host tensors."""
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    distributed.init_process_group("gloo")
    ctx = capi.Context(0)
    sd = scene_json.load_scene(cbox_path, 160, 96)
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, 160, 96)
    capi.pt_render(ctx, scene, distributed.shard_config(make_config(spp=8, spp_per_pass=4), rank, world), film)
    t = torch.from_numpy(film.read())
    distributed.reduce_film(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hip_render_under_torch_distributed(ctx, cbox_path):
    """World of 3 processes under torch.distributed (gloo; one GPU shared): every rank renders its tiles with the HIP path,
    the reduced film is the single-process film bit for bit -- the multi-GPU code path end to end except for the wire."""
    sd = scene_json.load_scene(cbox_path, 160, 96)
    scene = capi.Scene(ctx, sd)
    full = capi.Film(ctx, 160, 96)
    capi.pt_render(ctx, scene, make_config(spp=8, spp_per_pass=4), full)
    with tempfile.TemporaryDirectory() as d:
        out_path = os.path.join(d, "film.npy")
        sp = mp.get_context("spawn")
        port = 29600 + (os.getpid() % 300)
        procs = [sp.Process(target=_gloo_rank_main, args=(r, 3, port, cbox_path, out_path)) for r in range(3)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        assert n_bit_diff(np.load(out_path), full.read()) == 0


def test_bench_control_flow_with_eight_ranks_on_one_gpu(root):
    """`bench.py --gpus 8` exactly as the driver launches it (torch.distributed.run, one process per rank) but over gloo with all
    eight ranks on the one GPU of this box: the 8-rank control flow -- tile shards, per-rank all-gather, the C5 / weak-scaling /
    sample-range legs, the film reduces, the headline-first bookkeeping -- runs BEFORE the first real 8-GPU node does. (RCCL
    refuses two ranks on one device, so the wire itself cannot be exercised here.)"""
    import json
    import subprocess
    import sys

    port = 29900 + (os.getpid() % 90)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "1", "--warmup", "0"]
    res = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]   # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["value"] > 0
    cfg = out["config"]
    assert len(cfg["per_rank_samples"]) == 8 and sum(cfg["per_rank_samples"]) == 1920 * 1080 * 1024
    assert "gloo" in cfg["film_reduce"]            # the line names the reduce path that ran
    ex = out["extra_configs"]
    for leg in ("c5_strong", "c2_weak", "c2_sobol_sample_split"):
        assert "error" not in ex[leg], ex[leg]
        assert ex[leg]["n_gpus"] == 8 and ex[leg]["value"] > 0
    assert ex["c2_sobol_sample_split"]["weight_plane_ok"] is True


def test_bench_sample_split_headline_with_two_ranks_on_one_gpu(root):
    """`bench.py --gpus 2 --split samples --sampler sobol`: the headline itself sharded by sample ranges (two ranks over gloo on the
    one GPU): every rank renders all pixels and half of the samples, the reduced film holds every pixel's full weight (bench.py
    checks the weight plane itself before it prints), the ranks' sample counts are equal."""
    import json
    import subprocess
    import sys

    port = 29800 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1", "--split", "samples", "--sampler", "sobol",
           "--also", "none"]
    res = subprocess.run(cmd, cwd=root, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    cfg = out["config"]
    assert out["n_gpus"] == 2 and "sample ranges" in cfg["parallelism"] and "sobol" in out["data"]
    assert cfg["per_rank_samples"] == [1920 * 1080 * 512] * 2
