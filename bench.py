#!/usr/bin/env python
"""Benchmark of the `pt` hot path on MI355X: Msamples/s on scenes/cbox 1920x1080, force_diffuse (BASELINE.json
configs[1]); one step = one pass of spp_per_pass = 64 samples per pixel (the reference's kernel.dispatch,
pt.rs:1126-1133).

    python bench.py --gpus 1 --steps 16 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0. The path shards by independent units -- pixels and sample sets -- with no collective in
the render itself. N > 1, default (--scaling weak, per-GPU work fixed): every GPU renders the whole frame with its own
sampler seed, i.e. N independent sample sets of the same workload (N x 1024 spp at N GPUs, the spp of BASELINE's 8-GPU
config), and the films are sum-reduced onto rank 0 over RCCL inside the timed region. --scaling strong shards the ONE
frame's 32x32 pixel tiles over the ranks instead (same image for every N; DESIGN.md section 5 explains why its
efficiency is bounded by the per-pixel sequential sample streams: ~0.75 at 8 GPUs for 1080p).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H, SPP_PER_PASS = 1920, 1080, 64
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(d):
    """SURVEY.md 8(d) byte model: per-event record sizes x device counters. The BVH terms
    are zero for cbox (36 triangles, cache-resident) and counted in full for BVH scenes."""
    return (56 * d["n_closest"] + 292 * d["n_shaded"] + 64 * d["n_shadow"] + 156 * d["n_samples"] +
            64 * d["n_node_visits"] + (48 * d["n_tri_tests"] if d["n_node_visits"] else 0))


def host_threads():
    """Threads for the CPU baseline: the container may be limited by a cgroup CPU quota far below os.cpu_count() (the GPU
    box: 256 logical CPUs visible, cpu.max = 16 cores; 256 threads then run 40 % slower than 32). Two threads per core of
    quota, capped by the visible CPUs."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None and quota < n:
        return max(1, min(n, int(round(2 * quota)))), quota
    return n, None


def cpu_baseline(n_threads, quota=None):
    """The CPU oracle (restatement of the reference algorithm; the reference binary cannot run here) on the same
    workload, bounded sample: 1920x1080, force_diffuse, 4 spp."""
    from akari_render_amd import abi
    from oracle import pyoracle, scene_json

    sd = scene_json.load_scene(os.path.join(ROOT, "scenes", "cbox", "scene.json"), W, H)
    cfg = abi.PtConfig.default()
    cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth, cfg.force_diffuse = 4, 4, 12, 5, 1
    sc = pyoracle.OracleScene(sd)
    # calibrate the sample size to ~10-20 s of CPU work
    t0 = time.time()
    _, st = sc.render(cfg, n_threads=n_threads)
    dt = time.time() - t0
    spp = int(max(4, min(64, 4 * 12.0 / max(dt, 1e-3))))
    if spp > 4:
        cfg.spp = cfg.spp_per_pass = spp
        t0 = time.time()
        _, st = sc.render(cfg, n_threads=n_threads)
        dt = time.time() - t0
    return {"value": st["n_samples"] / dt / 1e6, "unit": "Msamples/s", "cores": n_threads, "kind": "port",
            "sample": f"cbox {W}x{H} force_diffuse {cfg.spp} spp ({st['n_samples']} camera paths, {dt:.1f} s), CPU oracle (C, pthreads)"
                      + (f", cgroup CPU quota {quota:g} cores" if quota is not None else "")}


def measured_traffic(args, d):
    """HBM bytes of one timed launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes,
    tools/pmc_run.sh; unit + gfx950 corrections of MI355X_MICROARCH.md "HBM" applied there). A counter pass cannot run
    inside this process, so the number is the one committed under profiles/ for exactly this launch shape (same
    workload, same number of fused passes); anything else reports null."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
    if args.gpus != 1 or not os.path.exists(path):
        return None
    try:
        t = json.load(open(path))
    except Exception:
        return None
    key = "full_graph" if args.full_graph else "force_diffuse"
    e = t.get(key)
    if not e or e.get("steps") != args.steps or d["n_launches"] != 1:
        return None
    return e.get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-graph", action="store_true", help="configs[2]: full Cycles-subset shader graph instead of force_diffuse")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every GPU renders the whole 1080p frame with its own sampler seed (N independent sample "
                         "sets, N x the samples; per-GPU work fixed), films sum-reduced; strong = the one frame's 32x32 pixel "
                         "tiles round-robin over the GPUs (identical image for every N)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only for smoke-testing the "
                         "multi-rank path on a box with fewer GPUs than ranks: all ranks then share device 0)")
    args = ap.parse_args()

    import torch

    from akari_render_amd import abi, capi, distributed

    rank, world, local_rank = distributed.env_rank_world()
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus} (WORLD_SIZE={world})"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        distributed.init_process_group(args.backend)
    import torch.distributed as dist

    ctx = capi.Context(local_rank if world > 1 else 0)
    scene = capi.Scene(ctx, os.path.join(ROOT, "scenes", "cbox", "scene.json"), W, H)  # akr_scene_load: C++ reader
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    film_t = torch.zeros(7 * W * H, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    film = capi.Film(ctx, W, H, device_ptr=film_t.data_ptr())

    cfg = abi.PtConfig.default()
    cfg.spp = (args.warmup + args.steps) * SPP_PER_PASS
    cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth, cfg.use_nee = SPP_PER_PASS, 12, 5, 1
    cfg.force_diffuse = 0 if args.full_graph else 1
    cfg.filter_type, cfg.filter_radius = abi.FILTER_GAUSSIAN, 1.5
    weak = world > 1 and args.scaling == "weak"
    if weak:
        cfg.sampler_seed = rank  # independent sample set per GPU (sampler/mod.rs:148-160 seeds the per-pixel streams from it)
    else:
        cfg.sampler_seed = 0
        cfg = distributed.shard_config(cfg, rank, world)

    se = capi.PtSession(ctx, scene, cfg, film)
    if args.warmup > 0:
        se.passes(args.warmup, blocking=True)
    if world > 1 and args.backend == "nccl":
        # warm the collective up too (RCCL builds its rings / proxy connections on the first reduce of a given size):
        # same message size as the film, on a scratch tensor, outside the timed region
        scratch = torch.zeros_like(film_t)
        distributed.reduce_film(scratch, dst=0)
        torch.cuda.synchronize(dev)
        del scratch
    s0 = se.stats()

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    sync()
    t0 = time.perf_counter()
    se.passes(args.steps, blocking=True)
    if world > 1:
        if args.backend == "gloo":
            host = film_t.cpu()
            distributed.reduce_film(host, dst=0)
            film_t.copy_(host)
        else:
            distributed.reduce_film(film_t, dst=0)
    sync()
    t1 = time.perf_counter()
    s1 = se.end()

    elapsed = t1 - t0
    d = {k: s1[k] - s0[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded", "n_node_visits", "n_tri_tests")}
    d["kernel_ms"] = s1["kernel_ms"] - s0["kernel_ms"]
    d["n_launches"] = s1["n_launches"] - s0["n_launches"]
    if world > 1:
        cdev = dev if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([d["n_samples"], d["n_closest"], d["n_shadow"], d["n_shaded"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_samples = int(c[0].item())
    else:
        total_samples = d["n_samples"]
    n_sets = world if weak else 1
    assert total_samples == n_sets * W * H * SPP_PER_PASS * args.steps, (total_samples, n_sets * W * H * SPP_PER_PASS * args.steps)

    if rank == 0:
        # frame sanity inside the bench: every pixel got its samples, film finite
        wsum = float(film_t[6 * W * H :].sum().item())
        assert wsum == float(n_sets * W * H) * (args.warmup + args.steps) * SPP_PER_PASS, wsum
        assert bool(torch.isfinite(film_t).all().item())
        # dominant kernel (k_pt_pass) of THIS rank: algorithmic bytes per launch / average launch duration (HIP events
        # recorded around each launch on the context's stream)
        launches = max(1, d["n_launches"])
        bytes_per_launch = algorithmic_bytes(d) / launches
        avg_launch_s = d["kernel_ms"] * 1e-3 / launches
        achieved = bytes_per_launch / avg_launch_s / 1e9
        out = {
            "metric": "Msamples/s (whole node), 1080p Cornell box, path tracer",
            "value": total_samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: scenes/cbox (36 triangles, reference scene data) at 1920x1080, independent sampler, seed " + ("= rank" if weak else "0"),
            "config": {
                "workload": ("cbox 1920x1080, full Cycles-subset shader graph" if args.full_graph else "cbox 1920x1080, diffuse-only BSDF (force_diffuse)")
                            + f", {SPP_PER_PASS} spp per step, max_depth 12, rr_depth 5, NEE, gaussian filter r=1.5",
                "spp_total": args.steps * SPP_PER_PASS * n_sets,
                "parallelism": ("single GPU" if args.gpus == 1 else
                                f"{args.gpus} independent sample sets of {args.steps * SPP_PER_PASS} spp (sampler seed = rank), one per GPU, films sum-reduced (RCCL)" if weak else
                                f"pixel tiles 32x32 round-robin over {args.gpus} GPU(s), film sum-reduce (RCCL)"),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(args, d),
                "kernel": "k_pt_pass",
                "launches": d["n_launches"],
                "avg_launch_ms": avg_launch_s * 1e3,
                "algorithmic_bytes_per_sample": algorithmic_bytes(d) / max(1, d["n_samples"]),
                "note": "byte model of SURVEY.md 8(d); on the 36-triangle cbox the kernel keeps path state in registers, so real HBM traffic is far below the model (see DESIGN.md)",
            },
            "counters": {k: d[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded")},
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(*host_threads())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
