#!/usr/bin/env python
"""Benchmark of the `pt` hot path on MI355X (BASELINE.json metric: Msamples/s, whole node, 1080p Cornell box 1024 spp).

    python bench.py                                   # N = 1: configs[1] (C2) timed, then C3 and C4 once each, CPU baseline
    python bench.py --gpus 1 --steps 20 --warmup 5    # what the driver runs
    python bench.py --config c3|c4                    # another BASELINE configuration as the timed workload
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

ONE STEP = ONE RENDER OF THE NAMED WORKLOAD'S SAMPLE BATCH: 1024 samples per pixel of the 1920x1080 frame, i.e. 16 passes
of spp_per_pass = 64 (the reference's kernel.dispatch granularity, pt.rs:1126-1133) which the library fuses into one
k_pt_pass launch. C2 = scenes/cbox, force_diffuse (configs[1], the configuration the metric is quoted on); C3 = the same
frame with the full Cycles-subset shader graph (configs[2]; its 4096 spp are 4 steps); C4 = procedural 10 M-triangle hall
(configs[3]). Inputs (scene, sampler states, film) are resident in HBM before the timed region.

Prints ONE JSON line on rank 0. N > 1: the path shards by pixel tiles (32x32, round-robin over the ranks; a pixel's sample
stream does not depend on who renders it), the image is the same for every N ("scaling": "strong", the default), and the
films are sum-reduced onto rank 0 over RCCL inside the timed region. --scaling weak instead gives every GPU the whole frame
with its own sampler seed (N independent sample sets; the metric name says so). --split samples --sampler sobol shards the
SAMPLES of every pixel instead of the pixels (akr_pt_config.sample_begin / sample_count: rank r renders samples
[r S / N, (r + 1) S / N) of all pixels -- perfectly balanced at any resolution; exact only for the index-based samplers, so it
is a secondary leg (extra_configs.c2_sobol_sample_split) next to the BASELINE configuration, which uses the independent sampler).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

W, H, SPP_PER_PASS, PASSES_PER_STEP = 1920, 1080, 64, 16
SPP_PER_STEP = SPP_PER_PASS * PASSES_PER_STEP
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
PT_PLANES = 5  # akr_film_reduce_planes: rgb + weight, the planes a pt film holds (4 N floats)
CONFIGS = {
    "c2": dict(name="C2", baseline_config=1, force_diffuse=1, scene="cbox",
               workload="scenes/cbox 1920x1080, diffuse-only BSDF (force_diffuse), 1024 spp per step"),
    "c3": dict(name="C3", baseline_config=2, force_diffuse=0, scene="cbox",
               workload="scenes/cbox 1920x1080, full Cycles-subset shader graph, 1024 spp per step (configs[2] = 4 steps)"),
    "c4": dict(name="C4", baseline_config=3, force_diffuse=0, scene="hall",
               workload="procedural Sponza-like hall, 10 M triangles (generator seed 1234), 1920x1080, 1024 spp per step"),
}
CONFIGS["c5"] = dict(name="C5", baseline_config=4, force_diffuse=0, scene="cbox", res=(3840, 2160),
                     workload="scenes/cbox 3840x2160, full Cycles-subset shader graph, 1024 spp per step (configs[4] = 8 steps), tiles over the ranks")
HALL_TRIS = 10_000_000


def log(*a):
    print(*a, file=sys.stderr, flush=True)


NODE_BYTES_8D, TRI_BYTES_8D = 64, 48  # the byte model prices a node visit at 64 B and a triangle test at 48 B


def algorithmic_bytes(d, fetched=False):
    """SURVEY.md 8(d) byte model: per-event record sizes x device counters. The BVH terms are zero
    for cbox (36 triangles, cache-resident: no node visits are counted) and counted in full for BVH scenes -- at the MODEL's
    record sizes (NODE = 64, TRI = 48), whatever this build's records weigh: fatter records must not raise the score.
    fetched=True prices them at what a node / triangle step of this build actually fetches (64 / 64 bytes, akr_scene_info)."""
    nb, tb = (d.get("node_bytes", NODE_BYTES_8D), d.get("tri_bytes", TRI_BYTES_8D)) if fetched else (NODE_BYTES_8D, TRI_BYTES_8D)
    return (56 * d["n_closest"] + 292 * d["n_shaded"] + 64 * d["n_shadow"] + 156 * d["n_samples"] +
            nb * d["n_node_visits"] + (tb * d["n_tri_tests"] if d["n_node_visits"] else 0))


def csrc_hash():
    """sha256 over the kernel and host sources of the library (sorted relative paths + contents): what a PMC summary under
    profiles/ must carry to be quoted next to a bench line -- the counters of a kernel that has since been edited describe
    another kernel."""
    import hashlib

    h = hashlib.sha256()
    base = os.path.join(ROOT, "akari_render_amd", "csrc")
    for d_, _, files in sorted(os.walk(base)):
        for f in sorted(files):
            if f.endswith((".h", ".hip", ".cpp")):
                path = os.path.join(d_, f)
                h.update(os.path.relpath(path, base).encode())
                h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def host_threads():
    """(threads, cores) for the CPU baseline. The container may be limited by a cgroup CPU quota far below os.cpu_count()
    (the GPU box: 256 logical CPUs visible, cpu.max = 16 cores; 256 threads then run 40 % slower than 32): two threads
    per core of quota, capped by the visible CPUs. `cores` = what the quota grants (the number to quote)."""
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
    return n, float(n)


def cpu_baseline(key, n_threads, cores):
    """The CPU oracle (C restatement of the reference algorithm; the reference binary cannot run here: Rust + LuisaCompute)
    on bounded samples of the same workloads: the timed configuration's 1080p frame at a few spp (~10-15 s), and C1
    (256x256, 64 spp, full graph: BASELINE configs[0], the reference's own CPU-runnable case) in full."""
    from akari_render_amd import abi
    from oracle import pyoracle, scene_json

    def run(w, h, spp, fd, budget_s):
        sd = scene_json.load_scene(os.path.join(ROOT, "scenes", "cbox", "scene.json"), w, h)
        cfg = abi.PtConfig.default()
        cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth, cfg.force_diffuse = spp, spp, 12, 5, fd
        sc = pyoracle.OracleScene(sd)
        t0 = time.time()
        _, st = sc.render(cfg, n_threads=n_threads)
        dt = time.time() - t0
        if budget_s:
            more = int(max(spp, min(64, spp * budget_s / max(dt, 1e-3))))
            if more > spp:
                cfg.spp = cfg.spp_per_pass = more
                t0 = time.time()
                _, st = sc.render(cfg, n_threads=n_threads)
                dt = time.time() - t0
        return st["n_samples"] / dt / 1e6, cfg.spp, st["n_samples"], dt

    fd = CONFIGS[key]["force_diffuse"] if CONFIGS[key]["scene"] == "cbox" else 1
    v, spp, ns, dt = run(W, H, 4, fd, 11.0)
    c1v, _, c1n, c1dt = run(256, 256, 64, 0, 0)
    what = "force_diffuse" if fd else "full graph"
    return {"value": v, "unit": "Msamples/s", "cores": cores, "threads": n_threads, "kind": "port",
            "sample": f"cbox {W}x{H} {what} {spp} spp ({ns} camera paths, {dt:.1f} s), CPU oracle (C, pthreads); cores = cgroup CPU quota",
            "c1": {"value": c1v, "unit": "Msamples/s", "sample": f"C1: cbox 256x256 64 spp full graph ({c1n} camera paths, {c1dt:.1f} s)"}}


def measured_counters(key):
    """What rocprofv3's PMC passes measured for this launch shape (they cannot run inside this process): the committed
    summary profiles/r4_pmc_<config>.json (tools/pmc_bench.sh) -- HBM bytes per launch (FETCH_SIZE x its calibrated factor +
    WRITE_SIZE, separate --pmc passes; the factor per access pattern is measured on a known byte count,
    profiles/r4_fetch_size_calibration.json), VALU busy share, lane utilisation. The summary
    carries the hash of the library sources it was measured on; if that is not the hash of the sources in this tree the
    block is dropped and the line says so. Returns (summary or None, note or None)."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{r}_pmc_{key}.json") for r in (6, 5, 4, 3)) if os.path.exists(q)), os.path.join(ROOT, "profiles", f"r6_pmc_{key}.json"))
    if not os.path.exists(path):
        return None, f"no PMC summary profiles/r6_pmc_{key}.json (tools/pmc_bench.sh {key})"
    try:
        m = json.load(open(path))
    except Exception as ex:  # noqa: BLE001
        return None, f"unreadable PMC summary: {ex}"
    have, want = m.get("csrc_hash"), csrc_hash()
    if have != want:
        return None, f"PMC summary {os.path.relpath(path, ROOT)} was measured on other kernel sources (csrc hash {have}, this tree {want}): not quoted"
    return m, None


def resolution(key):
    return CONFIGS[key].get("res", (W, H))


def build_scene(ctx, key):
    from akari_render_amd import capi

    if CONFIGS[key]["scene"] == "cbox":
        w, h = resolution(key)
        return capi.Scene(ctx, os.path.join(ROOT, "scenes", "cbox", "scene.json"), w, h), {}  # akr_scene_load: C++ reader
    from akari_render_amd import procedural

    t0 = time.time()
    sd = procedural.sponza_like(HALL_TRIS, seed=1234, width=W, height=H)
    t1 = time.time()
    scene = capi.Scene(ctx, sd)
    info = scene.info()
    return scene, {"n_triangles": int(info.n_triangles), "n_bvh_nodes": int(info.n_bvh_nodes), "scene_device_MB": info.device_bytes / 1e6,
                   "generate_s": t1 - t0, "compile_upload_s": time.time() - t1}


def _with_deadline(fn, seconds):
    """Runs fn() in a daemon thread; returns (finished, result or exception). A call that does not come back in time is
    abandoned (the thread may still be stuck inside RCCL): the caller must not touch what it was working on again."""
    import threading

    box = {}

    def work():
        try:
            box["result"] = fn()
        except BaseException as ex:  # noqa: BLE001
            box["error"] = ex

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return False, TimeoutError(f"no answer after {seconds:.0f} s")
    if "error" in box:
        return True, box["error"]
    return True, box.get("result")


def make_native_comm(ctx, rank, world, torch, dist, dev, local_rank, deadline_s):
    """The library's own RCCL communicator (akr_comm_create) + ONE warm-up reduce of a full-size film through it, under a
    wall-clock deadline: rank 0's unique id travels over torch.distributed, which is already up for the barrier and the timing
    reduction. Returns (comm or None, ctx, note). akr_film_reduce with more than one rank has never run before the first
    multi-GPU node appears, so nothing here may stall the run: if creation or the warm-up reduce fails or does not return
    within the deadline on ANY rank, every rank drops the communicator AND the context whose stream the stuck call may still
    occupy, makes a fresh context, and the films are reduced with torch.distributed instead (config.film_reduce says which)."""
    from akari_render_amd import capi

    # torch.distributed's own collective stays on the main thread (the current CUDA device is thread-local in PyTorch: a helper
    # thread would start on device 0); only the library's calls -- which bind their context's device themselves -- go under the deadline
    uid, why = None, None
    if rank == 0:
        try:
            uid = capi.comm_unique_id()
        except Exception as ex:  # noqa: BLE001 -- e.g. no librccl to dlopen: every rank must still get past the broadcast
            why = f"akr_comm_unique_id: {type(ex).__name__}: {ex}"
            log("rank 0: " + why + " -- falling back to torch.distributed.reduce")
    box = [uid, why]
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        return None, ctx, box[1]
    scratch = torch.zeros(7 * W * H, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)

    def create_and_warm():
        comm = capi.Comm(ctx, box[0], rank, world)
        sf = capi.Film(ctx, W, H, device_ptr=scratch.data_ptr())
        comm.reduce_film(sf, root=0, blocking=True, planes=PT_PLANES)  # RCCL builds its rings / proxy connections on the first reduce of a size
        del sf
        return comm

    finished, res = _with_deadline(create_and_warm, deadline_s)
    ok = 1 if (finished and not isinstance(res, BaseException)) else 0
    note = None
    if not ok:
        note = f"rank {rank}: akr_comm_create / first akr_film_reduce: {type(res).__name__}: {res}"
        log(note + " -- falling back to torch.distributed.reduce")
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return res, ctx, None
    if ok:
        try:
            res.close()
        except Exception:  # noqa: BLE001
            pass
    # a reduce that never returned still sits on the old context's stream: render on a new one
    return None, capi.Context(local_rank), (note or "another rank could not create the native communicator")


class LegSkipped(RuntimeError):
    pass


def agree(ok, world, torch, dist, cdev):
    """True iff every rank says ok. Called before a leg enters its collectives: a rank that failed to set the leg up (e.g. out of
    memory on the 4K film) must not leave the others waiting in a reduce."""
    if world <= 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


_SCENES = {}  # (config key, forced BVH) -> (scene, info): the 10 M-triangle hall takes 20 s to generate and compile


def run_config(ctx, key, steps, warmup, rank, world, scaling, film_t, torch, dist, backend, dev, comm=None, passes_per_step=PASSES_PER_STEP,
               force_bvh=False, keep_scene=False, split="tiles", sampler="independent"):
    """Times `steps` steps of configuration `key` on this rank. Returns (elapsed_s, per-rank counter deltas, extra info).
    split = "tiles": rank r renders the pixel tiles whose Morton code % world == r. split = "samples" (index-based samplers only): rank r renders
    samples [r S / world, (r + 1) S / world) of EVERY pixel, S = the run's total spp (akr_pt_config.sample_begin / sample_count)."""
    from akari_render_amd import abi, capi, distributed

    cdev = dev if backend == "nccl" else torch.device("cpu")
    weak = world > 1 and scaling == "weak"
    by_samples = world > 1 and split == "samples" and not weak
    setup_error = None
    try:
        if (key, force_bvh) in _SCENES:
            scene, sinfo = _SCENES[(key, force_bvh)]
            sinfo = dict(sinfo)
        else:
            with capi.options(force_bvh=1 if force_bvh else 0):
                scene, sinfo = build_scene(ctx, key)
            if keep_scene:
                _SCENES[(key, force_bvh)] = (scene, dict(sinfo))
        w, h = resolution(key)
        if film_t.numel() != 7 * w * h:  # a configuration with its own frame size (C5: 3840x2160) renders into its own film
            film_t = torch.zeros(7 * w * h, dtype=torch.float32, device=dev)
        film_t.zero_()
        torch.cuda.synchronize(dev)
        film = capi.Film(ctx, w, h, device_ptr=film_t.data_ptr())
        cfg = abi.PtConfig.default()
        cfg.spp = (warmup + steps) * SPP_PER_PASS * passes_per_step
        cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth, cfg.use_nee = SPP_PER_PASS, 12, 5, 1
        cfg.force_diffuse = CONFIGS[key]["force_diffuse"]
        cfg.filter_type, cfg.filter_radius = abi.FILTER_GAUSSIAN, 1.5
        cfg.sampler_type = {"independent": abi.SAMPLER_INDEPENDENT, "sobol": abi.SAMPLER_SOBOL, "pmj02bn": abi.SAMPLER_PMJ02BN}[sampler]
        my_passes = passes_per_step
        if weak:
            cfg.sampler_seed = rank  # independent sample set per GPU (sampler/mod.rs:148-160 seeds the per-pixel streams from it)
        elif by_samples:
            if passes_per_step % world:
                raise ValueError(f"--split samples: {passes_per_step} passes per step do not divide over {world} ranks")
            my_passes = passes_per_step // world
            cfg.sampler_seed = 0
            cfg.sample_count = (warmup + steps) * SPP_PER_PASS * my_passes
            cfg.sample_begin = rank * cfg.sample_count
        else:
            cfg.sampler_seed = 0
            cfg = distributed.shard_config(cfg, rank, world)
        se = capi.PtSession(ctx, scene, cfg, film)
        if warmup > 0:
            se.passes(warmup * my_passes, blocking=True)
    except Exception as ex:  # noqa: BLE001
        setup_error = ex
    if not agree(setup_error is None, world, torch, dist, cdev):
        if setup_error is not None:
            raise setup_error
        raise LegSkipped("another rank could not set this leg up")
    if world > 1 and backend == "nccl" and (comm is None or (w, h) != (W, H)):
        # warm the collective up too (RCCL builds its rings / proxy connections on the first reduce of a given size): same
        # message size as the film, on a scratch buffer, outside the timed region (the native communicator's 1080p warm-up
        # happened under a deadline when it was created, make_native_comm)
        scratch = torch.zeros_like(film_t)
        torch.cuda.synchronize(dev)
        if comm is not None:
            sf = capi.Film(ctx, w, h, device_ptr=scratch.data_ptr())
            comm.reduce_film(sf, root=0, blocking=True, planes=PT_PLANES)
            del sf
        else:
            distributed.reduce_film(scratch, dst=0, planes=PT_PLANES)
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
    se.passes(steps * my_passes, blocking=True)
    t_rendered = time.perf_counter()
    if world > 1:
        if backend == "gloo":
            host = film_t.cpu()
            distributed.reduce_film(host, dst=0, planes=PT_PLANES)
            film_t.copy_(host)
        elif comm is not None:
            comm.reduce_film(film, root=0, blocking=True, planes=PT_PLANES)  # akr_film_reduce_planes: ncclReduce of rgb + weight (one group) on the context's stream, after the render
        else:
            distributed.reduce_film(film_t, dst=0, planes=PT_PLANES)
        ctx.synchronize()
        torch.cuda.synchronize(dev)
    t_reduced = time.perf_counter()
    sync()
    t1 = time.perf_counter()
    arith_relaxed = bool(se.kernel_info()["kernel_flags"] & 16)  # which arithmetic tier the session's launches ran in
    s1 = se.end()
    d = {k: s1[k] - s0[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded", "n_node_visits", "n_tri_tests")}
    d["kernel_ms"] = s1["kernel_ms"] - s0["kernel_ms"]
    d["n_launches"] = s1["n_launches"] - s0["n_launches"]
    # this rank's share of the timed region: rendering (wall clock and kernel time) and its side of the film reduce (which
    # includes waiting for slower ranks): a bad scaling curve can then be attributed without a second run
    d["render_wall_ms"] = (t_rendered - t0) * 1e3
    d["reduce_wall_ms"] = (t_reduced - t_rendered) * 1e3
    info = scene.info()
    d["node_bytes"] = int(getattr(info, "node_bytes", 64) or 64)
    d["tri_bytes"] = int(getattr(info, "tri_bytes", 48) or 48)
    sinfo["weak"] = weak
    sinfo["film_reduce"] = ("none (one GPU)" if world == 1 else
                            "akr_film_reduce_planes (ncclReduce of rgb + weight over RCCL through the C ABI)" if (comm is not None and backend == "nccl") else
                            f"torch.distributed.reduce of rgb + weight ({'RCCL' if backend == 'nccl' else 'gloo, through host memory'})")
    sinfo["film_reduce_bytes"] = 0 if world == 1 else 4 * w * h * 4  # 4 N floats: SURVEY.md 8(e)'s count
    # what the warm-up launches (16 passes = ONE step = 1024 spp each: the session fuses more only once it has timed a pass) cost
    if warmup > 0 and s0["n_launches"] and warmup * my_passes == 16 * s0["n_launches"]:
        sinfo["warmup_launch_ms"] = s0["kernel_ms"] / s0["n_launches"]
    sinfo["arith_relaxed"] = arith_relaxed
    sinfo["spp_done"] = (warmup + steps) * SPP_PER_PASS * passes_per_step
    sinfo["film_tensor"] = film_t
    del film, scene
    return t1 - t0, d, sinfo


def roofline_block(key, d):
    """The dominant kernel (k_pt_pass) of this rank: algorithmic bytes per launch / average launch duration (HIP events
    around each launch on the context's own stream), next to what the PMC counters measured for the same launch shape."""
    launches = max(1, d["n_launches"])
    model_bytes = algorithmic_bytes(d)
    bytes_per_launch = model_bytes / launches
    avg_launch_s = d["kernel_ms"] * 1e-3 / launches
    achieved = bytes_per_launch / avg_launch_s / 1e9
    out = {
        "bound": "hbm",
        "achieved": achieved,
        "achieved_is": "ALGORITHMIC bytes of the SURVEY.md 8(d) byte model (device counters x the model's record sizes: node 64 B, "
                       "triangle 48 B) / launch time -- not measured traffic; frac_measured is the measured one",
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "frac_measured": None,
        "traffic": None,
        "kernel": "k_pt_pass",
        "launches": d["n_launches"],
        "avg_launch_ms": avg_launch_s * 1e3,
        "algorithmic_bytes_per_launch": bytes_per_launch,
        "algorithmic_bytes_per_sample": model_bytes / max(1, d["n_samples"]),
    }
    if d["n_node_visits"]:  # what this build's 64-byte nodes and 64-byte triangle records make of the same counters (not the score)
        out["fetched_bytes_frac"] = algorithmic_bytes(d, fetched=True) / launches / avg_launch_s / 1e9 / HBM_PEAK_GBS
        rays = max(1, d["n_closest"] + d["n_shadow"])
        out["nodes_per_ray"] = d["n_node_visits"] / rays
        out["tris_per_ray"] = d["n_tri_tests"] / rays
    m, note = measured_counters(key)
    samples_per_launch = d["n_samples"] / launches
    if m:
        if m.get("hbm_bytes_per_launch") is not None and (m.get("samples_per_launch") == samples_per_launch or not d["n_node_visits"]):
            # (a cbox launch reads and writes the sampler states and the film once, however many passes it fuses)
            out["traffic"] = m["hbm_bytes_per_launch"]
        elif m.get("hbm_bytes_per_sample") is not None:
            out["traffic"] = m["hbm_bytes_per_sample"] * samples_per_launch
        if out["traffic"] is not None:
            out["hbm_measured_gbs"] = out["traffic"] / avg_launch_s / 1e9
            out["frac_measured"] = out["hbm_measured_frac"] = out["hbm_measured_gbs"] / HBM_PEAK_GBS
        for k in ("valu_busy", "valu_lane_utilisation", "wait_share", "l2_hit", "ta_busy", "binding_limiter", "source", "csrc_hash", "kernel"):
            if k in m:
                out[k if k != "kernel" else "pmc_kernel"] = m[k]
    else:
        out["measured_counters"] = note
    return out


def reference_default_leg(ctx):
    """scenes/cbox/pt.json as it is, through akr_render_task (lib.rs:111-207): 1024 x 1024 (scene.json's own resolution), 4096 spp in
    passes of 64, pmj02bn sampler (the reference's shipped default), gaussian filter, film written to output/pt.exr."""
    import tempfile

    from akari_render_amd import capi

    scene = capi.Scene(ctx, os.path.join(ROOT, "scenes", "cbox", "scene.json"))
    info = scene.info()
    method = open(os.path.join(ROOT, "scenes", "cbox", "pt.json")).read()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            t0 = time.perf_counter()
            st = capi.render_task(ctx, scene, method)
            wall = time.perf_counter() - t0
            wrote = os.path.exists(os.path.join(d, "output", "pt.exr"))
        finally:
            os.chdir(cwd)
    dd = {k: st[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded", "n_node_visits", "n_tri_tests", "kernel_ms", "n_launches")}
    dd["node_bytes"], dd["tri_bytes"] = 64, 48
    return {"metric": "Msamples/s, scenes/cbox/pt.json unchanged (akr_render_task)", "value": st["n_samples"] / wall / 1e6, "unit": "Msamples/s",
            "value_kernels_only": st["n_samples"] / (st["kernel_ms"] * 1e-3) / 1e6, "wall_s": wall, "kernel_ms": st["kernel_ms"], "launches": st["n_launches"],
            "resolution": [int(info.width), int(info.height)], "spp": int(st["n_samples"] // (int(info.width) * int(info.height))), "sampler": "pmj02bn",
            "film_written": bool(wrote), "workload": "scenes/cbox/scene.json + scenes/cbox/pt.json (reference files, unchanged), output stage included",
            "roofline": roofline_block("reference_default", dd), "total_samples": st["n_samples"]}


def textured_room_leg(ctx, only=None):
    """The textured room of tests/helpers.py at 1080p with 2 x 64 MB images (tools/textured_bench.py's workload): the interpreter
    kernels against the per-scene kernel (hiprtc, option specialise), exhaustive (n_floor 1) and BVH (n_floor 8) intersectors;
    4 x 64 spp timed after a 64-spp warm-up. Films of the two settings are compared bit for bit."""
    from akari_render_amd import abi, capi
    from tests.helpers import textured_room

    out = {}
    rng = np.random.default_rng(0)
    big8 = rng.integers(0, 256, size=(4096, 4096, 4), dtype=np.uint8); big8[:, :, 3] = 255
    bigf = rng.random((2048, 2048, 4)).astype(np.float32); bigf[:, :, 2] = 0.5 + 0.5 * bigf[:, :, 2]; bigf[:, :, 3] = 1.0
    table = np.fromfile(os.path.join(ROOT, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    for nf, label in ((1, "exhaustive"), (8, "bvh")):
        if only and not only.startswith(label):
            continue
        sd = textured_room(W, H, n_floor=nf)
        sd.images[0] = abi.ImageData(big8, abi.TEX_FILTER_LINEAR, abi.TEX_REPEAT)
        sd.images[1] = abi.ImageData(bigf, abi.TEX_FILTER_LINEAR, abi.TEX_MIRROR)
        sd.ggx_table = table
        films = {}
        for mode, opt in (("interpreter", 0), ("per_scene", 1)):
            if only and only != f"{label}_{mode}":
                continue
            with capi.options(specialise=opt):
                scene = capi.Scene(ctx, sd)
                film = capi.Film(ctx, W, H)
                cfg = abi.PtConfig.default()
                cfg.spp, cfg.spp_per_pass, cfg.max_depth = 64 * 5, 64, 12
                se = capi.PtSession(ctx, scene, cfg, film)
            ki = se.kernel_info()
            se.passes(1, blocking=True)
            s0 = se.stats()
            t0 = time.perf_counter()
            se.passes(4, blocking=True)
            dt = time.perf_counter() - t0
            s1 = se.end()
            films[mode] = film.read()
            d = {k: s1[k] - s0[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded", "n_node_visits", "n_tri_tests", "n_launches", "kernel_ms")}
            rl = roofline_block(f"textured_{label}_{mode}", d)
            out[f"{label}_{mode}"] = {"value": (s1["n_samples"] - s0["n_samples"]) / dt / 1e6, "unit": "Msamples/s",
                                      "kernel": {k: ki[k] for k in ("specialised", "cache_hit", "min_waves", "vgprs", "scratch_bytes", "compile_ms", "load_ms", "status")},
                                      "roofline": {k: rl.get(k) for k in _LEG_ROOFLINE_KEYS if k in rl}, "total_samples": s1["n_samples"]}
            del se, film, scene
        if len(films) == 2:
            out[f"{label}_films_identical"] = bool(np.array_equal(films["interpreter"].view(np.uint32), films["per_scene"].view(np.uint32)))
    return out


def instanced_forest_leg(ctx, only=None):
    """procedural.instanced_forest(1000, 100_000) at 1080p -- 99.86 M instance-triangles, kept as meshes + instances (two-level
    acceleration structure; flattened it would take 21 GB): 2 x 8 spp, the second launch timed; the same forest at
    10 k triangles per mesh both ways, films compared bit for bit."""
    from akari_render_amd import abi, capi, procedural

    out = {}
    for tris, modes in ((100_000, ("kept",)), (10_000, ("kept", "flattened"))):
        if only and not only.startswith(f"{tris // 1000}k_"):
            continue
        sd = procedural.instanced_forest(1000, tris, width=W, height=H)
        films = {}
        for mode in modes:
            if only and only != f"{tris // 1000}k_{mode}":
                continue
            with capi.options(instancing=1 if mode == "kept" else 0):
                t0 = time.perf_counter()
                scene = capi.Scene(ctx, sd)
                t_load = time.perf_counter() - t0
                info = scene.info()

            def run(**opts):
                film = capi.Film(ctx, W, H)
                cfg = abi.PtConfig.default()
                cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth = 16, 8, 12, 5
                with capi.options(**opts):
                    se = capi.PtSession(ctx, scene, cfg, film)
                status = se.kernel_info()["status"]
                se.passes(1, blocking=True)
                s0 = se.stats()
                t0 = time.perf_counter()
                se.passes(1, blocking=True)
                dt = time.perf_counter() - t0
                s1 = se.end()
                d = {k: s1[k] - s0[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded", "n_node_visits", "n_tri_tests", "n_launches", "kernel_ms")}
                return d, dt, s1, film.read(), "wavefront" if "wavefront" in status else "megakernel"

            d, dt, s1, films[mode], schedule = run()  # the schedule the library chooses (kept scenes of this size: wavefront, api_pt.cpp choose_wavefront)
            rays = d["n_closest"] + d["n_shadow"]
            rl = roofline_block(f"forest_{tris // 1000}k_{mode}", d)  # the same byte model (8d) over the same counters: 64 B per node visit, 48 B per candidate
            rl["kernel"] = ("k_wf_trace<.., INST> + k_wf_shade" if schedule == "wavefront" else "k_pt_pass_inst") if mode == "kept" else "k_pt_pass"
            out[f"{tris // 1000}k_{mode}"] = {"value": d["n_samples"] / dt / 1e6, "unit": "Msamples/s", "schedule": schedule, "n_triangles": int(info.n_triangles), "uses_bvh": int(info.uses_bvh),
                                             "device_MB": info.device_bytes / 1e6, "compile_upload_s": t_load, "rays_per_s_G": rays / dt / 1e9,
                                             "node_visits_per_ray": d["n_node_visits"] / rays, "candidates_per_ray": d["n_tri_tests"] / rays,
                                             "roofline": {k: rl.get(k) for k in _LEG_ROOFLINE_KEYS if k in rl}, "total_samples": s1["n_samples"]}
            if not only:  # the other schedule on the same scene, kept or flattened (not under a profiler: --leg runs the chosen one alone)
                d2, dt2, _, film2, _ = run(wavefront=0 if schedule == "wavefront" else 1)
                out[f"{tris // 1000}k_{mode}"]["other_schedule"] = {"schedule": "megakernel" if schedule == "wavefront" else "wavefront", "value": d2["n_samples"] / dt2 / 1e6,
                                                                  "film_identical": bool(np.array_equal(film2.view(np.uint32), films[mode].view(np.uint32)))}
            del scene
        if len(films) == 2:
            out[f"{tris // 1000}k_films_identical"] = bool(np.array_equal(films["kept"].view(np.uint32), films["flattened"].view(np.uint32)))
    return out


# what a secondary leg's roofline block carries: the model fraction, the measured one (PMC summary of the same sources: tools/pmc_bench.sh <leg>), and what bounds the kernel
_LEG_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "frac_measured", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_sample", "hbm_measured_gbs",
                      "valu_busy", "valu_lane_utilisation", "wait_share", "l2_hit", "nodes_per_ray", "tris_per_ray", "csrc_hash", "measured_counters")
_NOT_REPORTED = ("weak", "spp_done", "film_tensor", "arith_relaxed")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4"], help="the timed workload (BASELINE.json configs[1..3])")
    ap.add_argument("--also", default=None, help="comma list of further configs measured once each after the timed one, reported under "
                                                   "extra_configs (default at N = 1 with --config c2: c3,c4; 'none' to skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--leg", default=None, help="run ONE secondary leg alone and print its JSON (what tools/pmc_bench.sh profiles): reference_default, "
                                                "forest_100k_kept, forest_10k_kept, forest_10k_flattened, textured_exhaustive_interpreter, textured_exhaustive_per_scene, "
                                                "textured_bvh_interpreter, textured_bvh_per_scene")
    ap.add_argument("--full-graph", action="store_true", help="same as --config c3")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong (default) = the ONE frame's 32x32 pixel tiles round-robin over the GPUs, identical image for "
                         "every N; weak = every GPU renders the whole frame with its own sampler seed (N independent sample sets)")
    ap.add_argument("--split", default="tiles", choices=["tiles", "samples"],
                    help="N > 1, strong scaling: what is sharded. tiles (default) = pixels; samples = rank r renders samples "
                         "[r S / N, (r + 1) S / N) of every pixel (needs --sampler sobol | pmj02bn: akr_pt_config.sample_begin / _count)")
    ap.add_argument("--sampler", default="independent", choices=["independent", "sobol", "pmj02bn"],
                    help="independent = BASELINE's configuration (default)")
    ap.add_argument("--reduce", default="native", choices=["native", "torch"],
                    help="N > 1: native = akr_film_reduce (the library's own RCCL call, csrc/host/comm.cpp; falls back to torch if the "
                         "communicator cannot be created or its first reduce does not return), torch = torch.distributed.reduce on the film tensor")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only for smoke-testing the "
                         "multi-rank path on a box with fewer GPUs than ranks: all ranks then share device 0)")
    ap.add_argument("--comm-deadline", type=float, default=120.0, help="seconds akr_comm_create + the first akr_film_reduce may take before the run falls back to --reduce torch")
    ap.add_argument("--legs-deadline", type=float, default=900.0, help="seconds the secondary legs (extra_configs, CPU baseline) may take after the "
                                                                        "headline is measured; past it rank 0 prints the headline line without them")
    args = ap.parse_args()
    if args.full_graph:
        args.config = "c3"
    if args.split == "samples" and args.sampler == "independent":
        ap.error("--split samples needs an index-based sampler (--sampler sobol | pmj02bn): the independent sampler's per-pixel PCG stream is sequential")

    import torch

    from akari_render_amd import capi, distributed

    if args.leg:  # one secondary leg alone (for the profiler): no headline, no other legs
        ctx = capi.Context(0)
        name = args.leg
        if name == "reference_default":
            leg = reference_default_leg(ctx)
        elif name.startswith("forest_"):
            leg = instanced_forest_leg(ctx, only=name[len("forest_"):])[name[len("forest_"):]]
        elif name.startswith("textured_"):
            leg = textured_room_leg(ctx, only=name[len("textured_"):])[name[len("textured_"):]]
        else:
            ap.error("unknown --leg " + name)
        print(json.dumps({"leg": name, **leg}), flush=True)
        return

    rank, world, local_rank = distributed.env_rank_world()
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus} (WORLD_SIZE={world})"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        distributed.init_process_group(args.backend)
    import torch.distributed as dist

    dev_index = local_rank if world > 1 else 0
    ctx = capi.Context(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")
    film_t = torch.zeros(7 * W * H, dtype=torch.float32, device=dev)
    key = args.config
    comm, comm_note = None, None
    if world > 1 and args.backend == "nccl" and args.reduce == "native":
        comm, ctx, comm_note = make_native_comm(ctx, rank, world, torch, dist, dev, dev_index, args.comm_deadline)
    elapsed, d, sinfo = run_config(ctx, key, args.steps, args.warmup, rank, world, args.scaling, film_t, torch, dist, args.backend, dev, comm,
                                   split=args.split, sampler=args.sampler)
    weak = sinfo["weak"]

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([d["n_samples"], d["n_closest"], d["n_shadow"], d["n_shaded"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_samples = int(c[0].item())
        per = torch.tensor([d["kernel_ms"], d["render_wall_ms"], d["reduce_wall_ms"], float(d["n_samples"])], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(per) for _ in range(world)]
        dist.all_gather(allr, per)
        per_rank = [[float(x) for x in t.tolist()] for t in allr]
    else:
        total_samples = d["n_samples"]
        per_rank = [[d["kernel_ms"], d["render_wall_ms"], d["reduce_wall_ms"], float(d["n_samples"])]]
    n_sets = world if weak else 1
    assert total_samples == n_sets * W * H * SPP_PER_STEP * args.steps, (total_samples, n_sets * W * H * SPP_PER_STEP * args.steps)

    # ---- the headline line is complete HERE, before any secondary leg runs: whatever happens below, rank 0 can print it ----
    out = None
    if rank == 0:
        # frame sanity inside the bench: every pixel got its samples (exact per-pixel comparison), film finite
        expect_w = float(n_sets * sinfo["spp_done"])
        wplane = film_t[6 * W * H:]
        assert bool((wplane == expect_w).all().item()), (float(wplane.min().item()), float(wplane.max().item()), expect_w)
        assert bool(torch.isfinite(film_t).all().item())
        cfgd = CONFIGS[key]
        scal = "weak" if weak else "strong"
        by_samples = world > 1 and args.split == "samples" and not weak
        out = {
            "metric": "Msamples/s (whole node), 1080p path tracer, " + cfgd["name"] + (" -- weak scaling: N independent sample sets" if weak else ""),
            "value": total_samples / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": scal,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: " + ("scenes/cbox (36 triangles, reference scene data)" if cfgd["scene"] == "cbox" else "procedural hall, generator seed 1234")
                    + f" at {W}x{H}, {args.sampler} sampler, seed " + ("= rank" if weak else "0"),
            "config": {
                "workload": cfgd["workload"] + ", max_depth 12, rr_depth 5, NEE, gaussian filter r=1.5",
                "baseline_config": f"BASELINE.json configs[{cfgd['baseline_config']}]",
                "spp_per_step": SPP_PER_STEP,
                "arithmetic": "relaxed tier (option arith = 1): NOT bit-exact with the oracle" if sinfo.get("arith_relaxed") else "AKR-F32 contract: bit-exact with the oracle",
                "spp_total": args.steps * SPP_PER_STEP * n_sets,
                "parallelism": ("single GPU" if args.gpus == 1 else
                                f"{args.gpus} independent sample sets of {args.steps * SPP_PER_STEP} spp (sampler seed = rank), one per GPU, films sum-reduced" if weak else
                                f"sample ranges: GPU r of {args.gpus} renders samples [r S / N, (r + 1) S / N) of every pixel, film sum-reduce" if by_samples else
                                f"pixel tiles 32x32 round-robin over {args.gpus} GPUs, film sum-reduce"),
                **{k: v for k, v in sinfo.items() if k not in _NOT_REPORTED},
                **({"film_reduce_fallback": comm_note} if comm_note else {}),
                # per rank over the timed region: kernel time (HIP events), render wall clock, film-reduce wall clock (incl. waiting)
                "per_rank_kernel_ms": {"max": max(r[0] for r in per_rank), "min": min(r[0] for r in per_rank), "all": [round(r[0], 2) for r in per_rank]},
                "per_rank_render_wall_ms": {"max": max(r[1] for r in per_rank), "min": min(r[1] for r in per_rank)},
                "film_reduce_wall_ms": {"max": max(r[2] for r in per_rank), "min": min(r[2] for r in per_rank), "rank0": per_rank[0][2]},
                "per_rank_samples": [int(r[3]) for r in per_rank],
            },
            "roofline": roofline_block(key, d),
            "counters": {k: d[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded", "n_node_visits", "n_tri_tests")},
        }
        # the library fuses the timed region's passes into launches of up to 64 once the warm-up has told the session what a pass costs
        # (akr_pt_passes): a launch is then several steps, and rocprofv3's per-kernel average mixes it with the 16-pass warm-up launches
        out["roofline"]["passes_per_launch"] = args.steps * PASSES_PER_STEP / max(1, d["n_launches"])
        out["roofline"]["steps_per_launch"] = args.steps / max(1, d["n_launches"])
        # ... so the line says what its value is the throughput OF: launches of `spp_per_launch` samples per pixel; and next to it what the
        # configuration's literal 1024-spp render (one 16-pass launch) gives, from the warm-up launches' HIP-event times on this rank
        out["config"]["passes_per_launch"] = out["roofline"]["passes_per_launch"]
        out["config"]["spp_per_launch"] = out["roofline"]["passes_per_launch"] * SPP_PER_PASS
        if world == 1 and sinfo.get("warmup_launch_ms"):
            out["value_1024spp_launch"] = W * H * SPP_PER_STEP / (sinfo["warmup_launch_ms"] * 1e-3) / 1e6
            out["config"]["value_1024spp_launch_is"] = "Msamples/s of a 16-pass (1024 spp) launch: mean HIP-event time of the warm-up launches"
    extra = {}
    printed = []
    import threading

    line_lock = threading.Lock()  # the watchdog thread serialises the line while the main thread may still be adding legs to it

    def emit(note=None):
        with line_lock:
            if rank == 0 and not printed:
                printed.append(1)
                line = dict(out)
                legs = dict(extra)
                if note:
                    legs["incomplete"] = note
                if legs:
                    line["extra_configs"] = legs
                print(json.dumps(line), flush=True)

    def add_leg(name, leg):
        with line_lock:
            extra[name] = leg

    # A secondary leg that never comes back (a rank lost inside a collective) must not cost the headline: past the deadline rank 0
    # prints what it has and every rank leaves.
    def give_up():
        try:
            log(f"rank {rank}: secondary legs did not finish within {args.legs_deadline:.0f} s; printing the headline line without them")
            emit(f"secondary legs did not finish within {args.legs_deadline:.0f} s")
        finally:
            os._exit(0)

    dog = threading.Timer(args.legs_deadline, give_up)
    dog.daemon = True
    dog.start()

    # Legs at EVERY N (all ranks take part; one step each, outside the timed region of the headline):
    #  c5_strong              BASELINE configs[4]: the 3840x2160 frame, full graph, its tiles over the N ranks -- the configuration the
    #                         reference's multi-GPU target is written for; 4x the pixels per rank of the 1080p frame (at N = 8 a rank of
    #                         the 1080p frame holds 2.2 waves per SIMD of long-running pixels, of the 4K frame 8.8)
    #  c2_weak                N > 1: every rank renders the whole 1080p frame with its own sampler seed (N independent sample sets)
    #  c2_sobol_sample_split  the 1080p frame with the sobol sampler, the SAMPLES of every pixel over the ranks (at N = 1: the same
    #                         frame on one GPU, the number the split's speed-up is measured against)
    if args.also != "none" and key == "c2" and args.split == "tiles" and args.sampler == "independent":
        legs = (("c5_strong", "c5", "strong", "tiles", "independent"),) + ((("c2_weak", "c2", "weak", "tiles", "independent"),) if world > 1 else ())
        legs += (("c2_sobol_sample_split", "c2", "strong", "samples", "sobol"),)
        for name, k2, scal2, split2, smp2 in legs:
            try:
                e2, d2, si2 = run_config(ctx, k2, 1, 1 if world > 1 else 0, rank, world, scal2, torch.zeros(1, dtype=torch.float32, device=dev), torch, dist,
                                         args.backend, dev, comm, split=split2, sampler=smp2)
                n2 = d2["n_samples"]
                if world > 1:
                    t2 = torch.tensor([e2, float(n2)], dtype=torch.float64, device=cdev)
                    tmax = t2.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                    tsum = t2.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
                    e2, n2 = float(tmax[0].item()), int(tsum[1].item())
                leg = {"metric": "Msamples/s (whole node), " + CONFIGS[k2]["name"] + (" -- weak scaling: N independent sample sets" if scal2 == "weak" else "")
                                 + (", sobol sampler, sample ranges over the ranks" if split2 == "samples" else ""),
                       "value": n2 / e2 / 1e6, "unit": "Msamples/s", "n_gpus": args.gpus, "scaling": scal2, "split": split2, "sampler": smp2, "steps": 1,
                       "ms_per_step": e2 * 1e3, "resolution": list(resolution(k2)), "workload": CONFIGS[k2]["workload"], "film_reduce": si2["film_reduce"]}
                if rank == 0 and split2 == "samples":  # every pixel holds all S samples after the reduce
                    w2, h2 = resolution(k2)
                    wp = si2["film_tensor"][6 * w2 * h2:]
                    leg["weight_plane_ok"] = bool((wp == float(si2["spp_done"])).all().item())
                add_leg(name, leg)
                del si2
            except Exception as ex:  # noqa: BLE001 -- a secondary leg must not cost the headline line
                add_leg(name, {"error": f"{type(ex).__name__}: {ex}"})
    if rank == 0:
        also = args.also
        if also is None:
            also = "c3,c4" if (args.gpus == 1 and key == "c2") else "none"
        for k2 in [k for k in also.split(",") if k and k != "none" and k != key]:
            try:
                e2, d2, si2 = run_config(ctx, k2, 1, 0 if k2 == "c4" else 1, 0, 1, "strong", film_t, torch, dist, args.backend, dev, keep_scene=(k2 == "c4"))
                add_leg(k2, {"metric": "Msamples/s, " + CONFIGS[k2]["name"], "value": d2["n_samples"] / e2 / 1e6, "unit": "Msamples/s",
                             "steps": 1, "ms_per_step": e2 * 1e3, "workload": CONFIGS[k2]["workload"],
                             "rays_per_s_G": (d2["n_closest"] + d2["n_shadow"]) / e2 / 1e9,
                             "roofline": roofline_block(k2, d2),
                             "counters": {k: d2[k] for k in ("n_samples", "n_closest", "n_shadow", "n_shaded", "n_node_visits", "n_tri_tests")},
                             **{k: v for k, v in si2.items() if k not in _NOT_REPORTED}})
            except Exception as ex:  # a secondary leg must not cost the headline line
                add_leg(k2, {"error": f"{type(ex).__name__}: {ex}"})
        # The reference's own shipped workload, unchanged: scenes/cbox/scene.json at the resolution the file names with scenes/cbox/pt.json
        # (4096 spp, pmj02bn, gaussian 1.5, full graph) through akr_render_task -- what `akari-cli -s scene.json -m pt.json` runs,
        # output stage included (the film written as OpenEXR into a scratch directory).
        if also != "none" and args.gpus == 1 and key == "c2":
            try:
                add_leg("reference_default", reference_default_leg(ctx))
            except Exception as ex:  # noqa: BLE001
                add_leg("reference_default", {"error": f"{type(ex).__name__}: {ex}"})
            try:
                add_leg("textured_room", textured_room_leg(ctx))
            except Exception as ex:  # noqa: BLE001
                add_leg("textured_room", {"error": f"{type(ex).__name__}: {ex}"})
            try:
                add_leg("instanced_forest", instanced_forest_leg(ctx))
            except Exception as ex:  # noqa: BLE001
                add_leg("instanced_forest", {"error": f"{type(ex).__name__}: {ex}"})
        # The wavefront schedule (wf_kernels.hip: trace / shade kernels, path state in HBM, ballot + prefix-sum compaction) next to the
        # megakernel on the same scenes, 4 passes (256 spp) each: the hall, and the cbox with a forced BVH (the wavefront schedule has
        # no exhaustive intersector). Measured every run so that the comparison in DESIGN.md is never a stale number.
        if also != "none" and args.gpus == 1 and key == "c2":
            from akari_render_amd import capi as _capi
            sched = {}
            for name, k2, fbvh in (("c2_forced_bvh", "c2", True), ("c4", "c4", False)):
                for mode in ("megakernel", "wavefront"):
                    try:
                        with _capi.options(wavefront=1 if mode == "wavefront" else 0):
                            e2, d2, _ = run_config(ctx, k2, 1, 1 if k2 == "c2" else 0, 0, 1, "strong", film_t, torch, dist, args.backend, dev, passes_per_step=4,
                                                   force_bvh=fbvh, keep_scene=True)
                        sched[f"{name}_{mode}"] = {"value": d2["n_samples"] / e2 / 1e6, "unit": "Msamples/s", "spp": 4 * SPP_PER_PASS,
                                                   "rays_per_s_G": (d2["n_closest"] + d2["n_shadow"]) / e2 / 1e9, "launches": d2["n_launches"]}
                    except Exception as ex:  # noqa: BLE001
                        sched[f"{name}_{mode}"] = {"error": f"{type(ex).__name__}: {ex}"}
            add_leg("schedules", sched)
            # The price of the bit-exact arithmetic contract: the same megakernels in the relaxed tier (option arith = 1: hardware rcp / sqrt /
            # sin / cos / log / exp, contraction; csrc/pt_kernels_relaxed.hip) on C2, C3 and C4, one step each next to a step of the contract
            # tier measured the same way. NOT the headline: against the oracle at fixed seed the relaxed tier is within 1e-3 on C2's shard
            # but not on C1 / C3 (comparisons that flip move the independent sampler's stream: tests/test_gpu_relaxed.py).
            tiers = {}
            for k2 in ("c2", "c3", "c4"):
                row = {}
                for tier, a in (("contract", 0), ("relaxed", 1)):
                    try:
                        with _capi.options(arith=a):
                            e2, d2, _ = run_config(ctx, k2, 1, 0 if k2 == "c4" else 1, 0, 1, "strong", film_t, torch, dist, args.backend, dev, keep_scene=True)
                        row[tier] = d2["n_samples"] / e2 / 1e6
                    except Exception as ex:  # noqa: BLE001
                        row[tier + "_error"] = f"{type(ex).__name__}: {ex}"
                if "contract" in row and "relaxed" in row:
                    row["relaxed_over_contract"] = row["relaxed"] / row["contract"]
                row["unit"] = "Msamples/s"
                tiers[k2] = row
            tiers["note"] = ("option arith: 0 = AKR-F32 contract (the headline, bit-exact with the oracle), 1 = relaxed tier (opt-in; relRMSE vs oracle at fixed seed: "
                             "C2 shards 0.8e-3 .. 1.3e-3, C1 2e-3 .. 9e-3 -- flipped comparisons, see tests/test_gpu_relaxed.py)")
            add_leg("arithmetic_tiers", tiers)
            _SCENES.clear()
        if args.gpus == 1 and not args.no_cpu_baseline:
            try:
                base = cpu_baseline(key, *host_threads())
            except Exception as ex:  # noqa: BLE001
                base = {"error": f"{type(ex).__name__}: {ex}"}
            with line_lock:
                out["cpu_baseline"] = base
    dog.cancel()
    emit()
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
