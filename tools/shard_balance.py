"""Single-GPU estimate of the multi-GPU strong-scaling efficiency: renders the cbox workload once unsharded and once per
rank of an N-way tile sharding (each on the whole GPU) and compares max-over-ranks kernel time with T1 / N.
python tools/shard_balance.py [N=8] [steps=16] [--fd | --full] [--4k] [--split samples [--sampler sobol|pmj02bn]]
(--fd = C2, force_diffuse, the default; --full = C3; --split samples: rank r renders samples [r S / N, (r + 1) S / N) of EVERY pixel --
akr_pt_config.sample_begin / sample_count, index-based samplers only -- instead of the pixel tiles morton(tx, ty) % N == r)
No multi-GPU hardware is involved: what this measures is how well 1/N of the frame fills ONE GPU -- the kernel-side term of
strong scaling. The film reduce (one ncclReduce of 7 W H floats) and launch overheads come on top."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from akari_render_amd import abi, capi, distributed
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nums = [a for a in sys.argv[1:] if a.isdigit()]
N = int(nums[0]) if len(nums) > 0 else 8
steps = int(nums[1]) if len(nums) > 1 else 16
W, H = (3840, 2160) if "--4k" in sys.argv else (1920, 1080)
FD = 0 if "--full" in sys.argv else 1
SPLIT = "samples" if ("--split" in sys.argv and sys.argv[sys.argv.index("--split") + 1] == "samples") else "tiles"
SAMPLER = sys.argv[sys.argv.index("--sampler") + 1] if "--sampler" in sys.argv else ("sobol" if SPLIT == "samples" else "independent")
SAMPLER_TYPE = {"independent": abi.SAMPLER_INDEPENDENT, "sobol": abi.SAMPLER_SOBOL, "pmj02bn": abi.SAMPLER_PMJ02BN}[SAMPLER]
ctx = capi.Context(0)
scene = capi.Scene(ctx, os.path.join(ROOT, "scenes/cbox/scene.json"), W, H)


def run(rank, world):
    film = capi.Film(ctx, W, H)
    cfg = abi.PtConfig.default()
    cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.force_diffuse = 64 * (steps + 1), 64, 12, FD
    cfg.sampler_type = SAMPLER_TYPE
    timed = steps
    if SPLIT == "samples" and world > 1:
        # the timed passes' samples are divided over the ranks (steps must be a multiple of `world`); one warm-up pass each
        assert steps % world == 0, "--split samples: steps must be a multiple of N"
        timed = steps // world
        cfg.spp = 64 * (timed + 1) * world
        cfg.sample_count = 64 * (timed + 1)
        cfg.sample_begin = rank * cfg.sample_count
    else:
        cfg = distributed.shard_config(cfg, rank, world)
    se = capi.PtSession(ctx, scene, cfg, film)
    se.passes(1, blocking=True)
    s0 = se.stats()
    se.passes(timed, blocking=True)
    s1 = se.end()
    return s1["kernel_ms"] - s0["kernel_ms"], s1["n_samples"] - s0["n_samples"]


t1, n1 = run(0, 1)
per = [run(r, N) for r in range(N)]
tmax = max(t for t, _ in per)
print(json.dumps({"config": "C2 (force_diffuse)" if FD else "C3 (full graph)", "resolution": [W, H], "steps": steps, "spp_per_step": 64,
                  "library": os.environ.get("AKR_HIP_LIB", "product"), "split": SPLIT, "sampler": SAMPLER, "T1_ms": t1, "ranks": N,
                  "per_rank_ms": [round(t, 2) for t, _ in per], "samples_per_rank": [n for _, n in per], "ideal_ms": t1 / N,
                  "kernel_scaling_efficiency": t1 / N / tmax, "predicted_speedup": t1 / tmax,
                  "msamples_per_s_full_gpu": n1 / t1 / 1e3}))
