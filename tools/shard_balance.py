"""Single-GPU estimate of the multi-GPU strong-scaling efficiency: renders the C2 workload once unsharded and once per
rank of an N-way tile sharding (each on the whole GPU) and compares max-over-ranks kernel time with T1 / N.
python tools/shard_balance.py [N=8] [steps=16]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from akari_render_amd import abi, capi, distributed
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H = (3840, 2160) if "--4k" in sys.argv else (1920, 1080)
ctx = capi.Context(0)
scene = capi.Scene(ctx, os.path.join(ROOT, "scenes/cbox/scene.json"), W, H)


def run(rank, world):
    film = capi.Film(ctx, W, H)
    cfg = abi.PtConfig.default()
    cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.force_diffuse = 64 * (steps + 1), 64, 12, 1
    cfg = distributed.shard_config(cfg, rank, world)
    se = capi.PtSession(ctx, scene, cfg, film)
    se.passes(1, blocking=True)
    s0 = se.stats()
    se.passes(steps, blocking=True)
    s1 = se.end()
    return s1["kernel_ms"] - s0["kernel_ms"], s1["n_samples"] - s0["n_samples"]


t1, n1 = run(0, 1)
per = [run(r, N) for r in range(N)]
tmax = max(t for t, _ in per)
print(json.dumps({"resolution": [W, H], "steps": steps, "T1_ms": t1, "ranks": N, "per_rank_ms": [round(t, 2) for t, _ in per],
                  "samples_per_rank": [n for _, n in per], "ideal_ms": t1 / N, "kernel_scaling_efficiency": t1 / N / tmax}))
