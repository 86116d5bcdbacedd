"""First-light check on a GPU box: HIP path vs oracle on cbox (bit-level diff stats), plus timing."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from akari_render_amd import abi, capi
from oracle import pyoracle, scene_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
ctx = capi.Context(0)
out["device"] = ctx.device_info()
print(out["device"], flush=True)

# math probes
x = np.concatenate([np.linspace(0, 6.2831855, 100001, dtype=np.float32), np.random.default_rng(1).random(100000, dtype=np.float32) * 6.2831855]).astype(np.float32)
s, c, l = capi.probe_math(ctx, x)
so = np.zeros_like(x); co = np.zeros_like(x); lo = np.zeros_like(x)
import ctypes as C
L = pyoracle.lib()
for i in range(0, x.size, 997):
    a, b = C.c_float(), C.c_float()
    L.or_kat_sincos(float(x[i]), C.byref(a), C.byref(b)); so[i], co[i] = a.value, b.value
    lo[i] = L.or_kat_log(float(x[i]))
idx = np.arange(0, x.size, 997)
out["math_bitexact"] = bool(np.array_equal(s[idx].view(np.uint32), so[idx].view(np.uint32)) and np.array_equal(c[idx].view(np.uint32), co[idx].view(np.uint32)) and np.array_equal(l[idx].view(np.uint32), lo[idx].view(np.uint32)))
out["sin_max_err_vs_f64"] = float(np.max(np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64)))))
print("math bitexact", out["math_bitexact"], out["sin_max_err_vs_f64"], flush=True)

def compare(W, H, spp, force_diffuse, tag, max_depth=12):
    sd = scene_json.load_scene(os.path.join(ROOT, "scenes/cbox/scene.json"), W, H)
    cfg = abi.PtConfig.default()
    cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth = spp, min(spp, 64), max_depth, 5
    cfg.force_diffuse = force_diffuse
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, W, H)
    t0 = time.time()
    st = capi.pt_render(ctx, scene, cfg, film)
    t_gpu = time.time() - t0
    g = film.read()
    osc = pyoracle.OracleScene(sd)
    t0 = time.time()
    o, ost = osc.render(cfg)
    t_cpu = time.time() - t0
    N = W * H
    diff = np.flatnonzero(g.view(np.uint32) != o.view(np.uint32))
    gi, oi = pyoracle.resolve(g, W, H), pyoracle.resolve(o, W, H)
    lum = oi @ np.array([0.2126, 0.7152, 0.0722])
    rel = float(np.sqrt(np.mean(np.sum((gi - oi) ** 2, axis=2))) / np.mean(lum))
    res = dict(tag=tag, W=W, H=H, spp=spp, n_diff_floats=int(diff.size), n_floats=int(g.size), relRMSE=rel, max_abs=float(np.max(np.abs(gi - oi))),
               gpu_stats=st, cpu_stats=ost, gpu_wall_s=t_gpu, cpu_wall_s=t_cpu, msamples_per_s_kernel=st["n_samples"] / (st["kernel_ms"] * 1e-3) / 1e6,
               cpu_msamples_per_s=ost["n_samples"] / t_cpu / 1e6, cpu_threads=os.cpu_count(), mean_rgb=[float(v) for v in gi.mean(axis=(0, 1))])
    print(json.dumps(res), flush=True)
    np.save(os.path.join(ROOT, "gpurun_out", f"img_{tag}.npy"), gi.astype(np.float32))
    return res

out["runs"] = []
out["runs"].append(compare(64, 64, 16, 1, "c64_diffuse"))
out["runs"].append(compare(64, 64, 16, 0, "c64_full"))
out["runs"].append(compare(256, 256, 64, 0, "c1_full"))
out["runs"].append(compare(256, 256, 64, 1, "c1_diffuse"))

# throughput at 1080p, a few passes
sd = scene_json.load_scene(os.path.join(ROOT, "scenes/cbox/scene.json"), 1920, 1080)
for fd in (1, 0):
    cfg = abi.PtConfig.default()
    cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth, cfg.force_diffuse = 128, 64, 12, 5, fd
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, 1920, 1080)
    st = capi.pt_render(ctx, scene, cfg, film)
    r = dict(tag=f"1080p_fd{fd}", stats=st, msamples_per_s=st["n_samples"] / (st["kernel_ms"] * 1e-3) / 1e6)
    print(json.dumps(r), flush=True)
    out["runs"].append(r)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "first_light.json"), "w"), indent=1)
