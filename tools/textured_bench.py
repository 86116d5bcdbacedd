"""Throughput of the TEX kernels: the textured room of tests/helpers.py at 1920x1080 with large images
(4096^2 RGBA8 = 64 MB, 2048^2 RGBA32F = 64 MB). python tools/textured_bench.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from akari_render_amd import abi, capi
from tests.helpers import textured_room

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_floor = int(sys.argv[2]) if len(sys.argv) > 2 else int(os.environ.get("TEXBENCH_NFLOOR", "1"))   # > 1 tessellates the floor: the scene then takes the BVH path
rng = np.random.default_rng(0)
sd = textured_room(1920, 1080, n_floor=n_floor)
big8 = rng.integers(0, 256, size=(4096, 4096, 4), dtype=np.uint8); big8[:, :, 3] = 255
bigf = rng.random((2048, 2048, 4)).astype(np.float32); bigf[:, :, 2] = 0.5 + 0.5 * bigf[:, :, 2]; bigf[:, :, 3] = 1.0
sd.images[0] = abi.ImageData(big8, abi.TEX_FILTER_LINEAR, abi.TEX_REPEAT)
sd.images[1] = abi.ImageData(bigf, abi.TEX_FILTER_LINEAR, abi.TEX_MIRROR)
sd.ggx_table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)
ctx = capi.Context(0)
out = {}
def constant_room(lobes_of_the_textured_room):
    v = textured_room(1920, 1080, n_floor=n_floor)
    for m in v.materials:
        m.graph = None
    if lobes_of_the_textured_room:  # what the graphs feed, as constants: the same lobes run, no graph is evaluated
        v.materials[0].metallic, v.materials[0].roughness = 0.25, 0.5
    v.materials[6].emission_color = (6.0, 6.0, 6.0)
    v.images = []
    v.ggx_table = sd.ggx_table
    return v

small = textured_room(1920, 1080, n_floor=n_floor)  # the 16x24 / 8x8 images of the tests: texel gathers hit in cache
small.ggx_table = sd.ggx_table
variants = [("textured", sd, None), ("textured, per-scene kernel", sd, None), ("textured, per-scene kernel, 4 waves", sd, None),
            ("textured, conductor deferral off", sd, "0"), ("textured, small images", small, None),
            ("same room, constant materials with the lobes the graphs select", constant_room(True), None),
            ("same room, constant materials", constant_room(False), None)]
only = os.environ.get("TEXBENCH_ONLY")  # substring of the variant's name
for name, variant, defer in variants:
    if only and only != name: continue
    capi.set_option("defer_metal", -1 if defer is None else int(defer))
    if os.environ.get("TEXBENCH_DEFER_ON"):  # which hits the BVH kernels' deferral puts off (1 conductor, 2 textured, 3 both); forces the deferral on
        capi.set_option("defer_on", int(os.environ["TEXBENCH_DEFER_ON"]))
        if defer is None: capi.set_option("defer_metal", int(os.environ.get("TEXBENCH_DEFER_MASK", "1")))
    capi.set_option("specialise", 1 if "per-scene" in name else 0)
    capi.set_option("specialise_waves", 4 if "4 waves" in name else 0)
    scene = capi.Scene(ctx, variant)
    film = capi.Film(ctx, 1920, 1080)
    cfg = abi.PtConfig.default(); cfg.spp = 64 * (steps + 1); cfg.spp_per_pass = 64; cfg.max_depth = 12
    se = capi.PtSession(ctx, scene, cfg, film)
    se.passes(1, blocking=True); s0 = se.stats()
    t0 = time.perf_counter(); se.passes(steps, blocking=True); t1 = time.perf_counter()
    ki = se.kernel_info()
    s1 = se.end()
    out[name] = {"msamples_per_s": (s1["n_samples"] - s0["n_samples"]) / (t1 - t0) / 1e6, "device_MB": scene.info().device_bytes / 1e6,
                 "kernel": {k: ki[k] for k in ("specialised", "cache_hit", "min_waves", "vgprs", "scratch_bytes", "compile_ms", "load_ms", "status")},
                 "shaded_per_sample": (s1["n_shaded"] - s0["n_shaded"]) / (s1["n_samples"] - s0["n_samples"])}
print(json.dumps(out))
