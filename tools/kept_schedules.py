"""Scenes kept as meshes + instances: the megakernel against the wavefront schedule (k_wf_trace<.., INST>), 1080p forest, 2 x 8 spp, second
launch timed; films compared bit for bit.  python tools/kept_schedules.py [tris_per_mesh ...]   (needs a GPU)
KS_REBRAID=1,4,16: the same for each value of option rebraid (top-level tree over that many (instance, subtree) pairs per instance);
KS_GROUPS=1,4,8: ... of option wf_groups; KS_SIZE=WxH; KS_INSTANCING=0: the forest flattened."""
import zlib, os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from akari_render_amd import capi, procedural, abi
W, H = (int(a) for a in os.environ.get("KS_SIZE", "1920x1080").split("x"))
SPP = int(os.environ.get("KS_SPP", "8"))  # samples per pixel and launch (the second launch is timed)
ctx = capi.Context(0)
for tris in [int(a) for a in sys.argv[1:]] or [10_000, 100_000]:
    sd = procedural.instanced_forest(1000, tris, width=W, height=H)
    if os.environ.get("KS_TEXTURED"):  # leaf and bark colours / roughness from image textures over the meshes' uvs (megakernel: per-scene kernel)
        rng = np.random.default_rng(5)
        N = abi.NodeData
        leaf = rng.integers(0, 256, size=(256, 256, 4), dtype=np.uint8); leaf[:, :, 3] = 255; leaf[:, :, 1] |= 128
        rough = rng.random((64, 64, 4)).astype(np.float32); rough[:, :, 3] = 1.0
        sd.images = [abi.ImageData(leaf, abi.TEX_FILTER_LINEAR, abi.TEX_REPEAT), abi.ImageData(rough, abi.TEX_FILTER_LINEAR, abi.TEX_MIRROR)]
        for mi in (2, 3):
            g = abi.GraphData([N(abi.NODE_IMAGE, (0, abi.NODE_NONE, 1)), N(abi.NODE_SPECTRAL_UPLIFT, (0,)),
                               N(abi.NODE_IMAGE, (1, abi.NODE_NONE, 0)), N(abi.NODE_SEPARATE_COLOR, (2,)), N(abi.NODE_EXTRACT, (3, abi.FIELD_GREEN))],
                              {"base_color": 1, "roughness": 4})
            sd.materials[mi].graph = g
    films = {}
    combos = [(rb, 0, 1) for rb in [int(a) for a in os.environ.get("KS_REBRAID", "1").split(",")]]
    combos += [(rb, 1, g) for rb, _, _ in list(combos) for g in [int(a) for a in os.environ.get("KS_GROUPS", "4").split(",")]]
    for rb, wf, groups in combos:
        with capi.options(instancing=int(os.environ.get("KS_INSTANCING", "1")), wavefront=wf, rebraid=rb, wf_groups=groups, specialise=int(os.environ.get("KS_SPECIALISE", "-1")), wf_carry=int(os.environ.get("KS_CARRY", "1"))):
            sc = capi.Scene(ctx, sd); f = capi.Film(ctx, W, H)
            cfg = abi.PtConfig.default(); cfg.spp, cfg.spp_per_pass, cfg.max_depth, cfg.rr_depth = 2 * SPP, SPP, int(os.environ.get("KS_DEPTH", "12")), 5
            se = capi.PtSession(ctx, sc, cfg, f)
        status = se.kernel_info()["status"]
        se.passes(1, blocking=True); s0 = se.stats(); t = time.perf_counter(); se.passes(1, blocking=True); dt = time.perf_counter() - t; s1 = se.end()
        films[(rb, wf, groups)] = f.read()
        rays = (s1["n_closest"] - s0["n_closest"]) + (s1["n_shadow"] - s0["n_shadow"])
        print(json.dumps({"size": [W, H], "spp_per_launch": SPP, "tris_per_mesh": tris, "rebraid": rb, "schedule": "wavefront" if wf else "megakernel", "kernel": status[:40], "wf_groups": groups, "msamples_per_s": (s1["n_samples"] - s0["n_samples"]) / dt / 1e6,
                          "rays_per_s_G": rays / dt / 1e9, "nodes_per_ray": (s1["n_node_visits"] - s0["n_node_visits"]) / rays, "candidates_per_ray": (s1["n_tri_tests"] - s0["n_tri_tests"]) / rays}), flush=True)
        del se, f, sc
    ref = next(iter(films.values())).view(np.uint32)
    print(json.dumps({"tris_per_mesh": tris, "films_identical": bool(all(np.array_equal(ref, f.view(np.uint32)) for f in films.values())),
                      "film_crc32": "%08x" % zlib.crc32(ref.tobytes())}), flush=True)
