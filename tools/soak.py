"""Randomised parity soak: random small scenes (random triangle soups + a few quads, random materials drawn from edge values,
random emitters, cameras, samplers, configs, colour pipelines; a third of them with random images and random shader-graph DAGs
feeding random inputs), the HIP path tracer against the oracle, film accumulators and
counters bit for bit. Prints the seeds that differ. python tools/soak.py [n_cases] [first_seed] [tex] [big] [aov | gpt | mcmc | shard | wavefront [carry] | inst]   (needs a GPU; uses oracle/)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from akari_render_amd import abi, capi
from oracle import pyoracle

EDGE = [0.0, 1e-5, 1e-4, 0.03, 0.25, 0.5, 0.75, 0.9999, 1.0]


def rand_material(rng, emissive=False):
    pick = lambda: float(rng.choice(EDGE)) if rng.random() < 0.5 else float(rng.random())  # noqa: E731
    col = lambda s=1.0: tuple(float(x) * s for x in (rng.random(3) if rng.random() < 0.8 else rng.choice([0.0, 1.0], 3)))  # noqa: E731
    kind = int(rng.choice([abi.MAT_PRINCIPLED] * 6 + [abi.MAT_DIFFUSE, abi.MAT_GLASS]))
    m = abi.MaterialData(kind=kind, base_color=col(), metallic=pick() if rng.random() < 0.4 else 0.0, roughness=pick(),
                         ior=float(rng.choice([1.0, 1.0001, 1.33, 1.45, 1.5, 2.4, 0.8])), specular_ior_level=float(rng.choice([0.0, 0.5, 0.5, 1.0, pick()])),
                         specular_tint=col() if rng.random() < 0.3 else (1.0, 1.0, 1.0), transmission_weight=pick() if rng.random() < 0.25 else 0.0,
                         coat_weight=pick() if rng.random() < 0.25 else 0.0, coat_roughness=pick(), coat_ior=float(rng.choice([1.0, 1.5, 1.8])),
                         coat_tint=col() if rng.random() < 0.3 else (1.0, 1.0, 1.0),
                         normal=tuple(float(x) for x in rng.normal(size=3)) if rng.random() < 0.15 else (0.0, 0.0, 0.0),
                         base_alpha=float(rng.choice([1.0, 1.0, 1.0, 0.5, 0.0])))
    if emissive:
        m.emission_color, m.emission_strength = col(float(rng.choice([1.0, 5.0, 40.0]))), float(rng.choice([1.0, 0.5, 3.0]))
        if kind != abi.MAT_PRINCIPLED and rng.random() < 0.5:
            m.kind = abi.MAT_EMISSION
    if rng.random() < 0.2:
        m.colorspaces = int(rng.integers(0, 16)) << 8
    return m


def rand_images(rng):
    out = []
    for _ in range(int(rng.integers(1, 4))):
        h, w = int(rng.integers(1, 10)), int(rng.integers(1, 10))
        if rng.random() < 0.5:
            t = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
            t[:, :, 3] = np.where(rng.random((h, w)) < (0.3 if rng.random() < 0.4 else 0.0), 0, 255)
        else:
            t = rng.random((h, w, 4)).astype(np.float32)
            t[:, :, 3] = np.where(rng.random((h, w)) < (0.3 if rng.random() < 0.3 else 0.0), 0.25, 1.0)
        out.append(abi.ImageData(t, int(rng.integers(0, 2)), int(rng.integers(0, 4))))
    return out


def rand_graph(rng, n_images):
    """An arbitrary DAG of the ten node kinds (values are float4 whatever the node: every wiring is legal) feeding a random
    subset of the surface node's inputs."""
    N = abi.NodeData
    nodes = []
    k3 = lambda hi=1.5: tuple(float(x) for x in rng.uniform(0.0, hi, size=3))  # noqa: E731
    for i in range(int(rng.integers(1, 11))):
        prev = lambda none_ok=False: (abi.NODE_NONE if (none_ok and (i == 0 or rng.random() < 0.5)) else int(rng.integers(0, i)))  # noqa: E731
        ops = [abi.NODE_CONST, abi.NODE_RGB, abi.NODE_TEXCOORDS, abi.NODE_IMAGE, abi.NODE_IMAGE]
        if i > 0:
            ops += [abi.NODE_MAPPING, abi.NODE_CHECKERBOARD, abi.NODE_SPECTRAL_UPLIFT, abi.NODE_SEPARATE_COLOR, abi.NODE_EXTRACT, abi.NODE_NORMAL_MAP, abi.NODE_IMAGE]
        op = int(rng.choice(ops))
        if op == abi.NODE_CONST:
            nodes.append(N(op, (), k3(3.0)))
        elif op == abi.NODE_RGB:
            nodes.append(N(op, (int(rng.integers(0, 2)),), k3(1.0)))
        elif op == abi.NODE_TEXCOORDS:
            nodes.append(N(op))
        elif op == abi.NODE_IMAGE:
            nodes.append(N(op, (int(rng.integers(0, n_images)), prev(True), int(rng.integers(0, 2)))))
        elif op == abi.NODE_MAPPING:
            nodes.append(N(op, (prev(), prev(), prev(), int(rng.integers(0, 2)))))
        elif op == abi.NODE_CHECKERBOARD:
            nodes.append(N(op, (prev(True), prev(), prev(), prev())))
        elif op in (abi.NODE_SPECTRAL_UPLIFT, abi.NODE_SEPARATE_COLOR):
            nodes.append(N(op, (prev(),)))
        elif op == abi.NODE_EXTRACT:
            nodes.append(N(op, (prev(), int(rng.integers(0, 4)))))
        else:
            nodes.append(N(op, (prev(), prev())))
    names = [n for n in abi.INPUT_NAMES if rng.random() < 0.25] or ["base_color"]
    return abi.GraphData(nodes, {n: int(rng.integers(0, len(nodes))) for n in names})


def rand_scene(seed, textures=None, big=False, inst=False):
    """big: meshes of up to a few thousand triangles (deep BVHs), larger frames, longer paths"""
    rng = np.random.default_rng(seed)
    textured = (rng.random() < 0.35) if textures is None else textures
    images = rand_images(rng) if textured else []
    w, h = (int(rng.integers(40, 90)), int(rng.integers(40, 90))) if big else (int(rng.integers(8, 40)), int(rng.integers(8, 40)))
    n_mats = int(rng.integers(1, 6))
    mats = [rand_material(rng) for _ in range(n_mats)] + [rand_material(rng, True) for _ in range(int(rng.integers(0, 3)))]
    meshes, insts = [], []
    eye = np.eye(4, dtype=np.float32)
    n_meshes = int(rng.integers(1, 5))
    many = rng.random() < 0.35  # enough triangles for the BVH path
    for mi in range(n_meshes):
        nt = int(rng.integers(1, 60 if many else 12)) * (int(rng.integers(1, 80)) if big else 1)
        if rng.random() < 0.5:  # soup of random triangles in [-1, 1]^3, some degenerate / tiny / huge
            c = rng.uniform(-1, 1, size=(nt, 1, 3))
            e = rng.normal(size=(nt, 3, 3)) * rng.choice([1e-3, 0.1, 0.5, 2.0], size=(nt, 1, 1))
            tri = (c + e).astype(np.float32)
            if rng.random() < 0.3:
                tri[0, 2] = tri[0, 1]  # a degenerate triangle
            verts = tri.reshape(-1, 3)
            idx = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
        else:  # a grid patch with shared vertices (coplanar neighbours: the plane-sharing rule)
            n = max(1, int(np.sqrt(nt / 2)))
            xs = np.linspace(-1, 1, n + 1, dtype=np.float32)
            y = np.float32(rng.uniform(-1, 1))
            verts = np.array([[xs[i], y, xs[j]] for j in range(n + 1) for i in range(n + 1)], dtype=np.float32)
            idx = []
            for j in range(n):
                for i in range(n):
                    a, b, c2, d = j * (n + 1) + i, j * (n + 1) + i + 1, (j + 1) * (n + 1) + i + 1, (j + 1) * (n + 1) + i
                    idx += [[a, c2, b], [a, d, c2]] if rng.random() < 0.5 else [[a, b, c2], [a, c2, d]]
            idx = np.array(idx, dtype=np.uint32)
        nt = idx.shape[0]
        n_slots = int(rng.integers(1, 4))
        slots = rng.integers(0, n_slots, size=nt).astype(np.uint32) if n_slots > 1 else None
        normals = None
        if rng.random() < 0.3:
            nn = rng.normal(size=(nt, 3, 3)).astype(np.float32)
            normals = (nn / np.linalg.norm(nn, axis=2, keepdims=True)).astype(np.float32)
        uvs = (rng.random((nt, 3, 2)) * rng.choice([1.0, 3.0, -2.0])).astype(np.float32) if rng.random() < (0.8 if textured else 0.3) else None
        meshes.append(abi.MeshData(vertices=np.ascontiguousarray(verts), indices=idx, material_slots=slots, normals=normals, uvs=uvs))
        for _ in range(int(rng.integers(1, 7 if inst else 3))):
            t = eye.copy()
            if inst and rng.random() < 0.7:  # "inst": any non-singular affine map -- rotation about a random axis, non-uniform scale, mirror, shear
                ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
                a = rng.uniform(0, 6.28)
                K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
                S = np.diag(rng.choice([0.05, 0.3, 1.0, 1.0, 2.5, -1.0, -0.4], size=3).astype(np.float64))
                M = R @ S
                if rng.random() < 0.3:
                    M[0] += rng.uniform(-1, 1) * M[1]
                t[:3, :3] = M.astype(np.float32)
                t[:3, 3] = (rng.uniform(-0.8, 0.8, size=3) * rng.choice([1.0, 1.0, 30.0])).astype(np.float32)
            elif rng.random() < 0.6:
                a = np.float32(rng.uniform(0, 6.28))
                t[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float32) * np.float32(rng.choice([0.5, 1.0, 1.7, -1.0]))
                t[:3, 3] = rng.uniform(-0.5, 0.5, size=3).astype(np.float32)
            insts.append(abi.InstanceData(mi, [int(rng.integers(0, len(mats))) for _ in range(n_slots)], t.T.reshape(16).copy()))
            if rng.random() < 0.15:  # an exact duplicate with other materials: every hit is a tie in t, the lower triangle id must win
                insts.append(abi.InstanceData(mi, [int(rng.integers(0, len(mats))) for _ in range(n_slots)], t.T.reshape(16).copy()))
    c2w = eye.copy()
    c2w[:3, 3] = rng.uniform(-0.5, 0.5, size=3).astype(np.float32) + np.array([0, 0, 2.5], dtype=np.float32) * np.float32(rng.random() < 0.7)
    cam = abi.CameraData(c2w=c2w.T.reshape(16).copy(), fov=float(rng.uniform(0.3, 2.2)), width=w, height=h)
    if textured:
        for m in mats:
            if rng.random() < 0.6:
                m.graph = rand_graph(rng, len(images))
    sd = abi.SceneData(meshes, insts, mats, cam, images=images)
    cfg = abi.PtConfig.default()
    cfg.spp = int(rng.integers(1, 7)); cfg.spp_per_pass = int(rng.integers(1, cfg.spp + 1))
    cfg.max_depth = int(rng.integers(1, 16 if big else 9)); cfg.rr_depth = int(rng.integers(0, 6))
    cfg.use_nee = int(rng.random() < 0.85); cfg.indirect_only = int(rng.random() < 0.1); cfg.force_diffuse = int(rng.random() < 0.25)
    cfg.filter_type = int(rng.choice([abi.FILTER_BOX, abi.FILTER_GAUSSIAN])); cfg.filter_radius = float(rng.choice([0.5, 1.0, 1.5]))
    cfg.sampler_type = int(rng.choice([abi.SAMPLER_INDEPENDENT] * 3 + [abi.SAMPLER_SOBOL, abi.SAMPLER_PMJ02BN]))
    cfg.sampler_seed = int(rng.integers(0, 1 << 40))
    if cfg.sampler_type != abi.SAMPLER_INDEPENDENT and cfg.spp > 1 and rng.random() < 0.4:  # a sample range of the render (index-based samplers)
        cfg.sample_begin = int(rng.integers(0, cfg.spp))
        cfg.sample_count = int(rng.integers(1, cfg.spp - cfg.sample_begin + 1))
    cfg.color = int(rng.choice([0, 0, 0, 1, 2, 3]))
    if rng.random() < 0.15:
        cfg.debug_depth = int(rng.integers(0, 4))
    return sd, cfg


def run_pt(ctx, scene, sd, cfg, rng):
    w, h = sd.camera.width, sd.camera.height
    film = capi.Film(ctx, w, h)
    st = capi.pt_render(ctx, scene, cfg, film)
    g = film.read()
    o, ost = pyoracle.OracleScene(sd).render(cfg)
    nd = int(np.count_nonzero(g.view(np.uint32) != o.view(np.uint32)))
    counts = {k: (int(st[k]), int(ost[k])) for k in ("n_samples", "n_closest", "n_shadow", "n_shaded")}
    return nd, all(a == b for a, b in counts.values()), counts


def run_aov(ctx, scene, sd, cfg, rng):
    c = abi.AovConfig.default()
    c.spp, c.aov, c.remap = int(rng.integers(1, 5)), int(rng.integers(0, 6)), int(rng.integers(0, 2))
    c.filter_type, c.filter_radius, c.sampler_type, c.sampler_seed, c.color = cfg.filter_type, cfg.filter_radius, cfg.sampler_type, cfg.sampler_seed, cfg.color
    film = capi.Film(ctx, sd.camera.width, sd.camera.height)
    st = capi.aov_render(ctx, scene, c, film)
    o, n_rays = pyoracle.OracleScene(sd).aov_render(c)
    g = film.read()
    return int(np.count_nonzero(g.view(np.uint32) != o.view(np.uint32))), int(st["n_samples"]) == int(n_rays), {"aov": c.aov}


def run_gpt(ctx, scene, sd, cfg, rng):
    c = abi.GptConfig.default()
    c.spp, c.max_depth, c.rr_depth, c.spp_per_pass = int(rng.integers(1, 4)), min(cfg.max_depth, 5), cfg.rr_depth, 2
    c.use_nee, c.indirect_only = cfg.use_nee, cfg.indirect_only
    c.reconstruction, c.reconstruction_iter, c.separate_weights = int(rng.integers(0, 3)), int(rng.integers(1, 5)), int(rng.integers(0, 2))
    c.stride = int(rng.integers(1, 3))
    c.filter_type, c.filter_radius, c.sampler_seed, c.color = cfg.filter_type, cfg.filter_radius, cfg.sampler_seed, cfg.color
    w, h = sd.camera.width, sd.camera.height
    if c.stride >= min(w, h):
        c.stride = 1
    film = capi.Film(ctx, w, h)
    capi.gpt_render(ctx, scene, c, film, want_aux=False)
    o, _ = pyoracle.OracleScene(sd).gpt_render(c)
    g = film.read()
    return int(np.count_nonzero(g.view(np.uint32) != o.view(np.uint32))), True, {"recon": c.reconstruction}


def run_shard(ctx, scene, sd, cfg, rng):
    """every rank of a random world renders its tiles; the films summed = the oracle's full frame"""
    w, h = sd.camera.width, sd.camera.height
    world = int(rng.integers(2, 5))
    tw, th = int(rng.choice([8, 16, 32, 64])), int(rng.choice([8, 16, 32]))  # multiples of 8 (akr_pt_config)
    total = np.zeros(7 * w * h, dtype=np.float32)
    counts = {k: 0 for k in ("n_samples", "n_closest", "n_shadow", "n_shaded")}
    for rank in range(world):
        c = abi.PtConfig.from_buffer_copy(bytes(cfg))
        c.shard_rank, c.shard_count, c.tile_w, c.tile_h = rank, world, tw, th
        film = capi.Film(ctx, w, h)
        st = capi.pt_render(ctx, scene, c, film)
        g = film.read()
        assert not np.any((total != 0) & (g != 0))  # disjoint pixels: the sum below is exact
        total += g
        for k in counts:
            counts[k] += int(st[k])
    o, ost = pyoracle.OracleScene(sd).render(cfg)
    nd = int(np.count_nonzero(total.view(np.uint32) != o.view(np.uint32)))
    # (-0.0 + 0.0 = +0.0: compare values where both are zero)
    if nd:
        nd = int(np.count_nonzero((total != o) & ~(np.isnan(total) & np.isnan(o))))
    return nd, all(counts[k] == int(ost[k]) for k in counts), {"world": world, "tile": (tw, th)}


def run_mcmc(ctx, scene, sd, cfg, rng):
    c = abi.McmcConfig.default()
    c.n_chains, c.n_bootstrap, c.spp, c.spp_per_pass, c.direct_spp = int(rng.integers(8, 64)), int(rng.integers(200, 800)), int(rng.integers(2, 8)), 4, int(rng.choice([-1, 0, 2]))
    c.max_depth, c.rr_depth, c.use_nee = min(cfg.max_depth, 5), cfg.rr_depth, cfg.use_nee
    c.filter_type, c.filter_radius, c.color, c.seed = cfg.filter_type, cfg.filter_radius, cfg.color, int(rng.integers(0, 1 << 30))
    c.sampler_type = abi.SAMPLER_INDEPENDENT
    film = capi.Film(ctx, sd.camera.width, sd.camera.height)
    try:
        o_film, o_res, o_chains = pyoracle.OracleScene(sd).mcmc_render(c)
    except AssertionError:  # "Bootstrap failed": no path of the bootstrap carries light; the library must refuse as well
        try:
            capi.mcmc_render(ctx, scene, c, film)
        except capi.AkariError:
            return 0, True, {"both refused": True}
        return 1, False, {"oracle refused, library rendered": True}
    st, res, chains = capi.mcmc_render(ctx, scene, c, film)
    nd = sum(int(np.count_nonzero(chains[name].view(np.uint32) != o_chains[name].view(np.uint32))) for name in chains.dtype.names)
    same = res["normalization"] == o_res["normalization"] and res["acceptance_rate"] == o_res["acceptance_rate"]
    return nd, bool(same), {"chains": c.n_chains}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    opts = sys.argv[3:]
    textures = True if "tex" in opts else None  # "tex": every scene with images and shader graphs
    runner = run_aov if "aov" in opts else run_gpt if "gpt" in opts else run_mcmc if "mcmc" in opts else run_shard if "shard" in opts else run_pt
    if "inst" in opts:  # scenes with a shared mesh are kept as meshes + instances (two-level traversal) instead of flattened
        capi.set_option("instancing", 1)
    if "wavefront" in opts:  # the path tracer's wavefront schedule instead of the megakernel
        capi.set_option("wavefront", 1)
        capi.set_option("force_bvh", 1)
        if "carry" in opts:  # rays carried from one trace launch into the next on these small frames too (option wf_carry: the launch size it starts at)
            capi.set_option("wf_carry", 2)
    table = np.fromfile(os.path.join(ROOT, "tests/golden/ggx_dielectric_s.f32"), dtype=np.float32)
    ctx = capi.Context(0)
    pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())
    bad, refused, t0 = [], 0, time.time()
    kinds = {"exhaustive": 0, "bvh": 0, "kept": 0, "textured": 0}
    for seed in range(first, first + n):
        sd, cfg = rand_scene(seed, textures, "big" in opts, "inst" in opts)
        if runner in (run_gpt, run_mcmc):
            cfg.sampler_type = abi.SAMPLER_INDEPENDENT  # gpt: independent sampler only
        sd.ggx_table = table
        try:
            scene = capi.Scene(ctx, sd)
        except capi.AkariError as e:
            refused += 1
            if refused <= 3:
                print("refused seed", seed, str(e)[:120], flush=True)
            continue
        kinds[("exhaustive", "bvh", "kept")[scene.info().uses_bvh]] += 1
        kinds["textured"] += int(bool(sd.images))
        try:
            nd, same_counts, info = runner(ctx, scene, sd, cfg, np.random.default_rng(seed + 7))
        except capi.AkariError as e:
            refused += 1
            if refused <= 3:
                print("render refused seed", seed, str(e)[:120], flush=True)
            continue
        if nd or not same_counts:
            bad.append((seed, nd, info))
            print("MISMATCH seed", seed, "floats", nd, info, flush=True)
    print(f"{runner.__name__} {opts}: {n} cases from seed {first}: {len(bad)} mismatches, {refused} refused, {kinds}, {time.time() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
