"""A numpy model of what the acceleration structure DECIDES (csrc/device/disect.h trav_step, dinst_trav.h trav_step_inst): the slab
test of a ray against the quantised child boxes of the 64-byte nodes host/bvh.cpp writes, in the device's f32 operations, and the
triangle test it has to be conservative for (disect.h tri_test on the f64-derived Woop rows). No GPU: the trees come from
`capi.Scene(None, ..)`. What it checks is the one property the box levels must have -- a triangle the exhaustive loop would accept
for a ray is never culled: every box on the way from the root to its leaf passes the slab test with the accepted t as the limit
(for the closest hit that is the tightest limit traversal ever applies to it; for a shadow ray the limit is tmax >= t).

fma(a, b, c) is modelled as f32(f64(a) * f64(b) + f64(c)): the product of two f32 is exact in f64, the sum is rounded twice
(to 53 bits, then to 24) -- different from the device's single rounding in about one case in 2^29, irrelevant for a test of
conservativeness. v_rcp_f32 (1 ulp) is modelled by the correctly rounded 1 / x and by its two neighbours."""
import numpy as np

F = np.float32
U = np.uint32
NODE_WORDS = 16


def fma(a, b, c):
    return (np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64) + np.asarray(c, dtype=np.float64)).astype(F)


def decode_tree(nodes: np.ndarray, node_off: int = 0, start: int = 0):
    """Per leaf triangle (index relative to the tree's triangle base): the list of (node index, entry) from the root down to the entry
    that holds it. nodes = u32 array of the whole node buffer; the tree starts at node `node_off` (child_base is relative to it).
    start: the node of the tree to start at instead of its root (a top-level leaf record of a re-braided scene names one): the
    triangles below that node, and their paths from it."""
    nodes = nodes.reshape(-1, NODE_WORDS)
    paths = {}
    stack = [(start, [])]
    while stack:
        idx, path = stack.pop()
        n = nodes[node_off + idx]
        child_base = int(n[3] >> 24) | (int(n[4] & 0xffff) << 8)
        tri_base = int(n[6])
        for e in range(6):
            meta = int((n[5] >> (8 * e)) & 0xff) if e < 4 else int((n[4] >> (16 + 8 * (e - 4))) & 0xff)
            if meta == 0:
                continue
            here = path + [(idx, e)]
            if (meta >> 5) == 1 and (meta & 0x1f) >= 24:  # inner child at octant position (meta & 0x1f) - 24
                stack.append((child_base + (meta & 0x1f) - 24, here))
            else:
                count = {1: 1, 3: 2, 7: 3}[meta >> 5]
                for k in range(count):
                    paths[tri_base + (meta & 0x1f) + k] = here
    return paths


def entry_box_params(nodes: np.ndarray, node_off: int, path):
    """For the (node, entry) pairs of a path: node origin (f32[D,3]), per-axis scale 2^e (f32[D,3]), quantised lo / hi (f32[D,3] each)."""
    nodes = nodes.reshape(-1, NODE_WORDS)
    D = len(path)
    origin = np.zeros((D, 3), F)
    scale = np.zeros((D, 3), F)
    qlo = np.zeros((D, 3), F)
    qhi = np.zeros((D, 3), F)
    for k, (idx, e) in enumerate(path):
        n = nodes[node_off + idx]
        origin[k] = n[0:3].view(F)
        for a in range(3):
            scale[k, a] = np.array([(int(n[3]) >> (8 * a) & 0xff) << 23], dtype=U).view(F)[0]
            if e < 4:
                qlo[k, a] = (int(n[7 + a]) >> (8 * e)) & 0xff
                qhi[k, a] = (int(n[10 + a]) >> (8 * e)) & 0xff
            else:
                qlo[k, a] = (int(n[13 + a]) >> (8 * (e - 4))) & 0xff
                qhi[k, a] = (int(n[13 + a]) >> (16 + 8 * (e - 4))) & 0xff
    return origin, scale, qlo, qhi


def _nudge(x, k):
    """x moved by k ulp (k in -1, 0, 1) away from / towards zero (f32, finite, non-zero)."""
    if k == 0:
        return x
    return (x.view(U).astype(np.int64) + k).astype(U).view(F)


def slab_pass(o, d, tmin, limit, params, rcp_ulp=0):
    """The device's slab test of rays (o, d: f32[R,3]; tmin, limit: f32[R]) against D boxes: bool[R,D].
    rcp_ulp: the reciprocal moved by that many ulp (v_rcp_f32 is accurate to 1 ulp)."""
    origin, scale, qlo, qhi = params
    with np.errstate(all="ignore"):
        a = np.where(np.abs(d) < F(1e-20), np.copysign(F(1e-20), d), d).astype(F)
        inv = _nudge((F(1.0) / a).astype(F), rcp_ulp)
        noi = (-o * inv).astype(F)
        b = (scale[None, :, :] * inv[:, None, :]).astype(F)                       # bx = 2^e * inv
        ax = fma(origin[None, :, :], inv[:, None, :], noi[:, None, :])          # fma(origin, inv, noi)
        tlo = fma(qlo[None, :, :], b, ax)
        thi = fma(qhi[None, :, :], b, ax)
        neg = (inv < 0)[:, None, :]
        tn = np.where(neg, thi, tlo)
        tf = np.where(neg, tlo, thi)
        tn = np.maximum(np.maximum(tn[..., 0], tn[..., 1]), np.maximum(tn[..., 2], tmin[:, None]))
        tf = np.minimum(np.minimum(tf[..., 0], tf[..., 1]), np.minimum(tf[..., 2], limit[:, None]))
        return tn <= tf


def tri_test(o, d, rows, tmin, tmax):
    """disect.h tri_test for rays (f32[R,3]) against ONE record (rows: f32[12]): accept bool[R], t f32[R]."""
    r0, r1, r2 = rows[0:4], rows[4:8], rows[8:12]
    with np.errstate(all="ignore"):
        dz = fma(r2[0], d[:, 0], fma(r2[1], d[:, 1], (r2[2] * d[:, 2]).astype(F)))
        oz = fma(r2[0], o[:, 0], fma(r2[1], o[:, 1], fma(r2[2], o[:, 2], r2[3])))
        t = (-oz / dz).astype(F)
        px, py, pz = fma(t, d[:, 0], o[:, 0]), fma(t, d[:, 1], o[:, 1]), fma(t, d[:, 2], o[:, 2])
        u = fma(r0[0], px, fma(r0[1], py, fma(r0[2], pz, r0[3])))
        v = fma(r1[0], px, fma(r1[1], py, fma(r1[2], pz, r1[3])))
        ok = (t >= tmin) & (t <= tmax) & (u >= 0) & (v >= 0) & ((u + v).astype(F) <= F(1.0))
    return ok, t


def adversarial_rays(A, B, C, n, rng, reach, extent):
    """n rays (f32 o, d) aimed at the rim of triangle ABC (f64 vertices): target points within a few round-off widths of an edge or a
    corner, inside and outside, at angles down to grazing, from origins up to `extent` away. reach = magnitude of the coordinates."""
    e1, e2 = B - A, C - A
    nrm = np.cross(e1, e2)
    nl = np.linalg.norm(nrm)
    if not nl > 0:
        return None, None
    nrm = nrm / nl
    # width of the rim in barycentric units: the round-off of u = r0 . p + c0 is ~ 2^-24 * |r0| * magnitude, |r0| = |e2| / |n|
    rim = 2.0 ** -24 * reach * max(np.linalg.norm(e1), np.linalg.norm(e2)) / nl
    w = 10.0 ** rng.uniform(-1.5, 2.0, size=n) * rim * rng.choice([-1.0, 1.0], size=n)
    s = rng.random(n)
    which = rng.integers(0, 3, size=n)
    u = np.where(which == 0, w, np.where(which == 1, s, s))
    v = np.where(which == 0, s, np.where(which == 1, w, 1.0 - s + w))
    corner = rng.random(n) < 0.25
    u = np.where(corner, rng.choice([0.0, 1.0], size=n) + w, u)
    v = np.where(corner, np.where(u > 0.5, 0.0, rng.choice([0.0, 1.0], size=n)) + w * rng.choice([-1.0, 1.0], size=n), v)
    q = A[None, :] + u[:, None] * e1[None, :] + v[:, None] * e2[None, :]
    # direction: in-plane unit vector tilted out of the plane by asin(sin_t), sin_t log-uniform down to grazing
    ang = rng.uniform(0, 2 * np.pi, size=n)
    t1 = e1 / np.linalg.norm(e1)
    t2 = np.cross(nrm, t1)
    inplane = np.cos(ang)[:, None] * t1[None, :] + np.sin(ang)[:, None] * t2[None, :]
    sin_t = 10.0 ** rng.uniform(-6, 0, size=n) * rng.choice([-1.0, 1.0], size=n)
    dirs = inplane * np.sqrt(np.maximum(0.0, 1.0 - sin_t ** 2))[:, None] + nrm[None, :] * sin_t[:, None]
    dist = extent * 10.0 ** rng.uniform(-3, 0, size=n)
    o = q - dirs * dist[:, None]
    scale_d = np.where(rng.random(n) < 0.5, 1.0, dist)  # unit directions (bounce rays) and unnormalised ones (shadow rays: d = target - origin)
    return o.astype(F), (dirs * scale_d[:, None]).astype(F)


def check_flattened(scene, n_rays_per_tri, rng, max_tris=None):
    """Flattened scene with a BVH: (pairs accepted by the triangle test, pairs among them a box on the way culls)."""
    from akari_render_amd import capi
    nodes = scene.array(capi.ARRAY_BVH_NODES, U)
    woop = scene.array(capi.ARRAY_WOOP, F).reshape(-1, 16)[: scene.info().n_triangles]
    shade = scene.array(capi.ARRAY_SHADE, F).reshape(-1, 32)
    gid = woop[:, 12].view(U)
    paths = decode_tree(nodes)
    verts = shade[:, :12].reshape(-1, 3, 4)[:, :, :3].astype(np.float64)  # object-space; the world triangle is rebuilt below
    inst = scene.array(capi.ARRAY_INSTANCES, F).reshape(-1, 32)
    c2w = scene.array(capi.ARRAY_C2W, F)
    world = _world_vertices(shade, inst)
    lo, hi = world.reshape(-1, 3).min(0), world.reshape(-1, 3).max(0)
    reach = float(np.maximum(np.maximum(np.abs(lo), np.abs(hi)), np.abs(c2w[12:15])).sum())
    extent = float(np.linalg.norm(hi - lo))
    order = np.arange(len(gid))
    if max_tris is not None and len(order) > max_tris:
        order = rng.choice(order, size=max_tris, replace=False)
    accepted = culled = 0
    worst = []
    for k in order:
        A, B, C = world[gid[k]]
        o, d = adversarial_rays(A, B, C, n_rays_per_tri, rng, reach, extent)
        if o is None:
            continue
        tmin = np.zeros(len(o), F)
        ok, t = tri_test(o, d, woop[k, :12], tmin, np.full(len(o), 1e20, F))
        if not ok.any():
            continue
        o, d, t = o[ok], d[ok], t[ok]
        params = entry_box_params(nodes, 0, paths[k])
        bad = np.zeros(len(o), bool)
        for ulp in (-1, 0, 1):
            bad |= ~slab_pass(o, d, np.zeros(len(o), F), t, params, ulp).all(axis=1)
        accepted += len(o)
        culled += int(bad.sum())
        if bad.any():
            worst.append((int(gid[k]), int(bad.sum())))
    return accepted, culled, worst


def _world_vertices(shade, inst):
    """f32 world vertices of every flattened triangle = xf_point(instance, object-space vertex) in the compiler's operation order."""
    v = shade[:, :12].reshape(-1, 3, 4)[:, :, :3]
    ii = shade[:, 26].view(U)
    m = inst[ii]
    c0, c1, c2, t = m[:, 0:3], m[:, 4:7], m[:, 8:11], m[:, 12:15]
    out = np.zeros((len(shade), 3, 3), F)
    for k in range(3):
        p = v[:, k, :]
        out[:, k, :] = ((((c0 * p[:, 0:1]).astype(F) + (c1 * p[:, 1:2]).astype(F)).astype(F) + (c2 * p[:, 2:3]).astype(F)).astype(F) + t).astype(F)
    return out.astype(np.float64)


def check_kept(kept, flat, n_rays_per_tri, rng, max_tris=None):
    """Scene kept as meshes + instances (`kept`; `flat` = the same scene flattened, for the exact records): the top-level boxes on the
    world ray, then the mesh's boxes on the ray taken through the instance's f32 inverse rows (dinst_trav.h trav_into_instance)."""
    from akari_render_amd import capi
    nodes = kept.array(capi.ARRAY_BVH_NODES, U)
    leaves = kept.array(capi.ARRAY_INST_LEAVES, F).reshape(-1, 16)
    mesh_tris = kept.array(capi.ARRAY_MESH_TRIS, F).reshape(-1, 16)
    tlas_paths = decode_tree(nodes, 0)
    fw = flat.array(capi.ARRAY_WOOP, F)
    ntri = flat.info().n_triangles
    if flat.info().uses_bvh:
        fw = fw.reshape(-1, 16)[:ntri]
        rows_of = {int(g): fw[k, :12] for k, g in enumerate(fw[:, 12].view(U))}
    else:
        fw = fw.reshape(-1, 12)[:ntri]
        rows_of = {k: fw[k] for k in range(ntri)}
    shade = flat.array(capi.ARRAY_SHADE, F).reshape(-1, 32)
    inst = flat.array(capi.ARRAY_INSTANCES, F).reshape(-1, 32)
    world = _world_vertices(shade, inst)
    c2w = kept.array(capi.ARRAY_C2W, F)
    lo, hi = world.reshape(-1, 3).min(0), world.reshape(-1, 3).max(0)
    reach = float(np.maximum(np.maximum(np.abs(lo), np.abs(hi)), np.abs(c2w[12:15])).sum())
    extent = float(np.linalg.norm(hi - lo))
    blas_paths = {}
    accepted = culled = 0
    worst = []
    todo = []
    kinst = kept.array(capi.ARRAY_INSTANCES, F).reshape(-1, 32)
    seen = set()
    for leaf in range(len(leaves)):
        lf = leaves[leaf]
        node_off, tri_off, ii, start = (int(lf[12 + c:13 + c].view(U)[0]) for c in range(4))
        gid_base = int(kinst[ii, 23:24].view(U)[0])
        if (node_off, start) not in blas_paths:
            blas_paths[(node_off, start)] = decode_tree(nodes, node_off, start)
        for k in blas_paths[(node_off, start)]:
            assert (ii, k) not in seen, "a triangle of an instance is reachable from two top-level leaf records"
            seen.add((ii, k))
            todo.append((leaf, (node_off, start), tri_off, gid_base, k))
    assert len(seen) == flat.info().n_triangles, "the top-level leaf records do not cover every instance-triangle"
    if max_tris is not None and len(todo) > max_tris:
        todo = [todo[i] for i in rng.choice(len(todo), size=max_tris, replace=False)]
    for leaf, node_off, tri_off, gid_base, k in todo:
        lf = leaves[leaf]
        prim = int(mesh_tris[tri_off + k, 15:16].view(U)[0])
        g = gid_base + prim
        A, B, C = world[g]
        o, d = adversarial_rays(A, B, C, n_rays_per_tri, rng, reach, extent)
        if o is None:
            continue
        ok, t = tri_test(o, d, rows_of[g], np.zeros(len(o), F), np.full(len(o), 1e20, F))
        if not ok.any():
            continue
        o, d, t = o[ok], d[ok], t[ok]
        z = np.zeros(len(o), F)
        # the ray in object space: dot(r, wo) + c with dot = (x x + y y) + z z, no contraction
        def dot3(r, p):
            return (((r[0] * p[:, 0]).astype(F) + (r[1] * p[:, 1]).astype(F)).astype(F) + (r[2] * p[:, 2]).astype(F)).astype(F)
        with np.errstate(all="ignore"):
            oo = np.stack([(dot3(lf[4 * r:4 * r + 3], o) + lf[4 * r + 3]).astype(F) for r in range(3)], axis=1)
            od = np.stack([dot3(lf[4 * r:4 * r + 3], d) for r in range(3)], axis=1)
        pt = entry_box_params(nodes, 0, tlas_paths[leaf])
        pb = entry_box_params(nodes, node_off[0], blas_paths[node_off][k])
        bad = np.zeros(len(o), bool)
        for ulp in (-1, 0, 1):
            bad |= ~slab_pass(o, d, z, t, pt, ulp).all(axis=1)
            bad |= ~slab_pass(oo, od, z, t, pb, ulp).all(axis=1)
        accepted += len(o)
        culled += int(bad.sum())
        if bad.any():
            worst.append((g, int(bad.sum())))
    return accepted, culled, worst
