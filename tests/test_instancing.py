"""Scenes kept as meshes + instances (csrc/host/scene_inst.cpp): host side. The reference keeps every scene that way
(mesh.rs:259-348: one `push_mesh(mesh, transform)` per instance into the accel); the flattening compiler is this build's choice for
scenes that fit, and must stay the default for them."""
import hashlib

import numpy as np
import pytest

from akari_render_amd import capi, procedural
from tests.helpers import grid_scene, instanced_scene

INST_ARRAYS = (capi.ARRAY_BVH_NODES, capi.ARRAY_INST_LEAVES, capi.ARRAY_MESH_TRIS, capi.ARRAY_MESH_POS, capi.ARRAY_MESH_META, capi.ARRAY_MESH_NORMALS,
               capi.ARRAY_INSTANCES, capi.ARRAY_AREA_ENTRIES, capi.ARRAY_AREA_PDF, capi.ARRAY_LIGHT_ENTRIES, capi.ARRAY_LIGHT_PDF)


def test_small_and_unshared_scenes_stay_flattened(hip_lib, cbox_path):
    """The automatic mode flattens whatever fits: scenes/cbox, a scene whose meshes are used once, a small scene with a shared mesh."""
    assert capi.get_option("instancing") == -1
    assert capi.Scene(None, cbox_path).info().uses_bvh == 0
    assert capi.Scene(None, grid_scene(n=12)).info().uses_bvh == 1
    assert capi.Scene(None, instanced_scene()).info().uses_bvh == 1
    with capi.options(instancing=1):
        assert capi.Scene(None, cbox_path).info().uses_bvh == 0          # no mesh is shared: nothing to keep
        assert capi.Scene(None, grid_scene(n=12)).info().uses_bvh == 1
        assert capi.Scene(None, instanced_scene()).info().uses_bvh == 2
    with pytest.raises(capi.AkariError):
        capi.set_option("instancing", 2)
    with pytest.raises(capi.AkariError):
        capi.set_option("instancing", -2)


def test_kept_scene_has_the_flattened_scene_s_lights_and_counts(hip_lib):
    sd = instanced_scene(n_inst=12, emissive_instances=2)
    flat = capi.Scene(None, sd)
    with capi.options(instancing=1):
        kept = capi.Scene(None, sd)
    a, b = flat.info(), kept.info()
    assert (a.n_triangles, a.n_instances, a.n_lights, a.n_materials) == (b.n_triangles, b.n_instances, b.n_lights, b.n_materials)
    assert b.uses_bvh == 2 and b.node_bytes == 64 and b.tri_bytes == 64
    for arr, dt in ((capi.ARRAY_LIGHT_ENTRIES, np.uint32), (capi.ARRAY_LIGHT_PDF, np.float32), (capi.ARRAY_AREA_ENTRIES, np.uint32),
                    (capi.ARRAY_AREA_PDF, np.float32), (capi.ARRAY_INST_TRI_OFFSET, np.uint32), (capi.ARRAY_MATERIALS, np.uint32)):
        assert np.array_equal(flat.array(arr, dt), kept.array(arr, dt)), arr
    for l in range(a.n_lights):
        assert flat.light(l) == kept.light(l)
    # the instance records agree in everything the flattened scene defines (the kept scene uses six spare words: dinst_trav.h)
    fi, ki = flat.array(capi.ARRAY_INSTANCES, np.uint32).reshape(-1, 32), kept.array(capi.ARRAY_INSTANCES, np.uint32).reshape(-1, 32)
    spare = [7, 11, 15, 23, 27, 28]
    keep = [k for k in range(32) if k not in spare]
    assert np.array_equal(fi[:, keep], ki[:, keep])
    # nothing per instance-triangle
    assert kept.array(capi.ARRAY_WOOP, np.uint32).size == 0 and kept.array(capi.ARRAY_SHADE, np.uint32).size == 0
    n_mesh_tris = sum(m.indices.shape[0] for m in sd.meshes)
    assert kept.array(capi.ARRAY_MESH_TRIS, np.float32).size == 16 * n_mesh_tris
    pos = kept.array(capi.ARRAY_MESH_POS, np.uint32)
    assert pos.size == n_mesh_tris
    # MESH_POS is a permutation inside each mesh, and MESH_TRIS holds the object-space vertices it points at
    tris = kept.array(capi.ARRAY_MESH_TRIS, np.float32).reshape(-1, 16)
    base = 0
    for m in sd.meshes:
        nt = m.indices.shape[0]
        p = pos[base:base + nt]
        assert np.array_equal(np.sort(p), np.arange(nt))
        rec = tris[base + p]
        v = np.asarray(m.vertices, dtype=np.float32)[np.asarray(m.indices)]
        assert np.array_equal(rec[:, 0:3], v[:, 0]) and np.array_equal(rec[:, 4:7], v[:, 1]) and np.array_equal(rec[:, 8:11], v[:, 2])
        assert np.array_equal(rec[:, 15].view(np.uint32), np.arange(nt, dtype=np.uint32))
        base += nt


def test_a_singular_transform_means_flattening(hip_lib):
    sd = instanced_scene(n_inst=4)
    t = np.asarray(sd.instances[3].transform, dtype=np.float32).reshape(4, 4).copy()
    t[1, :3] = 0.0  # second column of the matrix (stored transposed): the copy is squashed into a plane
    sd.instances[3].transform = t.reshape(16)
    with capi.options(instancing=1):
        assert capi.Scene(None, sd).info().uses_bvh == 1


def test_bad_slots_are_reported_either_way(hip_lib):
    sd = instanced_scene(n_inst=4)
    sd.instances[-1].materials = [2]  # the blob uses slots 0 and 1
    for mode in (0, 1):
        with capi.options(instancing=mode):
            with pytest.raises(capi.AkariError) as ei:
                capi.Scene(None, sd)
            assert "slot" in str(ei.value)


def test_a_hundred_million_instance_triangles_take_megabytes(hip_lib, monkeypatch):
    """1000 instances of two 100 k-triangle meshes: flattened, 21 GB of records; kept, under 1 GB -- in fact tens of MB,
    compiled in seconds. The arrays do not depend on the host's thread count."""
    sd = procedural.instanced_forest(1000, 100_000, width=64, height=36)
    assert sd.n_triangles() > 99_000_000

    def digest(threads):
        monkeypatch.setenv("AKR_HOST_THREADS", str(threads))
        sc = capi.Scene(None, sd)
        i = sc.info()
        assert i.uses_bvh == 2 and i.n_triangles == sd.n_triangles()
        h = hashlib.sha256()
        for arr in INST_ARRAYS:
            h.update(sc.array(arr, np.uint32).tobytes())
        return h.hexdigest(), i.device_bytes, i.bvh_depth, i.n_lights

    one = digest(1)
    assert one[1] < 64 << 20, one
    assert one[2] <= 40 and one[3] == 5
    assert digest(3) == one and digest(8) == one


def _pretest_cases(rng, n, offset, size, aspect, aim):
    """n (ray, triangle) pairs: triangles of edge ~`size` around `offset`, squashed by `aspect` along one edge (slivers), rays from a
    few sizes away aimed at a point of the triangle's plane spread `aim` x the triangle around its centre (aim 1: half hit, near
    misses at every edge and vertex)."""
    A = (offset + size * rng.normal(size=(n, 3))).astype(np.float32)
    e1 = size * rng.normal(size=(n, 3))
    e2 = size * rng.normal(size=(n, 3))
    e2 = e1 * rng.normal(size=(n, 1)) + aspect * e2
    B, C = (A + e1).astype(np.float32), (A + e2).astype(np.float32)
    bu, bv = aim * (rng.random(n) * 2.0 - 0.5), aim * (rng.random(n) * 2.0 - 0.5)
    k = rng.integers(0, 4, n)  # a quarter each: anywhere, on edge u = 0, on edge v = 0, on the hypotenuse
    bu = np.where(k == 1, 0.0, bu)
    bv = np.where(k == 2, 0.0, bv)
    bv = np.where(k == 3, 1.0 - bu, bv)
    target = A.astype(np.float64) + bu[:, None] * (B.astype(np.float64) - A) + bv[:, None] * (C.astype(np.float64) - A)
    dirs = rng.normal(size=(n, 3))
    nrm = np.cross(B.astype(np.float64) - A, C.astype(np.float64) - A)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    graze = rng.random(n) < 0.25  # a quarter of the rays nearly in the triangle's plane
    dirs = np.where(graze[:, None], dirs - (1.0 - 1e-3) * np.sum(dirs * nrm, axis=1, keepdims=True) * nrm, dirs)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dist = size * (0.5 + 20.0 * rng.random(n))
    o = (target - dirs * dist[:, None]).astype(np.float32)
    d = dirs.astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays = np.concatenate([o, d, np.zeros((n, 1), np.float32), np.full((n, 1), 1e20, np.float32)], axis=1).astype(np.float32)
    return rays, np.concatenate([A, B, C], axis=1)


def test_the_conservative_reject_never_drops_what_the_exact_test_accepts(hip_lib):
    """dinst.h tri_may_hit stands in front of the exact (f64 Woop rows) test of a candidate of a kept scene: it may only reject what
    that test rejects. Edges, vertices, slivers, grazing rays, triangles far from the origin, tiny and huge ones; and the t limits."""
    rng = np.random.default_rng(2024)
    total = hits = rejected = misses = 0
    for offset in (0.0, 10.0, 1e3, 1e5):
        for size in (1e-4, 1e-2, 1.0, 1e3):
            for aspect in (1.0, 1e-2, 1e-4, 1e-6):
                for aim in (1.0, 4.0):
                    rays, tris = _pretest_cases(rng, 4000, offset, size, aspect, aim)
                    may, exact, t = capi.host_tri_pretest(rays, tris)
                    assert not np.any(exact & ~may), (offset, size, aspect, aim, int(np.sum(exact & ~may)))
                    # the t limits: a limit just below / a tmin just above the exact t must not be rejected either way round
                    h = np.flatnonzero(exact)
                    if h.size:
                        r2 = rays[h].copy()
                        r2[:, 7] = t[h]                      # tlimit = t exactly: tri_test accepts (t <= tmax)
                        m2, e2, _ = capi.host_tri_pretest(r2, tris[h])
                        assert np.all(e2) and np.all(m2)
                        r2[:, 7] = 1e20
                        r2[:, 6] = t[h]                      # tmin = t exactly
                        m2, e2, _ = capi.host_tri_pretest(r2, tris[h])
                        assert np.all(e2) and np.all(m2)
                    total += rays.shape[0]
                    hits += int(exact.sum())
                    if offset <= 10.0 and aspect >= 1e-2:
                        misses += int((~exact).sum())
                        rejected += int((~exact & ~may).sum())
    assert hits > 0.05 * total
    # ... and it is worth having: on well-conditioned triangles it rejects half of these misses, three quarters of which sit exactly on
    # an edge or come from a grazing ray (in a render: 86 % of the candidates of the 10 M-triangle forest, profiles/r5_forest_*.json)
    assert rejected > 0.4 * misses, (rejected, misses)
    # a shared plane row (odd triangles): the plane may be off by plane_shift -- hits of the shifted plane must survive
    rays, tris = _pretest_cases(rng, 20000, 5.0, 0.5, 1.0, 1.0)
    shifted = tris.copy()
    n = np.cross(tris[:, 3:6] - tris[:, 0:3], tris[:, 6:9] - tris[:, 0:3])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    shift = 1e-4
    for k in range(3):
        shifted[:, 3 * k:3 * k + 3] += (shift * n).astype(np.float32)
    _, exact_shifted, _ = capi.host_tri_pretest(rays, shifted)
    may, _, _ = capi.host_tri_pretest(rays, tris, plane_shift=shift)
    assert not np.any(exact_shifted & ~may)


def test_more_instance_triangles_than_ids_are_refused(hip_lib):
    """Global triangle ids are 32 bits. A kept scene costs nothing per instance-triangle, so the count has to be checked: 43 100
    instances of a 100 k-triangle mesh (4.3 G instance-triangles) are refused, not rendered with wrapped ids."""
    sd = procedural.instanced_forest(43_100, 100_000, width=32, height=18, n_meshes=1, n_lanterns=0)
    assert sd.n_triangles() > 0xFFFFFFFF
    with pytest.raises(capi.AkariError) as ei:
        capi.Scene(None, sd)
    assert ei.value.code == capi.ERR_UNSUPPORTED and "32 bits" in str(ei.value)


def test_one_tiny_instance_does_not_inflate_the_trees_of_the_others(hip_lib):
    """a mesh's tree is padded for the worst of its instances (|M^-1| scales every term), so one copy scaled to 1e-4 among
    ordinary ones used to inflate the boxes of all of them until the tree stopped culling. The instances of a mesh are now sorted into
    padding classes and each class that occurs gets a tree of its own: the ordinary copies' tree is, byte for byte, the tree of the scene
    without the tiny copy; the tiny copy's own tree is the padded one; and no accepted (ray, triangle) pair is culled in either."""
    import copy

    from tests import bvh_model

    base = instanced_scene(n_inst=10, n=8)
    blob = next(i for i in base.instances if sum(1 for j in base.instances if j.mesh == i.mesh) > 1)
    tiny = copy.deepcopy(blob)
    m = np.array(tiny.transform, dtype=np.float32).reshape(4, 4).T.copy()  # (column-major in, row-major here)
    m[:3, :3] *= np.float32(1e-4)
    m[:3, 3] = np.float32(0.25)  # inside the scene's box: the scene's magnitude terms do not move
    tiny.transform = m.T.reshape(16).copy()
    with_tiny = copy.deepcopy(base)
    with_tiny.instances = list(base.instances) + [tiny]

    def trees(sd):
        with capi.options(instancing=1):
            sc = capi.Scene(None, sd)
        nodes = sc.array(capi.ARRAY_BVH_NODES, np.uint32).reshape(-1, 16)
        leaves = sc.array(capi.ARRAY_INST_LEAVES, np.float32).reshape(-1, 16)
        off = leaves[:, 12].view(np.uint32)
        inst = leaves[:, 14].view(np.uint32)
        return sc, nodes, {int(i): int(o) for i, o in zip(inst, off)}, sorted(set(int(o) for o in off))

    sc0, nodes0, off0, starts0 = trees(base)
    sc1, nodes1, off1, starts1 = trees(with_tiny)
    assert len(starts1) == len(starts0) + 1                      # one more per-mesh tree: the tiny copy's class
    blob_ids = [k for k, i in enumerate(base.instances) if i.mesh == blob.mesh]
    assert len({off1[k] for k in blob_ids}) == 1 and off1[len(base.instances)] not in {off1[k] for k in blob_ids}

    def tree_at(nodes, starts, o):  # the nodes of the tree that starts at node o
        nxt = [x for x in starts if x > o]
        return nodes[o:(nxt[0] if nxt else len(nodes))]

    ordinary0, ordinary1 = tree_at(nodes0, starts0, off0[blob_ids[0]]), tree_at(nodes1, starts1, off1[blob_ids[0]])
    assert ordinary0.shape == ordinary1.shape and np.array_equal(ordinary0, ordinary1)
    with capi.options(instancing=0, force_bvh=1):
        flat = capi.Scene(None, with_tiny)
    a, c, worst = bvh_model.check_kept(sc1, flat, 48, np.random.default_rng(11), max_tris=700)
    assert a > 3000 and c == 0, (a, c, worst[:5])
