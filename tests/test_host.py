"""Host-side logic of libakari_hip.so (no GPU): seed stream, PCG closed form, alias tables, scene compiler,
BVH builder -- compared with the oracle's independent implementations."""
import ctypes as C

import numpy as np
import pytest

from akari_render_amd import abi, capi
from oracle import pyoracle, scene_json
from tests.helpers import cbox_variant, grid_scene


def test_seed_stream_matches_oracle(hip_lib, oracle_lib):
    for seed in (0, 7, 2**63 + 5):
        a = capi.host_stdrng_u64(seed, 300)
        b = np.zeros(300, dtype=np.uint64)
        oracle_lib.or_kat_stdrng_u64(seed, 300, b.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert np.array_equal(a, b)
        assert np.array_equal(capi.host_pcg32_states(seed, 257), pyoracle.init_pcg32_states(257, seed))


def test_chacha_core(hip_lib):
    key = np.arange(32, dtype=np.uint8).view("<u4").copy()
    out = capi.host_chacha_block(key, 1 | (0x09000000 << 32), 0x4A000000, 20)
    assert out[0] == 0xE4E7F110 and out[15] == 0x4E3C50A2
    assert capi.host_chacha_block(np.zeros(8, np.uint32), 0, 0, 12).astype("<u4").tobytes()[:8].hex() == "9bf49a6a0755f953"


def test_pcg_start_closed_form(hip_lib, oracle_lib):
    """The kernels replace sampler.start() = advance(16384) by state' = A state + (A + C inc)."""
    rng = np.random.default_rng(1)
    for _ in range(200):
        s0, inc = int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2)), int(rng.integers(0, 2**62)) * 2 + 1
        st = C.c_uint64(s0)
        oracle_lib.or_kat_pcg32_advance(C.byref(st), inc, 16384)
        assert capi.host_pcg_start(s0, inc) == st.value


def test_alias_table_matches_oracle(hip_lib, oracle_lib):
    rng = np.random.default_rng(4)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    for n in (1, 2, 5, 100, 1000):
        w = rng.random(n).astype(np.float32)
        if n > 3:
            w[1] = 0.0
        j, t, pdf = capi.host_alias_table(w)
        jo = np.zeros(n, np.uint32); to = np.zeros(n, np.float32); po = np.zeros(n, np.float32)
        oracle_lib.or_kat_alias_build(w.ctypes.data_as(fp), n, jo.ctypes.data_as(up), to.ctypes.data_as(fp), po.ctypes.data_as(fp))
        assert np.array_equal(j, jo) and np.array_equal(t, to) and np.array_equal(pdf, po)
        mass = np.zeros(n)
        for i in range(n):
            mass[i] += t[i] / n; mass[j[i]] += (1 - t[i]) / n
        assert np.max(np.abs(mass - w / w.sum())) < 1e-3  # util/distribution.rs:125-146


def _check_shade_records(sd, sc):
    osc = pyoracle.OracleScene(sd)
    shade = sc.array(capi.ARRAY_SHADE, np.float32).reshape(-1, 32)
    off = sc.array(capi.ARRAY_INST_TRI_OFFSET, np.uint32)
    assert off[-1] == sd.n_triangles() == sc.info().n_triangles
    rng = np.random.default_rng(0)
    for inst in range(len(sd.instances)):
        mesh = sd.meshes[sd.instances[inst].mesh]
        nt = mesh.indices.shape[0]
        for prim in (range(nt) if nt <= 64 else rng.integers(0, nt, 64)):
            prim = int(prim)
            o = osc.surface_interaction(inst, prim, 0.3, 0.2)
            r = shade[off[inst] + prim]
            assert np.array_equal(o[3:6], r[12:15])       # ng
            assert o[17] == r[24]                          # prim_area
            assert int(o[18]) == int(r[25:26].view(np.uint32)[0])  # material
            if mesh.normals is None:
                assert np.array_equal(o[9:12], r[16:19]) and np.array_equal(o[12:15], r[20:23])  # frame t, s
    # lights
    assert sc.info().n_lights == osc.num_lights()
    for i in range(osc.num_lights()):
        assert sc.light(i) == osc.light_info(i)
    return osc


def test_scene_compiler_matches_oracle_cbox(hip_lib, cbox_path):
    sd = scene_json.load_scene(cbox_path, 64, 64)
    sc = capi.Scene(None, sd)
    assert sc.info().uses_bvh == 0 and sc.info().n_triangles == 36 and sc.info().n_lights == 1
    _check_shade_records(sd, sc)
    inst, power, pdf = sc.light(0)
    assert inst == 0 and abs(power - 2 * 17 * 0.0893) < 0.01 and pdf == 1.0


@pytest.mark.parametrize("normals", [False, True])
def test_scene_compiler_matches_oracle_grid(hip_lib, normals):
    sd = grid_scene(n=12, with_normals=normals)
    sc = capi.Scene(None, sd)
    assert sc.info().uses_bvh == 1
    _check_shade_records(sd, sc)


def test_material_flags(hip_lib, cbox_path):
    sd = cbox_variant(scene_json.load_scene(cbox_path, 32, 32), "glass_coat")
    sc = capi.Scene(None, sd)
    mats = sc.array(capi.ARRAY_MATERIALS, np.uint32).reshape(-1, 64)
    names = sd.material_names
    SPEC, COAT, BASE, METAL, DIFF, DIEL = 1, 2, 4, 8, 16, 32
    def fl(n):
        return int(mats[names.index(n)][1])
    assert fl("leftWall_001") & (SPEC | COAT | METAL | DIEL) == 0 and fl("leftWall_001") & BASE and fl("leftWall_001") & DIFF
    assert fl("shortBox_001") & DIEL and not fl("shortBox_001") & DIFF
    assert fl("floor_001") & COAT and fl("backWall_001") & SPEC
    assert fl("tallBox_001") & METAL and fl("tallBox_001") & BASE


def _decode_bvh8(sc):
    """The scene's compressed wide BVH (csrc/host/bvh.cpp: 64-byte nodes, six entries in eight octant positions) as python
    objects: per node origin, scale, child_base, tri_base, meta[8], q[6][8] (entries 6, 7 are always empty)."""
    info = sc.info()
    stride = info.node_stride_bytes // 4
    raw = sc.array(capi.ARRAY_BVH_NODES, np.uint32).reshape(-1, stride)
    assert raw.shape[0] == info.n_bvh_nodes and info.node_bytes == 64 and stride == 16
    return raw, info


def _node(raw, ni):
    """Node words: origin xyz | exponents + child_base[7:0] | child_base[23:8] + meta[4..5] | meta[0..3] | tri_base | six words = the
    planes lo.x lo.y lo.z hi.x hi.y hi.z of entries 0..3 | three words = x, y, z of entries 4, 5 as (lo4, lo5, hi4, hi5)."""
    nd = raw[ni]
    origin = nd[0:3].view(np.float32).astype(np.float32)
    e = [(int(nd[3]) >> (8 * a)) & 0xFF for a in range(3)]
    scale = np.array([np.float32(2.0) ** np.float32(x - 127) for x in e], dtype=np.float32)
    child_base = (int(nd[3]) >> 24) | ((int(nd[4]) & 0xFFFF) << 8)
    meta = np.zeros(8, dtype=np.uint8)
    meta[0:4] = nd[5:6].view(np.uint8)
    meta[4:6] = nd[4:5].view(np.uint8)[2:4]
    q = np.zeros((6, 8), dtype=np.uint8)
    q[0:3, 6:8] = 255  # the two entries that do not exist: inverted boxes, like every empty entry
    q[:, 0:4] = nd[7:13].view(np.uint8).reshape(6, 4)
    b = nd[13:16].view(np.uint8).reshape(3, 4)  # per axis: lo4 lo5 hi4 hi5
    q[0:3, 4:6] = b[:, 0:2]
    q[3:6, 4:6] = b[:, 2:4]
    return origin, scale, child_base, int(nd[6]), meta, q


def test_bvh_structure(hip_lib):
    """64-byte 6-wide compressed nodes (csrc/host/bvh.cpp): every triangle under exactly one leaf, decoded child boxes
    contain their triangles and nest inside the parent's decoded box, inner children live at child_base + octant position, leaf
    triangles of a node are contiguous from tri_base, depth = what akr_scene_info reports and fits the traversal stack."""
    sd = grid_scene(n=20)
    sc = capi.Scene(None, sd)
    raw, info = _decode_bvh8(sc)
    gid = sc.array(capi.ARRAY_TRI_GID, np.uint32)
    rec = sc.array(capi.ARRAY_WOOP, np.float32).reshape(-1, 16)
    n_tris = info.n_triangles
    assert info.tri_bytes == 64 and rec.shape[0] >= n_tris
    assert sorted(gid.tolist()) == list(range(n_tris))
    assert np.array_equal(rec[:n_tris, 12].view(np.uint32), gid)  # the record carries its global id
    shade = sc.array(capi.ARRAY_SHADE, np.float32).reshape(-1, 32)
    inst = sc.array(capi.ARRAY_INSTANCES, np.float32).reshape(-1, 32)

    def world(g):
        r = shade[g]; m = inst[int(r[26:27].view(np.uint32)[0])]
        M = np.stack([m[0:3], m[4:7], m[8:11]], axis=1); t = m[12:15]
        return np.stack([M @ r[0:3] + t, M @ r[4:7] + t, M @ r[8:11] + t])

    seen = np.zeros(n_tris, dtype=int)
    visited = set()
    stack = [(0, np.full(3, -np.inf), np.full(3, np.inf), 1)]
    depth = 0
    while stack:
        ni, plo, phi, d = stack.pop()
        assert ni not in visited
        visited.add(ni)
        depth = max(depth, d)
        origin, scale, child_base, tri_base, meta, q = _node(raw, ni)
        assert child_base < (1 << 24) and not meta[6] and not meta[7]
        next_offset, last_pos = 0, -1
        for s in range(8):
            m = int(meta[s])
            lo = origin + q[0:3, s].astype(np.float32) * scale
            hi = origin + q[3:6, s].astype(np.float32) * scale
            if m == 0:
                assert np.all(q[0:3, s] == 255) and np.all(q[3:6, s] == 0)  # empty slot: inverted box
                continue
            if np.all(np.isfinite(plo)):  # children are quantised in their own node's frame: nested up to one step
                tol = (phi - plo) / 100.0 + 1e-3
                assert np.all(lo >= plo - tol) and np.all(hi <= phi + tol)
            if (m & 0x18) == 0x18:  # inner: 0x20 | (24 + octant position); entries are stored in ascending position
                pos = (m & 31) - 24
                assert (m >> 5) == 1 and 0 <= pos < 8 and pos > last_pos
                last_pos = pos
                stack.append((child_base + pos, lo, hi, d + 1))
            else:
                unary, offset = m >> 5, m & 31
                count = {1: 1, 3: 2, 7: 3}[unary]
                assert offset == next_offset and offset + count <= 18   # leaves of a node are packed in entry order
                next_offset += count
                for k in range(tri_base + offset, tri_base + offset + count):
                    seen[k] += 1
                    w = world(int(gid[k]))
                    assert np.all(w >= lo - 1e-6) and np.all(w <= hi + 1e-6)  # conservative: decoded box contains the triangle
    assert np.all(seen == 1)
    assert depth == info.bvh_depth and depth <= 24
    # slots the tree does not use are holes (all zero)
    holes = [i for i in range(raw.shape[0]) if i not in visited]
    assert all(not raw[i].any() for i in holes)
    # the 48-byte records map each triangle to the unit triangle
    for k in (0, 17, n_tris - 1):
        w = world(int(gid[k])).astype(np.float64)
        R = rec[k, :12].reshape(3, 4).astype(np.float64)
        loc = (R[:, :3] @ w.T + R[:, 3:4]).T
        assert np.allclose(loc, [[0, 0, 0], [1, 0, 0], [0, 1, 0]], atol=2e-4)


def test_bvh_slot_order_is_front_to_back(hip_lib):
    """Children sit in the slot whose bits say on which side of the node centre they lie, so visiting slots in the order
    slot ^ octant is (approximately) near-to-far: for axis-aligned rays the first visited of two children that are separated
    along that axis is the nearer one in at least 90 % of the node pairs."""
    sd = grid_scene(n=20)
    sc = capi.Scene(None, sd)
    raw, info = _decode_bvh8(sc)
    good = total = 0
    for ni in range(raw.shape[0]):
        if not raw[ni].any():
            continue
        origin, scale, _, _, meta, q = _node(raw, ni)
        slots = [s for s in range(8) if (int(meta[s]) & 0x18) == 0x18]  # inner children: their octant position is in the meta byte
        pos = {s: (int(meta[s]) & 31) - 24 for s in slots}
        for a in range(3):
            for s1 in slots:
                for s2 in slots:
                    if s1 >= s2:
                        continue
                    c1 = int(q[a, s1]) + int(q[3 + a, s1])
                    c2 = int(q[a, s2]) + int(q[3 + a, s2])
                    if q[3 + a, s1] <= q[a, s2] or q[3 + a, s2] <= q[a, s1]:  # separated along axis a
                        # a ray towards +a visits the position whose bit a is 0 first
                        first = s1 if ((pos[s1] >> a) & 1) < ((pos[s2] >> a) & 1) else (s2 if ((pos[s2] >> a) & 1) < ((pos[s1] >> a) & 1) else None)
                        if first is None:
                            continue
                        total += 1
                        nearer = s1 if c1 < c2 else s2
                        good += first == nearer
    print(f"slot order agrees with the distance order for {good} of {total} separated child pairs")
    assert total > 50 and good > 0.9 * total


def test_coplanar_neighbours_share_their_plane_row(hip_lib, cbox_path):
    """Triangles 2j, 2j+1 of an instance whose vertices are coplanar to 1e-6 of their size carry the same third row (the
    exhaustive intersector solves the plane once per such pair); the oracle applies the same rule independently."""
    from tests.helpers import box_scene

    sd = scene_json.load_scene(cbox_path, 32, 32)
    sc = capi.Scene(None, sd)
    w = sc.array(capi.ARRAY_WOOP, np.float32)[: 12 * 36].reshape(36, 12)
    same = [k for k in range(1, 36, 2) if np.array_equal(w[k, 8:].view(np.uint32), w[k - 1, 8:].view(np.uint32))]
    osc = pyoracle.OracleScene(sd)
    assert len(same) == osc.shared_plane_rows() == 17  # every quad of the Cornell box but one whose halves are 3e-6 apart
    # the first two rows stay the triangle's own
    assert all(not np.array_equal(w[k, :8], w[k - 1, :8]) for k in same)
    # a cube of exact rectangles: all six faces; a grid of displaced vertices: (almost) none
    assert pyoracle.OracleScene(box_scene()).shared_plane_rows() == 6
    assert pyoracle.OracleScene(grid_scene(n=12)).shared_plane_rows() <= 2


def _traverse_bvh8(raw, stride_unused, o, d, tmin, tmax):
    """A scalar python mirror of device/disect.h trav_step (node part, no culling by hits): the set of traversal-order
    triangle indices whose leaf box the ray enters, and the number of nodes visited."""
    inv = np.where(np.abs(d) < 1e-20, np.copysign(1e-20, d), d)
    inv = (1.0 / inv.astype(np.float64))
    oi = int(inv[0] >= 0) | (int(inv[1] >= 0) << 1) | (int(inv[2] >= 0) << 2)
    octinv4 = oi * 0x01010101
    G = 1 << (24 + oi)
    stack, tris, n_nodes = [], [], 0
    while True:
        if (G >> 24) == 0:
            if not stack:
                break
            G = stack.pop()
        j = G.bit_length() - 1
        G &= ~(1 << j)
        if (G >> 24) != 0:
            stack.append(G)
        slot = (j - 24) ^ (octinv4 & 7)
        origin, scale, child_base, tri_base, meta, q = _node(raw, (G & 0xFFFFFF) + slot)
        n_nodes += 1
        hitmask = 0
        for h in range(2):
            meta4 = int.from_bytes(bytes(meta[4 * h:4 * h + 4]), "little")
            is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010
            inner_mask4 = (is_inner4 >> 4) * 0xFF
            bit_index4 = (meta4 ^ (octinv4 & inner_mask4)) & 0x1F1F1F1F
            child_bits4 = (meta4 >> 5) & 0x07070707
            for i in range(4):
                s = 4 * h + i
                lo = origin.astype(np.float64) + q[0:3, s].astype(np.float64) * scale
                hi = origin.astype(np.float64) + q[3:6, s].astype(np.float64) * scale
                t0, t1 = (lo - o) * inv, (hi - o) * inv
                tn = max(np.minimum(t0, t1).max(), tmin)
                tf = min(np.maximum(t0, t1).min(), tmax)
                if tn <= tf:
                    hitmask |= ((child_bits4 >> (8 * i)) & 0xFF) << ((bit_index4 >> (8 * i)) & 0xFF)
        G = (child_base & 0xFFFFFF) | (hitmask & 0xFF000000)
        T = hitmask & 0xFFFFFF
        while T:
            b = (T & -T).bit_length() - 1
            T &= T - 1
            tris.append(tri_base + b)
    return tris, n_nodes


def test_bvh_traversal_reaches_every_hit(hip_lib):
    """The traversal algorithm of device/disect.h, mirrored in python on the host-built tree: for 1500 rays (random and aimed
    at edges) every triangle the oracle's exhaustive loop reports is among the triangles the traversal would test, and the
    node-group bookkeeping (octant bit order, SWAR meta decode, one stack entry per level) never revisits a node."""
    from tests.helpers import probe_rays

    sd = grid_scene(n=20)
    sc = capi.Scene(None, sd)
    raw, info = _decode_bvh8(sc)
    gid = sc.array(capi.ARRAY_TRI_GID, np.uint32)
    osc = pyoracle.OracleScene(sd)
    rays = probe_rays(osc.world_vertices(), 750, 750, seed=9)
    out, _ = osc.intersect_many(rays)
    off = osc.tri_offsets()
    n_hit = 0
    max_nodes = 0
    for r, h in zip(rays, out):
        tris, n_nodes = _traverse_bvh8(raw, None, r[0:3].astype(np.float64), r[3:6].astype(np.float32), 0.0, 1e20)
        assert len(tris) == len(set(tris))          # no triangle (hence no node) reached twice
        max_nodes = max(max_nodes, n_nodes)
        if h[0]:
            n_hit += 1
            g = int(off[h[1]] + h[2])
            assert g in set(int(gid[k]) for k in tris)
    assert n_hit > 300 and max_nodes < info.n_bvh_nodes


def test_non_finite_vertices_are_refused(hip_lib):
    """Round-2 advisor finding: one +-inf / NaN coordinate made the 8-wide builder's slot costs NaN, its greedy assignment write
    out of bounds and the tree drop geometry. compile_scene refuses such scenes, whatever their size."""
    from tests.helpers import grid_scene
    for n, bad in ((2, np.inf), (12, -np.inf), (12, np.nan)):
        sd = grid_scene(n=n, width=16, height=16)
        v = sd.meshes[0].vertices.copy()
        v[v.shape[0] // 2, 1] = bad
        sd.meshes[0].vertices = v
        with pytest.raises(capi.AkariError) as ei:
            capi.Scene(None, sd)
        assert ei.value.code == -1 and "non-finite" in str(ei.value)  # AKR_ERR_INVALID_ARGUMENT


def test_options_are_a_table_not_the_environment(hip_lib, cbox_path, monkeypatch):
    """akr_option_set / akr_option_get: the test hooks are read from the environment once; later changes of the environment
    are not seen, akr_option_set is."""
    assert capi.get_option("force_bvh") in (0, 1)
    monkeypatch.setenv("AKR_FORCE_BVH", "1")  # after the first look: ignored
    before = capi.get_option("force_bvh")
    assert capi.Scene(None, cbox_path).info().uses_bvh == before
    with capi.options(force_bvh=1):
        assert capi.get_option("force_bvh") == 1
        assert capi.Scene(None, cbox_path).info().uses_bvh == 1
    assert capi.get_option("force_bvh") == before
    assert capi.get_option("defer_metal") == -1 and capi.get_option("wavefront") == -1
    with pytest.raises(capi.AkariError):
        capi.set_option("no_such_option", 1)
    # the schedule options of round 6's last session: defaults, and values out of range are refused
    assert capi.get_option("wf_carry") == 1 and capi.get_option("sched_trial") == -1 and capi.get_option("wf_groups") == 0
    for name, bad in (("wf_carry", -1), ("sched_trial", 2), ("sched_trial", -2), ("wf_groups", 33), ("wavefront", 2)):
        with pytest.raises(capi.AkariError):
            capi.set_option(name, bad)
    with capi.options(wf_carry=4096, sched_trial=1):  # (wf_carry above 1: the launch size from which rays are carried -- the tests' hook)
        assert capi.get_option("wf_carry") == 4096 and capi.get_option("sched_trial") == 1
    assert capi.get_option("wf_carry") == 1 and capi.get_option("sched_trial") == -1


def test_scene_compile_does_not_depend_on_the_thread_count(hip_lib, monkeypatch):
    """host/bvh.cpp and the per-triangle records run on the host's threads above 2^17 triangles: the tree, the triangle order and
    the records are the same bytes for 1, 3 and 8 threads (every quantity a split depends on is a min, a max or an integer count;
    the partition is the sequential one)."""
    import hashlib

    from akari_render_amd import procedural

    sd = procedural.sponza_like(300_000, seed=7, width=32, height=32)

    def digest(threads):
        monkeypatch.setenv("AKR_HOST_THREADS", str(threads))
        sc = capi.Scene(None, sd)
        h = hashlib.sha256()
        for arr in (capi.ARRAY_BVH_NODES, capi.ARRAY_TRI_GID, capi.ARRAY_WOOP, capi.ARRAY_SHADE):
            h.update(sc.array(arr, np.uint32).tobytes())
        return h.hexdigest(), sc.info().n_bvh_nodes, sc.info().bvh_depth

    one = digest(1)
    assert digest(3) == one and digest(8) == one
