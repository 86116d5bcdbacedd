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


def test_bvh_structure(hip_lib):
    """64-byte quantised BVH4 nodes (csrc/host/bvh.cpp): every triangle in exactly one leaf, decoded child boxes
    contain their triangles and nest inside the parent's decoded box."""
    sd = grid_scene(n=20)
    sc = capi.Scene(None, sd)
    nodes = sc.array(capi.ARRAY_BVH_NODES, np.uint32).reshape(-1, 16)
    assert nodes.shape[0] == sc.info().n_bvh_nodes
    gid = sc.array(capi.ARRAY_TRI_GID, np.uint32)
    woop = sc.array(capi.ARRAY_WOOP, np.float32).reshape(-1, 12)
    n_tris = sc.info().n_triangles
    assert sorted(gid.tolist()) == list(range(n_tris))
    shade = sc.array(capi.ARRAY_SHADE, np.float32).reshape(-1, 32)
    inst = sc.array(capi.ARRAY_INSTANCES, np.float32).reshape(-1, 32)
    def world(g):
        r = shade[g]; m = inst[int(r[26:27].view(np.uint32)[0])]
        M = np.stack([m[0:3], m[4:7], m[8:11]], axis=1); t = m[12:15]
        return np.stack([M @ r[0:3] + t, M @ r[4:7] + t, M @ r[8:11] + t])
    def decode(nd):
        origin = nd[0:3].view(np.float32).astype(np.float32)
        e = np.array([(int(nd[3]) >> (8 * a)) & 0xFF for a in range(3)])
        scale = np.array([np.float32(2.0) ** np.float32(int(x) - 127) for x in e], dtype=np.float32)
        q = [[(int(nd[4 + k]) >> (8 * i)) & 0xFF for i in range(4)] for k in range(6)]  # lo.x lo.y lo.z hi.x hi.y hi.z
        refs = [int(nd[10]), int(nd[11]), int(nd[12]), int(nd[13])]
        boxes = []
        for i in range(4):
            lo = np.array([origin[a] + np.float32(q[a][i]) * scale[a] for a in range(3)], dtype=np.float32)
            hi = np.array([origin[a] + np.float32(q[3 + a][i]) * scale[a] for a in range(3)], dtype=np.float32)
            boxes.append((lo, hi))
        return boxes, refs
    seen = np.zeros(n_tris, dtype=int)
    stack = [(0, np.full(3, -np.inf), np.full(3, np.inf))]
    n_visited = 0
    while stack:
        ni, plo, phi = stack.pop()
        n_visited += 1
        boxes, refs = decode(nodes[ni])
        for (lo, hi), ref in zip(boxes, refs):
            if ref == 0xFFFFFFFF:
                assert np.all(lo > hi)  # empty slot: inverted box
                continue
            # children are quantised in their own node's frame: they nest in the parent's decoded box up to one step
            if np.all(np.isfinite(plo)):
                tol = (phi - plo) / 100.0 + 1e-3
                assert np.all(lo >= plo - tol) and np.all(hi <= phi + tol)
            if ref & 0x80000000:
                first, count = ref & 0x0FFFFFFF, (ref >> 28) & 7
                assert 1 <= count <= 4
                for k in range(first, first + count):
                    seen[k] += 1
                    w = world(int(gid[k]))
                    assert np.all(w >= lo - 1e-6) and np.all(w <= hi + 1e-6)  # conservative: decoded box contains the triangle
            else:
                stack.append((ref, lo, hi))
    assert np.all(seen == 1) and n_visited == nodes.shape[0]
    # the 48-byte records map each triangle to the unit triangle
    for k in (0, 17, n_tris - 1):
        w = world(int(gid[k])).astype(np.float64)
        R = woop[k].reshape(3, 4).astype(np.float64)
        loc = (R[:, :3] @ w.T + R[:, 3:4]).T
        assert np.allclose(loc, [[0, 0, 0], [1, 0, 0], [0, 1, 0]], atol=2e-4)


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
