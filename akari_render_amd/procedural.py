"""Procedural "Sponza-like" hall for BASELINE.json configs[3] (10 M triangles, BVH resident in HBM).

SURVEY.md 8(d) C4: a 30 x 12 x 15 m hall -- displaced floor / ceiling / wall grids, two rows of 12 fluted columns,
arches between them, hanging curtains -- one mesh, identity transform, no uvs/normals, six diffuse Principled materials
chosen by a hash of the triangle index, one 2 x 2 m ceiling area light with emission (17, 12, 4). The generator is a
pure function of (target triangle count, seed); the reference has no such scene (it ships only scenes/cbox), so this is
synthetic input of the kind the north star asks for, not a port of anything.
"""
from __future__ import annotations

import numpy as np

from . import abi

HALL = (30.0, 12.0, 15.0)  # x (length), y (height), z (width)


def _grid(nu: int, nv: int, fn, flip=False):
    u, v = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="xy")
    p = fn(u.ravel(), v.ravel()).astype(np.float32)
    i = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).ravel().astype(np.uint32)
    a, b, c, d = i, i + 1, i + nu + 2, i + nu + 1
    if flip:
        tri = np.stack([np.stack([a, c, b], 1), np.stack([a, d, c], 1)], 1).reshape(-1, 3)
    else:
        tri = np.stack([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 1).reshape(-1, 3)
    return p, tri.astype(np.uint32)


def _hash_u32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    x = (x ^ (x >> 16)) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> 15)) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    return (x ^ (x >> 16)).astype(np.uint32)


def sponza_like(target_tris: int = 10_000_000, seed: int = 1234, width: int = 1920, height: int = 1080) -> abi.SceneData:
    rng = np.random.default_rng(seed)
    L, H, Wd = HALL
    # the tessellations below add up to 11.62 M triangles at scale 1: normalise so that `target_tris` is met
    scale = np.sqrt(target_tris / 11_616_000.0)
    parts = []
    ph = rng.random(16) * 2 * np.pi  # phases of the displacement waves

    def wave(a, b, amp, k):
        return amp * (np.sin(2 * np.pi * 7 * a + ph[k]) * np.cos(2 * np.pi * 5 * b + ph[k + 1]) + 0.5 * np.sin(2 * np.pi * 23 * (a + b) + ph[k + 2]))

    def n(x):
        return max(2, int(round(x * scale)))

    # floor / ceiling: 2 x ~2.1 M triangles
    nf = (n(1450), n(725))
    parts.append(_grid(*nf, lambda u, v: np.stack([L * (u - 0.5), wave(u, v, 0.03, 0), Wd * (v - 0.5)], 1), flip=True))
    parts.append(_grid(*nf, lambda u, v: np.stack([L * (u - 0.5), H + wave(u, v, 0.05, 3), Wd * (v - 0.5)], 1)))
    # long walls (z = +-Wd/2) and end walls (x = +-L/2): ~1.9 M
    nw = (n(1100), n(440))
    parts.append(_grid(*nw, lambda u, v: np.stack([L * (u - 0.5), H * v, -Wd / 2 + wave(u, v, 0.04, 6)], 1)))
    parts.append(_grid(*nw, lambda u, v: np.stack([L * (u - 0.5), H * v, Wd / 2 + wave(u, v, 0.04, 7)], 1), flip=True))
    ne = (n(350), n(280))
    parts.append(_grid(*ne, lambda u, v: np.stack([-L / 2 + wave(u, v, 0.04, 8), H * v, Wd * (u - 0.5)], 1), flip=True))
    parts.append(_grid(*ne, lambda u, v: np.stack([L / 2 + wave(u, v, 0.04, 9), H * v, Wd * (u - 0.5)], 1)))
    # 2 x 12 fluted columns: ~2.4 M
    nc = (n(360), n(140))
    col_x = np.linspace(-L / 2 + 1.5, L / 2 - 1.5, 12)
    for zc in (-Wd / 4, Wd / 4):
        for xc in col_x:
            def col(u, v, xc=xc, zc=zc):
                ang = 2 * np.pi * u
                r = 0.45 * (1.0 + 0.06 * np.cos(20 * ang)) * (1.0 + 0.25 * np.exp(-((v - 0.03) / 0.03) ** 2) + 0.25 * np.exp(-((v - 0.97) / 0.03) ** 2))
                return np.stack([xc + r * np.cos(ang), 0.75 * H * v, zc + r * np.sin(ang)], 1)
            parts.append(_grid(*nc, col))
    # arches between consecutive columns (half tori): ~1.1 M
    na = (n(160), n(160))
    for zc in (-Wd / 4, Wd / 4):
        for x0, x1 in zip(col_x[:-1], col_x[1:]):
            def arch(u, v, x0=x0, x1=x1, zc=zc):
                R, r = 0.5 * (x1 - x0), 0.22
                a, b = np.pi * u, 2 * np.pi * v
                return np.stack([0.5 * (x0 + x1) - (R + r * np.cos(b)) * np.cos(a), 0.75 * H + (R + r * np.cos(b)) * np.sin(a) * 0.8,
                                 zc + r * np.sin(b)], 1)
            parts.append(_grid(*na, arch))
    # curtains: 8 displaced sheets hanging between columns: ~1.5 M
    ncu = (n(310), n(310))
    for k in range(8):
        xk = -L / 2 + 3.0 + k * (L - 6.0) / 7.0
        def curtain(u, v, xk=xk, k=k):
            fold = 0.12 * np.sin(2 * np.pi * 9 * u + ph[10 + k % 5]) * (0.3 + 0.7 * (1 - v))
            return np.stack([xk + fold, 2.0 + 0.55 * H * v, (Wd / 4 - 0.6) * (2 * u - 1)], 1)
        parts.append(_grid(*ncu, curtain))

    verts, tris, off = [], [], 0
    for p, t in parts:
        verts.append(p)
        tris.append(t + np.uint32(off))
        off += p.shape[0]
    verts = np.concatenate(verts).astype(np.float32)
    tris = np.concatenate(tris).astype(np.uint32)
    slots = (_hash_u32(np.arange(tris.shape[0], dtype=np.uint64) + np.uint64(seed)) % np.uint32(6)).astype(np.uint32)
    hall = abi.MeshData(vertices=verts, indices=tris, material_slots=slots)
    lv = np.array([[-1, H - 0.35, -1], [1, H - 0.35, -1], [1, H - 0.35, 1], [-1, H - 0.35, 1]], dtype=np.float32)
    light = abi.MeshData(vertices=lv, indices=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32))
    palette = [(0.75, 0.72, 0.68), (0.62, 0.45, 0.33), (0.4, 0.48, 0.55), (0.7, 0.62, 0.4), (0.52, 0.6, 0.45), (0.8, 0.78, 0.75)]
    mats = [abi.MaterialData(base_color=c, roughness=0.9, ior=1.0, specular_ior_level=0.0) for c in palette]
    mats.append(abi.MaterialData(base_color=(0.8, 0.8, 0.8), ior=1.0, specular_ior_level=0.0, emission_color=(17.0, 12.0, 4.0), emission_strength=1.0))
    eye = np.eye(4, dtype=np.float32).T.reshape(16).copy()
    # camera near one end, 1.7 m above the floor, looking down the hall (-x -> +x), slightly upwards
    fwd = np.array([1.0, 0.08, 0.05]); fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -fwd, [-L / 2 + 1.0, 1.7, 0.3]
    cam = abi.CameraData(c2w=c2w.T.reshape(16).astype(np.float32).copy(), fov=np.deg2rad(70.0), width=width, height=height)
    insts = [abi.InstanceData(0, list(range(6)), eye), abi.InstanceData(1, [6], eye)]
    return abi.SceneData([hall, light], insts, mats, cam)


def instanced_forest(n_instances: int = 1000, tris_per_mesh: int = 100_000, seed: int = 77, width: int = 1920, height: int = 1080,
                     n_meshes: int = 2, n_lanterns: int = 4) -> abi.SceneData:
    """A field of `n_instances` copies of `n_meshes` "plants" (closed, lumpy, fluted blobs of ~`tris_per_mesh` triangles each, with
    smooth corner normals, uvs and two material slots) on a displaced ground, under a sky-light quad. Transforms: a rotation about
    a tilted axis, non-uniform scale, every fifth copy mirrored; the first `n_lanterns` copies carry an emissive material in slot 1.
    The kind of scene the reference's accel is built for (mesh.rs:259-348: one `push_mesh` per instance) and a flattening scene
    compiler cannot hold: 1000 x 100 k = 100 M instance-triangles. A pure function of its arguments."""
    rng = np.random.default_rng(seed)
    meshes = []
    for mi in range(n_meshes):
        segs = max(4, int(round(np.sqrt(tris_per_mesh))))
        rings = max(3, int(round(tris_per_mesh / (2.0 * segs))) + 1)
        th = np.linspace(0.0, np.pi, rings + 1)[:, None]
        ph = (np.linspace(0.0, 2.0 * np.pi, segs + 1)[:-1])[None, :]
        k = rng.random(8) * 2 * np.pi
        r = 1.0 + 0.18 * np.sin(5 * ph + k[0]) * np.sin(3 * th + k[1]) + 0.08 * np.sin(17 * ph + 9 * th + k[2]) + 0.03 * np.sin(61 * ph + k[3]) * np.sin(47 * th + k[4])
        r = r * (0.55 + 0.45 * np.sin(th) ** (0.5 + mi))  # plant 0 is round, the next ones more spindle-shaped
        x, y, z = r * np.sin(th) * np.cos(ph), (1.0 + 0.6 * mi) * np.cos(th) * np.ones_like(ph), r * np.sin(th) * np.sin(ph)
        verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
        vuv = np.stack([np.broadcast_to(ph / (2 * np.pi), x.shape), np.broadcast_to(th / np.pi, x.shape)], -1).reshape(-1, 2).astype(np.float32)
        j, i = np.meshgrid(np.arange(rings), np.arange(segs), indexing="ij")
        a, b = j * segs + i, j * segs + (i + 1) % segs
        c, d = (j + 1) * segs + (i + 1) % segs, (j + 1) * segs + i
        t1 = np.stack([a, b, c], -1)[1:].reshape(-1, 3)      # (the first ring's upper triangles are degenerate at the pole: left out)
        t2 = np.stack([a, c, d], -1)[:-1].reshape(-1, 3)     # (and the last ring's lower ones)
        idx = np.concatenate([t1, t2]).astype(np.uint32)
        idx = idx[np.argsort(_hash_u32(np.arange(idx.shape[0]) + 977 * mi), kind="stable")]  # exporter order is not spatial order
        fn = np.cross(verts[idx[:, 1]] - verts[idx[:, 0]], verts[idx[:, 2]] - verts[idx[:, 0]]).astype(np.float64)
        vn = np.zeros((verts.shape[0], 3))
        for kk in range(3):
            np.add.at(vn, idx[:, kk], fn)
        vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-30)
        slots = (_hash_u32(np.arange(idx.shape[0]) // 64 + 31 * mi) % 5 == 0).astype(np.uint32)  # patches of 64 triangles in slot 1
        meshes.append(abi.MeshData(vertices=verts, indices=idx, material_slots=slots, normals=vn[idx].astype(np.float32), uvs=vuv[idx].astype(np.float32)))
    side = int(np.ceil(np.sqrt(n_instances)))
    extent = 2.6 * side
    gp, gi = _grid(64, 64, lambda u, v: np.stack([extent * (u - 0.5) * 1.3, 0.15 * np.sin(9 * u) * np.cos(7 * v), extent * (v - 0.5) * 1.3], 1), flip=True)
    ground = abi.MeshData(vertices=gp, indices=gi)
    lq = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32) * np.float32(0.35 * extent) + np.array([0, 0.55 * extent, 0], dtype=np.float32)
    sky = abi.MeshData(vertices=lq, indices=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32))
    meshes += [ground, sky]
    mats = [
        abi.MaterialData(base_color=(0.45, 0.4, 0.3), roughness=0.9, ior=1.45, specular_ior_level=0.5),     # 0 ground
        abi.MaterialData(base_color=(0.8, 0.8, 0.8), ior=1.0, specular_ior_level=0.0, emission_color=(6.0, 6.5, 8.0), emission_strength=1.0),  # 1 sky
        abi.MaterialData(base_color=(0.15, 0.5, 0.2), roughness=0.6, ior=1.45, specular_ior_level=0.5),     # 2 leaf
        abi.MaterialData(base_color=(0.45, 0.3, 0.15), roughness=0.8, ior=1.45, specular_ior_level=0.3),    # 3 bark
        abi.MaterialData(base_color=(0.7, 0.75, 0.3), roughness=0.35, ior=1.5, specular_ior_level=0.5, coat_weight=0.5, coat_roughness=0.1, coat_ior=1.5),  # 4 waxy
        abi.MaterialData(base_color=(0.9, 0.8, 0.5), roughness=0.25, metallic=1.0, ior=1.5),                # 5 brass ornament
        abi.MaterialData(base_color=(0.5, 0.5, 0.5), ior=1.0, specular_ior_level=0.0, emission_color=(9.0, 5.0, 2.0), emission_strength=1.0),  # 6 lantern
    ]
    eye = np.eye(4, dtype=np.float32)
    insts = [abi.InstanceData(n_meshes, [0], eye.T.reshape(16).copy()), abi.InstanceData(n_meshes + 1, [1], eye.T.reshape(16).copy())]
    for kk in range(n_instances):
        gx, gz = kk % side, kk // side
        ax = np.array([0.25 * (rng.random() - 0.5), 1.0, 0.25 * (rng.random() - 0.5)])
        ax /= np.linalg.norm(ax)
        ang = rng.random() * 2 * np.pi
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        S = np.diag(0.6 + 0.7 * rng.random(3))
        if kk % 5 == 4:
            S[2, 2] = -S[2, 2]
        M = np.eye(4)
        M[:3, :3] = R @ S
        M[:3, 3] = [(gx + 0.5 + 0.6 * (rng.random() - 0.5)) * 2.6 - 0.5 * extent, 1.2 * S[1, 1] + 0.2, (gz + 0.5 + 0.6 * (rng.random() - 0.5)) * 2.6 - 0.5 * extent]
        slot1 = 6 if kk < n_lanterns else (3, 4, 5)[kk % 3]
        insts.append(abi.InstanceData(kk % n_meshes, [2, slot1], M.astype(np.float32).T.reshape(16).copy()))
    ca = -0.45
    c2w = np.array([[1, 0, 0, 0], [0, np.cos(ca), -np.sin(ca), 0.32 * extent], [0, np.sin(ca), np.cos(ca), 0.62 * extent], [0, 0, 0, 1]], dtype=np.float32)
    cam = abi.CameraData(c2w=c2w.T.reshape(16).copy(), fov=0.9, width=width, height=height)
    return abi.SceneData(meshes, insts, mats, cam)
