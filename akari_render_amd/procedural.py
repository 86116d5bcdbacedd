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
