"""Independent Python reader for the reference's scene-graph JSON (TEST INFRASTRUCTURE ONLY).

Second implementation of what the product does in C++ (akari_render_amd/csrc/host/scene_json.cpp); the two
are compared in tests/test_scene_loader.py. Follows crates/akari_scenegraph/src/scene.rs:86-117,333-340
(schema), crates/akari_render/src/load.rs:129-194 (transforms, camera) and svm/compiler.rs:116-337 /
svm/eval.rs:97-269 for constant folding of shader graphs.
"""
from __future__ import annotations

import base64
import json
import math
import os
from typing import Dict

import numpy as np

from akari_render_amd.abi import (
    MAT_DIFFUSE,
    MAT_EMISSION,
    MAT_GLASS,
    MAT_PRINCIPLED,
    CameraData,
    InstanceData,
    MaterialData,
    MeshData,
    SceneData,
)

f32 = np.float32

# color.rs: srgb <-> ACEScg with CAT (only needed for non-sRGB constants; the default pipeline is sRGB)
_ACES_TO_SRGB = None


def _axis_angle(axis, angle) -> np.ndarray:
    """glam 0.25 Mat4::from_axis_angle, f32, returns 4x4 (row, col)."""
    s, c = f32(math.sin(f32(angle))), f32(math.cos(f32(angle)))
    ax = np.asarray(axis, dtype=f32)
    axis_sin = ax * s
    axis_sq = ax * ax
    omc = f32(1.0) - c
    xyomc = ax[0] * ax[1] * omc
    xzomc = ax[0] * ax[2] * omc
    yzomc = ax[1] * ax[2] * omc
    m = np.zeros((4, 4), dtype=f32)
    m[:, 0] = [axis_sq[0] * omc + c, xyomc + axis_sin[2], xzomc - axis_sin[1], 0]
    m[:, 1] = [xyomc - axis_sin[2], axis_sq[1] * omc + c, yzomc + axis_sin[0], 0]
    m[:, 2] = [xzomc + axis_sin[1], yzomc - axis_sin[0], axis_sq[2] * omc + c, 0]
    m[:, 3] = [0, 0, 0, 1]
    return m


def _matmul(a, b):
    return (a.astype(f32) @ b.astype(f32)).astype(f32)


def load_transform(t: dict, is_camera: bool) -> np.ndarray:
    """load.rs:129-171 -> 4x4 f32 (row, col)."""
    if t["type"] == "matrix":
        # from_cols_array_2d(m).transpose(): JSON rows are matrix rows
        return np.asarray(t["data"], dtype=f32).reshape(4, 4)
    trs = t["data"]
    tr = np.asarray(trs["translation"], dtype=f32)
    r = np.asarray(trs["rotation"], dtype=f32)
    s = np.asarray(trs["scale"], dtype=f32)
    m = np.eye(4, dtype=f32)
    if not is_camera:
        m = _matmul(np.diag(np.array([s[0], s[1], s[2], 1], dtype=f32)), m)
    cs = trs["coordinate_system"]
    tm = np.eye(4, dtype=f32)
    if cs == "Akari":
        m = _matmul(_axis_angle((0, 0, 1), r[2]), m)
        m = _matmul(_axis_angle((1, 0, 0), r[0]), m)
        m = _matmul(_axis_angle((0, 1, 0), r[1]), m)
        tm[:3, 3] = tr
    elif cs == "Blender":
        if is_camera:
            m = _matmul(_axis_angle((1, 0, 0), -f32(math.pi) / f32(2.0)), m)
        m = _matmul(_axis_angle((1, 0, 0), r[0]), m)
        m = _matmul(_axis_angle((0, 0, 1), -r[1]), m)
        m = _matmul(_axis_angle((0, 1, 0), r[2]), m)
        tm[:3, 3] = [tr[0], tr[2], -tr[1]]
    else:
        raise ValueError(cs)
    return _matmul(tm, m)


def _col_major(m4: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(m4.T, dtype=f32).reshape(16)


class _Buffers:
    def __init__(self, scene: dict, base_dir: str):
        self.scene = scene
        self.base = base_dir
        self.cache: Dict[str, bytes] = {}

    def buffer(self, bid: str) -> bytes:
        if bid in self.cache:
            return self.cache[bid]
        b = self.scene["buffers"][bid]
        ty = b["type"]
        if ty == "path":
            p = b["path"]
            cand = p if os.path.isabs(p) and os.path.exists(p) else os.path.join(self.base, p)
            if not os.path.exists(cand):
                # scenes/cbox stores an absolute Windows path; fall back to the basename next to the JSON
                cand = os.path.join(self.base, p.replace("\\", "/").split("/")[-1])
            with open(cand, "rb") as f:
                data = f.read()
        elif ty == "base64":
            data = base64.b64decode(b["data"])
        elif ty == "binary":
            data = bytes(b["data"])
        else:
            raise ValueError(ty)
        self.cache[bid] = data
        return data

    def view(self, ref, dtype, cols):
        if ref is None:
            return None
        v = self.scene["buffer_views"][ref["id"]]
        data = self.buffer(v["buffer"]["id"])
        raw = data[v["offset"] : v["offset"] + v["length"]]
        return np.frombuffer(raw, dtype=dtype).reshape(-1, cols).copy()


def _fold_material(shader: dict) -> MaterialData:
    nodes = shader["nodes"]

    def const(ref):
        """Evaluate a constant node the way svm/eval.rs does; returns (values[list], alpha)."""
        n = nodes[ref["id"]]
        ty = n["type"]
        if ty == "float":
            return [n["value"]], 1.0
        if ty == "float3":
            return list(n["value"]), 1.0
        if ty == "rgb":
            if n.get("colorspace", "srgb") != "srgb":
                raise NotImplementedError("non-sRGB constant colours")
            return list(n["value"]), 1.0  # RgbTex: rgb.extend(1.0), svm/eval.rs:125-135
        if ty == "spectral_uplift":
            return const(n["rgb"])
        raise NotImplementedError(f"shader node '{ty}' (only constant inputs are supported)")

    def f(ref):  # eval_float_auto_convert
        return float(const(ref)[0][0])

    def c3(ref):
        v = const(ref)[0]
        return tuple(v[:3]) if len(v) >= 3 else (v[0], 0.0, 0.0)

    out = nodes[shader["output"]["id"]]
    assert out["type"] == "output"
    n = nodes[out["node"]["id"]]
    ty = n["type"]
    m = MaterialData()
    if ty == "principled":
        m.kind = MAT_PRINCIPLED
        m.base_color = c3(n["base_color"])
        m.base_alpha = const(n["base_color"])[1]
        m.metallic = f(n["metallic"])
        m.roughness = f(n["roughness"])
        m.ior = f(n["ior"])
        m.specular_ior_level = f(n["specular_ior_level"])
        m.specular_tint = c3(n["specular_tint"])
        m.transmission_weight = f(n["transmission_weight"])
        m.coat_weight = f(n["coat_weight"])
        m.coat_roughness = f(n["coat_roughness"])
        m.coat_ior = f(n["coat_ior"])
        m.coat_tint = c3(n["coat_tint"])
        m.emission_color = c3(n["emission_color"])
        m.emission_strength = f(n["emission_strength"])
        m.normal = c3(n["normal"])
    elif ty == "diffuse":
        m.kind = MAT_DIFFUSE
        m.base_color = c3(n["color"])
        m.base_alpha = const(n["color"])[1]
    elif ty == "glass":
        m.kind = MAT_GLASS
        m.base_color = c3(n["color"])
        m.ior = f(n["ior"])
        m.roughness = f(n["roughness"])
    elif ty == "emission":
        m.kind = MAT_EMISSION
        m.emission_color = c3(n["color"])
        m.emission_strength = f(n["strength"])
    else:
        raise NotImplementedError(f"surface shader '{ty}'")
    return m


def load_scene(path: str, width: int = 0, height: int = 0) -> SceneData:
    with open(path, "r") as fh:
        scene = json.load(fh)
    bufs = _Buffers(scene, os.path.dirname(os.path.abspath(path)))
    # Collections are BTreeMaps: lexicographic (byte-wise) key order (akari_scenegraph/src/lib.rs:71)
    geom_ids = sorted(scene["geometries"].keys())
    geom_index = {g: i for i, g in enumerate(geom_ids)}
    meshes = []
    for gid in geom_ids:
        g = scene["geometries"][gid]
        assert g["type"] == "mesh"
        uvs = bufs.view(g.get("uvs"), np.float32, 2)
        normals = bufs.view(g.get("normals"), np.float32, 3)
        tangents = bufs.view(g.get("tangents"), np.float32, 3)
        slots = bufs.view(g.get("materials"), np.uint32, 1)
        meshes.append(
            MeshData(
                vertices=bufs.view(g["vertices"], np.float32, 3),
                indices=bufs.view(g["indices"], np.uint32, 3),
                uvs=None if uvs is None else uvs.reshape(-1, 3, 2),
                normals=None if normals is None else normals.reshape(-1, 3, 3),
                # supplied tangents are bound (mesh.rs:145-165, has_tangents = args.tangents.is_some()); the
                # mikktspace ones the reference generates when they are absent never are (mesh.rs:182 vs 277-281)
                tangents=None if tangents is None else tangents.reshape(-1, 3, 3),
                # a one-entry slot buffer means "slot 0 for every triangle" (mesh.rs:139, load.rs:227-229)
                material_slots=None if (slots is None or slots.size <= 1) else slots.reshape(-1),
            )
        )
    mat_ids = sorted(scene["materials"].keys())
    mat_index = {m: i for i, m in enumerate(mat_ids)}
    materials = [_fold_material(scene["materials"][m]["shader"]) for m in mat_ids]
    inst_ids = sorted(scene["instances"].keys())
    instances = []
    for iid in inst_ids:
        inst = scene["instances"][iid]
        instances.append(
            InstanceData(
                mesh=geom_index[inst["geometry"]["id"]],
                materials=[mat_index[m["id"]] for m in inst["materials"]],
                transform=_col_major(load_transform(inst["transform"], False)),
            )
        )
    cam = scene["camera"]
    assert cam["type"] == "perspective"
    cd = cam["data"]
    c2w = load_transform(cd["transform"], True)
    fov = f32(cd["fov"]) * (f32(math.pi) / f32(180.0))  # f32::to_radians = x * (PI / 180.0f32)
    camera = CameraData(
        c2w=_col_major(c2w),
        fov=float(fov),
        width=int(width or cd["sensor_width"]),
        height=int(height or cd["sensor_height"]),
    )
    return SceneData(meshes, instances, materials, camera, None, inst_ids, mat_ids)


def load_method(path_or_text: str) -> dict:
    """RenderTask JSON (akari_integrator/src/lib.rs:93-109) -> dict with reference defaults applied."""
    text = open(path_or_text).read() if os.path.exists(path_or_text) else path_or_text
    j = json.loads(text)
    if isinstance(j, list):
        j = j[0]
    method = dict(spp=256, max_depth=7, rr_depth=5, spp_per_pass=64, use_nee=True, indirect_only=False,
                  force_diffuse=False, pixel_offset=[0, 0], debug_depth=None)
    m = j.get("method", {})
    assert m.get("type", "pt") == "pt"
    method.update({k: v for k, v in m.items() if k != "type"})
    sampler = {"type": "independent", "seed": 0}
    sampler.update(j.get("sampler", {}))
    film = {"out": "out.png", "filter": {"type": "gaussian", "radius": 1.5}}
    fj = j.get("film", {})
    film.update({k: v for k, v in fj.items()})
    return {"method": method, "sampler": sampler, "film": film}
