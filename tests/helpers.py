"""Shared helpers for the parity tests."""
import numpy as np

from akari_render_amd import abi


def make_config(spp=16, spp_per_pass=None, max_depth=12, rr_depth=5, **kw) -> abi.PtConfig:
    c = abi.PtConfig.default()
    c.spp = spp
    c.spp_per_pass = spp_per_pass if spp_per_pass is not None else min(spp, 64)
    c.max_depth, c.rr_depth = max_depth, rr_depth
    for k, v in kw.items():
        if k == "pixel_offset":
            c.pixel_offset[0], c.pixel_offset[1] = v
        else:
            setattr(c, k, v)
    return c


def resolve_np(film: np.ndarray, w: int, h: int) -> np.ndarray:
    """Film resolve (film.rs:128-143, hdr = true) in numpy."""
    n = w * h
    rgb = film[: 3 * n].reshape(n, 3)
    wt = film[6 * n : 7 * n]
    inv = np.where(wt == 0, np.float32(1), wt).astype(np.float32)
    out = (rgb / inv[:, None]).astype(np.float32) + film[3 * n : 6 * n].reshape(n, 3) * np.float32(1.0)
    return out.reshape(h, w, 3)


def rel_rmse(img: np.ndarray, ref: np.ndarray) -> float:
    """BASELINE.md parity metric: sqrt(mean_p |rgb - ref|^2) / mean_p luminance(ref)."""
    lum = ref.astype(np.float64) @ np.array([0.2126, 0.7152, 0.0722])
    if np.mean(lum) == 0.0:  # black reference (e.g. max_depth = 0 with no emitter in view): absolute error
        return float(np.sqrt(np.mean(np.sum((img.astype(np.float64) - ref.astype(np.float64)) ** 2, axis=-1))))
    return float(np.sqrt(np.mean(np.sum((img.astype(np.float64) - ref.astype(np.float64)) ** 2, axis=-1))) / np.mean(lum))


def n_bit_diff(a: np.ndarray, b: np.ndarray) -> int:
    return int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32)))


def box_scene(albedo=0.5, emission=1.0, width=32, height=32, kind=abi.MAT_PRINCIPLED) -> abi.SceneData:
    """A closed unit cube seen from inside (white-furnace test): every wall has the same material."""
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    idx = []
    for a, b, c, d in quads:
        idx += [[a, c, b], [a, d, c]]  # wound so that the geometric normals point into the box (emitters are one-sided)
    mesh = abi.MeshData(vertices=v, indices=np.array(idx, dtype=np.uint32))
    m = abi.MaterialData(kind=kind, base_color=(albedo,) * 3, roughness=1.0, ior=1.0, specular_ior_level=0.0,
                         emission_color=(emission,) * 3, emission_strength=1.0)
    eye = np.eye(4, dtype=np.float32)
    cam = abi.CameraData(c2w=eye.T.reshape(16).copy(), fov=1.0, width=width, height=height)
    return abi.SceneData([mesh], [abi.InstanceData(0, [0], eye.T.reshape(16).copy())], [m], cam)


def grid_scene(n=24, width=64, height=64, seed=3, with_normals=False) -> abi.SceneData:
    """A bumpy n x n heightfield floor (2 n^2 triangles -> takes the BVH path) under a small area light,
    two materials chosen per triangle, one rotated + scaled instance. Exercises multi-material slots,
    a non-identity instance transform and (optionally) per-corner shading normals."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-1, 1, n + 1, dtype=np.float32)
    hgt = (0.15 * rng.random((n + 1, n + 1))).astype(np.float32)
    verts = np.array([[xs[i], hgt[j, i], xs[j]] for j in range(n + 1) for i in range(n + 1)], dtype=np.float32)
    idx = []
    for j in range(n):
        for i in range(n):
            a, b, c, d = j * (n + 1) + i, j * (n + 1) + i + 1, (j + 1) * (n + 1) + i + 1, (j + 1) * (n + 1) + i
            idx += [[a, c, b], [a, d, c]]
    idx = np.array(idx, dtype=np.uint32)
    slots = (rng.random(idx.shape[0]) < 0.3).astype(np.uint32)
    normals = None
    if with_normals:
        # smooth per-vertex normals -> per-corner array
        vn = np.zeros_like(verts)
        for t in idx:
            p0, p1, p2 = verts[t[0]], verts[t[1]], verts[t[2]]
            fn = np.cross(p1 - p0, p2 - p0)
            for k in t:
                vn[k] += fn
        vn /= np.linalg.norm(vn, axis=1, keepdims=True)
        normals = vn[idx].astype(np.float32)
    floor = abi.MeshData(vertices=verts, indices=idx, material_slots=slots, normals=normals)
    lv = np.array([[-0.3, 1.2, -0.3], [0.3, 1.2, -0.3], [0.3, 1.2, 0.3], [-0.3, 1.2, 0.3]], dtype=np.float32)
    light = abi.MeshData(vertices=lv, indices=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32))
    mats = [
        abi.MaterialData(base_color=(0.7, 0.6, 0.5), roughness=0.8, ior=1.0, specular_ior_level=0.0),
        abi.MaterialData(base_color=(0.9, 0.9, 0.9), roughness=0.25, metallic=1.0, ior=1.5),
        abi.MaterialData(base_color=(0.8, 0.8, 0.8), ior=1.0, specular_ior_level=0.0, emission_color=(12.0, 10.0, 8.0), emission_strength=1.0),
    ]
    ang = np.float32(0.3)
    rot = np.array([[np.cos(ang), 0, np.sin(ang), 0], [0, 1, 0, 0], [-np.sin(ang), 0, np.cos(ang), 0], [0, 0, 0, 1]], dtype=np.float32)
    rot[:3, :3] *= np.float32(1.25)
    eye = np.eye(4, dtype=np.float32)
    # camera at (0, 1, 3) looking at -z, slightly down
    ca = np.float32(-0.3)
    c2w = np.array([[1, 0, 0, 0], [0, np.cos(ca), -np.sin(ca), 1.0], [0, np.sin(ca), np.cos(ca), 3.0], [0, 0, 0, 1]], dtype=np.float32)
    cam = abi.CameraData(c2w=c2w.T.reshape(16).copy(), fov=0.9, width=width, height=height)
    insts = [abi.InstanceData(0, [0, 1], rot.T.reshape(16).copy()), abi.InstanceData(1, [2], eye.T.reshape(16).copy())]
    return abi.SceneData([floor, light], insts, mats, cam)


def cbox_variant(sd: abi.SceneData, which: str) -> abi.SceneData:
    """scenes/cbox with materials that exercise the Principled branches the stock scene folds away."""
    names = sd.material_names
    def mat(n):
        return sd.materials[names.index(n)]
    if which == "glass_coat":
        m = mat("shortBox_001"); m.transmission_weight, m.ior, m.roughness = 1.0, 1.45, 0.15
        m = mat("floor_001"); m.coat_weight, m.coat_roughness, m.coat_ior, m.coat_tint = 0.8, 0.1, 1.5, (0.9, 0.8, 1.0)
        m = mat("backWall_001"); m.specular_ior_level, m.ior, m.roughness = 0.5, 1.5, 0.4
        m = mat("tallBox_001"); m.metallic = 0.6
    elif which == "kinds":
        m = mat("shortBox_001"); m.kind, m.ior, m.roughness = abi.MAT_GLASS, 1.33, 0.05
        m = mat("floor_001"); m.kind = abi.MAT_DIFFUSE
        m = mat("light_001"); m.kind = abi.MAT_EMISSION
        m = mat("leftWall_001"); m.normal = (0.2, -0.1, 0.9)
    elif which == "alpha":
        m = mat("shortBox_001"); m.base_alpha = 0.5
        m = mat("tallBox_001"); m.base_alpha = 0.25
    else:
        raise ValueError(which)
    return sd
