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


# ---------------------------------------------------------------------------------------------- textured scenes
def make_png(pixels: np.ndarray, color_type: int, depth: int = 8, filters=None, palette=None, trns: bytes = None, level: int = 6, interlace: bool = False) -> bytes:
    """Encodes a PNG from raw samples: pixels (H, W, C) of integer samples (or (H, W) palette indices / grey).
    `filters` = per-row filter types (default: cycling 0..4), encoded exactly as the PNG spec defines them.
    interlace: Adam7 (PNG specification 8.2): seven reduced images, each filtered on its own."""
    import struct
    import zlib

    px = np.asarray(pixels)
    if px.ndim == 2:
        px = px[:, :, None]
    h, w, ch = px.shape
    bits = ch * depth
    bpp = max(1, bits // 8)

    def scanlines(sub):  # filtered scanlines of one (reduced) image
        sh, sw = sub.shape[:2]
        stride = (sw * bits + 7) // 8
        rows = []
        for y in range(sh):
            if depth == 8:
                row = sub[y].astype(np.uint8).reshape(-1)
            elif depth == 16:
                row = np.stack([(sub[y] >> 8) & 255, sub[y] & 255], axis=-1).astype(np.uint8).reshape(-1)
            else:
                b = ((sub[y].reshape(-1)[:, None] >> np.arange(depth - 1, -1, -1)) & 1).astype(np.uint8).reshape(-1)
                row = np.packbits(b)
            assert row.size == stride
            rows.append(row.astype(np.int32))
        out = bytearray()
        prev = np.zeros(stride, dtype=np.int32)
        for y, row in enumerate(rows):
            ft = (filters[y] if filters is not None else y % 5)
            a = np.concatenate([np.zeros(bpp, dtype=np.int32), row[:-bpp]]) if stride > bpp else np.zeros(stride, dtype=np.int32)
            c = np.concatenate([np.zeros(bpp, dtype=np.int32), prev[:-bpp]]) if stride > bpp else np.zeros(stride, dtype=np.int32)
            if ft == 0:
                f = row
            elif ft == 1:
                f = row - a
            elif ft == 2:
                f = row - prev
            elif ft == 3:
                f = row - ((a + prev) >> 1)
            else:
                p = a + prev - c
                pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
                f = row - pred
            out.append(ft)
            out += bytes((f & 255).astype(np.uint8))
            prev = row
        return out

    if interlace:
        out = bytearray()
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = px[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                out += scanlines(sub)
    else:
        out = scanlines(px)

    def chunk(ty, body):
        return struct.pack(">I", len(body)) + ty + body + struct.pack(">I", zlib.crc32(ty + body) & 0xFFFFFFFF)

    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(np.asarray(palette, dtype=np.uint8).reshape(-1)))
    if trns is not None:
        data += chunk(b"tRNS", trns)
    comp = zlib.compress(bytes(out), level)
    half = len(comp) // 2
    data += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")
    return data


def textured_room(width=48, height=48, n_floor=1, seed=11, alpha_cutout=False, textured_light=True) -> abi.SceneData:
    """A closed room with per-corner uvs and texture-fed materials: checkerboard floor (colours + roughness from a
    separate_color channel), sRGB byte image on the back wall through a mapping node, float image with a normal map on
    a side wall, a textured emitter on the ceiling, optionally an alpha-cutout quad in front of the camera.
    n_floor > 1 tessellates the floor (n_floor^2 quads) so that the scene takes the BVH path."""
    rng = np.random.default_rng(seed)
    N = abi.NodeData
    meshes, insts, mats = [], [], []
    eye = np.eye(4, dtype=np.float32).T.reshape(16).copy()

    def quad(p0, p1, p2, p3, nu=1, nv=1, uv_scale=1.0):
        """quad p0->p1 (u) x p0->p3 (v), tessellated nu x nv; normals = (p1-p0) x (p3-p0)"""
        p0, p1, p3 = np.asarray(p0, np.float32), np.asarray(p1, np.float32), np.asarray(p3, np.float32)
        verts, idx, uvs = [], [], []
        for j in range(nv + 1):
            for i in range(nu + 1):
                verts.append(p0 + (p1 - p0) * np.float32(i / nu) + (p3 - p0) * np.float32(j / nv))
        uvf = lambda i, j: (np.float32(uv_scale * i / nu), np.float32(uv_scale * j / nv))  # noqa: E731
        for j in range(nv):
            for i in range(nu):
                a, b, c, d = j * (nu + 1) + i, j * (nu + 1) + i + 1, (j + 1) * (nu + 1) + i + 1, (j + 1) * (nu + 1) + i
                idx += [[a, b, c], [a, c, d]]
                uvs += [[uvf(i, j), uvf(i + 1, j), uvf(i + 1, j + 1)], [uvf(i, j), uvf(i + 1, j + 1), uvf(i, j + 1)]]
        return abi.MeshData(vertices=np.array(verts, np.float32), indices=np.array(idx, np.uint32), uvs=np.array(uvs, np.float32))

    def add(mesh, mat):
        meshes.append(mesh)
        mats.append(mat)
        insts.append(abi.InstanceData(len(meshes) - 1, [len(mats) - 1], eye))

    # images
    img8 = rng.integers(0, 256, size=(16, 24, 4), dtype=np.uint8)
    img8[:, :, 3] = np.where(rng.random((16, 24)) < 0.4, 0, 255).astype(np.uint8) if alpha_cutout else 255
    imgf = rng.random((8, 8, 4)).astype(np.float32)
    imgf[:, :, 2] = 0.5 + 0.5 * imgf[:, :, 2]  # normal-map-ish: z > 0.5
    imgf[:, :, 3] = 1.0
    img_e = (rng.random((4, 4, 4)) * 6.0).astype(np.float32)
    img_e[:, :, 3] = 1.0
    images = [abi.ImageData(img8, abi.TEX_FILTER_LINEAR, abi.TEX_REPEAT), abi.ImageData(imgf, abi.TEX_FILTER_LINEAR, abi.TEX_MIRROR),
              abi.ImageData(img_e, abi.TEX_FILTER_NEAREST, abi.TEX_EXTEND), abi.ImageData(img8, abi.TEX_FILTER_NEAREST, abi.TEX_CLIP)]

    # floor (y = -1, normal +y): checkerboard colours, roughness = green channel of the float image
    g = abi.GraphData([
        N(abi.NODE_RGB, (), (0.9, 0.85, 0.8)), N(abi.NODE_SPECTRAL_UPLIFT, (0,)),          # 0, 1 colour 1
        N(abi.NODE_RGB, (), (0.15, 0.2, 0.3)), N(abi.NODE_SPECTRAL_UPLIFT, (2,)),          # 2, 3 colour 2
        N(abi.NODE_CONST, (), (3.0, 0.0, 0.0)),                                            # 4 scale
        N(abi.NODE_CHECKERBOARD, (abi.NODE_NONE, 4, 1, 3)),                                # 5 checker on si.uv
        N(abi.NODE_IMAGE, (1, abi.NODE_NONE, 0)), N(abi.NODE_SEPARATE_COLOR, (6,)), N(abi.NODE_EXTRACT, (7, abi.FIELD_GREEN)),  # 6, 7, 8
        N(abi.NODE_CONST, (), (0.25, 0.0, 0.0)),                                           # 9 constant metallic through the graph
    ], {"base_color": 5, "roughness": 8, "metallic": 9})
    add(quad([-1, -1, 1], [1, -1, 1], [1, -1, -1], [-1, -1, -1], n_floor, n_floor, 2.0),
        abi.MaterialData(base_color=(0.5, 0.5, 0.5), roughness=0.5, ior=1.5, graph=g))
    # back wall (z = -1, normal +z): sRGB byte image through texcoords -> mapping(point)
    g = abi.GraphData([
        N(abi.NODE_TEXCOORDS), N(abi.NODE_EXTRACT, (0, abi.FIELD_UV)),
        N(abi.NODE_CONST, (), (0.125, -0.25, 0.0)), N(abi.NODE_CONST, (), (1.5, 2.0, 1.0)),
        N(abi.NODE_MAPPING, (1, 2, 3, abi.MAPPING_POINT)),
        N(abi.NODE_IMAGE, (0, 4, 1)), N(abi.NODE_SPECTRAL_UPLIFT, (5,)),
    ], {"base_color": 6})
    add(quad([-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1]), abi.MaterialData(roughness=0.9, ior=1.0, specular_ior_level=0.0, graph=g))
    # left wall (x = -1, normal +x): normal map from the float image (mapping type texture), glossy
    g = abi.GraphData([
        N(abi.NODE_TEXCOORDS), N(abi.NODE_CONST, (), (0.1, 0.2, 0.0)), N(abi.NODE_CONST, (), (0.5, 0.25, 1.0)),
        N(abi.NODE_MAPPING, (0, 1, 2, abi.MAPPING_TEXTURE)),
        N(abi.NODE_IMAGE, (1, 3, 0)), N(abi.NODE_CONST, (), (0.7, 0.0, 0.0)), N(abi.NODE_NORMAL_MAP, (4, 5)),
    ], {"normal": 6})
    add(quad([-1, -1, 1], [-1, -1, -1], [-1, 1, -1], [-1, 1, 1]), abi.MaterialData(base_color=(0.7, 0.3, 0.25), roughness=0.35, ior=1.45, graph=g))
    # right wall, front wall, ceiling: constant diffuse-ish principled
    plain = lambda c: abi.MaterialData(base_color=c, roughness=1.0, ior=1.0, specular_ior_level=0.0)  # noqa: E731
    add(quad([1, -1, -1], [1, -1, 1], [1, 1, 1], [1, 1, -1]), plain((0.25, 0.6, 0.3)))
    add(quad([1, -1, 1], [-1, -1, 1], [-1, 1, 1], [1, 1, 1]), plain((0.7, 0.7, 0.7)))
    add(quad([-1, 1, -1], [1, 1, -1], [1, 1, 1], [-1, 1, 1]), plain((0.8, 0.8, 0.8)))
    # ceiling light (y = 0.98, normal -y): emission colour from a float image (nearest), strength constant
    if textured_light:
        g = abi.GraphData([N(abi.NODE_IMAGE, (2, abi.NODE_NONE, 0)), N(abi.NODE_SPECTRAL_UPLIFT, (0,))], {"emission_color": 1})
        lm = abi.MaterialData(base_color=(0.8, 0.8, 0.8), ior=1.0, specular_ior_level=0.0, emission_strength=2.0, graph=g)
    else:
        lm = abi.MaterialData(base_color=(0.8, 0.8, 0.8), ior=1.0, specular_ior_level=0.0, emission_color=(9.0, 8.0, 7.0), emission_strength=1.0)
    add(quad([-0.4, 0.98, -0.4], [0.4, 0.98, -0.4], [0.4, 0.98, 0.4], [-0.4, 0.98, 0.4], 2, 2), lm)
    if alpha_cutout:  # quad at z = 0.2 facing the camera, base colour + alpha from the byte image (clip addressing)
        g = abi.GraphData([N(abi.NODE_IMAGE, (3, abi.NODE_NONE, 1)), N(abi.NODE_SPECTRAL_UPLIFT, (0,))], {"base_color": 1})
        add(quad([-0.6, -0.6, 0.2], [0.6, -0.6, 0.2], [0.6, 0.6, 0.2], [-0.6, 0.6, 0.2], 1, 1, 1.2), abi.MaterialData(roughness=0.8, ior=1.0, specular_ior_level=0.0, graph=g))
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 3] = [0.0, 0.0, 0.95]
    cam = abi.CameraData(c2w=c2w.T.reshape(16).copy(), fov=1.3, width=width, height=height)
    return abi.SceneData(meshes, insts, mats, cam, images=images)


def pxr24_round(a: np.ndarray) -> np.ndarray:
    """What a FLOAT channel holds after PXR24: the value rounded to 24 bits (ImfPxr24Compressor floatToFloat24 for finite values whose
    rounding does not overflow the mantissa into infinity)."""
    b = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((b + 0x80) >> 8) << 8).astype(np.uint32).view(np.float32)


def piz_compress(rows_of_planes, n_rows: int, names_per_row: int, use_runs: bool = True) -> bytes:
    """One PIZ block from the block's channel rows in file order (n_rows x names_per_row arrays). Written from the description of the
    format (OpenEXR's ImfPizCompressor / ImfHuf / ImfWav): value bitmap + forward table, in-place 2-D Haar wavelet per 16-bit word plane
    (14-bit variant when the largest table index is below 2^14, else the modular 16-bit one), canonical Huffman codes with the packed
    length table (zero runs) and the run-length symbol. An ENCODER: independent of the library's decoder by construction."""
    import heapq
    import struct

    # per channel: a (rows, cols, size) array of 16-bit words (FLOAT / UINT: low word first)
    chans = []
    for c in range(names_per_row):
        rows = [rows_of_planes[r * names_per_row + c] for r in range(n_rows)]
        a = np.stack(rows)
        size = 1 if a.dtype == np.float16 else 2
        chans.append(np.ascontiguousarray(a).view(np.uint16).reshape(n_rows, a.shape[1], size).astype(np.int64))
    allw = np.concatenate([c.ravel() for c in chans])
    present = np.zeros(65536, bool)
    present[allw] = True
    bitmap = np.packbits(present, bitorder="little")
    bitmap[0] &= 0xFE  # zero is never stored explicitly
    nz = np.nonzero(bitmap)[0]
    present[0] = True
    lut = np.cumsum(present) - 1
    max_value = int(present.sum()) - 1
    w14 = max_value < (1 << 14)

    def s16(x):
        return ((x + 32768) & 0xFFFF) - 32768

    def wenc(a, b):
        if w14:
            a_, b_ = s16(a), s16(b)
            return ((a_ + b_) >> 1) & 0xFFFF, (a_ - b_) & 0xFFFF
        ao = (a + 0x8000) & 0xFFFF
        m = (ao + b) >> 1
        d = ao - b
        if d < 0:
            m = (m + 0x8000) & 0xFFFF
        return m, d & 0xFFFF

    def wav2_encode(pl):  # pl: (ny, nx) python lists of ints, in place
        ny, nx = len(pl), len(pl[0])
        n = min(nx, ny)
        p, p2 = 1, 2
        while p2 <= n:
            y = 0
            while y <= ny - p2:
                x = 0
                while x <= nx - p2:
                    i00, i01 = wenc(pl[y][x], pl[y][x + p])
                    i10, i11 = wenc(pl[y + p][x], pl[y + p][x + p])
                    pl[y][x], pl[y + p][x] = wenc(i00, i10)
                    pl[y][x + p], pl[y + p][x + p] = wenc(i01, i11)
                    x += p2
                if nx & p:
                    pl[y][x], pl[y + p][x] = wenc(pl[y][x], pl[y + p][x])
                y += p2
            if ny & p:
                x = 0
                while x <= nx - p2:
                    pl[y][x], pl[y][x + p] = wenc(pl[y][x], pl[y][x + p])
                    x += p2
            p, p2 = p2, p2 << 1

    words = []
    for c in chans:
        m = lut[c]
        for j in range(c.shape[2]):
            pl = [[int(v) for v in row] for row in m[:, :, j]]
            wav2_encode(pl)
            m[:, :, j] = np.array(pl)
        words += [int(v) for v in m.ravel()]
    # Huffman: code lengths from the word frequencies + the run-length symbol
    freq = {}
    for v in words:
        freq[v] = freq.get(v, 0) + 1
    im, iM = min(freq), max(freq) + 1
    freq[iM] = 1
    heap = [(f, i, (sym,)) for i, (sym, f) in enumerate(sorted(freq.items()))]
    heapq.heapify(heap)
    length = {sym: 0 for sym in freq}
    tick = len(heap)
    if len(heap) == 1:
        length[heap[0][2][0]] = 1
    while len(heap) > 1:
        f1, _, s1 = heapq.heappop(heap)
        f2, _, s2 = heapq.heappop(heap)
        for sym in s1 + s2:
            length[sym] += 1
        heapq.heappush(heap, (f1 + f2, tick, s1 + s2))
        tick += 1
    assert max(length.values()) <= 58
    count = [0] * 59
    for l in length.values():
        count[l] += 1
    base, c = [0] * 59, 0
    for l in range(58, 0, -1):
        base[l] = c
        c = (c + count[l]) >> 1
    code, nxt = {}, list(base)
    for sym in sorted(length):
        l = length[sym]
        code[sym] = (nxt[l], l)
        nxt[l] += 1
    bits = []

    def put(v, n):
        bits.append((v, n))

    i = im
    while i <= iM:  # the packed table of code lengths
        l = length.get(i, 0)
        if l == 0:
            run = 1
            while i + run <= iM and length.get(i + run, 0) == 0 and run < 255 + 6:
                run += 1
            if run >= 6:
                put(63, 6); put(run - 6, 8); i += run; continue
            if run >= 2:
                put(59 + run - 2, 6); i += run; continue
        put(l, 6)
        i += 1

    def flush(bl):
        acc = n = 0
        out = bytearray()
        for v, k in bl:
            acc = (acc << k) | v
            n += k
            while n >= 8:
                out.append((acc >> (n - 8)) & 255)
                n -= 8
            acc &= (1 << n) - 1
        if n:
            out.append((acc << (8 - n)) & 255)
        return bytes(out), sum(k for _, k in bl)

    table, _ = flush(bits)
    bits = []
    k = 0
    while k < len(words):
        sym, run = words[k], 0
        while k + run + 1 < len(words) and words[k + run + 1] == sym and run < 255:
            run += 1
        cl, rl = code[sym][1], code[iM][1]
        if use_runs and cl + rl + 8 < cl * run:
            put(*code[sym]); put(*code[iM]); put(run, 8)
        else:
            for _ in range(run + 1):
                put(*code[sym])
        k += run + 1
    data, n_bits = flush(bits)
    huf = struct.pack("<IIIII", im, iM, len(table), n_bits, 0) + table + data
    head = struct.pack("<HH", int(nz[0]) if len(nz) else 8191, int(nz[-1]) if len(nz) else 0)
    if len(nz):
        head += bitmap[nz[0]:nz[-1] + 1].tobytes()
    return head + struct.pack("<i", len(huf)) + huf


def _b44_ordered(s: np.ndarray) -> np.ndarray:
    """half bit patterns -> B44's ordered 16-bit representation (larger value = larger number; NaN / infinity -> 0x8000)."""
    s = s.astype(np.int64)
    t = np.where(s & 0x8000, (~s) & 0xffff, s | 0x8000)
    return np.where((s & 0x7c00) == 0x7c00, 0x8000, t)


def b44_pack_block(s16: np.ndarray, flat_fields: bool) -> bytes:
    """One 4 x 4 block of half bit patterns (row-major, 16 values) -> 14 bytes (or 3: B44A, all sixteen equal), from the format's
    description: the first value in the ordered representation, the smallest shift for which the fifteen running differences (down the first
    column, then along the rows), in units of 2^shift and rounded, fit 6 bits around a bias of 32."""
    t = _b44_ordered(np.asarray(s16))
    t_max = int(t.max())
    shift = -1
    while True:
        shift += 1
        # distance from the block's maximum in units of 2^shift, rounded to nearest, ties to even
        x2 = (t_max - t) << 1
        d = (x2 + ((1 << shift) - 1) + ((x2 >> (shift + 1)) & 1)) >> (shift + 1)
        pairs = [(0, 4), (4, 8), (8, 12)] + [(4 * r + c - 1, 4 * r + c) for c in (1, 2, 3) for r in range(4)]
        r = [int(d[i] - d[j]) + 0x20 for i, j in pairs]
        if min(r) >= 0 and max(r) <= 0x3f:
            break
    if flat_fields and min(r) == 0x20 and max(r) == 0x20:
        return bytes([int(t[0]) >> 8, int(t[0]) & 255, 0xfc])
    t0 = t_max - (int(d[0]) << shift)  # the first value as the decoder will rebuild the others from it
    bits = (t0 & 0xffff) << (6 + 90) | shift << 90
    for k, rk in enumerate(r):
        bits |= rk << (90 - 6 * (k + 1))
    return bits.to_bytes(14, "big")


def b44_unpack_block(b: bytes):
    """-> (16 half bit patterns, bytes consumed): the inverse, written from the same description (tests' own reference decoder)."""
    if b[2] >= 13 << 2:
        t = [(b[0] << 8) | b[1]] * 16
        used = 3
    else:
        bits = int.from_bytes(b[:14], "big")
        shift = (bits >> 90) & 0x3f
        r = [(bits >> (90 - 6 * (k + 1))) & 0x3f for k in range(15)]
        t = [0] * 16
        t[0] = bits >> 96
        pairs = [(0, 4), (4, 8), (8, 12)] + [(4 * rr + c - 1, 4 * rr + c) for c in (1, 2, 3) for rr in range(4)]
        for (i, j), rk in zip(pairs, r):
            t[j] = (t[i] + (rk << shift) - (0x20 << shift)) & 0xffff
        used = 14
    return [(v & 0x7fff) if v & 0x8000 else (~v) & 0xffff for v in t], used


def b44_compress(planes_in_order, flat_fields: bool, p_linear=()) -> bytes:
    """A block's channels one after the other: HALF channels as 4 x 4 blocks (edges padded by repetition), the others raw."""
    out = bytearray()
    for k, pl in enumerate(planes_in_order):
        if pl.dtype != np.float16:
            out += np.ascontiguousarray(pl).tobytes()
            continue
        v = pl.view(np.uint16)
        if k in p_linear:  # the writer's side of pLinear: 8 log(x), 0 for negative or non-finite values
            f = pl.astype(np.float64)
            with np.errstate(all="ignore"):
                g = np.where(np.isfinite(f) & (f >= 0), 8.0 * np.log(f), 0.0)
            v = np.where(np.isfinite(g), g, 0.0).astype(np.float32).astype(np.float16).view(np.uint16)
        h, w = v.shape
        for y in range(0, h, 4):
            for x in range(0, w, 4):
                ys = np.minimum(np.arange(y, y + 4), h - 1)
                xs = np.minimum(np.arange(x, x + 4), w - 1)
                out += b44_pack_block(v[np.ix_(ys, xs)].reshape(16), flat_fields)
    return bytes(out)


def b44_reference_decode(blob: bytes, dtypes, shape, exp_table=None, p_linear=()):
    """The planes a reader must get from one block's B44 data (tests' reference; exp_table: the 65536-entry pLinear table)."""
    h, w = shape
    pos, out = 0, []
    for k, dt in enumerate(dtypes):
        if dt != np.float16:
            out.append(np.frombuffer(blob, dtype=dt, count=h * w, offset=pos).reshape(h, w).copy())
            pos += 4 * h * w
            continue
        pl = np.zeros((h, w), np.uint16)
        for y in range(0, h, 4):
            for x in range(0, w, 4):
                s, used = b44_unpack_block(blob[pos:pos + 14])
                pos += used
                blk = np.array(s, np.uint16).reshape(4, 4)
                if k in p_linear:
                    blk = exp_table[blk]
                pl[y:y + 4, x:x + 4] = blk[:min(4, h - y), :min(4, w - x)]
        out.append(pl.view(np.float16))
    assert pos == len(blob)
    return out


def make_exr(planes: dict, compression: int = 3, line_order: int = 0, tiles=None, mipmap: bool = False, p_linear=(), blobs_out=None) -> bytes:
    """Single-part OpenEXR from {channel name: (H, W) array of float16 / float32 / uint32}; compression 0 none,
    1 RLE, 2 ZIPS, 3 ZIP (the file-format definitions: per block, channel rows one after the other, byte de-interleave,
    delta predictor, then RLE / deflate; a block that does not shrink is stored raw), 5 PXR24 (per row and channel the byte planes,
    most significant first, of the running differences of the values -- FLOAT cut to 24 bits -- deflated). tiles = (w, h): a TILED
    file (version flag 0x200, `tiles` attribute, one chunk per tile in row-major order: tile x, tile y, level x, level y, size);
    mipmap: the level mode says MIPMAP and a half-resolution level's tiles follow level 0's (readers of the full resolution skip them).
    6 / 7: B44 / B44A (b44_compress; p_linear = names of the channels flagged pLinear; blobs_out collects (compressed bytes or None, region))."""
    import struct
    import zlib

    names = sorted(planes)
    h, w = planes[names[0]].shape
    tcode = {np.dtype(np.uint32): 0, np.dtype(np.float16): 1, np.dtype(np.float32): 2}

    def attr(name, ty, v):
        return name.encode() + b"\0" + ty.encode() + b"\0" + struct.pack("<I", len(v)) + v

    chl = b"".join(n.encode() + b"\0" + struct.pack("<IIII", tcode[planes[n].dtype], 1 if n in p_linear else 0, 1, 1) for n in names) + b"\0"
    head = struct.pack("<II", 20000630, 2 | (0x200 if tiles else 0))
    head += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
    if tiles:
        head += attr("tiles", "tiledesc", struct.pack("<IIB", tiles[0], tiles[1], 1 if mipmap else 0))
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    head += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", bytes([line_order]))
    head += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    head += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lpb = 32 if compression in (4, 6, 7) else (16 if compression in (3, 5) else 1)

    def pxr24(rows_of_planes):  # [(channel array row), ...] in file order
        out = bytearray()
        for row in rows_of_planes:
            if row.dtype == np.float32:
                v = ((row.view(np.uint32).astype(np.uint64) + 0x80) >> 8).astype(np.uint64)  # 24-bit values
                nb, mod = 3, 1 << 24
            elif row.dtype == np.float16:
                v, nb, mod = row.view(np.uint16).astype(np.uint64), 2, 1 << 16
            else:
                v, nb, mod = row.astype(np.uint64), 4, 1 << 32
            d = v.copy()
            d[1:] = (v[1:] - v[:-1]) % mod
            for k in range(nb):
                out += ((d >> (8 * (nb - 1 - k))) & 255).astype(np.uint8).tobytes()
        return zlib.compress(bytes(out), 6)

    regions = []  # (header bytes before the size, x0, x1, y0, y1)
    if tiles:
        tw, th = tiles
        for ty in range((h + th - 1) // th):
            for tx in range((w + tw - 1) // tw):
                regions.append((struct.pack("<iiii", tx, ty, 0, 0), tx * tw, min(w, tx * tw + tw), ty * th, min(h, ty * th + th)))
    else:
        for y0 in range(0, h, lpb):
            regions.append((struct.pack("<i", y0), 0, w, y0, min(h, y0 + lpb)))
    blocks = []
    for hdr, x0, x1, y0, y1 in regions:
        rows = [np.ascontiguousarray(planes[n][y, x0:x1]) for y in range(y0, y1) for n in names]
        raw = b"".join(r.tobytes() for r in rows)
        blob = raw
        if compression in (6, 7):
            comp = b44_compress([np.ascontiguousarray(planes[n][y0:y1, x0:x1]) for n in names], compression == 7, [k for k, n in enumerate(names) if n in p_linear])
            blob = comp if len(comp) < len(raw) else raw
            if blobs_out is not None:
                blobs_out.append((comp if blob is comp else None, (x0, x1, y0, y1)))
        elif compression == 5:
            comp = pxr24(rows)
            blob = comp if len(comp) < len(raw) else raw
        elif compression == 4:
            comp = piz_compress(rows, y1 - y0, len(names))
            blob = comp if len(comp) < len(raw) else raw
        elif compression:
            t = np.frombuffer(raw, dtype=np.uint8)
            t = np.concatenate([t[0::2], t[1::2]]).astype(np.int64)
            d = t.copy()
            d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
            pre = d.astype(np.uint8).tobytes()
            if compression == 1:
                out, i = bytearray(), 0
                while i < len(pre):  # runs of >= 3 equal bytes, literals otherwise
                    j = i
                    while j + 1 < len(pre) and pre[j + 1] == pre[i] and j - i < 126:
                        j += 1
                    if j - i >= 2:
                        out += bytes([j - i, pre[i]]); i = j + 1
                    else:
                        k = i
                        while k < len(pre) and k - i < 127 and not (k + 2 < len(pre) and pre[k] == pre[k + 1] == pre[k + 2]):
                            k += 1
                        out += bytes([(256 - (k - i)) & 255]) + pre[i:k]; i = k
                comp = bytes(out)
            else:
                comp = zlib.compress(pre, 6)
            blob = comp if len(comp) < len(raw) else raw
        blocks.append(hdr + struct.pack("<I", len(blob)) + blob)
    if tiles and mipmap:  # one more level: what it holds does not matter to a reader of level 0
        lw, lh = max(1, w // 2), max(1, h // 2)
        for ty in range((lh + tiles[1] - 1) // tiles[1]):
            for tx in range((lw + tiles[0] - 1) // tiles[0]):
                blocks.append(struct.pack("<iiiiI", tx, ty, 1, 1, 4) + b"\xde\xad\xbe\xef")
    order = range(len(blocks)) if line_order == 0 else reversed(range(len(blocks)))
    table_pos = len(head)
    offsets = [0] * len(blocks)
    body = b""
    cur = table_pos + 8 * len(blocks)
    for k in order:
        offsets[k] = cur
        body += blocks[k]
        cur += len(blocks[k])
    return head + b"".join(struct.pack("<Q", o) for o in offsets) + body


# ---------------------------------------------------------------------------------------------- ray sets for intersector tests
def probe_rays(world_vertices: np.ndarray, n_random: int, n_adversarial: int, seed: int = 7) -> np.ndarray:
    """(n, 8) f32 rays = origin, direction (unit), tmin = 0, tmax = 1e20 for intersector cross-checks.
    `world_vertices` (T, 3, 3): the scene's f32 world-space triangles. Random rays: origins inside the (slightly grown)
    bounding box, isotropic directions. Adversarial rays: aimed at points ON triangle edges and AT vertices (where two or
    more triangles meet and the inside tests of neighbouring triangles decide by the last bit), from random origins."""
    rng = np.random.default_rng(seed)
    v = world_vertices.reshape(-1, 3).astype(np.float64)
    lo, hi = v.min(axis=0), v.max(axis=0)
    ext = np.maximum(hi - lo, 1e-3)
    lo, hi = lo - 0.05 * ext, hi + 0.05 * ext

    def dirs(n):
        d = rng.normal(size=(n, 3))
        return d / np.linalg.norm(d, axis=1, keepdims=True)

    o1 = lo + rng.random((n_random, 3)) * (hi - lo)
    d1 = dirs(n_random)
    T = world_vertices.shape[0]
    tri = world_vertices[rng.integers(0, T, size=n_adversarial)].astype(np.float64)
    kind = rng.integers(0, 3, size=n_adversarial)           # 0: on an edge, 1: at a vertex, 2: interior
    e = rng.integers(0, 3, size=n_adversarial)
    s = rng.random(n_adversarial)[:, None]
    a, b = tri[np.arange(n_adversarial), e], tri[np.arange(n_adversarial), (e + 1) % 3]
    w = rng.dirichlet((1, 1, 1), size=n_adversarial)
    target = np.where((kind == 0)[:, None], a + s * (b - a), np.where((kind == 1)[:, None], a, (tri * w[:, :, None]).sum(axis=1)))
    o2 = lo + rng.random((n_adversarial, 3)) * (hi - lo)
    d2 = target - o2
    ln = np.linalg.norm(d2, axis=1, keepdims=True)
    d2 = np.where(ln > 0, d2 / np.maximum(ln, 1e-30), dirs(n_adversarial))
    rays = np.zeros((n_random + n_adversarial, 8), dtype=np.float32)
    rays[:, 0:3] = np.concatenate([o1, o2]).astype(np.float32)
    rays[:, 3:6] = np.concatenate([d1, d2]).astype(np.float32)
    rays[:, 7] = 1e20
    return rays


def _conditioning(world_vertices, rays, gid):
    """Per ray, for triangle gid[i]: (|cos| between the ray and the triangle's normal, R / smallest altitude) with R the
    largest coordinate magnitude in play. An f32 plane solve t = -(n.o + c) / (n.d) carries a few ulp / |cos| of relative
    error; u = r0.p + c0 with |r0| = 1 / altitude and |p| <= R adds a few ulp x R / altitude. (1, 1) where gid is invalid."""
    ok = gid != 0xFFFFFFFF
    tri = world_vertices[np.where(ok, gid, 0)].astype(np.float64)
    e0, e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], tri[:, 2] - tri[:, 1]
    nrm = np.cross(e0, e1)
    ln = np.linalg.norm(nrm, axis=1)
    longest = np.maximum(np.maximum(np.linalg.norm(e0, axis=1), np.linalg.norm(e1, axis=1)), np.linalg.norm(e2, axis=1))
    alt = ln / np.maximum(longest, 1e-300)
    R = max(float(np.abs(world_vertices).max()), float(np.abs(rays[:, 0:3]).max()))
    c = np.abs((rays[:, 3:6].astype(np.float64) * (nrm / np.maximum(ln, 1e-300)[:, None])).sum(axis=1))
    return np.where(ok, np.maximum(c, 1e-12), 1.0), np.where(ok, np.maximum(1.0, R / np.maximum(alt, 1e-300)), 1.0)


def check_against_mt_f64(world_vertices, rays, hit, gid, tuv, mt_gid, mt, mt_own, eps: float = 1e-6):
    """Compares an f32 intersector (hit flag, global triangle id, (t, u, v)) with the independent f64 Moeller-Trumbore
    closest hit (mt_gid, mt = (t, u, v, margin)); mt_own = the f64 solve for the triangle the f32 intersector chose.
    Every tolerance is `eps x conditioning` with eps = 1e-6 (16 ulp of f32) and the conditioning of _conditioning():
    1 / |cos| for t, (R / altitude) / |cos| for u, v and the inside margin. Measured worst cases on 10^6 rays: 0.3e-6 (t) and
    0.2e-6 (u, v) in these units. Two intersectors may legitimately disagree only on a razor's edge: the f64 inside margin of
    the triangle one accepted and the other rejected is within tolerance of zero, or two candidates are within tolerance of
    the same distance. Everything else is counted as `unexplained` -- a defect. Returns counts and worst cases."""
    hit = hit.astype(bool)
    mt_hit = mt_gid != 0xFFFFFFFF
    cos_own, k_own = _conditioning(world_vertices, rays, np.where(hit, gid, 0xFFFFFFFF).astype(np.uint32))
    cos_mt, k_mt = _conditioning(world_vertices, rays, mt_gid)
    same = hit & mt_hit & (gid == mt_gid)
    out = {"n": int(hit.size), "both_miss": int((~hit & ~mt_hit).sum()), "same_triangle": int(same.sum())}
    dt = np.abs(tuv[same, 0].astype(np.float64) - mt[same, 0]) / np.maximum(1.0, np.abs(mt[same, 0])) * cos_own[same]
    du = np.abs(tuv[same, 1].astype(np.float64) - mt[same, 1]) * cos_own[same] / k_own[same]
    dv = np.abs(tuv[same, 2].astype(np.float64) - mt[same, 2]) * cos_own[same] / k_own[same]
    out["max_dt_scaled"], out["max_du_scaled"], out["max_dv_scaled"] = (float(x.max()) if x.size else 0.0 for x in (dt, du, dv))
    only_f32 = hit & ~mt_hit          # f32 accepted a triangle f64 rejects everywhere
    only_f64 = ~hit & mt_hit
    other = hit & mt_hit & (gid != mt_gid)
    tol_own, tol_mt = eps * k_own / cos_own, eps * k_mt / cos_mt
    bad = 0
    m = mt_own[only_f32]
    bad += int((~((m[:, 3] > -tol_own[only_f32]) & (m[:, 0] > -tol_own[only_f32]))).sum())   # chosen triangle: a hair outside at most
    # f64's triangle: a hair inside at most -- or a hair inside the ray's range (an origin ON a triangle: t = +-1e-8)
    t_edge = np.minimum(np.abs(mt[:, 0] - rays[:, 6]), np.abs(mt[:, 0] - rays[:, 7])) < tol_mt
    bad += int((~((mt[only_f64, 3] < tol_mt[only_f64]) | t_edge[only_f64])).sum())
    if other.any():  # different triangles: (nearly) the same distance, or one of the two is an edge case
        t32, t64 = mt_own[other, 0], mt[other, 0]
        near = np.abs(t32 - t64) <= eps / np.minimum(cos_own[other], cos_mt[other]) * np.maximum(1.0, np.abs(t64))
        edge = (np.abs(mt_own[other, 3]) < tol_own[other]) | (np.abs(mt[other, 3]) < tol_mt[other])
        bad += int((~(near | edge)).sum())
    out["only_f32"], out["only_f64"], out["other_triangle"], out["unexplained"] = int(only_f32.sum()), int(only_f64.sum()), int(other.sum()), bad
    return out


# ---------------------------------------------------------------------------------------------- instanced scenes
def instanced_scene(n_inst=12, n=6, width=48, height=48, seed=5, with_normals=True, with_uvs=True, mirror=True, emissive_instances=1,
                    alpha=False, textured=False, tangents=False) -> abi.SceneData:
    """`n_inst` copies of one bumpy blob (a closed latitude/longitude mesh of 2 n (2n - 1)... triangles, two material slots, corner
    normals, uvs) with rotated, non-uniformly scaled and (every third) mirrored transforms, over a floor quad, under a light quad.
    `emissive_instances` of the copies carry an emissive material in slot 1 (lights on an instanced mesh)."""
    rng = np.random.default_rng(seed)
    # the blob: rings x segments grid on a sphere with a radial bump
    rings, segs = n, 2 * n
    th = np.linspace(0.0, np.pi, rings + 1, dtype=np.float32)
    ph = np.linspace(0.0, 2.0 * np.pi, segs + 1, dtype=np.float32)[:-1]
    bump = (1.0 + 0.25 * rng.random((rings + 1, segs))).astype(np.float32)
    bump[0, :] = bump[0, 0]
    bump[-1, :] = bump[-1, 0]
    verts = np.array([[bump[j, i] * np.sin(th[j]) * np.cos(ph[i]), bump[j, i] * np.cos(th[j]), bump[j, i] * np.sin(th[j]) * np.sin(ph[i])]
                      for j in range(rings + 1) for i in range(segs)], dtype=np.float32)
    vuv = np.array([[i / segs, j / rings] for j in range(rings + 1) for i in range(segs)], dtype=np.float32)
    idx = []
    for j in range(rings):
        for i in range(segs):
            a, b = j * segs + i, j * segs + (i + 1) % segs
            c, d = (j + 1) * segs + (i + 1) % segs, (j + 1) * segs + i
            if j > 0:
                idx.append([a, b, c])
            if j < rings - 1:
                idx.append([a, c, d])
    idx = np.array(idx, dtype=np.uint32)
    slots = (rng.random(idx.shape[0]) < 0.35).astype(np.uint32)
    normals = None
    if with_normals:
        vn = np.zeros_like(verts)
        for t in idx:
            fn = np.cross(verts[t[1]] - verts[t[0]], verts[t[2]] - verts[t[0]])
            for k in t:
                vn[k] += fn
        vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-20)
        normals = vn[idx].astype(np.float32)
    uvs = vuv[idx].astype(np.float32) if with_uvs else None
    tang = None
    if tangents:  # per-corner tangents (mesh.rs:557-571: used only when all nine floats of a triangle are finite)
        tang = rng.normal(size=(idx.shape[0], 3, 3)).astype(np.float32)
        tang /= np.linalg.norm(tang, axis=2, keepdims=True)
        tang[::7, 1, 2] = np.nan
    blob = abi.MeshData(vertices=verts, indices=idx, material_slots=slots, normals=normals, uvs=uvs, tangents=tang)
    quad = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32)
    floor = abi.MeshData(vertices=quad * np.float32(6.0), indices=np.array([[0, 2, 1], [0, 3, 2]], dtype=np.uint32))
    light = abi.MeshData(vertices=quad * np.float32(1.5) + np.array([0, 6.0, 0], dtype=np.float32), indices=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32))
    mats = [
        abi.MaterialData(base_color=(0.7, 0.65, 0.6), roughness=0.7, ior=1.45, specular_ior_level=0.5),                       # 0 floor
        abi.MaterialData(base_color=(0.8, 0.8, 0.8), ior=1.0, specular_ior_level=0.0, emission_color=(9.0, 8.0, 7.0), emission_strength=1.0),  # 1 light
        abi.MaterialData(base_color=(0.2, 0.6, 0.3), roughness=0.5, ior=1.45, specular_ior_level=0.5),                        # 2 blob a
        abi.MaterialData(base_color=(0.9, 0.85, 0.6), roughness=0.2, metallic=1.0, ior=1.5),                                  # 3 blob b
        abi.MaterialData(base_color=(0.6, 0.3, 0.2), roughness=0.4, ior=1.45, coat_weight=0.6, coat_roughness=0.1, coat_ior=1.5),  # 4 blob c
        abi.MaterialData(base_color=(0.5, 0.5, 0.5), ior=1.0, specular_ior_level=0.0, emission_color=(2.0, 3.0, 4.0), emission_strength=1.0),  # 5 glow
    ]
    if alpha:
        mats[2].base_alpha = 0.5
    images = []
    if textured:  # slot 0: byte image with holes (alpha cut-out) through a mapping; coat material: normal map; glow: textured emission
        N = abi.NodeData
        img8 = rng.integers(0, 256, size=(16, 24, 4), dtype=np.uint8)
        img8[:, :, 3] = np.where(rng.random((16, 24)) < 0.3, 0, 255).astype(np.uint8)
        imgf = rng.random((8, 8, 4)).astype(np.float32)
        imgf[:, :, 2] = 0.5 + 0.5 * imgf[:, :, 2]
        imgf[:, :, 3] = 1.0
        img_e = (rng.random((4, 4, 4)) * 5.0).astype(np.float32)
        img_e[:, :, 3] = 1.0
        images = [abi.ImageData(img8, abi.TEX_FILTER_LINEAR, abi.TEX_REPEAT), abi.ImageData(imgf, abi.TEX_FILTER_LINEAR, abi.TEX_MIRROR),
                  abi.ImageData(img_e, abi.TEX_FILTER_NEAREST, abi.TEX_REPEAT)]
        mats[2].graph = abi.GraphData([
            N(abi.NODE_TEXCOORDS), N(abi.NODE_EXTRACT, (0, abi.FIELD_UV)), N(abi.NODE_CONST, (), (0.125, -0.25, 0.0)), N(abi.NODE_CONST, (), (3.0, 2.0, 1.0)),
            N(abi.NODE_MAPPING, (1, 2, 3, abi.MAPPING_POINT)), N(abi.NODE_IMAGE, (0, 4, 1)), N(abi.NODE_SPECTRAL_UPLIFT, (5,))], {"base_color": 6})
        mats[4].graph = abi.GraphData([
            N(abi.NODE_IMAGE, (1, abi.NODE_NONE, 0)), N(abi.NODE_CONST, (), (0.8, 0.0, 0.0)), N(abi.NODE_NORMAL_MAP, (0, 1)),
            N(abi.NODE_RGB, (), (0.9, 0.5, 0.2)), N(abi.NODE_SPECTRAL_UPLIFT, (3,)), N(abi.NODE_RGB, (), (0.1, 0.2, 0.5)), N(abi.NODE_SPECTRAL_UPLIFT, (5,)),
            N(abi.NODE_CONST, (), (6.0, 0.0, 0.0)), N(abi.NODE_CHECKERBOARD, (abi.NODE_NONE, 7, 4, 6))], {"normal": 2, "base_color": 8})
        mats[5].graph = abi.GraphData([N(abi.NODE_IMAGE, (2, abi.NODE_NONE, 0)), N(abi.NODE_SPECTRAL_UPLIFT, (0,))], {"emission_color": 1})
        mats[5].emission_strength = 1.5
    eye = np.eye(4, dtype=np.float32)
    insts = [abi.InstanceData(1, [0], eye.T.reshape(16).copy()), abi.InstanceData(2, [1], eye.T.reshape(16).copy())]
    for k in range(n_inst):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = rng.random() * 2 * np.pi
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        S = np.diag(0.35 + 0.5 * rng.random(3))
        if mirror and k % 3 == 2:
            S[0, 0] = -S[0, 0]
        M = np.eye(4)
        M[:3, :3] = R @ S
        M[:3, 3] = [(rng.random() - 0.5) * 7.0, 0.6 + rng.random() * 2.5, (rng.random() - 0.5) * 7.0]
        slot1 = 5 if k < emissive_instances else (3 if k % 2 else 4)
        insts.append(abi.InstanceData(0, [2, slot1], M.astype(np.float32).T.reshape(16).copy()))
    ca = np.float32(-0.35)
    c2w = np.array([[1, 0, 0, 0], [0, np.cos(ca), -np.sin(ca), 4.0], [0, np.sin(ca), np.cos(ca), 9.0], [0, 0, 0, 1]], dtype=np.float32)
    cam = abi.CameraData(c2w=c2w.T.reshape(16).copy(), fov=0.8, width=width, height=height)
    return abi.SceneData([blob, floor, light], insts, mats, cam, images=images)


def shift_scene(sd: abi.SceneData, offset) -> abi.SceneData:
    """The whole scene, camera included, moved by `offset` (a scene modelled far from the origin)."""
    off = np.asarray(offset, dtype=np.float32)
    for inst in sd.instances:
        t = np.asarray(inst.transform, dtype=np.float32).reshape(4, 4).copy()  # stored transposed: row 3 is the translation
        t[3, :3] += off
        inst.transform = t.reshape(16)
    c = np.asarray(sd.camera.c2w, dtype=np.float32).reshape(4, 4).copy()
    c[3, :3] += off
    sd.camera.c2w = c.reshape(16)
    return sd


def extreme_instanced_scene(seed: int, width: int = 40, height: int = 32):
    """An instanced scene under transforms far outside the soak's range: log-uniform scales 1e-4 .. 1e4 (non-uniform, mirrored,
    sheared), offsets up to 1e4 x the scene's unit, meshes squashed into slivers. What stresses the CULLING of a tree (the padding
    of its boxes): tools/inst_extreme_check.py, tests/test_bvh_conservative.py, tests/test_gpu_parity.py. Returns (scene, config)."""
    rng = np.random.default_rng(seed)
    sd = instanced_scene(n_inst=int(rng.integers(3, 20)), n=int(rng.integers(3, 10)), width=width, height=height, seed=seed,
                         emissive_instances=int(rng.integers(0, 3)), with_normals=bool(rng.random() < 0.5), alpha=bool(rng.random() < 0.3),
                         textured=bool(rng.random() < 0.3))
    if rng.random() < 0.3:  # slivers
        v = sd.meshes[0].vertices.copy()
        v[:, int(rng.integers(0, 3))] *= np.float32(10.0 ** rng.uniform(-5, -1))
        sd.meshes[0].vertices = v
        sd.meshes[0].normals = None
    world = 10.0 ** rng.uniform(-3, 3)      # the whole scene's unit
    offset = rng.uniform(-1, 1, size=3) * 10.0 ** rng.uniform(0, 4) * world * float(rng.random() < 0.6)
    for k, inst in enumerate(sd.instances):
        t = np.asarray(inst.transform, dtype=np.float64).reshape(4, 4).copy()  # transposed: rows are columns
        if k >= 2 and rng.random() < 0.5:     # a blob: its own extreme, non-uniform scale (the camera still looks at the cluster)
            s3 = 10.0 ** rng.uniform(-2, 2, size=3) * rng.choice([1.0, 1.0, -1.0], size=3)
            t[:3, :3] = t[:3, :3] * s3[:, None]
            if rng.random() < 0.3:
                t[0, :3] += rng.uniform(-2, 2) * t[1, :3]
        t[:3, :3] *= world
        t[3, :3] = t[3, :3] * world + offset
        inst.transform = t.astype(np.float32).reshape(16)
    c = np.asarray(sd.camera.c2w, dtype=np.float64).reshape(4, 4).copy()
    c[3, :3] = c[3, :3] * world + offset
    sd.camera.c2w = c.astype(np.float32).reshape(16)
    cfg = make_config(spp=4, spp_per_pass=4, max_depth=int(rng.integers(2, 10)), force_diffuse=int(rng.random() < 0.3), sampler_type=int(rng.integers(0, 3)))
    return sd, cfg


def far_modelled_mesh_scene(offset: float, **kw) -> abi.SceneData:
    """instanced_scene with its shared mesh modelled `offset` away from its own origin and every instance's translation taking it
    back: world coordinates as before, object-space coordinates and translations ~ offset -- what cancels when a ray is taken
    through an instance's inverse (the padding of the per-mesh trees has to follow THAT magnitude)."""
    sd = instanced_scene(**kw)
    off = np.array([offset, -0.5 * offset, 0.25 * offset], dtype=np.float64)
    v = np.asarray(sd.meshes[0].vertices, dtype=np.float64) + off[None, :]
    sd.meshes[0].vertices = v.astype(np.float32)
    for inst in sd.instances:
        if inst.mesh != 0:
            continue
        t = np.asarray(inst.transform, dtype=np.float64).reshape(4, 4).copy()  # stored transposed: rows 0..2 are the matrix columns
        t[3, :3] -= off @ t[:3, :3]
        inst.transform = t.astype(np.float32).reshape(16)
    return sd
