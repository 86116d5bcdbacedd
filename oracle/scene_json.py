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
from typing import Dict, List

import numpy as np

from akari_render_amd import abi
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

# which MaterialData.colorspaces bit an ACEScg constant of this input sets (abi.MAT_CS_*)
_CS_BIT = {"base_color": abi.MAT_CS_BASE_COLOR, "specular_tint": abi.MAT_CS_SPECULAR_TINT, "coat_tint": abi.MAT_CS_COAT_TINT,
           "emission_color": abi.MAT_CS_EMISSION_COLOR}


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


def decode_png(data: bytes) -> np.ndarray:
    """PNG -> (H, W, 4) uint8, file order, with the expansions of the image crate's decode().to_rgba8()
    (palette, tRNS, low bit depths, 16 -> 8 bit with rounding). Independent of the C++ reader: zlib does the inflate."""
    import struct
    import zlib

    assert data[:8] == b"\x89PNG\r\n\x1a\n", "bad PNG signature"
    pos, idat, plte, trns, hdr = 8, b"", b"", b"", None
    while pos + 12 <= len(data):
        (ln,) = struct.unpack(">I", data[pos : pos + 4])
        ty = data[pos + 4 : pos + 8]
        body = data[pos + 8 : pos + 8 + ln]
        if ty == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif ty == b"PLTE":
            plte = body
        elif ty == b"tRNS":
            trns = body
        elif ty == b"IDAT":
            idat += body
        elif ty == b"IEND":
            break
        pos += 12 + ln
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace:  # Adam7: through Pillow (an independent decoder); its 8-bit RGBA conversion is the image crate's for depths <= 8
        if depth > 8:
            raise NotImplementedError("interlaced 16-bit PNG")
        import io

        from PIL import Image

        return np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"), dtype=np.uint8).copy()
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bits = ch * depth
    stride, bpp = (w * bits + 7) // 8, (bits + 7) // 8
    raw = zlib.decompress(idat)
    img = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft = raw[(stride + 1) * y]
        line = np.frombuffer(raw[(stride + 1) * y + 1 : (stride + 1) * (y + 1)], dtype=np.uint8).astype(np.int32)
        cur = np.zeros(stride, dtype=np.int32)
        if ft == 0:
            cur = line.copy()
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if ft == 1:
                    p_ = a
                elif ft == 3:
                    p_ = (a + b) >> 1
                elif ft == 4:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    p_ = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError("bad PNG filter")
                cur[x] = (line[x] + p_) & 255
        img[y] = cur.astype(np.uint8)
        prev = cur
    # samples as integers
    if depth == 8:
        smp = img.astype(np.uint32)
    elif depth == 16:
        smp = (img[:, 0::2].astype(np.uint32) << 8) | img[:, 1::2].astype(np.uint32)
    else:
        bitsarr = np.unpackbits(img, axis=1)[:, : w * ch * depth].reshape(h, w * ch, depth)
        smp = np.zeros((h, w * ch), dtype=np.uint32)
        for k in range(depth):
            smp = (smp << 1) | bitsarr[:, :, k]
    smp = smp[:, : w * ch].reshape(h, w, ch)

    def to8(v):
        if depth == 8:
            return v.astype(np.uint8)
        if depth == 16:
            return ((v + 128) // 257).astype(np.uint8)
        return (v * 255 // ((1 << depth) - 1)).astype(np.uint8)

    out = np.full((h, w, 4), 255, dtype=np.uint8)
    if ctype == 3:
        pal = np.frombuffer(plte, dtype=np.uint8).reshape(-1, 3)
        idx = smp[:, :, 0]
        out[:, :, :3] = pal[idx]
        al = np.full(256, 255, dtype=np.uint8)
        al[: len(trns)] = np.frombuffer(trns, dtype=np.uint8)
        out[:, :, 3] = al[idx]
    elif ctype == 0:
        out[:, :, 0] = out[:, :, 1] = out[:, :, 2] = to8(smp[:, :, 0])
        if len(trns) >= 2:
            key = (trns[0] << 8) | trns[1]
            out[:, :, 3] = np.where(smp[:, :, 0] == key, 0, 255)
    elif ctype == 4:
        out[:, :, 0] = out[:, :, 1] = out[:, :, 2] = to8(smp[:, :, 0])
        out[:, :, 3] = to8(smp[:, :, 1])
    elif ctype == 2:
        out[:, :, :3] = to8(smp)
        if len(trns) >= 6:
            key = [(trns[2 * k] << 8) | trns[2 * k + 1] for k in range(3)]
            hit = (smp[:, :, 0] == key[0]) & (smp[:, :, 1] == key[1]) & (smp[:, :, 2] == key[2])
            out[:, :, 3] = np.where(hit, 0, 255)
    else:
        out[:, :, :] = to8(smp)
    return out


def decode_tiff_pil(data: bytes) -> np.ndarray:
    """TIFF -> (H, W, 4) uint8 through Pillow / libtiff (an independent decoder), with the image crate's `to_rgba8` rules applied
    to what it returns: grey replicated, alpha 255 when absent, 16-bit samples (v + 128) // 257."""
    import io

    from PIL import Image

    im = Image.open(io.BytesIO(data))
    a = np.asarray(im)
    if a.dtype == np.uint16:
        a = ((a.astype(np.uint32) + 128) // 257).astype(np.uint8)
    elif a.dtype != np.uint8:
        raise NotImplementedError(f"tiff sample type {a.dtype}")
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    out = np.full((h, w, 4), 255, dtype=np.uint8)
    if c <= 2:
        out[:, :, :3] = a[:, :, :1]
        if c == 2:
            out[:, :, 3] = a[:, :, 1]
    else:
        out[:, :, :c] = a
    return out


def decode_dds(data: bytes) -> np.ndarray:
    """DDS with DXT1 / DXT3 / DXT5 blocks -> (H, W, 4) uint8, the integer rules of the image crate's decoder (image 0.24
    src/codecs/dxt.rs: 565 channels c * 255 // 31 (63); (2a + b + 1) // 3; DXT1 3-colour mode (a + b + 1) // 2 and black, alpha
    255; DXT5 alpha ((8 - i) a0 + (i - 1) a1) // 7 or ((6 - i) a0 + (i - 1) a1) // 5, 0, 255; DXT3 nibble * 17)."""
    import struct

    if data[:4] != b"DDS " or struct.unpack_from("<I", data, 4)[0] != 124:
        raise ValueError("not a DDS file")
    h, w = struct.unpack_from("<II", data, 12)
    fourcc = data[84:88]
    at = 128
    if fourcc == b"DX10":
        dxgi = struct.unpack_from("<I", data, 128)[0]
        at = 148
        kind = {70: 1, 71: 1, 72: 1, 73: 3, 74: 3, 75: 3, 76: 5, 77: 5, 78: 5}[dxgi]
    else:
        kind = {b"DXT1": 1, b"DXT3": 3, b"DXT5": 5}[fourcc]
    bw, bh = (w + 3) // 4, (h + 3) // 4
    out = np.full((bh * 4, bw * 4, 4), 255, dtype=np.uint8)
    size = 8 if kind == 1 else 16

    def colours(s, dxt1):
        c0, c1, table = struct.unpack_from("<HHI", s, 0)
        dec = lambda v: [((v >> 11) & 31) * 255 // 31, ((v >> 5) & 63) * 255 // 63, (v & 31) * 255 // 31]  # noqa: E731
        col = [dec(c0), dec(c1), [0, 0, 0], [0, 0, 0]]
        if c0 > c1 or not dxt1:
            col[2] = [(2 * a + b + 1) // 3 for a, b in zip(col[0], col[1])]
            col[3] = [(a + 2 * b + 1) // 3 for a, b in zip(col[0], col[1])]
        else:
            col[2] = [(a + b + 1) // 2 for a, b in zip(col[0], col[1])]
        return [col[(table >> (2 * i)) & 3] for i in range(16)]

    for by in range(bh):
        for bx in range(bw):
            s = data[at + (by * bw + bx) * size: at + (by * bw + bx + 1) * size]
            alpha = [255] * 16
            if kind == 1:
                rgb = colours(s, True)
            elif kind == 3:
                alpha = [((s[i // 2] >> (4 * (i & 1))) & 15) * 17 for i in range(16)]
                rgb = colours(s[8:], False)
            else:
                a0, a1 = s[0], s[1]
                tab = [a0, a1, 0, 0, 0, 0, 0, 255]
                if a0 > a1:
                    for i in range(2, 8):
                        tab[i] = ((8 - i) * a0 + (i - 1) * a1) // 7
                else:
                    for i in range(2, 6):
                        tab[i] = ((6 - i) * a0 + (i - 1) * a1) // 5
                bits = int.from_bytes(s[2:8], "little")
                alpha = [tab[(bits >> (3 * i)) & 7] for i in range(16)]
                rgb = colours(s[8:], False)
            for i in range(16):
                out[by * 4 + i // 4, bx * 4 + i % 4] = rgb[i] + [alpha[i]]
    return out[:h, :w].copy()


def decode_exr(data: bytes) -> np.ndarray:
    """OpenEXR (single-part scanline, none / RLE / ZIPS / ZIP, half / float / uint channels) -> (H, W, 4) float32 in file
    order; R, G, B (a lone Y is replicated), A = 1 when absent. Independent of the C++ reader (numpy + zlib)."""
    import struct
    import zlib

    assert struct.unpack_from("<I", data, 0)[0] == 20000630, "bad EXR magic"
    version = struct.unpack_from("<I", data, 4)[0]
    if version & 0x1A00:
        raise NotImplementedError("tiled / deep / multi-part EXR")
    pos, chans, comp, dw = 8, [], None, None
    while data[pos] != 0:
        e = data.index(b"\0", pos); name = data[pos:e].decode(); pos = e + 1
        e = data.index(b"\0", pos); pos = e + 1
        (size,) = struct.unpack_from("<I", data, pos); pos += 4
        v = data[pos : pos + size]; pos += size
        if name == "channels":
            q = 0
            while v[q] != 0:
                e = v.index(b"\0", q); cn = v[q:e].decode(); q = e + 1
                ty, _, xs, ys = struct.unpack_from("<IIII", v, q); q += 16
                assert xs == 1 and ys == 1
                chans.append((cn, ty))
        elif name == "compression":
            comp = v[0]
        elif name == "dataWindow":
            dw = struct.unpack("<iiii", v)
    pos += 1
    if comp not in (0, 1, 2, 3):
        raise NotImplementedError(f"EXR compression {comp}")
    W, H = dw[2] - dw[0] + 1, dw[3] - dw[1] + 1
    lpb = 16 if comp == 3 else 1
    nblk = (H + lpb - 1) // lpb
    sizes = [2 if ty == 1 else 4 for _, ty in chans]
    row_bytes = W * sum(sizes)
    planes = {cn: np.zeros((H, W), dtype=np.float32) for cn, _ in chans}

    def unpredict(b: bytes) -> bytes:
        t = np.frombuffer(b, dtype=np.uint8).astype(np.int64)
        t[1:] -= 128
        t = np.cumsum(t) & 255
        half = (len(t) + 1) // 2
        out = np.zeros(len(t), dtype=np.uint8)
        out[0::2] = t[:half]
        out[1::2] = t[half:]
        return out.tobytes()

    def unrle(b: bytes) -> bytes:
        out, i = bytearray(), 0
        while i < len(b):
            c = b[i] - 256 if b[i] > 127 else b[i]
            i += 1
            if c < 0:
                out += b[i : i - c]; i += -c
            else:
                out += bytes([b[i]]) * (c + 1); i += 1
        return bytes(out)

    for k in range(nblk):
        (off,) = struct.unpack_from("<Q", data, pos + 8 * k)
        y0, csize = struct.unpack_from("<iI", data, off)
        rows = min(lpb, dw[3] - y0 + 1)
        blob = data[off + 8 : off + 8 + csize]
        if comp != 0 and csize != row_bytes * rows:
            blob = unpredict(unrle(blob) if comp == 1 else zlib.decompress(blob))
        p = 0
        for r in range(rows):
            for (cn, ty), sz in zip(chans, sizes):
                seg = blob[p : p + W * sz]; p += W * sz
                arr = np.frombuffer(seg, dtype={0: np.uint32, 1: np.float16, 2: np.float32}[ty]).astype(np.float32)
                planes[cn][y0 - dw[1] + r] = arr
    out = np.zeros((H, W, 4), dtype=np.float32)
    out[:, :, 3] = 1.0
    names = [cn for cn, _ in chans]
    if not any(c in names for c in "RGB") and "Y" in names:
        out[:, :, 0] = out[:, :, 1] = out[:, :, 2] = planes["Y"]
    for k, cn in enumerate("RGBA"):
        if cn in planes:
            out[:, :, k] = planes[cn]
    return out


class _Graph:
    """Translation of the texture-fed inputs of a surface node (svm/compiler.rs:116-337) into abi.GraphData."""

    def __init__(self, nodes: dict, bufs: "_Buffers", images: list, image_index: dict):
        self.nodes, self.bufs, self.images, self.image_index = nodes, bufs, images, image_index
        self.out: List[abi.NodeData] = []
        self.inputs: dict = {}
        self.memo: dict = {}

    def is_const(self, ref) -> bool:
        n = self.nodes[ref["id"]]
        if n["type"] in ("float", "float3", "rgb"):
            return True
        if n["type"] == "spectral_uplift":
            return self.is_const(n["rgb"])
        return False

    def push(self, op, args=(), k=(0.0, 0.0, 0.0)) -> int:
        self.out.append(abi.NodeData(op, tuple(args), tuple(k)))
        return len(self.out) - 1

    def image(self, im: dict) -> int:
        key = (im["data"]["id"], im["format"], im["extension"], im["interpolation"], im["width"], im["height"], im["channels"])
        if key in self.image_index:
            return self.image_index[key]
        address = {"repeat": abi.TEX_REPEAT, "clip": abi.TEX_CLIP, "mirror": abi.TEX_MIRROR, "extend": abi.TEX_EXTEND}[im["extension"]]
        filt = {"linear": abi.TEX_FILTER_LINEAR, "cubic": abi.TEX_FILTER_LINEAR, "nearest": abi.TEX_FILTER_NEAREST}[im["interpolation"]]
        v = self.bufs.scene["buffer_views"][im["data"]["id"]]
        raw = self.bufs.buffer(v["buffer"]["id"])[v["offset"] : v["offset"] + v["length"]]
        if im["format"] == "float":
            w, h, ch = im["width"], im["height"], im["channels"]
            src = np.frombuffer(raw, dtype=np.float32).reshape(h, w, ch)
            tex = np.zeros((h, w, 4), dtype=np.float32)
            tex[:, :, 3] = 1.0
            tex[:, :, :ch] = src  # load.rs:552-569; not flipped
        elif im["format"] == "png":
            tex = decode_png(raw)[::-1].copy()  # flipv, load.rs:596
        elif im["format"] == "exr":
            tex = decode_exr(raw)[::-1].copy()
        elif im["format"] == "jpeg":
            import io

            from PIL import Image  # independent decoder (libjpeg); texels may differ from the library's by a few LSB

            tex = np.asarray(Image.open(io.BytesIO(raw)).convert("RGBA"), dtype=np.uint8)[::-1].copy()
        elif im["format"] == "tiff":
            tex = decode_tiff_pil(raw)[::-1].copy()
        elif im["format"] == "dds":
            tex = decode_dds(raw)[::-1].copy()
        else:
            raise NotImplementedError(f"image format '{im['format']}'")
        self.images.append(abi.ImageData(tex, filt, address))
        self.image_index[key] = len(self.images) - 1
        return self.image_index[key]

    def emit(self, ref) -> int:
        nid = ref["id"]
        if nid in self.memo:
            return self.memo[nid]
        n = self.nodes[nid]
        ty = n["type"]
        opt = lambda key: self.emit(n[key]) if n.get(key) is not None else abi.NODE_NONE  # noqa: E731
        if ty == "float":
            r = self.push(abi.NODE_CONST, (), (n["value"], 0.0, 0.0))
        elif ty == "float3":
            r = self.push(abi.NODE_CONST, (), tuple(n["value"]))
        elif ty == "rgb":
            cs = {"srgb": 0, "aces": 1}[n.get("colorspace", "srgb")]  # RgbColorSpace, color.rs:6-11
            r = self.push(abi.NODE_RGB, (cs,), tuple(n["value"]))
        elif ty == "spectral_uplift":
            r = self.push(abi.NODE_SPECTRAL_UPLIFT, (self.emit(n["rgb"]),))
        elif ty == "texcoords":
            r = self.push(abi.NODE_TEXCOORDS)
        elif ty == "image":
            uv = opt("uv")
            cs = n["image"]["colorspace"]
            assert cs in ("srgb", "none"), cs
            r = self.push(abi.NODE_IMAGE, (self.image(n["image"]), uv, 1 if cs == "srgb" else 0))
        elif ty == "mapping":
            v, loc, sc = self.emit(n["vector"]), self.emit(n["location"]), self.emit(n["scale"])
            r = self.push(abi.NODE_MAPPING, (v, loc, sc, {"point": abi.MAPPING_POINT, "texture": abi.MAPPING_TEXTURE}[n["mapping"]]))
        elif ty == "checkerboard":
            v = opt("vector")
            sc, c1, c2 = self.emit(n["scale"]), self.emit(n["color1"]), self.emit(n["color2"])
            r = self.push(abi.NODE_CHECKERBOARD, (v, sc, c1, c2))
        elif ty == "normal_map":
            assert n["space"] == "tangent"
            nn, st = self.emit(n["normal"]), self.emit(n["strength"])
            r = self.push(abi.NODE_NORMAL_MAP, (nn, st))
        elif ty == "separate_color":
            r = self.push(abi.NODE_SEPARATE_COLOR, (self.emit(n["color"]),))
        elif ty == "extract":
            field_ = {"Red": abi.FIELD_RED, "Green": abi.FIELD_GREEN, "Blue": abi.FIELD_BLUE, "uv": abi.FIELD_UV, "UV": abi.FIELD_UV}[n["field"]]
            r = self.push(abi.NODE_EXTRACT, (self.emit(n["node"]), field_))
        else:
            raise NotImplementedError(f"shader node '{ty}'")
        self.memo[nid] = r
        return r


def _fold_material(shader: dict, bufs: "_Buffers" = None, images: list = None, image_index: dict = None) -> MaterialData:
    nodes = shader["nodes"]
    graph = _Graph(nodes, bufs, images if images is not None else [], image_index if image_index is not None else {})

    def const(ref):
        """Evaluate a constant node the way svm/eval.rs does; returns (values[list], alpha)."""
        n = nodes[ref["id"]]
        ty = n["type"]
        if ty == "float":
            return [n["value"]], 1.0
        if ty == "float3":
            return list(n["value"]), 1.0
        if ty == "rgb":
            if {"srgb": 0, "aces": 1}[n.get("colorspace", "srgb")] and current[0] in _CS_BIT:
                cs_flags[0] |= _CS_BIT[current[0]]  # the constant stays in ACEScg; the render's ColorPipeline converts it
            return list(n["value"]), 1.0  # RgbTex: rgb.extend(1.0), svm/eval.rs:125-135
        if ty == "spectral_uplift":
            return const(n["rgb"])
        raise NotImplementedError(f"shader node '{ty}' (only constant inputs are supported)")

    current = [None]  # name of the input being read (abi.INPUT_NAMES)
    cs_flags = [0]

    def f(ref, default=0.0):  # eval_float_auto_convert
        if not graph.is_const(ref):
            graph.inputs[current[0]] = graph.emit(ref)
            return default
        return float(const(ref)[0][0])

    def c3(ref, default=(0.0, 0.0, 0.0)):
        if not graph.is_const(ref):
            graph.inputs[current[0]] = graph.emit(ref)
            return default
        v = const(ref)[0]
        return tuple(v[:3]) if len(v) >= 3 else (v[0], 0.0, 0.0)

    def alpha(ref):
        return const(ref)[1] if graph.is_const(ref) else 1.0

    def rd(name, fn, ref, *a):
        current[0] = name
        return fn(ref, *a)

    out = nodes[shader["output"]["id"]]
    assert out["type"] == "output"
    n = nodes[out["node"]["id"]]
    ty = n["type"]
    m = MaterialData()
    # the library's loader initialises every input it does not fold: same defaults here
    m.base_color, m.base_alpha, m.metallic, m.roughness, m.ior, m.specular_ior_level = (0.0, 0.0, 0.0), 1.0, 0.0, 0.0, 1.0, 0.5
    m.specular_tint, m.transmission_weight, m.coat_weight, m.coat_roughness, m.coat_ior = (1.0, 1.0, 1.0), 0.0, 0.0, 0.0, 0.0
    m.coat_tint, m.emission_color, m.emission_strength, m.normal = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), 0.0, (0.0, 0.0, 0.0)
    if ty == "principled":
        m.kind = MAT_PRINCIPLED
        m.base_color = rd("base_color", c3, n["base_color"])
        m.base_alpha = alpha(n["base_color"])
        m.metallic = rd("metallic", f, n["metallic"])
        m.roughness = rd("roughness", f, n["roughness"])
        m.ior = rd("ior", f, n["ior"], 1.0)
        m.specular_ior_level = rd("specular_ior_level", f, n["specular_ior_level"], 0.5)
        m.specular_tint = rd("specular_tint", c3, n["specular_tint"], (1.0, 1.0, 1.0))
        m.transmission_weight = rd("transmission_weight", f, n["transmission_weight"])
        m.coat_weight = rd("coat_weight", f, n["coat_weight"])
        m.coat_roughness = rd("coat_roughness", f, n["coat_roughness"])
        m.coat_ior = rd("coat_ior", f, n["coat_ior"])
        m.coat_tint = rd("coat_tint", c3, n["coat_tint"], (1.0, 1.0, 1.0))
        m.emission_color = rd("emission_color", c3, n["emission_color"])
        m.emission_strength = rd("emission_strength", f, n["emission_strength"])
        m.normal = rd("normal", c3, n["normal"])
    elif ty == "diffuse":
        m.kind = MAT_DIFFUSE
        m.base_color = rd("base_color", c3, n["color"])
        m.base_alpha = alpha(n["color"])
    elif ty == "glass":
        m.kind = MAT_GLASS
        m.base_color = rd("base_color", c3, n["color"])
        m.ior = rd("ior", f, n["ior"], 1.0)
        m.roughness = rd("roughness", f, n["roughness"])
    elif ty == "emission":
        m.kind = MAT_EMISSION
        m.emission_color = rd("emission_color", c3, n["color"])
        m.emission_strength = rd("emission_strength", f, n["strength"])
    else:
        raise NotImplementedError(f"surface shader '{ty}'")
    if graph.out:
        m.graph = abi.GraphData(graph.out, graph.inputs)
    m.colorspaces = cs_flags[0]
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
    images, image_index = [], {}
    materials = [_fold_material(scene["materials"][m]["shader"], bufs, images, image_index) for m in mat_ids]
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
    return SceneData(meshes, instances, materials, camera, None, inst_ids, mat_ids, images)


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
