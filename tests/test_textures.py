"""Shader graphs with texture-fed inputs: PNG reader, sampler, node evaluation, loaders, light tables.
CPU only: the library's host build of device/dtex.h against the oracle (oracle/or_tex.h) and numpy."""
import base64
import json
import zlib

import numpy as np
import pytest

from akari_render_amd import abi, capi
from oracle import pyoracle, scene_json

from tests.helpers import make_png, n_bit_diff, textured_room


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def test_exp_pow_accuracy():
    L = pyoracle.lib()
    xs = np.concatenate([np.linspace(-87.0, 88.0, 4001), np.linspace(-1.0, 1.0, 2001), [-103.0, -100.0, 0.0, 88.7]]).astype(np.float32)
    got = np.array([L.or_kat_exp(float(x)) for x in xs], dtype=np.float32)
    want = np.exp(xs.astype(np.float64)).astype(np.float32)
    ok = np.isfinite(want) & (want > 1e-37)
    assert ulp_diff(got[ok], want[ok]).max() <= 2
    assert L.or_kat_exp(float("inf")) == float("inf") and L.or_kat_exp(-200.0) == 0.0 and np.isnan(L.or_kat_exp(float("nan")))
    # the sRGB decode range
    s = np.linspace(0.04046, 1.0, 5000).astype(np.float32)
    base = ((s + np.float32(0.055)) / np.float32(1.055)).astype(np.float32)
    got = np.array([L.or_kat_pow(float(b), 2.4) for b in base], dtype=np.float32)
    want = np.power(base.astype(np.float64), 2.4)
    assert np.max(np.abs(got - want) / want) < 1e-6
    assert L.or_kat_pow(0.0, 2.4) == 0.0


PNG_CASES = [
    ("rgba8", 6, 8, 4), ("rgb8", 2, 8, 3), ("grey8", 0, 8, 1), ("greya8", 4, 8, 2), ("rgb16", 2, 16, 3), ("rgba16", 6, 16, 4),
    ("grey16", 0, 16, 1), ("grey1", 0, 1, 1), ("grey2", 0, 2, 1), ("grey4", 0, 4, 1), ("pal8", 3, 8, 1), ("pal4", 3, 4, 1), ("pal2", 3, 2, 1),
]


@pytest.mark.parametrize("name,ctype,depth,ch", PNG_CASES)
def test_png_reader(name, ctype, depth, ch):
    rng = np.random.default_rng(hash(name) & 0xFFFF)
    h, w = 13, 11
    palette = trns = None
    if ctype == 3:
        n_pal = min(1 << depth, 23)
        palette = rng.integers(0, 256, size=(n_pal, 3), dtype=np.uint8)
        trns = bytes(rng.integers(0, 256, size=n_pal - 3, dtype=np.uint8))
        px = rng.integers(0, n_pal, size=(h, w, 1))
    else:
        px = rng.integers(0, 1 << depth, size=(h, w, ch))
    if name == "grey8":
        trns = bytes([0, int(px[3, 4, 0])])
    if name == "rgb16":
        trns = b"".join(int(v).to_bytes(2, "big") for v in px[2, 5])
    data = make_png(px, ctype, depth, palette=palette, trns=trns)
    got = capi.host_decode_png(data)
    ref = scene_json.decode_png(data)
    assert got.shape == (h, w, 4) and np.array_equal(got, ref)
    # against the definition
    to8 = (lambda v: v) if depth == 8 else (lambda v: (v + 128) // 257) if depth == 16 else (lambda v: v * 255 // ((1 << depth) - 1))
    want = np.full((h, w, 4), 255, dtype=np.int64)
    if ctype == 3:
        want[:, :, :3] = palette[px[:, :, 0]]
        al = np.full(256, 255)
        al[: len(trns)] = list(trns)
        want[:, :, 3] = al[px[:, :, 0]]
    elif ctype == 0:
        want[:, :, :3] = to8(px[:, :, :1])
        if trns is not None:
            want[:, :, 3] = np.where(px[:, :, 0] == int.from_bytes(trns, "big"), 0, 255)
    elif ctype == 4:
        want[:, :, :3] = to8(px[:, :, :1])
        want[:, :, 3] = to8(px[:, :, 1])
    elif ctype == 2:
        want[:, :, :3] = to8(px)
        if trns is not None:
            key = [int.from_bytes(trns[2 * k : 2 * k + 2], "big") for k in range(3)]
            want[:, :, 3] = np.where((px[:, :, 0] == key[0]) & (px[:, :, 1] == key[1]) & (px[:, :, 2] == key[2]), 0, 255)
    else:
        want[:, :, :] = to8(px)
    assert np.array_equal(got.astype(np.int64), want)


@pytest.mark.parametrize("size", [(1, 1), (2, 3), (5, 7), (8, 8), (9, 17), (33, 20), (3, 1), (1, 9)])
def test_png_reader_adam7(size):
    """Interlaced files (PNG specification 8.2): seven reduced images, each filtered on its own; every colour type and depth at sizes
    that leave some passes empty. Against the non-interlaced encoding of the same pixels and, for 8-bit files, Pillow's decoder."""
    import io

    from PIL import Image

    w, h = size
    rng = np.random.default_rng(w * 100 + h)
    for ct, depth, ch in ((2, 8, 3), (6, 8, 4), (0, 8, 1), (0, 1, 1), (0, 4, 1), (0, 16, 1), (4, 8, 2), (2, 16, 3), (6, 16, 4), (3, 2, 1), (3, 8, 1)):
        kw = {}
        if ct == 3:
            kw["palette"] = rng.integers(0, 256, size=(1 << depth, 3))
            px = rng.integers(0, 1 << depth, size=(h, w))
        else:
            px = rng.integers(0, 1 << depth, size=(h, w, ch))
        inter, plain = make_png(px, ct, depth, interlace=True, **kw), make_png(px, ct, depth, **kw)
        got = capi.host_decode_png(inter)
        assert np.array_equal(got, capi.host_decode_png(plain)), (ct, depth)
        if depth == 8 and ct != 3:
            assert np.array_equal(got, np.asarray(Image.open(io.BytesIO(inter)).convert("RGBA"))), (ct, depth)
    with pytest.raises(capi.AkariError):
        capi.host_decode_png(inter[:-30])


def test_png_reader_stored_and_fixed_blocks_and_errors():
    px = np.arange(5 * 7 * 3, dtype=np.int64).reshape(5, 7, 3) % 256
    for level in (0, 1, 9):  # stored blocks, fast (mostly fixed Huffman on tiny inputs), dynamic
        assert np.array_equal(capi.host_decode_png(make_png(px, 2, 8, level=level))[:, :, :3], px)
    big = (np.arange(200 * 300 * 4).reshape(200, 300, 4) * 7919 % 251).astype(np.int64)
    assert np.array_equal(capi.host_decode_png(make_png(big, 6, 8, level=9)), big)
    with pytest.raises(capi.AkariError):
        capi.host_decode_png(b"not a png at all")
    bad = bytearray(make_png(px, 2, 8))
    bad[60] ^= 0xFF
    with pytest.raises(capi.AkariError):
        capi.host_decode_png(bytes(bad))


def np_sample(img: abi.ImageData, uv):
    """float32 restatement of the sampler definition (dtex.h / or_tex.h) in numpy."""
    t = img.texels
    h, w = t.shape[:2]
    f32 = np.float32

    def wrap_coord(u):
        if img.address == abi.TEX_REPEAT:
            return f32(u - np.floor(u))
        if img.address == abi.TEX_MIRROR:
            t = f32(u - f32(f32(2.0) * np.floor(f32(u * f32(0.5)))))
            return f32(f32(2.0) - t) if t > 1.0 else t
        return u

    def wrap(i, n):
        if img.address == abi.TEX_REPEAT:
            i = i + n if i < 0 else (i - n if i >= n else i)
            return min(max(i, 0), n - 1), True
        if img.address in (abi.TEX_MIRROR, abi.TEX_EXTEND):
            return min(max(i, 0), n - 1), True
        return i, 0 <= i < n

    def fetch(i, j):
        i, oki = wrap(i, w)
        j, okj = wrap(j, h)
        if not (oki and okj):
            return np.zeros(4, dtype=f32)
        p = t[j, i]
        return (p.astype(f32) / f32(255.0)).astype(f32) if t.dtype == np.uint8 else p.astype(f32)

    out = []
    for u, v in np.asarray(uv, dtype=f32):
        x, y = f32(wrap_coord(u) * f32(w)), f32(wrap_coord(v) * f32(h))
        if img.filter == abi.TEX_FILTER_NEAREST:
            out.append(fetch(int(np.floor(x)), int(np.floor(y))))
            continue
        x, y = f32(x - f32(0.5)), f32(y - f32(0.5))
        i, j = int(np.floor(x)), int(np.floor(y))
        tx, ty = f32(x - f32(i)), f32(y - f32(j))
        lerp = lambda a, b, s: (a + (b - a).astype(f32) * s).astype(f32)  # noqa: E731
        r0, r1 = lerp(fetch(i, j), fetch(i + 1, j), tx), lerp(fetch(i, j + 1), fetch(i + 1, j + 1), tx)
        out.append(lerp(r0, r1, ty))
    return np.array(out, dtype=f32)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("filt", [abi.TEX_FILTER_NEAREST, abi.TEX_FILTER_LINEAR])
@pytest.mark.parametrize("address", [abi.TEX_REPEAT, abi.TEX_CLIP, abi.TEX_MIRROR, abi.TEX_EXTEND])
def test_sampler_matches_numpy_definition(dtype, filt, address):
    rng = np.random.default_rng(5)
    tex = rng.integers(0, 256, size=(5, 7, 4)).astype(np.uint8) if dtype == np.uint8 else rng.random((5, 7, 4)).astype(np.float32)
    img = abi.ImageData(tex, filt, address)
    uv = np.concatenate([rng.uniform(-2.5, 3.5, size=(400, 2)), [[0.0, 0.0], [1.0, 1.0], [0.5 / 7, 0.5 / 5], [-0.0, 2.0], [6.5 / 7, 4.5 / 5]]]).astype(np.float32)
    got = pyoracle.tex_sample(img, uv)
    want = np_sample(img, uv)
    assert n_bit_diff(got, want) == 0


def test_graph_evaluation_host_build_matches_oracle():
    sd = textured_room(alpha_cutout=True)
    osc = pyoracle.OracleScene(sd)
    sc = capi.Scene(None, sd)
    rng = np.random.default_rng(2)
    uv = np.concatenate([rng.uniform(-1.0, 3.0, size=(2000, 2)), [[0, 0], [1, 1], [0.5, 0.5]]]).astype(np.float32)
    for m in range(len(sd.materials)):
        a = capi.probe_material_inputs(None, sc, m, uv)
        b = osc.material_inputs(m, uv)
        assert n_bit_diff(a, b) == 0, f"material {m}"
    # the graph really drives the inputs: floor colour takes both checker colours, roughness varies, metallic is the folded constant
    fl = capi.probe_material_inputs(None, sc, 0, uv)
    assert len(np.unique(fl[:, 1])) == 2 and fl[:, 6].std() > 0.05 and np.all(fl[:, 5] == np.float32(0.25))
    # back wall: sRGB-decoded bytes in [0, 1]; left wall: normal = 2 c - 1 scaled by strength
    bw = capi.probe_material_inputs(None, sc, 1, uv)
    assert bw[:, 1:4].min() >= 0.0 and bw[:, 1:4].max() <= 1.0 and bw[:, 1].std() > 0.05
    lw = capi.probe_material_inputs(None, sc, 2, uv)
    assert np.all(np.abs(lw[:, 23:25]) <= 0.7 + 1e-6) and lw[:, 25].min() >= -1e-6


def test_light_tables_with_textured_emission_match_oracle():
    sd = textured_room()
    osc = pyoracle.OracleScene(sd)
    sc = capi.Scene(None, sd)
    info = sc.info()
    assert info.n_lights == osc.num_lights() == 1
    inst, power, pdf = sc.light(0)
    oi, op, opdf = osc.light_info(0)
    assert (inst, np.float32(power).view(np.uint32), np.float32(pdf).view(np.uint32)) == (oi, np.float32(op).view(np.uint32), np.float32(opdf).view(np.uint32))
    area_pdf = sc.array(9, np.float32)  # AKR_ARRAY_AREA_PDF
    assert area_pdf.size == 8 and abs(float(area_pdf.sum()) - 1.0) < 1e-5 and area_pdf.std() > 1e-3  # the texture modulates the per-triangle power


def _scene_json_with_textures(tmp_path, png_bytes, float_img):
    """A two-quad scene in the reference's scene-graph format whose materials use image / checkerboard / mapping /
    separate_color / normal_map nodes, with base64 and binary buffers."""
    verts = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1], [-1, 0.5, -1], [1, 0.5, -1], [1, 2, -1], [-1, 2, -1]], dtype=np.float32)
    idx = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
    uvs = np.array([[[0, 0], [1, 0], [1, 1]], [[0, 0], [1, 1], [0, 1]]], dtype=np.float32)
    blob = verts.tobytes() + idx.tobytes() + uvs.tobytes() + np.zeros(1, np.uint32).tobytes()
    off = [0, verts.nbytes, verts.nbytes + idx.nbytes, verts.nbytes + idx.nbytes + uvs.nbytes]

    def view(buf, o, ln):
        return {"buffer": {"id": buf}, "offset": o, "length": ln}

    def const_f(v):
        return {"type": "float", "value": v}

    def principled(nodes, **over):
        defaults = {"base_color": "c_base", "metallic": "f0", "roughness": "f_half", "ior": "f_ior", "alpha": "f1", "normal": "n0",
                    "subsurface_weight": "f0", "subsurface_radius": "n0", "subsurface_scale": "f0", "subsurface_ior": "f_ior",
                    "subsurface_anisotropy": "f0", "specular_ior_level": "f_half", "specular_tint": "c_white", "anisotropic": "f0",
                    "anisotropic_rotation": "f0", "tangent": "n0", "transmission_weight": "f0", "sheen_weight": "f0", "sheen_tint": "c_white",
                    "coat_weight": "f0", "coat_roughness": "f0", "coat_ior": "f_ior", "coat_tint": "c_white", "coat_normal": "n0",
                    "emission_color": "c_black", "emission_strength": "f0"}
        defaults.update(over)
        nodes.update({"f0": const_f(0.0), "f1": const_f(1.0), "f_half": const_f(0.5), "f_ior": const_f(1.45), "n0": {"type": "float3", "value": [0, 0, 0]},
                      "rgb_base": {"type": "rgb", "value": [0.8, 0.7, 0.6], "colorspace": "srgb"}, "c_base": {"type": "spectral_uplift", "rgb": {"id": "rgb_base"}},
                      "rgb_white": {"type": "rgb", "value": [1, 1, 1], "colorspace": "srgb"}, "c_white": {"type": "spectral_uplift", "rgb": {"id": "rgb_white"}},
                      "rgb_black": {"type": "rgb", "value": [0, 0, 0], "colorspace": "srgb"}, "c_black": {"type": "spectral_uplift", "rgb": {"id": "rgb_black"}}})
        p = {"type": "principled"}
        p.update({k: {"id": v} for k, v in defaults.items()})
        nodes["bsdf"] = p
        nodes["out"] = {"type": "output", "node": {"id": "bsdf"}}
        return {"shader": {"kind": "surface", "nodes": nodes, "output": {"id": "out"}}}

    png_image = {"data": {"id": "v_png"}, "format": "png", "colorspace": "srgb", "extension": "repeat", "interpolation": "cubic",
                 "width": 6, "height": 5, "channels": 3}
    flt_image = {"data": {"id": "v_flt"}, "format": "float", "colorspace": "none", "extension": "mirror", "interpolation": "nearest",
                 "width": float_img.shape[1], "height": float_img.shape[0], "channels": float_img.shape[2]}
    m_floor = principled({
        "tc": {"type": "texcoords"}, "tc_uv": {"type": "extract", "node": {"id": "tc"}, "field": "UV"},
        "loc": {"type": "float3", "value": [0.25, 0.5, 0]}, "rot": {"type": "float3", "value": [0, 0, 0]}, "scl": {"type": "float3", "value": [2, 3, 1]},
        "map": {"type": "mapping", "vector": {"id": "tc_uv"}, "mapping": "point", "location": {"id": "loc"}, "rotation": {"id": "rot"}, "scale": {"id": "scl"}},
        "img": {"type": "image", "image": png_image, "uv": {"id": "map"}}, "img_c": {"type": "spectral_uplift", "rgb": {"id": "img"}},
        "img2": {"type": "image", "image": flt_image}, "sep": {"type": "separate_color", "mode": "rgb", "color": {"id": "img2"}},
        "rough": {"type": "extract", "node": {"id": "sep"}, "field": "Blue"},
        "nm_s": const_f(0.5), "nm": {"type": "normal_map", "normal": {"id": "img2"}, "strength": {"id": "nm_s"}, "space": "tangent"},
    }, base_color="img_c", roughness="rough", normal="nm")
    m_wall = principled({
        "cs": const_f(4.0), "chk": {"type": "checkerboard", "vector": None, "scale": {"id": "cs"}, "color1": {"id": "c_base"}, "color2": {"id": "c_black"}},
        "img": {"type": "image", "image": png_image}, "img_c": {"type": "spectral_uplift", "rgb": {"id": "img"}},
        "es": const_f(3.0),
    }, base_color="chk", emission_color="img_c", emission_strength="es")
    trs = {"type": "trs", "data": {"translation": [0, 0, 0], "rotation": [0, 0, 0], "scale": [1, 1, 1], "coordinate_system": "Akari"}}
    scene = {
        "camera": {"type": "perspective", "data": {"transform": {"type": "trs", "data": {"translation": [0, 1, 3], "rotation": [0, 0, 0], "scale": [1, 1, 1], "coordinate_system": "Akari"}},
                                                  "fov": 50.0, "focal_distance": 1.0, "fstop": 2.8, "sensor_width": 40, "sensor_height": 30}},
        "instances": {"a_floor": {"geometry": {"id": "g_floor"}, "transform": trs, "materials": [{"id": "m_floor"}]},
                      "b_wall": {"geometry": {"id": "g_wall"}, "transform": trs, "materials": [{"id": "m_wall"}]}},
        "geometries": {"g_floor": {"type": "mesh", "vertices": {"id": "v_pos0"}, "indices": {"id": "v_idx"}, "uvs": {"id": "v_uv"}, "materials": {"id": "v_slot"}},
                       "g_wall": {"type": "mesh", "vertices": {"id": "v_pos1"}, "indices": {"id": "v_idx"}, "uvs": {"id": "v_uv"}, "materials": {"id": "v_slot"}}},
        "materials": {"m_floor": m_floor, "m_wall": m_wall},
        "lights": {}, "images": {},
        "buffers": {"b_geo": {"type": "base64", "data": base64.b64encode(blob).decode(), "length": len(blob)},
                    "b_png": {"type": "binary", "data": list(png_bytes), "length": len(png_bytes)},
                    "b_flt": {"type": "base64", "data": base64.b64encode(float_img.tobytes()).decode(), "length": float_img.nbytes}},
        "buffer_views": {"v_pos0": view("b_geo", 0, 48), "v_pos1": view("b_geo", 48, 48), "v_idx": view("b_geo", off[1], idx.nbytes),
                         "v_uv": view("b_geo", off[2], uvs.nbytes), "v_slot": view("b_geo", off[3], 4),
                         "v_png": view("b_png", 0, len(png_bytes)), "v_flt": view("b_flt", 0, float_img.nbytes)},
    }
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(scene))
    return str(path)


def test_scene_loader_reads_shader_graphs_and_images(tmp_path):
    rng = np.random.default_rng(9)
    px = rng.integers(0, 256, size=(5, 6, 3))
    png = make_png(px, 2, 8)
    fimg = rng.random((4, 3, 3)).astype(np.float32)
    path = _scene_json_with_textures(tmp_path, png, fimg)
    ref = scene_json.load_scene(path)
    sc = capi.Scene(None, path)
    got = sc.to_scene_data()
    assert len(got.images) == len(ref.images) == 2
    for a, b in zip(got.images, ref.images):
        assert a.texels.dtype == b.texels.dtype and np.array_equal(a.texels, b.texels) and (a.filter, a.address) == (b.filter, b.address)
    # png: flipped vertically, alpha 255, "cubic" -> linear; float: not flipped, alpha 1
    ipng = [im for im in got.images if im.texels.dtype == np.uint8][0]
    assert np.array_equal(ipng.texels[:, :, :3], px[::-1]) and np.all(ipng.texels[:, :, 3] == 255) and ipng.filter == abi.TEX_FILTER_LINEAR
    iflt = [im for im in got.images if im.texels.dtype == np.float32][0]
    assert np.array_equal(iflt.texels[:, :, :3], fimg) and np.all(iflt.texels[:, :, 3] == 1.0) and iflt.address == abi.TEX_MIRROR
    for mg, mr in zip(got.materials, ref.materials):
        assert (mg.graph is None) == (mr.graph is None)
        assert mg.graph.inputs == mr.graph.inputs
        assert len(mg.graph.nodes) == len(mr.graph.nodes)
        for x, y in zip(mg.graph.nodes, mr.graph.nodes):
            na = {abi.NODE_CONST: 0, abi.NODE_RGB: 0, abi.NODE_TEXCOORDS: 0, abi.NODE_IMAGE: 3, abi.NODE_MAPPING: 4, abi.NODE_CHECKERBOARD: 4,
                  abi.NODE_SPECTRAL_UPLIFT: 1, abi.NODE_SEPARATE_COLOR: 1, abi.NODE_EXTRACT: 2, abi.NODE_NORMAL_MAP: 2}[x.op]
            pad = lambda t: tuple(t) + (abi.NODE_NONE,) * (4 - len(t))  # noqa: E731
            assert x.op == y.op and pad(x.args)[:na] == pad(y.args)[:na]
            assert np.array_equal(np.float32(x.k), np.float32(y.k))
        for f in ("kind", "base_alpha", "metallic", "ior", "specular_ior_level", "emission_strength"):
            assert np.float32(getattr(mg, f)) == np.float32(getattr(mr, f)), f
    assert set(got.materials[0].graph.inputs) == {"base_color", "roughness", "normal"}
    assert set(got.materials[1].graph.inputs) == {"base_color", "emission_color"}  # the constant strength was folded
    assert got.materials[1].emission_strength == 3.0
    # both loaders drive identical evaluations
    osc = pyoracle.OracleScene(ref)
    uv = rng.uniform(-1, 2, size=(500, 2)).astype(np.float32)
    for m in range(2):
        assert n_bit_diff(capi.probe_material_inputs(None, sc, m, uv), osc.material_inputs(m, uv)) == 0
    # the wall became a light through its textured emission
    assert sc.info().n_lights == 1 and osc.num_lights() == 1
    assert np.float32(sc.light(0)[1]).view(np.uint32) == np.float32(osc.light_info(0)[1]).view(np.uint32)


def test_loader_rejects_unsupported_texture_inputs(tmp_path):
    rng = np.random.default_rng(1)
    path = _scene_json_with_textures(tmp_path, make_png(rng.integers(0, 256, size=(5, 6, 3)), 2, 8), rng.random((4, 3, 3)).astype(np.float32))
    scene = json.loads(open(path).read())
    nodes = scene["materials"]["m_wall"]["shader"]["nodes"]
    nodes["img"]["image"]["format"] = "webp"  # not one of load.rs:585-592 (float, png, jpeg, tiff, exr, dds)
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(scene))
    with pytest.raises(capi.AkariError) as e:
        capi.Scene(None, str(bad))
    assert "webp" in str(e.value) and e.value.code == -6  # AKR_ERR_UNSUPPORTED


def test_graph_validation_errors():
    sd = textured_room()
    sd.materials[0].graph.nodes[5] = abi.NodeData(abi.NODE_CHECKERBOARD, (abi.NODE_NONE, 7, 1, 3))  # forward reference
    with pytest.raises(capi.AkariError):
        capi.Scene(None, sd)
    sd = textured_room()
    sd.materials[1].graph.nodes[5] = abi.NodeData(abi.NODE_IMAGE, (17, 4, 1))  # image index out of range
    with pytest.raises(capi.AkariError):
        capi.Scene(None, sd)


# ---------------------------------------------------------------------------------------------- JPEG reader
def _jpeg_cases():
    cases = []
    for (h, w) in [(16, 16), (33, 47), (7, 5)]:
        for sub in (0, 1, 2):
            for prog in (False, True):
                cases.append((h, w, sub, prog, "RGB", {}))
    cases += [(40, 50, 2, False, "L", {}), (40, 50, 0, True, "L", {}), (70, 90, 2, False, "RGB", {"restart_marker_blocks": 3}),
              (70, 90, 1, True, "RGB", {"restart_marker_rows": 1}), (31, 29, 2, False, "RGB", {"quality": 30}),
              (31, 29, 2, True, "RGB", {"quality": 98, "optimize": True}), (24, 40, 0, False, "RGB", {"keep_rgb": True})]
    return cases


def _make_jpeg(h, w, sub, prog, mode, kw, seed=1):
    import io

    from PIL import Image

    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 128 + 90 * np.cos(x / 11.0 + y / 9.0), 100 + 60 * np.sin((x + y) / 13.0)], -1)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    kw = dict(kw)
    args = dict(quality=kw.pop("quality", 90), subsampling=sub, progressive=prog)
    args.update(kw)
    if mode == "L":
        args.pop("subsampling")
    buf = io.BytesIO()
    Image.fromarray(img if mode == "RGB" else img[:, :, 0], mode).save(buf, "JPEG", **args)
    data = buf.getvalue()
    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
    return data, ref


@pytest.mark.parametrize("case", _jpeg_cases(), ids=lambda c: f"{c[0]}x{c[1]}-s{c[2]}-{'prog' if c[3] else 'base'}-{c[4]}-{'-'.join(c[5])}")
def test_jpeg_reader_against_libjpeg(case):
    """Baseline and progressive files, 4:4:4 / 4:2:2 / 4:2:0, grey, restart intervals, RGB-coded: within 3 LSB of libjpeg
    (the IDCT and the chroma filter differ in their rounding; parity with the reference's jpeg-decoder crate is unpinned)."""
    pytest.importorskip("PIL")
    data, ref = _make_jpeg(*case)
    got = capi.host_decode_jpeg(data)
    assert got.shape == ref.shape and np.all(got[:, :, 3] == 255)
    d = np.abs(got.astype(int) - ref.astype(int))
    assert d.max() <= 3 and d.mean() < 0.5


def test_jpeg_reader_errors():
    pytest.importorskip("PIL")
    data, _ = _make_jpeg(33, 47, 2, False, "RGB", {})
    with pytest.raises(capi.AkariError):
        capi.host_decode_jpeg(b"\xff\xd8 nothing here")
    with pytest.raises(capi.AkariError):
        capi.host_decode_jpeg(data[:3] + data[40:])  # tables cut out
    for cut in (len(data) // 2, len(data) * 3 // 4, len(data) - 5):  # truncated files: an error or the part that is there, never a crash
        try:
            assert capi.host_decode_jpeg(data[:cut]).shape == (33, 47, 4)
        except capi.AkariError:
            pass


def test_scene_loader_reads_jpeg_textures(tmp_path):
    pytest.importorskip("PIL")
    rng = np.random.default_rng(9)
    data, ref = _make_jpeg(20, 24, 2, True, "RGB", {})
    path = _scene_json_with_textures(tmp_path, data, rng.random((4, 3, 3)).astype(np.float32))
    scene = json.loads(open(path).read())
    for m in ("m_floor", "m_wall"):
        scene["materials"][m]["shader"]["nodes"]["img"]["image"].update(format="jpeg", width=24, height=20)
    p2 = tmp_path / "jpeg_scene.json"
    p2.write_text(json.dumps(scene))
    got = capi.Scene(None, str(p2)).to_scene_data()
    pyl = scene_json.load_scene(str(p2))
    a = [im for im in got.images if im.texels.dtype == np.uint8][0].texels
    b = [im for im in pyl.images if im.texels.dtype == np.uint8][0].texels
    assert a.shape == b.shape == (20, 24, 4)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 3          # two decoders (this library's, libjpeg)
    assert np.abs(a.astype(int) - ref[::-1].astype(int)).max() <= 3  # flipped vertically like every encoded image


def test_alpha_textured_flag_is_exact():
    """MF_ALPHA_TEXTURED (bit 9 of DMaterial.flags) only where the alpha of a fed base colour can differ from 1: opaque images and
    constant checkerboards are provably opaque, an image with transparent texels or Zero addressing is not."""
    def flags(sd):
        mats = capi.Scene(None, sd).array(capi.ARRAY_MATERIALS, np.uint32).reshape(-1, 64)
        return [(int(f) >> 8) & 3 for f in mats[:, 1]]  # bit 0: MF_TEXTURED, bit 1: MF_ALPHA_TEXTURED

    opaque = flags(textured_room())
    assert opaque[:3] == [1, 1, 1] and opaque[6] == 1 and opaque[3:6] == [0, 0, 0]
    cut = flags(textured_room(alpha_cutout=True))
    assert cut[1] == 3            # the back wall samples the byte image whose alpha now has zeros
    assert cut[7] == 3            # the cut-out quad (Zero addressing as well)
    assert cut[0] == 1 and cut[2] == 1  # checkerboard of constants / normal map only
    sd = textured_room()
    sd.images[0].address = abi.TEX_CLIP  # opaque texels, but outside [0, 1]^2 the sampler returns alpha 0
    assert flags(sd)[1] == 3


# ---------------------------------------------------------------------------------------------- OpenEXR reader
@pytest.mark.parametrize("compression", [0, 1, 2, 3], ids=["none", "rle", "zips", "zip"])
@pytest.mark.parametrize("layout", ["rgb_half", "rgba_float", "y_half", "mixed"])
def test_exr_reader(compression, layout):
    from tests.helpers import make_exr

    rng = np.random.default_rng(compression * 7 + len(layout))
    h, w = 37, 21
    smooth = lambda: (np.linspace(0, 4, w)[None, :] * np.linspace(0.5, 2, h)[:, None] + rng.random((h, w)) * 0.01)  # noqa: E731
    if layout == "rgb_half":
        planes = {c: smooth().astype(np.float16) for c in "RGB"}
    elif layout == "rgba_float":
        planes = {c: smooth().astype(np.float32) for c in "RGBA"}
    elif layout == "y_half":
        planes = {"Y": smooth().astype(np.float16)}
    else:
        planes = {"R": smooth().astype(np.float32), "G": smooth().astype(np.float16), "B": (smooth() * 100).astype(np.uint32),
                  "Z": smooth().astype(np.float32)}
        planes["G"][3, 4] = np.float16(6e-6)  # a subnormal half
        planes["G"][5, 6] = np.float16(np.inf)
    data = make_exr(planes, compression, line_order=compression % 2)
    got = capi.host_decode_exr(data)
    ref = scene_json.decode_exr(data)
    assert got.shape == (h, w, 4) and n_bit_diff(got, ref) == 0
    want = np.zeros((h, w, 4), dtype=np.float32)
    want[:, :, 3] = 1.0
    for k, c in enumerate("RGBA"):
        if c in planes:
            want[:, :, k] = planes[c].astype(np.float32)
    if layout == "y_half":
        want[:, :, :3] = planes["Y"].astype(np.float32)[:, :, None]
    assert n_bit_diff(got, want) == 0


@pytest.mark.parametrize("compression", [0, 1, 3, 5], ids=["none", "rle", "zip", "pxr24"])
@pytest.mark.parametrize("tiles", [None, (16, 16), (32, 8), (64, 64)], ids=["scanlines", "tiles16", "tiles32x8", "one_tile"])
def test_exr_reader_tiled_and_pxr24(compression, tiles):
    """Round 6: tiled single-part files and PXR24, which the exr crate behind load.rs:586-600 reads. The fixture
    writer (tests/helpers.py make_exr) is written from the file-format description, independently of the reader."""
    from tests.helpers import make_exr, pxr24_round

    if tiles is None and compression != 5:
        pytest.skip("covered by test_exr_reader")
    rng = np.random.default_rng(compression + (tiles[0] if tiles else 0))
    h, w = 45, 52
    smooth = lambda: (np.linspace(0, 4, w)[None, :] * np.linspace(0.5, 2, h)[:, None] + rng.random((h, w)) * 0.01)  # noqa: E731
    planes = {"R": smooth().astype(np.float32), "G": smooth().astype(np.float16), "B": smooth().astype(np.float32), "A": (smooth() * 50).astype(np.uint32)}
    for mip in ((False, True) if tiles else (False,)):
        data = make_exr(planes, compression, tiles=tiles, mipmap=mip)
        got = capi.host_decode_exr(data)
        want = np.zeros((h, w, 4), dtype=np.float32)
        for k, c in enumerate("RGBA"):
            v = planes[c]
            want[:, :, k] = pxr24_round(v) if (compression == 5 and v.dtype == np.float32) else v.astype(np.float32)
        assert got.shape == (h, w, 4) and n_bit_diff(got, want) == 0


@pytest.mark.parametrize("layout", ["rgb_half_smooth", "rgba_mixed_noisy", "y_float", "constant", "wide_range_16bit"])
@pytest.mark.parametrize("tiles", [None, (16, 24)], ids=["scanlines", "tiles"])
def test_exr_reader_piz(layout, tiles):
    """PIZ (OpenEXR's wavelet + Huffman method, a common default of exporters; read by the exr crate behind load.rs:586-600): the decoder
    against an encoder written from the format's description (tests/helpers.py piz_compress). Lossless: the planes come back bit for bit.
    Covered: 14-bit and 16-bit wavelet variants, the run-length symbol, zero runs in the code-length table, odd block sizes, a block that
    is stored raw because PIZ does not shrink it."""
    from tests.helpers import make_exr

    rng = np.random.default_rng(len(layout) + (1 if tiles else 0))
    h, w = 71, 39  # three blocks of 32 scanlines, the last one 7 high
    smooth = lambda: (np.linspace(0, 4, w)[None, :] * np.linspace(0.5, 2, h)[:, None] + rng.random((h, w)) * 0.01)  # noqa: E731
    if layout == "rgb_half_smooth":
        planes = {c: smooth().astype(np.float16) for c in "RGB"}
    elif layout == "rgba_mixed_noisy":
        planes = {"R": rng.random((h, w)).astype(np.float32), "G": rng.random((h, w)).astype(np.float16), "B": rng.integers(0, 2 ** 32, size=(h, w), dtype=np.uint32),
                  "A": smooth().astype(np.float16)}
    elif layout == "y_float":
        planes = {"Y": smooth().astype(np.float32)}
    elif layout == "constant":
        planes = {c: np.full((h, w), 0.25, dtype=np.float16) for c in "RGB"}
        planes["R"][10:20, 5:9] = np.float16(0.0)
    else:  # more than 2^14 distinct 16-bit values in a block: the modular 16-bit wavelet
        planes = {"R": rng.integers(0, 65536, size=(h, w)).astype(np.uint16).view(np.float16), "G": np.arange(h * w, dtype=np.uint16).reshape(h, w).view(np.float16)}
        planes["R"] = np.where(np.isnan(planes["R"]), np.float16(1.0), planes["R"])
    data = make_exr(planes, 4, tiles=tiles)
    got = capi.host_decode_exr(data)
    want = np.zeros((h, w, 4), dtype=np.float32)
    want[:, :, 3] = 1.0
    for k, c in enumerate("RGBA"):
        if c in planes:
            want[:, :, k] = planes[c].astype(np.float32)
    if "Y" in planes:
        want[:, :, :3] = planes["Y"].astype(np.float32)[:, :, None]
    assert got.shape == (h, w, 4) and n_bit_diff(got, want) == 0
    # damaged streams are refused, not read past their end
    i = data.rindex(b"\0\0\0\0") if False else len(data) - 40
    for cut in (len(data) - 7, len(data) - 200):
        with pytest.raises(capi.AkariError):
            capi.host_decode_exr(data[:cut])


@pytest.mark.parametrize("variant", ["b44", "b44a", "b44a_tiles", "b44_plinear"])
def test_exr_reader_b44(variant):
    """B44 / B44A (OpenEXR's fixed-rate lossy method for HALF channels; the exr crate behind load.rs:586-600 reads both): 4 x 4 blocks of
    14 bytes -- first value, a shift, fifteen 6-bit running differences -- or 3 bytes for a block of one value (B44A); FLOAT / UINT channels
    stored as they are; pLinear channels through the exp(x / 8) table. The reader against the tests' own encoder AND reference decoder
    (tests/helpers.py b44_*), both written from the format's description: bit for bit what the reference decoder makes of the same bytes,
    and within the block's quantisation step of what was written."""
    from tests.helpers import b44_reference_decode, make_exr

    rng = np.random.default_rng(len(variant))
    h, w = 45, 30  # two blocks of 32 scanlines; neither size a multiple of 4
    smooth = (np.linspace(0.1, 3, w)[None, :] * np.linspace(0.5, 2, h)[:, None])
    planes = {"R": (smooth + rng.random((h, w)) * 0.02).astype(np.float16), "G": (rng.standard_normal((h, w)) * 40).astype(np.float16),
              "B": np.full((h, w), 0.375, np.float16), "A": rng.random((h, w)).astype(np.float32)}
    planes["B"][20:24, 8:12] = np.float16(-2.0)            # a flat block of another value, and blocks that straddle the step
    planes["G"][3, 5] = np.float16(np.inf)                  # non-finite values pack as the ordered representation's 0x8000
    p_linear = ("R",) if variant == "b44_plinear" else ()
    blobs = []
    data = make_exr(planes, 7 if variant.startswith("b44a") else 6, tiles=(16, 20) if variant.endswith("tiles") else None, p_linear=p_linear, blobs_out=blobs)
    got = capi.host_decode_exr(data)
    assert got.shape == (h, w, 4)
    # the tests' reference decode of the very bytes in the file
    hb = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float64)
    with np.errstate(all="ignore"):
        et = np.where(~np.isfinite(hb), 0.0, np.where(hb >= 8 * np.log(65504.0), 65504.0, np.exp(hb / 8.0)))
    exp_table = et.astype(np.float32).astype(np.float16).view(np.uint16)
    names = sorted(planes)
    want = np.zeros((h, w, 4), np.float32)
    assert all(b is not None for b, _ in blobs)  # (every block shrank: none stored raw)
    for blob, (x0, x1, y0, y1) in blobs:
        dec = b44_reference_decode(blob, [planes[n].dtype for n in names], (y1 - y0, x1 - x0), exp_table, [k for k, n in enumerate(names) if n in p_linear])
        for n, pl in zip(names, dec):
            want[y0:y1, x0:x1, "RGBA".index(n)] = pl.astype(np.float32)
    assert n_bit_diff(got, want) == 0
    # ... and it is the image that was written, to the method's precision
    assert np.array_equal(got[:, :, 3], planes["A"])                                   # FLOAT channels are not touched
    if variant == "b44a":
        assert np.array_equal(got[:, :, 2], planes["B"].astype(np.float32))            # flat blocks are exact
        assert len(data) < len(make_exr(planes, 6))                                    # ... and three bytes instead of fourteen
    r = planes["R"].astype(np.float32)
    assert np.max(np.abs(got[:, :, 0] - r) / r) < (0.25 if p_linear else 0.05)  # (a steep 4 x 4 block shares one step size: coarse for its small values)
    for cut in (len(data) - 5, len(data) - 300):
        with pytest.raises(capi.AkariError):
            capi.host_decode_exr(data[:cut])


def test_exr_reader_reads_the_writer_and_rejects_what_it_cannot_read(tmp_path):
    rgb = np.random.default_rng(0).random((9, 14, 3)).astype(np.float32)
    path = str(tmp_path / "out.exr")
    capi.image_write(path, rgb)  # the library's own uncompressed writer
    got = capi.host_decode_exr(open(path, "rb").read())
    assert np.array_equal(got[:, :, :3], rgb) and np.all(got[:, :, 3] == 1.0)
    from tests.helpers import make_exr

    data = bytearray(make_exr({"R": np.zeros((4, 4), np.float32)}, 2))
    i = data.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    data[i] = 8  # DWAA: refused by name (B44 / B44A are read since round 6)
    with pytest.raises(capi.AkariError) as e:
        capi.host_decode_exr(bytes(data))
    assert e.value.code == -6
    with pytest.raises(capi.AkariError):
        capi.host_decode_exr(b"v/1\x01 not an exr")


def test_scene_loader_reads_exr_textures(tmp_path):
    from tests.helpers import make_exr

    rng = np.random.default_rng(4)
    planes = {c: rng.random((6, 5)).astype(np.float16) for c in "RGB"}
    data = make_exr(planes, 3)
    path = _scene_json_with_textures(tmp_path, data, rng.random((4, 3, 3)).astype(np.float32))
    scene = json.loads(open(path).read())
    for m in ("m_floor", "m_wall"):
        scene["materials"][m]["shader"]["nodes"]["img"]["image"].update(format="exr", width=5, height=6, colorspace="none")
    p2 = tmp_path / "exr_scene.json"
    p2.write_text(json.dumps(scene))
    got = capi.Scene(None, str(p2)).to_scene_data()
    pyl = scene_json.load_scene(str(p2))
    a = [im.texels for im in got.images if im.texels.shape[:2] == (6, 5)][0]
    b = [im.texels for im in pyl.images if im.texels.shape[:2] == (6, 5)][0]
    assert a.dtype == np.float32 and n_bit_diff(a, b) == 0
    assert np.array_equal(a[::-1, :, 0], planes["R"].astype(np.float32))  # flipped vertically


def test_unorm8_reciprocal_form_is_a_correctly_rounded_division():
    """dtex.h unorm8 (device): with y = RN(1 / 255), q = RN(b y), r = b - 255 q (exact), RN(q + r y) == RN(b / 255) for every
    byte -- checked in exact rational arithmetic."""
    from fractions import Fraction

    def rn32(fr):
        f = np.float32(float(fr))
        cands = [f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))]
        return Fraction(float(min(cands, key=lambda c: (abs(Fraction(float(c)) - fr), int(np.float32(c).view(np.uint32)) & 1))))

    y = rn32(Fraction(1, 255))
    assert float(y) == float(np.float32(0.003921568859368563))
    for b in range(256):
        q = rn32(Fraction(b) * y)
        r = Fraction(b) - 255 * q
        assert rn32(r) == r
        assert float(rn32(q + r * y)) == float(np.float32(b) / np.float32(255.0)), b


def test_constant_alpha_next_to_an_opaque_texture_fed_base_colour():
    """The alpha of a node-fed base colour is the node's (principled.rs:15-21), never the description's constant: with an opaque
    image feeding base_color the folded record carries alpha 1 (and no per-candidate graph evaluation), whatever base_alpha says.
    Found by tools/soak.py -- every designed scene had left the constant at 1."""
    sd = textured_room()
    for m in sd.materials:
        m.base_alpha = 0.25
    mats = capi.Scene(None, sd).array(capi.ARRAY_MATERIALS, np.uint32).reshape(-1, 64)
    flags = [(int(f) >> 8) & 3 for f in mats[:, 1]]
    alpha = mats[:, 2].view(np.float32)
    assert flags[1] == 1 and alpha[1] == 1.0          # back wall: opaque byte image -> base colour
    assert flags[0] == 1 and alpha[0] == 1.0          # floor: checkerboard of two constants
    assert flags[3] == 0 and alpha[3] == np.float32(0.25)  # constant material: the constant counts
    osc = pyoracle.OracleScene(sd)
    uv = np.random.default_rng(0).random((16, 2), dtype=np.float32)
    assert np.all(osc.material_inputs(1, uv)[:, 4] == 1.0) and np.all(osc.material_inputs(3, uv)[:, 4] == np.float32(0.25))
