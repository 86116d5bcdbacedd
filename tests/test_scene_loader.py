"""The C++ reader of the reference's scene-graph JSON against the independent Python reader and the committed
golden flattening of scenes/cbox (tests/golden/cbox_flat.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np

from akari_render_amd import abi, capi
from oracle import scene_json


def _same(sd_a, sd_b, cam_tol=0.0):
    assert len(sd_a.meshes) == len(sd_b.meshes) and len(sd_a.instances) == len(sd_b.instances)
    for a, b in zip(sd_a.meshes, sd_b.meshes):
        assert np.array_equal(a.vertices, b.vertices) and np.array_equal(a.indices, b.indices)
        for f in ("uvs", "normals", "tangents", "material_slots"):
            x, y = getattr(a, f), getattr(b, f)
            assert (x is None) == (y is None), f
            if x is not None:
                assert np.array_equal(np.asarray(x).ravel(), np.asarray(y).ravel())
    for a, b in zip(sd_a.instances, sd_b.instances):
        assert a.mesh == b.mesh and list(a.materials) == list(b.materials)
        assert np.array_equal(a.transform, b.transform)
    for a, b in zip(sd_a.materials, sd_b.materials):
        assert bytes(a.to_struct()) == bytes(b.to_struct())
    assert np.max(np.abs(sd_a.camera.c2w - sd_b.camera.c2w)) <= cam_tol
    assert abs(sd_a.camera.fov - sd_b.camera.fov) <= cam_tol
    assert (sd_a.camera.width, sd_a.camera.height) == (sd_b.camera.width, sd_b.camera.height)


def test_cpp_loader_equals_python_loader(hip_lib, cbox_path):
    sc = capi.Scene(None, cbox_path)
    assert (sc.info().width, sc.info().height) == (1024, 1024)  # sensor_width/height of scenes/cbox
    # transforms go through correctly rounded f32 sin / cos in both readers: the camera matrix is the same bit for bit
    _same(sc.to_scene_data(), scene_json.load_scene(cbox_path), cam_tol=0.0)
    sc2 = capi.Scene(None, cbox_path, 1920, 1080)
    assert (sc2.info().width, sc2.info().height) == (1920, 1080)


def test_cbox_matches_golden_flattening(hip_lib, cbox_path, root):
    g = np.load(os.path.join(root, "tests", "golden", "cbox_flat.npz"))
    sd = capi.Scene(None, cbox_path).to_scene_data()
    assert len(sd.instances) == 8 and sd.n_triangles() == 36
    verts = np.concatenate([sd.meshes[i.mesh].vertices[sd.meshes[i.mesh].indices].reshape(-1, 3) for i in sd.instances])
    assert np.array_equal(verts, g["corner_vertices"])
    assert np.array_equal(np.stack([i.transform for i in sd.instances]), g["transforms"])
    assert np.array_equal(np.array([bytes(m.to_struct()) for m in sd.materials]), g["materials"])
    assert np.allclose(sd.camera.c2w, g["c2w"], atol=2e-7) and abs(sd.camera.fov - float(g["fov"])) < 1e-7
    # world-space facts from SURVEY.md Appendix B: eye (0,1,9) looking down -z; light quad at y = 1.98
    c2w = sd.camera.c2w.reshape(4, 4).T
    assert np.allclose(c2w[:3, 3], [0, 1, 9], atol=1e-6) and np.allclose(c2w[:3, :3], np.eye(3), atol=1e-6)
    light = sd.meshes[sd.instances[0].mesh]
    assert np.allclose(light.vertices[:, 1], 1.98, atol=1e-3)


def test_base64_and_relative_buffers(hip_lib, tmp_path, cbox_path):
    """Buffer::EmbeddedBase64 and a relative Buffer::Path resolve like the absolute Windows path of cbox."""
    import base64, json, shutil
    scene = json.load(open(cbox_path))
    raw = open(os.path.join(os.path.dirname(cbox_path), "Scene.bin"), "rb").read()
    a = dict(scene); a["buffers"] = {"Scene": {"type": "base64", "data": base64.b64encode(raw).decode(), "length": len(raw)}}
    (tmp_path / "a.json").write_text(json.dumps(a))
    b = dict(scene); b["buffers"] = {"Scene": {"type": "path", "path": "sub/S.bin", "length": len(raw)}}
    (tmp_path / "sub").mkdir(); (tmp_path / "sub" / "S.bin").write_bytes(raw)
    (tmp_path / "b.json").write_text(json.dumps(b))
    ref = capi.Scene(None, cbox_path).to_scene_data()
    _same(capi.Scene(None, str(tmp_path / "a.json")).to_scene_data(), ref)
    _same(capi.Scene(None, str(tmp_path / "b.json")).to_scene_data(), ref)
    _same(scene_json.load_scene(str(tmp_path / "a.json")), scene_json.load_scene(cbox_path))


def test_buffer_views_outside_their_buffer_are_refused(hip_lib, tmp_path, cbox_path):
    """A negative or huge offset / length must not wrap around in the bounds check (found by tools/fuzz/scene_json.cpp)."""
    import json
    import shutil

    import pytest

    src = os.path.dirname(cbox_path)
    dst = tmp_path / "cbox"
    shutil.copytree(src, dst)
    scene = json.load(open(dst / "scene.json"))
    key = sorted(scene["buffer_views"])[0]
    for field, value in (("offset", -1), ("offset", 99999999999), ("offset", 1e30), ("length", -4), ("length", 4294967296.0), ("offset", 18446744073709551615)):
        bad = json.loads(json.dumps(scene))
        bad["buffer_views"][key][field] = value
        (dst / "bad.json").write_text(json.dumps(bad))
        with pytest.raises(capi.AkariError) as e:
            capi.Scene(None, str(dst / "bad.json"), 16, 16)
        assert "buffer view" in str(e.value) or "json" in str(e.value).lower(), str(e.value)


def _random_shader(rng, images):
    """A random, well-typed surface shader in the reference's JSON node language: every input of the surface node fed by a constant,
    an image (through uplift / separate_color / extract as the type requires), a checkerboard of colours, a mapping chain, a normal map."""
    nodes, counter = {}, [0]

    def add(n):
        counter[0] += 1
        key = f"n{counter[0]:03d}"
        nodes[key] = n
        return {"id": key}

    def cfloat(lo=0.0, hi=1.0):
        return add({"type": "float", "value": float(np.float32(rng.uniform(lo, hi)))})

    def vec3(lo=0.0, hi=1.0):
        return add({"type": "float3", "value": [float(np.float32(x)) for x in rng.uniform(lo, hi, size=3)]})

    def vector():
        r = rng.random()
        if r < 0.4:
            return None
        tc = add({"type": "extract", "node": add({"type": "texcoords"}), "field": "UV"})
        if r < 0.6:
            return tc
        return add({"type": "mapping", "vector": tc, "mapping": str(rng.choice(["point", "texture"])), "location": vec3(-1, 1),
                    "rotation": add({"type": "float3", "value": [0, 0, 0]}), "scale": vec3(0.5, 3.0)})

    def image():
        n = {"type": "image", "image": images[int(rng.integers(0, len(images)))]}
        v = vector()
        if v is not None:
            n["uv"] = v
        return add(n)

    def colour(depth=0):
        r = rng.random()
        if r < 0.45 or depth > 1:
            return add({"type": "spectral_uplift", "rgb": add({"type": "rgb", "value": [float(np.float32(x)) for x in rng.random(3)],
                                                              "colorspace": str(rng.choice(["srgb", "srgb", "aces"]))})})
        if r < 0.75:
            return add({"type": "spectral_uplift", "rgb": image()})
        return add({"type": "checkerboard", "vector": vector(), "scale": cfloat(0.5, 6.0), "color1": colour(depth + 1), "color2": colour(depth + 1)})

    def scalar(lo=0.0, hi=1.0):
        if rng.random() < 0.6:
            return cfloat(lo, hi)
        return add({"type": "extract", "node": add({"type": "separate_color", "mode": "rgb", "color": image()}), "field": str(rng.choice(["Red", "Green", "Blue"]))})

    def normal():
        if rng.random() < 0.7:
            return add({"type": "float3", "value": [0, 0, 0]})
        return add({"type": "normal_map", "normal": image(), "strength": cfloat(0.2, 1.5), "space": "tangent"})

    kind = str(rng.choice(["principled"] * 5 + ["diffuse", "glass", "emission"]))
    if kind == "principled":
        zero3 = add({"type": "float3", "value": [0, 0, 0]})
        f0 = add({"type": "float", "value": 0.0})
        white = add({"type": "spectral_uplift", "rgb": add({"type": "rgb", "value": [1, 1, 1], "colorspace": "srgb"})})
        s = {"type": "principled", "base_color": colour(), "metallic": scalar(), "roughness": scalar(), "ior": scalar(1.0, 2.5), "alpha": add({"type": "float", "value": 1.0}),
             "normal": normal(), "subsurface_weight": f0, "subsurface_radius": zero3, "subsurface_scale": f0, "subsurface_ior": f0, "subsurface_anisotropy": f0,
             "specular_ior_level": scalar(), "specular_tint": colour(), "anisotropic": f0, "anisotropic_rotation": f0, "tangent": zero3,
             "transmission_weight": scalar(), "sheen_weight": f0, "sheen_tint": white, "coat_weight": scalar(), "coat_roughness": scalar(), "coat_ior": scalar(1.0, 2.0),
             "coat_tint": colour(), "coat_normal": zero3, "emission_color": colour(), "emission_strength": scalar(0.0, 4.0)}
    elif kind == "diffuse":
        s = {"type": "diffuse", "color": colour()}
    elif kind == "glass":
        s = {"type": "glass", "color": colour(), "ior": scalar(1.0, 2.0), "roughness": scalar()}
    else:
        s = {"type": "emission", "color": colour(), "strength": scalar(0.0, 5.0)}
    nodes["zz_bsdf"] = s
    nodes["zz_out"] = {"type": "output", "node": {"id": "zz_bsdf"}}
    return {"shader": {"kind": "surface", "nodes": nodes, "output": {"id": "zz_out"}}}


def test_random_typed_shader_graphs_read_alike_by_both_loaders(hip_lib, tmp_path):
    """200 random well-typed shaders (every node kind of svm/compiler.rs:116-337, every surface kind, srgb / aces constants, png and
    1-4 channel float images with every extension / interpolation) written as scene.json: the library's C++ reader and the oracle's
    Python reader must produce materials that EVALUATE identically at random uvs, the same images and the same light tables."""
    import json

    from oracle import pyoracle
    from tests.helpers import make_png
    from tests.test_textures import _scene_json_with_textures

    rng = np.random.default_rng(2024)
    compared = 0
    for case in range(200):
        d = tmp_path / f"c{case}"
        d.mkdir()
        fimg = rng.random((int(rng.integers(1, 6)), int(rng.integers(1, 6)), int(rng.integers(1, 5)))).astype(np.float32)
        png = make_png(rng.integers(0, 256, size=(5, 6, 3)), 2, 8)
        path = _scene_json_with_textures(d, png, fimg)
        scene = json.load(open(path))
        ext = lambda: str(rng.choice(["repeat", "clip", "mirror", "extend"]))  # noqa: E731
        interp = lambda: str(rng.choice(["linear", "cubic", "nearest"]))  # noqa: E731
        images = [{"data": {"id": "v_png"}, "format": "png", "colorspace": "srgb", "extension": ext(), "interpolation": interp(), "width": 6, "height": 5, "channels": 3},
                  {"data": {"id": "v_flt"}, "format": "float", "colorspace": "none", "extension": ext(), "interpolation": interp(),
                   "width": fimg.shape[1], "height": fimg.shape[0], "channels": fimg.shape[2]}]
        scene["materials"] = {"m_floor": _random_shader(rng, images), "m_wall": _random_shader(rng, images)}
        (d / "s.json").write_text(json.dumps(scene))
        try:
            sc = capi.Scene(None, str(d / "s.json"))
        except capi.AkariError as e:
            # both readers must agree on what they cannot take (e.g. more live values than the kernels' slots)
            try:
                ref = scene_json.load_scene(str(d / "s.json"))
                capi.Scene(None, ref)  # the python reader's result through the flat API: refused for the same reason
            except (capi.AkariError, NotImplementedError, AssertionError):
                continue
            raise AssertionError(f"case {case}: the C++ reader refused ({e}) what the python reader and the flat API accept")
        ref = scene_json.load_scene(str(d / "s.json"))
        got = sc.to_scene_data()
        assert len(got.images) == len(ref.images), case
        for a, b in zip(got.images, ref.images):
            assert a.texels.dtype == b.texels.dtype and np.array_equal(a.texels, b.texels) and (a.filter, a.address) == (b.filter, b.address), case
        osc = pyoracle.OracleScene(ref)
        uv = rng.uniform(-2, 3, size=(64, 2)).astype(np.float32)
        for m in range(2):
            for color in (0, 3):
                x, y = capi.probe_material_inputs_host(sc, m, uv, color), osc.material_inputs(m, uv, color)
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (case, m, color)
        assert sc.info().n_lights == osc.num_lights(), case
        for i in range(sc.info().n_lights):
            assert np.float32(sc.light(i)[1]).view(np.uint32) == np.float32(osc.light_info(i)[1]).view(np.uint32), (case, i)
        compared += 1
    assert compared >= 150, compared


def test_random_transforms_and_cameras_read_alike_by_both_loaders(hip_lib, tmp_path):
    """TRS (both coordinate systems) and matrix transforms on instances and the camera, random values: the two readers agree on every
    matrix transform bit for bit and on the TRS ones and the camera to the rounding of the trigonometric functions they use."""
    import json

    from tests.helpers import make_png
    from tests.test_textures import _scene_json_with_textures

    rng = np.random.default_rng(77)
    path = _scene_json_with_textures(tmp_path, make_png(rng.integers(0, 256, size=(5, 6, 3)), 2, 8), rng.random((2, 2, 3)).astype(np.float32))
    base = json.load(open(path))

    def transform():
        if rng.random() < 0.3:
            m = rng.normal(size=(4, 4)).astype(np.float32)
            m[3] = [0, 0, 0, 1]
            return {"type": "matrix", "data": [[float(x) for x in row] for row in m]}
        return {"type": "trs", "data": {"translation": [float(np.float32(x)) for x in rng.uniform(-3, 3, 3)], "rotation": [float(np.float32(x)) for x in rng.uniform(-3.2, 3.2, 3)],
                                        "scale": [float(np.float32(x)) for x in rng.uniform(0.2, 3, 3)], "coordinate_system": str(rng.choice(["Akari", "Blender"]))}}

    for case in range(150):
        scene = json.loads(json.dumps(base))
        for inst in scene["instances"].values():
            inst["transform"] = transform()
        scene["camera"]["data"]["transform"] = transform()
        scene["camera"]["data"]["fov"] = float(np.float32(rng.uniform(10, 120)))
        scene["camera"]["data"]["sensor_width"] = int(rng.integers(8, 64))
        scene["camera"]["data"]["sensor_height"] = int(rng.integers(8, 64))
        p = tmp_path / f"t{case}.json"
        p.write_text(json.dumps(scene))
        a = capi.Scene(None, str(p)).to_scene_data()
        b = scene_json.load_scene(str(p))
        for x, y in zip(a.instances, b.instances):  # (sin / cos come from two maths libraries: a few ulp; matrices are copied exactly)
            tx, ty = np.float32(x.transform), np.float32(y.transform)
            assert np.allclose(tx, ty, rtol=0, atol=4e-6 * max(1.0, float(np.abs(ty).max()))), (case, float(np.abs(tx - ty).max()))
            if scene["instances"][sorted(scene["instances"])[a.instances.index(x)]]["transform"]["type"] == "matrix":
                assert np.array_equal(tx.view(np.uint32), ty.view(np.uint32)), case
        assert np.allclose(a.camera.c2w, b.camera.c2w, rtol=0, atol=2e-6 * max(1.0, float(np.abs(b.camera.c2w).max()))), case
        assert abs(a.camera.fov - b.camera.fov) <= 1e-6 and (a.camera.width, a.camera.height) == (b.camera.width, b.camera.height), case
