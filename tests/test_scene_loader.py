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
    # transforms go through sin/cos of the host libm in both readers: allow 1 ulp on the camera matrix
    _same(sc.to_scene_data(), scene_json.load_scene(cbox_path), cam_tol=2e-7)
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
