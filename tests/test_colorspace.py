"""The ColorPipeline (crates/akari_render/src/color.rs:663-676): Rgb constants declared in sRGB or ACEScg, converted to
`rgb_colorspace` at the Rgb node and to the space of `color_repr` at the spectral_uplift node (svm/texture/mod.rs:9-43),
shading in that space, every sample converted back to sRGB primaries for the film (film.rs:218, color.rs:262-275)."""
import numpy as np
import pytest

from akari_render_amd import abi, capi
from oracle import pyoracle, scene_json
from tests.helpers import box_scene, make_config, n_bit_diff, rel_rmse, resolve_np, textured_room

S2A = np.array([[0.612494199, 0.338737252, 0.048855526], [0.070594252, 0.917671484, 0.011704306], [0.020727335, 0.106882232, 0.872338062]], dtype=np.float32)
A2S = np.array([[1.707062673, -0.619959540, -0.087259850], [-0.130976829, 1.139032275, -0.007956297], [-0.024510601, -0.124810932, 1.149395971]], dtype=np.float32)


def mat_vec(m, v):  # (c0 x + c1 y) + c2 z in f32, the AKR-F32 matrix product
    v = np.asarray(v, dtype=np.float32)
    return (m[:, 0] * v[0] + m[:, 1] * v[1]) + m[:, 2] * v[2]


def test_cat_matrices_are_inverse_and_keep_white():
    assert np.allclose(A2S.astype(np.float64) @ S2A.astype(np.float64), np.eye(3), atol=1e-4)  # the reference's constants: inverse to 4e-5
    assert np.allclose(S2A.sum(axis=1), 1.0, atol=2e-4) and np.allclose(A2S.sum(axis=1), 1.0, atol=2e-4)  # D65 white -> white


@pytest.mark.parametrize("color", [0, abi.COLOR_RGB_ACESCG, abi.COLOR_REPR_ACESCG, abi.COLOR_RGB_ACESCG | abi.COLOR_REPR_ACESCG])
def test_constant_inputs_through_the_pipeline(cbox_path, color):
    """Oracle: a constant tagged ACEScg / sRGB under each of the four pipelines = the two matrix steps by hand."""
    sd = scene_json.load_scene(cbox_path, 16, 16)
    m = sd.materials[0]
    m.base_color, m.emission_color, m.emission_strength = (0.8, 0.3, 0.1), (2.0, 1.0, 0.5), 1.5
    m.colorspaces = abi.MAT_CS_BASE_COLOR  # base colour declared in ACEScg, emission in sRGB
    osc = pyoracle.OracleScene(sd)
    got = osc.material_inputs(0, [[0.3, 0.7]], color)[0]
    rgb_aces, repr_aces = bool(color & abi.COLOR_RGB_ACESCG), bool(color & abi.COLOR_REPR_ACESCG)

    def pipeline(v, tag_aces):
        v = np.asarray(v, dtype=np.float32)
        if tag_aces != rgb_aces:
            v = mat_vec(S2A if rgb_aces else A2S, v)
        if rgb_aces != repr_aces:
            v = mat_vec(S2A if repr_aces else A2S, v)
        return v

    assert np.array_equal(got[1:4], pipeline((0.8, 0.3, 0.1), True))
    assert np.array_equal(got[19:22], pipeline((2.0, 1.0, 0.5), False))
    assert got.view(np.uint32)[0] == abi.MAT_PRINCIPLED  # the flags do not leak into the kind


def test_host_fold_matches_the_oracle_for_tagged_constants(hip_lib, cbox_path):
    """The product's host-side material compiler (default pipeline) against the oracle: an ACEScg-tagged constant is
    converted to sRGB, directly and through a graph (Rgb node with colour space, spectral_uplift)."""
    sd = textured_room(16, 16)
    sd.materials[3].colorspaces = abi.MAT_CS_BASE_COLOR | abi.MAT_CS_SPECULAR_TINT      # plain constant material
    g = sd.materials[0].graph                                                              # checkerboard floor: its Rgb nodes in ACEScg
    for nd in g.nodes:
        if nd.op == abi.NODE_RGB:
            nd.args = (1,) + tuple(nd.args[1:])
    sc = capi.Scene(None, sd)
    osc = pyoracle.OracleScene(sd)
    uv = np.random.default_rng(1).random((64, 2), dtype=np.float32)
    for mi in (0, 3):
        a = capi.probe_material_inputs(None, sc, mi, uv)
        b = osc.material_inputs(mi, uv, 0)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), mi
    # and the tag matters
    sd2 = textured_room(16, 16)
    assert not np.array_equal(pyoracle.OracleScene(sd2).material_inputs(3, uv[:1], 0), osc.material_inputs(3, uv[:1], 0))


def test_tagged_constant_equals_converted_constant(cbox_path):
    """Default pipeline: base colour (r, g, b) declared in ACEScg renders exactly like aces_to_srgb(r, g, b) declared in sRGB."""
    a = scene_json.load_scene(cbox_path, 24, 24)
    b = scene_json.load_scene(cbox_path, 24, 24)
    i = a.material_names.index("leftWall_001")
    col = a.materials[i].base_color
    a.materials[i].colorspaces = abi.MAT_CS_BASE_COLOR
    b.materials[i].base_color = tuple(float(x) for x in mat_vec(A2S, col))
    cfg = make_config(spp=4, spp_per_pass=4)
    fa, _ = pyoracle.OracleScene(a).render(cfg)
    fb, _ = pyoracle.OracleScene(b).render(cfg)
    assert n_bit_diff(fa, fb) == 0
    f0, _ = pyoracle.OracleScene(scene_json.load_scene(cbox_path, 24, 24)).render(cfg)
    assert n_bit_diff(fa, f0) > 0


def test_grey_furnace_is_the_same_in_every_pipeline():
    """Grey materials and a white emitter: changing the working space changes nothing but rounding (white is preserved by
    the CAT matrices to 1e-4); a saturated colour is NOT invariant (products of colours depend on the primaries)."""
    sd = box_scene(albedo=0.5, emission=1.0, width=12, height=12)
    cfg = make_config(spp=32, max_depth=8)
    ref = resolve_np(pyoracle.OracleScene(sd).render(cfg)[0], 12, 12)
    for color in (abi.COLOR_RGB_ACESCG | abi.COLOR_REPR_ACESCG, abi.COLOR_REPR_ACESCG):
        c2 = cfg.copy()
        c2.color = color
        img = resolve_np(pyoracle.OracleScene(sd).render(c2)[0], 12, 12)
        assert rel_rmse(img, ref) < 1e-3
    sd.materials[0].base_color = (0.9, 0.2, 0.1)
    ref = resolve_np(pyoracle.OracleScene(sd).render(cfg)[0], 12, 12)
    c2 = cfg.copy()
    c2.color = abi.COLOR_RGB_ACESCG | abi.COLOR_REPR_ACESCG
    img = resolve_np(pyoracle.OracleScene(sd).render(c2)[0], 12, 12)
    assert rel_rmse(img, ref) > 1e-2


def test_scene_json_readers_keep_the_colour_space(hip_lib, tmp_path, root):
    """An Rgb node with "colorspace": "aces" in scene.json: both readers (C++ and the oracle's python one) tag the constant."""
    import json
    import os
    import shutil

    src = os.path.join(root, "scenes", "cbox")
    dst = tmp_path / "cbox"
    shutil.copytree(src, dst)
    j = json.load(open(dst / "scene.json"))
    n_changed = 0
    for mat in j["materials"].values():
        for node in mat["shader"]["nodes"].values():
            if node.get("type") == "rgb" and n_changed < 3:
                node["colorspace"] = "aces"
                n_changed += 1
    assert n_changed == 3
    json.dump(j, open(dst / "scene.json", "w"))
    a = capi.Scene(None, str(dst / "scene.json"), 16, 16).to_scene_data()
    b = scene_json.load_scene(str(dst / "scene.json"), 16, 16)
    fa = [m.colorspaces for m in a.materials]
    fb = [m.colorspaces for m in b.materials]
    assert fa == fb and sum(1 for f in fa if f) >= 1


@pytest.mark.parametrize("color", [0, 1, 2, 3])
def test_host_material_tables_match_the_oracle_in_every_pipeline(hip_lib, color):
    """The tables a session compiles for its ColorPipeline (constants folded through the graph's Rgb / uplift nodes, pruned
    node lists, raw inputs), evaluated on the host by the code the kernels run, against the oracle: every material, bit for bit."""
    sd = textured_room(32, 24)
    for m in sd.materials:
        if m.graph is None:
            continue
        for k, nd in enumerate(m.graph.nodes):
            if nd.op == abi.NODE_RGB and k % 2 == 0:
                nd.args = (1,) + tuple(nd.args[1:])
    sd.materials[3].colorspaces = abi.MAT_CS_BASE_COLOR
    sc = capi.Scene(None, sd)
    osc = pyoracle.OracleScene(sd)
    uv = np.random.default_rng(1).random((64, 2), dtype=np.float32)
    for mi in range(len(sd.materials)):
        a = capi.probe_material_inputs_host(sc, mi, uv, color)
        b = osc.material_inputs(mi, uv, color)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (color, mi)
