"""HIP path against the oracle under every ColorPipeline (color.rs:663-676): constants and graph nodes in ACEScg / sRGB,
shading in the space of color_repr, film in sRGB. Films bit for bit."""
import json
import os
import shutil

import numpy as np
import pytest

from akari_render_amd import abi, capi
from oracle import pyoracle, scene_json
from tests.helpers import cbox_variant, make_config, n_bit_diff, rel_rmse, resolve_np, textured_room

pytestmark = pytest.mark.gpu

PIPELINES = {"srgb_srgb": 0, "rgb_aces": abi.COLOR_RGB_ACESCG, "repr_aces": abi.COLOR_REPR_ACESCG,
             "acescg": abi.COLOR_RGB_ACESCG | abi.COLOR_REPR_ACESCG}


def both(ctx, sd, cfg):
    scene = capi.Scene(ctx, sd)
    w, h = sd.camera.width, sd.camera.height
    film = capi.Film(ctx, w, h)
    gst = capi.pt_render(ctx, scene, cfg, film)
    o, ost = pyoracle.OracleScene(sd).render(cfg)
    return film.read(), o, gst, ost, scene


@pytest.mark.parametrize("name", list(PIPELINES))
def test_cbox_full_graph_in_every_pipeline(ctx, cbox_path, root, name):
    sd = cbox_variant(scene_json.load_scene(cbox_path, 96, 72), "glass_coat")
    sd.ggx_table = np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    names = sd.material_names
    sd.materials[names.index("leftWall_001")].colorspaces = abi.MAT_CS_BASE_COLOR          # some constants declared in ACEScg
    sd.materials[names.index("light_001")].colorspaces = abi.MAT_CS_EMISSION_COLOR
    sd.materials[names.index("floor_001")].colorspaces = abi.MAT_CS_COAT_TINT | abi.MAT_CS_SPECULAR_TINT
    cfg = make_config(spp=16, spp_per_pass=8, color=PIPELINES[name])
    g, o, gst, ost, _ = both(ctx, sd, cfg)
    assert n_bit_diff(g, o) == 0
    for k in ("n_samples", "n_closest", "n_shadow", "n_shaded"):
        assert gst[k] == ost[k]


def test_pipelines_differ_and_force_diffuse_too(ctx, cbox_path):
    sd = scene_json.load_scene(cbox_path, 64, 48)
    films = {}
    for name, color in PIPELINES.items():
        g, o, _, _, _ = both(ctx, sd, make_config(spp=8, spp_per_pass=8, color=color, force_diffuse=1))
        assert n_bit_diff(g, o) == 0   # force_diffuse: grey 0.8 in the working space, the light's emission still converted
        films[name] = resolve_np(g, 64, 48)
    # grey walls and one emitter: the image is LINEAR in the emitter's colour, so the working space changes only the rounding
    assert 0 < rel_rmse(films["acescg"], films["srgb_srgb"]) < 1e-3
    # coloured walls multiply colours component-wise, which depends on the primaries: the full graph differs visibly
    full = {}
    for name in ("srgb_srgb", "acescg"):
        g, o, _, _, _ = both(ctx, sd, make_config(spp=8, spp_per_pass=8, color=PIPELINES[name]))
        assert n_bit_diff(g, o) == 0
        full[name] = resolve_np(g, 64, 48)
    assert 1e-3 < rel_rmse(full["acescg"], full["srgb_srgb"]) < 0.5


@pytest.mark.parametrize("name", ["acescg", "rgb_aces"])
@pytest.mark.parametrize("bvh", [False, True])
def test_textured_graphs_in_acescg(ctx, root, name, bvh):
    """Rgb nodes declared in ACEScg, spectral_uplift nodes, image + checkerboard inputs, textured emitter: the graph is
    evaluated per hit on the device with the pipeline's conversions at its Rgb / uplift nodes."""
    sd = textured_room(64, 48, n_floor=8 if bvh else 1)
    sd.ggx_table = np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)  # both sides read the committed table
    for m in sd.materials:
        if m.graph is None:
            continue
        for k, nd in enumerate(m.graph.nodes):
            if nd.op == abi.NODE_RGB and k % 2 == 0:
                nd.args = (1,) + tuple(nd.args[1:])
    sd.materials[3].colorspaces = abi.MAT_CS_BASE_COLOR
    cfg = make_config(spp=8, spp_per_pass=4, max_depth=6, color=PIPELINES[name])
    g, o, gst, ost, scene = both(ctx, sd, cfg)
    assert scene.info().uses_bvh == (1 if bvh else 0)
    assert n_bit_diff(g, o) == 0
    # the same scene object renders in the default pipeline afterwards (the per-pipeline tables do not leak)
    cfg0 = make_config(spp=8, spp_per_pass=4, max_depth=6)
    film = capi.Film(ctx, 64, 48)
    capi.pt_render(ctx, scene, cfg0, film)
    o0, _ = pyoracle.OracleScene(sd).render(cfg0)
    assert n_bit_diff(film.read(), o0) == 0


def test_method_file_with_colour_pipeline(ctx, tmp_path, root):
    """scene.json with "colorspace": "aces" constants + a method file with a "color" block through akr_render_task."""
    src = os.path.join(root, "scenes", "cbox")
    dst = tmp_path / "cbox"
    shutil.copytree(src, dst)
    j = json.load(open(dst / "scene.json"))
    k = 0
    for mat in j["materials"].values():
        for node in mat["shader"]["nodes"].values():
            if node.get("type") == "rgb":
                if k % 3 == 0:
                    node["colorspace"] = "aces"
                k += 1
    json.dump(j, open(dst / "scene.json", "w"))
    method = {"method": {"type": "pt", "spp": 8, "spp_per_pass": 8, "max_depth": 5}, "sampler": {"type": "independent", "seed": 3},
              "color": {"color_repr": {"type": "rgb", "colorspace": "aces"}, "rgb_colorspace": "aces"}, "film": {"out": str(tmp_path / "out.exr")}}
    scene = capi.Scene(ctx, str(dst / "scene.json"), 80, 60)
    capi.render_task(ctx, scene, json.dumps(method))
    cfg, _ = capi.config_from_json(json.dumps(method))
    assert cfg.color == abi.COLOR_RGB_ACESCG | abi.COLOR_REPR_ACESCG
    film = capi.Film(ctx, 80, 60)
    capi.pt_render(ctx, scene, cfg, film)
    sd = scene_json.load_scene(str(dst / "scene.json"), 80, 60)
    o, _ = pyoracle.OracleScene(sd).render(cfg)
    # two independent readers of the same file, both with correctly rounded sin / cos in their transforms: the same bits
    assert n_bit_diff(film.read(), o) == 0
    assert os.path.exists(tmp_path / "out.exr")
