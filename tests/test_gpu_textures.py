"""Shader graphs with texture-fed inputs on the GPU: the TEX instantiations of the kernels against
the oracle, film accumulators bit for bit."""
import os

import numpy as np
import pytest

from akari_render_amd import abi, capi
from oracle import pyoracle
from tests.helpers import make_config, n_bit_diff, textured_room
from tests.test_gpu_parity import assert_parity, render_both

pytestmark = pytest.mark.gpu


def with_table(sd, root):
    """Both sides read the committed ggx_dielectric_s table (the oracle has no GPU to compute it on)."""
    sd.ggx_table = np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)
    return sd


def test_node_evaluation_on_device_matches_oracle(ctx):
    sd = textured_room(alpha_cutout=True)
    scene = capi.Scene(ctx, sd)
    osc = pyoracle.OracleScene(sd)
    rng = np.random.default_rng(4)
    uv = np.concatenate([rng.uniform(-1.5, 3.0, size=(20000, 2)), [[0, 0], [1, 1], [0.5, 0.5], [1e9, -1e9], [np.nan, 0.3]]]).astype(np.float32)
    for m in range(len(sd.materials)):
        d = capi.probe_material_inputs(ctx, scene, m, uv)
        o = osc.material_inputs(m, uv)
        h = capi.probe_material_inputs(None, scene, m, uv)
        assert n_bit_diff(d, o) == 0, f"material {m}: device vs oracle"
        assert n_bit_diff(d, h) == 0, f"material {m}: device vs host build"


@pytest.mark.parametrize("variant", ["exhaustive", "bvh", "cutout", "cutout_bvh", "constant_light"])
def test_textured_room_parity(ctx, root, variant):
    sd = with_table(textured_room(48, 48, n_floor=8 if "bvh" in variant else 1, alpha_cutout="cutout" in variant, textured_light=variant != "constant_light"), root)
    scene = capi.Scene(ctx, sd)
    assert bool(scene.info().uses_bvh) == ("bvh" in variant)
    g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=16, spp_per_pass=8, max_depth=8))
    assert_parity(g, o, 48, 48, gst, ost)
    # the textures are visible: the floor is not uniform
    img = (g[: 3 * 48 * 48].reshape(48, 48, 3) / np.maximum(g[6 * 48 * 48 :].reshape(48, 48, 1), 1)).astype(np.float32)
    assert img[40:, :, :].std() > 0.01


@pytest.mark.parametrize("defer_on,mask", [(1, 1), (2, 1), (3, 1), (3, 3), (0, 0)])
def test_deferred_shading_in_the_textured_bvh_kernel_changes_no_bit(ctx, root, defer_on, mask):
    """The BVH kernels of scenes with textures put hits on conductor / texture-fed materials off to even iterations (pt_kernels.hip:
    DEFER, option defer_on): per lane only the iteration a vertex is shaded in changes -- film, sampler states and counters are
    the oracle's for every choice of what is deferred and every period."""
    sd = with_table(textured_room(48, 40, n_floor=8, alpha_cutout=True), root)
    with capi.options(defer_on=defer_on, defer_metal=mask):
        g, o, gst, ost, gs, os_ = render_both(ctx, sd, make_config(spp=12, spp_per_pass=4, max_depth=8), want_states=True)
    assert_parity(g, o, 48, 40, gst, ost)
    assert np.array_equal(gs, os_)


def test_textured_room_force_diffuse_keeps_textured_emission_and_alpha(ctx, root):
    sd = with_table(textured_room(40, 40, alpha_cutout=True), root)
    g, o, gst, ost, _, _ = render_both(ctx, sd, make_config(spp=8, spp_per_pass=8, max_depth=6, force_diffuse=1))
    assert_parity(g, o, 40, 40, gst, ost)


def test_textured_room_wavefront_schedule(ctx, root):
    sd = with_table(textured_room(40, 40, n_floor=8, alpha_cutout=True), root)
    cfg = make_config(spp=8, spp_per_pass=4, max_depth=8)
    with capi.options(wavefront=1):
        g, o, gst, ost, _, _ = render_both(ctx, sd, cfg)
    assert_parity(g, o, 40, 40, gst, ost)


def test_textured_scene_through_the_json_loader(ctx, root, tmp_path):
    from tests.test_textures import _scene_json_with_textures
    from oracle import scene_json

    rng = np.random.default_rng(3)
    from tests.helpers import make_png
    path = _scene_json_with_textures(tmp_path, make_png(rng.integers(0, 256, size=(5, 6, 3)), 2, 8), rng.random((4, 3, 3)).astype(np.float32))
    sd = with_table(scene_json.load_scene(path, 40, 30), root)
    scene = capi.Scene(ctx, sd)  # the library's own reader is compared with this one in tests/test_textures.py
    film = capi.Film(ctx, 40, 30)
    cfg = make_config(spp=16, spp_per_pass=16, max_depth=5)
    gst = capi.pt_render(ctx, scene, cfg, film)
    o, ost = pyoracle.OracleScene(sd).render(cfg)
    assert_parity(film.read(), o, 40, 30, gst, ost)


def test_every_byte_value_decodes_like_a_division_by_255(ctx):
    """dtex.h unorm8: the device decodes RGBA8 texels with q = b y, q + (b - 255 q) y (two fma) instead of an IEEE division;
    a 256 x 1 nearest-filtered image holding every byte value, looked up at every texel centre, against the oracle's b / 255."""
    sd = textured_room()
    ramp = np.zeros((1, 256, 4), dtype=np.uint8)
    ramp[0, :, 0] = np.arange(256)
    ramp[0, :, 1] = np.arange(256)[::-1]
    ramp[0, :, 2] = (np.arange(256) * 7 + 3) % 256
    ramp[0, :, 3] = 255
    sd.images[2] = abi.ImageData(ramp, abi.TEX_FILTER_NEAREST, abi.TEX_EXTEND)  # the emitter's image
    scene = capi.Scene(ctx, sd)
    osc = pyoracle.OracleScene(sd)
    uv = np.stack([(np.arange(256, dtype=np.float32) + 0.5) / 256.0, np.full(256, 0.5, np.float32)], axis=1).astype(np.float32)
    d = capi.probe_material_inputs(ctx, scene, 6, uv)
    o = osc.material_inputs(6, uv)
    assert n_bit_diff(d, o) == 0
    assert np.array_equal(d[:, 19], np.arange(256, dtype=np.float32) / np.float32(255.0))  # emission colour, red


def test_long_shader_graph(ctx, root):
    """A material whose five texture-fed inputs each bring their own mapping chain and image lookup (40 nodes after pruning: more
    than the 24 the first implementation's private value array allowed; the values live in at most 8 LDS slots whatever the length)."""
    sd = with_table(textured_room(40, 32), root)
    N = abi.NodeData
    nodes, inputs = [], {}
    for k, name in enumerate(("base_color", "roughness", "metallic", "specular_tint", "coat_weight")):
        b = len(nodes)
        nodes += [N(abi.NODE_TEXCOORDS), N(abi.NODE_EXTRACT, (b, abi.FIELD_UV)), N(abi.NODE_CONST, (), (0.1 * k, 0.05 * k, 0.0)),
                  N(abi.NODE_CONST, (), (1.0 + 0.5 * k, 2.0 - 0.2 * k, 1.0)), N(abi.NODE_MAPPING, (b + 1, b + 2, b + 3, k % 2)),
                  N(abi.NODE_IMAGE, (k % 2, b + 4, 1 if k == 0 else 0))]
        if name in ("base_color", "specular_tint"):
            nodes += [N(abi.NODE_SPECTRAL_UPLIFT, (b + 5,)), N(abi.NODE_SEPARATE_COLOR, (b + 6,))]
            inputs[name] = b + 7
        else:
            nodes += [N(abi.NODE_SEPARATE_COLOR, (b + 5,)), N(abi.NODE_EXTRACT, (b + 6, k % 3))]
            inputs[name] = b + 7
    assert len(nodes) == 40
    sd.materials[0].graph = abi.GraphData(nodes, inputs)
    scene = capi.Scene(ctx, sd)
    osc = pyoracle.OracleScene(sd)
    uv = np.random.default_rng(2).uniform(-1, 2, size=(4000, 2)).astype(np.float32)
    assert n_bit_diff(capi.probe_material_inputs(ctx, scene, 0, uv), osc.material_inputs(0, uv)) == 0
    film = capi.Film(ctx, 40, 32)
    cfg = make_config(spp=4, spp_per_pass=4, max_depth=6)
    capi.pt_render(ctx, scene, cfg, film)
    o, _ = osc.render(cfg)
    assert n_bit_diff(film.read(), o) == 0
