"""The `aov` integrator (akari_integrator/src/aov.rs) on the GPU against the oracle, film accumulators bit for bit."""
import json
import os

import numpy as np
import pytest

from akari_render_amd import abi, capi
from oracle import pyoracle, scene_json
from tests.helpers import cbox_variant, grid_scene, n_bit_diff, textured_room

pytestmark = pytest.mark.gpu


def both(ctx, sd, cfg):
    scene = capi.Scene(ctx, sd)
    film = capi.Film(ctx, sd.camera.width, sd.camera.height)
    st = capi.aov_render(ctx, scene, cfg, film)
    o, n_rays = pyoracle.OracleScene(sd).aov_render(cfg)
    g = film.read()
    assert st["n_samples"] == n_rays == sd.camera.width * sd.camera.height * cfg.spp
    assert n_bit_diff(g, o) == 0, f"aov {abi.AOV_NAMES[cfg.aov]} remap={cfg.remap}: {n_bit_diff(g, o)} floats differ"
    return g


def table(root):
    return np.fromfile(os.path.join(root, "tests", "golden", "ggx_dielectric_s.f32"), dtype=np.float32)


@pytest.mark.parametrize("aov", range(6), ids=abi.AOV_NAMES)
@pytest.mark.parametrize("scene_name", ["cbox", "glass_coat", "kinds", "grid_normals", "textured"])
def test_aov_parity(ctx, cbox_path, root, scene_name, aov):
    if scene_name == "cbox":
        sd = scene_json.load_scene(cbox_path, 56, 40)
    elif scene_name in ("glass_coat", "kinds"):
        sd = cbox_variant(scene_json.load_scene(cbox_path, 56, 40), scene_name)
    elif scene_name == "grid_normals":
        sd = grid_scene(n=10, width=56, height=40, with_normals=True)
    else:
        sd = textured_room(56, 40, alpha_cutout=True)
    sd.ggx_table = table(root)
    cfg = abi.AovConfig.default()
    cfg.spp, cfg.aov, cfg.remap = 5, aov, 1 if aov % 2 == 0 else 0
    g = both(ctx, sd, cfg)
    n = 56 * 40
    assert np.all(g[6 * n :] == 5.0)
    if aov in (abi.AOV_NS, abi.AOV_NG) and cfg.remap:
        assert g[: 3 * n].min() >= 0.0 and g[: 3 * n].max() <= 5.0 + 1e-4


def test_aov_sharded_and_through_render_task(ctx, cbox_path, tmp_path, monkeypatch):
    sd = scene_json.load_scene(cbox_path, 64, 64)
    scene = capi.Scene(ctx, sd)
    cfg = abi.AovConfig.default()
    cfg.spp, cfg.aov = 4, abi.AOV_ALBEDO
    full = capi.Film(ctx, 64, 64)
    capi.aov_render(ctx, scene, cfg, full)
    total = np.zeros(7 * 64 * 64, dtype=np.float32)
    for r in range(3):
        c = abi.AovConfig.default()
        c.spp, c.aov, c.shard_rank, c.shard_count = 4, abi.AOV_ALBEDO, r, 3
        f = capi.Film(ctx, 64, 64)
        capi.aov_render(ctx, scene, c, f)
        total += f.read()
    assert n_bit_diff(total, full.read()) == 0
    # the method-file route: {"method": {"type": "aov", ...}} through akr_render_task writes the resolved image
    monkeypatch.chdir(tmp_path)
    method = {"method": {"type": "aov", "spp": 4, "aov": "albedo", "remap": False}, "film": {"out": "albedo.exr", "filter": {"type": "gaussian", "radius": 1.5}}}
    capi.render_task(ctx, scene, json.dumps(method))
    assert os.path.getsize(tmp_path / "albedo.exr") > 64 * 64 * 12


@pytest.mark.parametrize("color", [abi.COLOR_REPR_ACESCG, abi.COLOR_RGB_ACESCG, abi.COLOR_RGB_ACESCG | abi.COLOR_REPR_ACESCG])
@pytest.mark.parametrize("aov", [abi.AOV_NS, abi.AOV_ALBEDO, abi.AOV_ROUGHNESS], ids=["ns", "albedo", "roughness"])
def test_aov_in_a_non_default_colour_pipeline(ctx, cbox_path, root, aov, color):
    """akr_aov_config.color: the materials are folded for the pipeline and every value -- normals included -- goes through the
    film's conversion to sRGB primaries like a colour (aov.rs:98-124, film.rs:196-229)."""
    sd = textured_room(40, 32) if aov == abi.AOV_ALBEDO else scene_json.load_scene(cbox_path, 40, 32)
    sd.ggx_table = table(root)
    cfg = abi.AovConfig.default()
    cfg.spp, cfg.aov, cfg.remap, cfg.color = 4, aov, 1, color
    g = both(ctx, sd, cfg)
    cfg.color = 0
    g0 = both(ctx, sd, cfg)
    if aov != abi.AOV_ALBEDO and not (color & abi.COLOR_REPR_ACESCG):
        assert n_bit_diff(g, g0) == 0      # no colour constant involved and the film conversion is the identity
    else:
        assert n_bit_diff(g, g0) > 0
