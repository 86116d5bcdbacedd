"""Every BASELINE.json configuration at ITS OWN resolution / triangle count / sample count, HIP path against the oracle,
bit for bit.

A whole C2..C5 frame is hours of CPU oracle time, but the path shards by pixel tiles and a pixel's sample stream does not
depend on which tiles are rendered with it (akr_pt_config.shard_* on the GPU side, or_pixel_owned in the oracle;
pt.rs:1075-1103: one thread per pixel, no cross-pixel state). So the GPU renders tile shard k of n of the real frame with
the real spp / depth / filter, the oracle renders the same shard, and the two 7 N-float films must be identical -- inside
the shard (all samples of all owned pixels) and outside it (zeros). The shard is a few tiles spread over the frame
(a tile belongs to shard morton(tx, ty) % n), so it mixes background, walls, boxes and light.

C4 (10 M triangles): the oracle's exhaustive loop is the definition of a hit; it runs here on one 8x8 tile at 1 spp. The
1024-spp shard uses the oracle-side BVH of oracle/or_accel.h, which is pinned to the exhaustive loop ray by ray on the CPU
(tests/test_oracle_accel.py) and, below, on a tile of this very scene.
"""
import numpy as np
import pytest

from akari_render_amd import capi, distributed
from oracle import pyoracle, scene_json
from tests.helpers import make_config, n_bit_diff, rel_rmse, resolve_np

pytestmark = pytest.mark.gpu

REL_RMSE_TOL = 1e-3  # BASELINE.md contract; the assertions below are stricter (0 differing floats)


def shard_parity(ctx, sd, cfg, osc=None, scene=None, min_owned=1):
    w, h = sd.camera.width, sd.camera.height
    scene = scene or capi.Scene(ctx, sd)
    film = capi.Film(ctx, w, h)
    gst = capi.pt_render(ctx, scene, cfg, film)
    g = film.read()
    del film
    osc = osc or pyoracle.OracleScene(sd)
    o, ost = osc.render(cfg)
    n = w * h
    owned = distributed.owned_pixel_mask(w, h, cfg.shard_rank, cfg.shard_count, cfg.tile_w, cfg.tile_h).ravel()
    assert owned.sum() >= min_owned
    assert np.array_equal(g[6 * n:] == cfg.spp, owned)           # every owned pixel got all its samples, nobody else any
    for k in ("n_samples", "n_closest", "n_shadow", "n_shaded"):
        assert gst[k] == ost[k], k
    assert gst["n_samples"] == int(owned.sum()) * cfg.spp
    nd = n_bit_diff(g, o)
    err = rel_rmse(resolve_np(g, w, h), resolve_np(o, w, h))
    assert err < REL_RMSE_TOL
    assert nd == 0, f"{nd} film floats differ (relRMSE {err:.3e})"
    return gst


def test_c2_1080p_force_diffuse_1024spp_shard(ctx, cbox_path):
    """configs[1]: scenes/cbox 1920x1080, 1024 spp, diffuse-only BSDF. 8 of the 2040 tiles (8192 pixels x 1024 spp)."""
    sd = scene_json.load_scene(cbox_path, 1920, 1080)
    cfg = distributed.shard_config(make_config(spp=1024, spp_per_pass=64, max_depth=12, rr_depth=5, force_diffuse=1), 7, 255, 32, 32)
    st = shard_parity(ctx, sd, cfg, min_owned=8 * 1024 - 512)
    assert st["n_launches"] == 1


def test_c3_1080p_full_graph_4096spp_shard(ctx, cbox_path):
    """configs[2]: scenes/cbox 1920x1080, 4096 spp, full Cycles-subset shader graph. 4 tiles x 4096 spp = 16.8 M paths."""
    sd = scene_json.load_scene(cbox_path, 1920, 1080)
    cfg = distributed.shard_config(make_config(spp=4096, spp_per_pass=64, max_depth=12, rr_depth=5), 100, 510, 32, 32)
    st = shard_parity(ctx, sd, cfg, min_owned=3 * 1024)
    assert st["n_launches"] == 4   # 64 passes, 16 fused per launch


def test_c5_4k_rank_of_8(ctx, cbox_path):
    """configs[4] geometry: 3840x2160, full graph, the tile partition 8 GPUs would use (rank 5 of 8: 1020 tiles, 1.04 M
    pixels), at 4 spp so the oracle finishes in seconds."""
    sd = scene_json.load_scene(cbox_path, 3840, 2160)
    cfg = distributed.shard_config(make_config(spp=4, spp_per_pass=4, max_depth=12, rr_depth=5), 5, 8, 32, 32)
    shard_parity(ctx, sd, cfg, min_owned=1_000_000)


def test_c5_4k_8192spp_shard(ctx, cbox_path):
    """configs[4] sample count: 3840x2160, 8192 spp (128 passes), full graph, 2 of the 8160 tiles."""
    sd = scene_json.load_scene(cbox_path, 3840, 2160)
    cfg = distributed.shard_config(make_config(spp=8192, spp_per_pass=64, max_depth=12, rr_depth=5), 3500, 4080, 32, 32)
    st = shard_parity(ctx, sd, cfg, min_owned=2048)
    assert st["n_launches"] == 8


# ---- C4: the procedural hall at 1 M and 10 M triangles -------------------------------------------------------------------
@pytest.fixture(scope="module", params=[1_000_000, 10_000_000], ids=["1M", "10M"])
def hall(request, ctx):
    from akari_render_amd import procedural

    sd = procedural.sponza_like(request.param, seed=1234, width=1920, height=1080)
    assert abs(sd.n_triangles() - request.param) < 0.01 * request.param
    scene = capi.Scene(ctx, sd)
    info = scene.info()
    assert info.uses_bvh == 1 and info.n_triangles == sd.n_triangles()
    return sd, scene, pyoracle.OracleScene(sd, bvh=True)


def test_c4_hall_exhaustive_tile(ctx, hall):
    """One 8x8 tile, 1 spp, two bounces against the oracle's EXHAUSTIVE loop over all triangles (the definition), and the
    oracle-side BVH against the same."""
    sd, scene, acc = hall
    cfg = distributed.shard_config(make_config(spp=1, spp_per_pass=1, max_depth=2), 16000, 32400, 8, 8)
    ex = pyoracle.OracleScene(sd)
    shard_parity(ctx, sd, cfg, osc=ex, scene=scene, min_owned=64)
    a, _ = ex.render(cfg)
    b, _ = acc.render(cfg)
    assert n_bit_diff(a, b) == 0


def test_c4_hall_1080p_1024spp_shard(ctx, hall):
    """configs[3]: 1080p, 1024 spp, max_depth 12: 4 tiles of 8x8 spread over the frame, 262 144 paths, ~3 M rays."""
    sd, scene, acc = hall
    cfg = distributed.shard_config(make_config(spp=1024, spp_per_pass=64, max_depth=12, rr_depth=5), 4321, 8100, 8, 8)
    st = shard_parity(ctx, sd, cfg, osc=acc, scene=scene, min_owned=256)
    assert st["n_node_visits"] > 0


def test_c4_hall_wide_shard_low_spp(ctx, hall):
    """The same frame, many more pixels (506 tiles of 8x8 = 32 384 pixels over the whole image), 2 spp."""
    sd, scene, acc = hall
    cfg = distributed.shard_config(make_config(spp=2, spp_per_pass=2, max_depth=12, rr_depth=5), 11, 64, 8, 8)
    shard_parity(ctx, sd, cfg, osc=acc, scene=scene, min_owned=30_000)
