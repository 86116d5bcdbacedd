"""The box levels of the acceleration structures only CULL: a triangle the exhaustive loop (the oracle's definition of a hit) accepts
for a ray must be reachable, i.e. every box between the root and its leaf must pass the device's slab test with the accepted t as
the limit. tests/bvh_model.py models that test in the device's f32 operations on the trees `capi.Scene(None, ..)` compiles -- no
GPU -- and aims rays at the rims of the triangles, down to grazing angles.

Round 6: this model found what one film pixel in 400 extreme scenes had hinted at (HISTORY R5.7): the inside
test of a NEEDLE is ill-conditioned -- the triangle it sees is displaced along its long axis by 1 / sin(angle at the first vertex)
times the usual round-off -- and a flat padding cannot cover that. Boxes now follow each triangle's conditioning (scene_build.h
tri_conditioning); the same model reproduced ADVICE r5's object-space defect (a mesh modelled far from its own origin)."""
import numpy as np
import pytest

from akari_render_amd import capi
from tests import bvh_model
from tests.helpers import extreme_instanced_scene, far_modelled_mesh_scene, grid_scene, instanced_scene, shift_scene


def both_scenes(sd, rebraid=1):
    with capi.options(instancing=0, force_bvh=1):
        flat = capi.Scene(None, sd)
    with capi.options(instancing=1, rebraid=rebraid):
        kept = capi.Scene(None, sd)
    assert flat.info().uses_bvh == 1 and kept.info().uses_bvh == 2
    return flat, kept


# 50, 52, 63, 90, 93, 107: scenes in which the round-5 trees culled an accepted pair (93 is the scene of HISTORY R5.7); 88: the kept tree did
@pytest.mark.parametrize("seed", [50, 52, 63, 88, 90, 93, 107, 3, 17])
def test_extreme_transforms_no_accepted_pair_is_culled(hip_lib, seed):
    sd, _ = extreme_instanced_scene(seed)
    flat, kept = both_scenes(sd)
    rng = np.random.default_rng(1000 + seed)  # (the stream the defects were found with)
    a, c, worst = bvh_model.check_flattened(flat, 64, rng, max_tris=600)
    assert a > 1000 and c == 0, (a, c, worst[:5])
    a, c, worst = bvh_model.check_kept(kept, flat, 64, rng, max_tris=600)
    assert a > 1000 and c == 0, (a, c, worst[:5])


@pytest.mark.parametrize("rebraid", [4, 16])
def test_rebraided_top_level_tree(hip_lib, rebraid):
    """Option rebraid: the top-level tree over (instance, subtree) pairs. check_kept asserts that the pairs' subtrees partition every
    instance's triangles, then the usual property from each pair's entry node down."""
    from akari_render_amd import procedural
    sd = procedural.instanced_forest(12, 3000, width=64, height=64)
    flat, kept = both_scenes(sd, rebraid)
    n_leaves = len(kept.array(capi.ARRAY_INST_LEAVES, np.float32)) // 16
    assert 14 * rebraid // 2 < n_leaves <= 14 * rebraid  # 12 plants + ground + sky
    a, c, worst = bvh_model.check_kept(kept, flat, 32, np.random.default_rng(3), max_tris=500)
    assert a > 3000 and c == 0, (a, c, worst[:5])


@pytest.mark.parametrize("offset", [1e3, 1e5])
def test_mesh_modelled_far_from_its_own_origin(hip_lib, offset):
    """vertices at ~offset, instance translations taking them back. World coordinates ~ 1, object-space rays ~ offset:
    the round-5 padding of the per-mesh trees culled 175 of 75 000 accepted pairs at 1e4 and 5 947 at 1e5 in this model."""
    sd = far_modelled_mesh_scene(offset)
    flat, kept = both_scenes(sd)
    rng = np.random.default_rng(5)
    a, c, worst = bvh_model.check_kept(kept, flat, 96, rng, max_tris=800)
    assert a > 10000 and c == 0, (a, c, worst[:5])
    a, c, worst = bvh_model.check_flattened(flat, 96, rng, max_tris=800)
    assert a > 10000 and c == 0, (a, c, worst[:5])


def test_scenes_far_from_the_origin(hip_lib):
    for sd in (shift_scene(grid_scene(n=12), (1e4, 2e4, -5e3)), shift_scene(instanced_scene(n_inst=8), (-3e3, 1e3, 2e4))):
        with capi.options(instancing=0):
            flat = capi.Scene(None, sd)
        a, c, worst = bvh_model.check_flattened(flat, 64, np.random.default_rng(9), max_tris=500)
        assert a > 1000 and c == 0, (a, c, worst[:5])


def test_the_model_sees_a_tree_that_is_too_tight(hip_lib, monkeypatch):
    """The check is not vacuous: child boxes pulled in by one quantisation step on every side lose accepted pairs."""
    sd = instanced_scene(n_inst=6)
    with capi.options(instancing=0):
        flat = capi.Scene(None, sd)
    real = bvh_model.entry_box_params

    def shrunk(nodes, node_off, path):
        origin, scale, qlo, qhi = real(nodes, node_off, path)
        return origin, scale, qlo + np.float32(1.0), qhi - np.float32(1.0)

    a, c, _ = bvh_model.check_flattened(flat, 64, np.random.default_rng(2), max_tris=300)
    assert a > 1000 and c == 0
    monkeypatch.setattr(bvh_model, "entry_box_params", shrunk)
    a, c, _ = bvh_model.check_flattened(flat, 64, np.random.default_rng(2), max_tris=300)
    assert c > 0


def test_well_shaped_triangles_keep_the_flat_padding(hip_lib, cbox_path):
    """tri_conditioning is 2 for a right angle at the first vertex and 2.31 for 60 degrees: below kTriCondFree, nothing is added --
    the Cornell box's tree and the grid's are what the flat padding alone gives (checked through the boxes of the leaves: every
    leaf box decodes to within one quantisation step of its triangles' bounds + the flat padding)."""
    from oracle import scene_json
    for sd in (scene_json.load_scene(cbox_path, 32, 32), grid_scene(n=8)):
        with capi.options(force_bvh=1):
            sc = capi.Scene(None, sd)
        nodes = sc.array(capi.ARRAY_BVH_NODES, np.uint32)
        woop = sc.array(capi.ARRAY_WOOP, np.float32).reshape(-1, 16)[: sc.info().n_triangles]
        shade = sc.array(capi.ARRAY_SHADE, np.float32).reshape(-1, 32)
        world = bvh_model._world_vertices(shade, sc.array(capi.ARRAY_INSTANCES, np.float32).reshape(-1, 32))
        lo, hi = world.reshape(-1, 3).min(0), world.reshape(-1, 3).max(0)
        c2w = sc.array(capi.ARRAY_C2W, np.float32)
        reach = float(np.maximum(np.maximum(np.abs(lo), np.abs(hi)), np.abs(c2w[12:15])).sum())
        pad = 4e-6 * max(reach, float(np.linalg.norm(hi - lo)))
        paths = bvh_model.decode_tree(nodes)
        leaves = {}
        for k, path in paths.items():
            leaves.setdefault(path[-1], []).append(k)
        for (idx, e), tris in leaves.items():
            origin, scale, qlo, qhi = bvh_model.entry_box_params(nodes, 0, [(idx, e)])
            blo = origin[0].astype(np.float64) + qlo[0] * scale[0].astype(np.float64)
            bhi = origin[0].astype(np.float64) + qhi[0] * scale[0].astype(np.float64)
            g = woop[tris, 12].view(np.uint32)
            tlo, thi = world[g].reshape(-1, 3).min(0) - pad, world[g].reshape(-1, 3).max(0) + pad
            step = scale[0].astype(np.float64) * 1.01 + 1e-6 * reach
            assert np.all(blo <= tlo + 1e-7 * reach) and np.all(bhi >= thi - 1e-7 * reach)
            assert np.all(blo >= tlo - step) and np.all(bhi <= thi + step), (idx, e)
