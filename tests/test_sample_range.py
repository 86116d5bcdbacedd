"""Sample-range split (akr_pt_config.sample_begin / sample_count; SURVEY.md 8b "sample range", 8e): the checker side.
For the index-based samplers, sample s of a pixel is a pure function of (pixel, s, seed, spp) -- Pmj02BnState.sample_index
(sampler/mod.rs:451-466, 650-663) -- so ranges rendered separately draw exactly the samples the one-shot render draws; only
the order of the film's f32 additions differs. The independent sampler is refused: its start() continues the pixel's PCG stream
(sampler/mod.rs:115-131, 192-203)."""
import numpy as np
import pytest

from akari_render_amd import abi
from oracle import pyoracle, scene_json
from tests.helpers import make_config, rel_rmse, resolve_np


def _render(osc, cfg):
    return osc.render(cfg, n_threads=8)


@pytest.mark.parametrize("sampler", [abi.SAMPLER_SOBOL, abi.SAMPLER_PMJ02BN])
def test_ranges_partition_the_one_shot_render(cbox_path, sampler):
    w, h, spp = 40, 30, 24
    sd = scene_json.load_scene(cbox_path, w, h)
    osc = pyoracle.OracleScene(sd)
    if sampler == abi.SAMPLER_PMJ02BN:
        from akari_render_amd import capi
        pyoracle.set_pmj_tables(*capi.host_pmj02bn_tables())  # regenerated tables (the reference's are absent from its tree)
    whole, wst = _render(osc, make_config(spp=spp, spp_per_pass=8, max_depth=6, sampler_type=sampler, sampler_seed=3))
    n = w * h
    acc = np.zeros(7 * n, dtype=np.float64)
    tot = {k: 0 for k in wst}
    for b, c in ((0, 7), (7, 9), (16, 8)):  # ragged on purpose: the ranges do not line up with the passes of the one-shot render
        f, st = _render(osc, make_config(spp=spp, spp_per_pass=8, max_depth=6, sampler_type=sampler, sampler_seed=3, sample_begin=b, sample_count=c))
        assert np.all(f[6 * n:] == c)  # weight plane: exactly the range's sample count
        acc += f
        for k in st:
            tot[k] += st[k]
    assert tot == wst  # the very same paths: every counter adds up
    assert np.array_equal(acc[6 * n:], whole[6 * n:].astype(np.float64))
    summed = acc.astype(np.float32)
    assert rel_rmse(resolve_np(summed, w, h), resolve_np(whole, w, h)) < 1e-6  # same samples, another order of f32 additions


def test_a_range_equals_the_tail_of_a_progressive_render(cbox_path):
    """Samples [8, 16) rendered as a range = what the passes 8..15 of the one-shot render add to the film, bit for bit
    (film of 16 samples minus nothing: compare through the sampler states and by rendering the head first)."""
    w, h = 32, 24
    sd = scene_json.load_scene(cbox_path, w, h)
    osc = pyoracle.OracleScene(sd)
    n = w * h
    states = np.zeros(2 * n, dtype=np.uint64)
    states[0::2] = 0xFFFFFFFF
    states[1::2] = (np.arange(n, dtype=np.uint64) % np.uint64(w)) | ((np.arange(n, dtype=np.uint64) // np.uint64(w)) << np.uint64(32))
    cfg = make_config(spp=16, spp_per_pass=8, max_depth=5, sampler_type=abi.SAMPLER_SOBOL, sampler_seed=1)
    head_cfg = cfg.copy(); head_cfg.sample_count = 8
    osc.render(head_cfg, n_threads=4, states=states)          # samples 0..7; states now hold sample_index 7
    assert np.all(states[0::2] == 7)
    tail_via_states, _ = osc.render(head_cfg, n_threads=4, states=states)   # continues: samples 8..15 (count 8 from where the states are)
    tail_cfg = cfg.copy(); tail_cfg.sample_begin, tail_cfg.sample_count = 8, 8
    tail_via_range, _ = osc.render(tail_cfg, n_threads=4)
    assert np.array_equal(tail_via_states.view(np.uint32), tail_via_range.view(np.uint32))


def test_the_independent_sampler_refuses_a_range(cbox_path):
    import ctypes as C
    sd = scene_json.load_scene(cbox_path, 16, 12)
    osc = pyoracle.OracleScene(sd)
    cfg = make_config(spp=16, spp_per_pass=8, sample_begin=8, sample_count=8)
    film = np.zeros(7 * 16 * 12, dtype=np.float32)
    rc = pyoracle.lib().or_pt_render(osc.h, C.byref(cfg), film.ctypes.data_as(C.POINTER(C.c_float)), C.POINTER(C.c_uint64)(), 1, None)
    assert rc == -4
    bad = make_config(spp=16, spp_per_pass=8, sampler_type=abi.SAMPLER_SOBOL, sample_begin=10, sample_count=8)  # runs past spp
    assert pyoracle.lib().or_pt_render(osc.h, C.byref(bad), film.ctypes.data_as(C.POINTER(C.c_float)), C.POINTER(C.c_uint64)(), 1, None) == -4


def test_caller_states_must_stand_where_the_range_begins(cbox_path):
    """or_pt_render with caller-supplied sampler states and sample_begin != 0: the states must hold sample index begin - 1 (what the
    previous range left); anything else is refused instead of silently rendering other samples."""
    import ctypes as C
    w, h = 16, 12
    sd = scene_json.load_scene(cbox_path, w, h)
    osc = pyoracle.OracleScene(sd)
    n = w * h
    states = np.zeros(2 * n, dtype=np.uint64)
    states[0::2] = 0xFFFFFFFF
    states[1::2] = (np.arange(n, dtype=np.uint64) % np.uint64(w)) | ((np.arange(n, dtype=np.uint64) // np.uint64(w)) << np.uint64(32))
    cfg = make_config(spp=16, spp_per_pass=8, max_depth=4, sampler_type=abi.SAMPLER_SOBOL, sampler_seed=1, sample_begin=8, sample_count=8)
    film = np.zeros(7 * n, dtype=np.float32)
    sp = states.ctypes.data_as(C.POINTER(C.c_uint64))
    assert pyoracle.lib().or_pt_render(osc.h, C.byref(cfg), film.ctypes.data_as(C.POINTER(C.c_float)), sp, 2, None) == -4  # fresh states: index -1
    assert not film.any()
    states[0::2] = 7
    assert pyoracle.lib().or_pt_render(osc.h, C.byref(cfg), film.ctypes.data_as(C.POINTER(C.c_float)), sp, 2, None) == 0
    ref, _ = osc.render(cfg, n_threads=2)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32)) and np.all(states[0::2] == 15)
