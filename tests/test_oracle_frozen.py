"""The checker does not move with the kernels.

Commit aa0bc1f introduced the coplanar-neighbour plane-row rule (triangles 2j, 2j+1 of an instance share their third Woop row
when coplanar) in the product's scene compiler AND in the oracle, and regenerated the golden films. These tests
pin what that did: `cbox_64x64_16spp_prerule.npz` is the golden file as committed BEFORE the rule (git 3d05f86, byte for
byte); the oracle with the rule switched off must still reproduce it exactly, and the films with the rule must stay far
inside the 1e-3 contract of the films without it. Any later change that edits oracle/ together with a kernel has to add a
bound of this kind ("frozen checker").
"""
import os

import numpy as np
import pytest

from oracle import pyoracle, scene_json
from tests.helpers import make_config, n_bit_diff, rel_rmse, resolve_np

RULE_BOUND = 1e-4  # relRMSE the rule may cost; the contract is 1e-3. Measured: 5.6e-7 .. 8.3e-7 at 64x64x16, 2.2e-5 on C1


@pytest.fixture(scope="module")
def goldens(root):
    g = os.path.join(root, "tests", "golden")
    return np.load(os.path.join(g, "cbox_64x64_16spp_prerule.npz")), np.load(os.path.join(g, "cbox_64x64_16spp.npz"))


@pytest.mark.parametrize("key,fd", [("full", 0), ("force_diffuse", 1)])
def test_oracle_without_the_rule_is_the_pre_rule_golden(cbox_path, goldens, key, fd):
    pre, cur = goldens
    sd = scene_json.load_scene(cbox_path, 64, 64)
    cfg = make_config(spp=16, spp_per_pass=16, force_diffuse=fd)
    off, _ = pyoracle.OracleScene(sd, share_plane_rows=False).render(cfg)
    on, _ = pyoracle.OracleScene(sd).render(cfg)
    assert n_bit_diff(off, pre[key]) == 0      # the unmodified restatement still gives the pre-rule film, bit for bit
    assert n_bit_diff(on, cur[key]) == 0       # and the current golden is the film with the rule
    err = rel_rmse(resolve_np(on, 64, 64), resolve_np(off, 64, 64))
    nd = n_bit_diff(on, off)
    print(f"{key}: rule changes {nd} of {on.size} film floats, relRMSE {err:.3e}")
    assert 0 < nd < on.size // 2
    assert err < RULE_BOUND


def test_rule_bound_on_c1(cbox_path):
    """BASELINE.json configs[0] (256x256, 64 spp, full graph): with vs without the rule."""
    sd = scene_json.load_scene(cbox_path, 256, 256)
    cfg = make_config(spp=64, spp_per_pass=64)
    a = pyoracle.OracleScene(sd)
    b = pyoracle.OracleScene(sd, share_plane_rows=False)
    # (4 axis-aligned quads get bit-identical rows from their own vertices anyway)
    assert a.shared_plane_rows() == 17 and b.shared_plane_rows() == 4
    on, s_on = a.render(cfg)
    off, s_off = b.render(cfg)
    err = rel_rmse(resolve_np(on, 256, 256), resolve_np(off, 256, 256))
    nd = n_bit_diff(on, off)
    print(f"C1: rule changes {nd} of {on.size} film floats, relRMSE {err:.3e}; rays {s_on['n_closest']} vs {s_off['n_closest']}")
    assert err < RULE_BOUND
    # the paths themselves are the same paths: ray counts differ by at most a few per million
    assert abs(s_on["n_closest"] - s_off["n_closest"]) < 1e-4 * s_on["n_closest"]
