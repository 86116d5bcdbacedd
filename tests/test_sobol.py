"""The "sobol" sampler (akr_sampler_type 2): Owen-scrambled, padded Sobol' (0,2)-sequence with Pmj02BnSampler's state and
interface (sampler/mod.rs:329-700). The reference ships only a data stub for Sobol' (akari_data/src/lib.rs:13,19), so there
is nothing to be bit-compatible with: the tests pin the defining properties (every dimension pair of a pixel is a (0,2)-net,
pixels and dimensions are decorrelated) and that it converges faster than the independent sampler; HIP-vs-oracle parity is
bit-exact like everywhere else (tests/test_gpu_sobol.py)."""
import ctypes as C

import numpy as np
import pytest

from akari_render_amd import abi
from oracle import pyoracle, scene_json
from tests.helpers import make_config, resolve_np


def points(px, py, dim, seed, spp):
    L = pyoracle.lib()
    out = np.zeros((spp, 2), dtype=np.float32)
    buf = (C.c_float * 2)()
    for i in range(spp):
        L.or_kat_sobol_2d(px, py, dim, seed, spp, i, buf)
        out[i] = buf[0], buf[1]
    return out


@pytest.mark.parametrize("spp", [16, 64, 256, 1024])
def test_every_dimension_pair_is_a_02_net(spp):
    m = int(np.log2(spp))
    for (px, py, dim, seed) in [(0, 0, 4, 0), (3, 5, 4, 1), (100, 7, 12, 99), (1919, 1079, 30, 12345)]:
        p = points(px, py, dim, seed, spp).astype(np.float64)
        assert np.all((p >= 0) & (p < 1))
        for a in range(m + 1):  # elementary intervals 2^-a x 2^-(m-a): exactly one point each
            b = m - a
            cell = np.floor(p[:, 0] * (1 << a)).astype(np.int64) * (1 << b) + np.floor(p[:, 1] * (1 << b)).astype(np.int64)
            assert np.array_equal(np.sort(cell), np.arange(spp)), (px, py, dim, seed, a)


def test_pixels_and_dimensions_are_decorrelated():
    a, b, c = points(10, 10, 4, 0, 64), points(11, 10, 4, 0, 64), points(10, 10, 6, 0, 64)
    assert not np.array_equal(a, b) and not np.array_equal(a, c)
    # the i-th sample of neighbouring pixels / dimension pairs is not the same point in another order either
    assert not np.array_equal(np.sort(a[:, 0]), np.sort(b[:, 0]))
    # a non-power-of-two spp still gives distinct, in-range points
    p = points(1, 2, 4, 3, 48)
    assert len({tuple(x) for x in p}) == 48 and np.all((p >= 0) & (p < 1))


def test_converges_faster_than_independent(cbox_path):
    """Direct lighting of the Cornell box: RMSE against a 4096-spp reference at 16 and 64 spp, averaged over three seeds."""
    sd = scene_json.load_scene(cbox_path, 40, 40)
    osc = pyoracle.OracleScene(sd)

    def render(spp, sampler, seed):
        cfg = make_config(spp=spp, spp_per_pass=spp if spp <= 64 else 256, max_depth=1, sampler_type=sampler, sampler_seed=seed)
        return resolve_np(osc.render(cfg)[0], 40, 40).astype(np.float64)

    ref = render(4096, abi.SAMPLER_INDEPENDENT, 777)
    for spp in (16, 64):
        e_ind = np.mean([np.sqrt(np.mean((render(spp, abi.SAMPLER_INDEPENDENT, s) - ref) ** 2)) for s in (1, 2, 3)])
        e_sob = np.mean([np.sqrt(np.mean((render(spp, abi.SAMPLER_SOBOL, s) - ref) ** 2)) for s in (1, 2, 3)])
        print(f"spp {spp}: rmse independent {e_ind:.4f}, sobol {e_sob:.4f}")
        assert e_sob < 0.85 * e_ind
    # unbiased: the 1024-spp sobol image agrees with the reference to within its (smaller) noise
    assert abs(render(1024, abi.SAMPLER_SOBOL, 5).mean() - ref.mean()) < 0.01 * ref.mean()


def test_second_dimension_without_the_loop():
    """drng.h sobol_dim1_reversed (five butterfly steps) = reverse_bits32(sobol_dim1(i)) (the defining loop): every index below
    2^20, random 32-bit indices, and the single-bit ones."""
    from akari_render_amd import capi

    idx = np.concatenate([np.arange(1 << 20, dtype=np.uint32), np.random.default_rng(0).integers(0, 1 << 32, size=200000, dtype=np.uint64).astype(np.uint32),
                          (np.uint32(1) << np.arange(32, dtype=np.uint32)), np.array([0xffffffff, 0x80000001], dtype=np.uint32)])
    a, b = np.zeros_like(idx), np.zeros_like(idx)
    up = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint32))  # noqa: E731
    capi.check(capi.lib().akr_host_sobol_dim1(idx.size, up(idx), up(a), up(b)))
    assert np.array_equal(a, b)
    # and the loop is the Pascal matrix: index 2^k gives column k = (1 + x)^k, bit j set iff j is a submask of k
    for k in range(32):
        col = int(a[(1 << 20) + 200000 + k])
        assert col == sum(1 << j for j in range(32) if (j & ~k) == 0)


def test_remainder_without_a_division():
    """drng.h fastmod_u32 (the last line of permute_element for a sample count that is not a power of two) = a % d."""
    from akari_render_amd import capi

    rng = np.random.default_rng(3)
    d = np.concatenate([rng.integers(1, 1 << 32, size=300000, dtype=np.uint64), rng.integers(1, 70000, size=300000, dtype=np.uint64),
                        np.array([1, 2, 3, 25600, 65535, 65536, 0xffffffff, 0x80000000, 0x7fffffff], dtype=np.uint64)]).astype(np.uint32)
    a = rng.integers(0, 1 << 32, size=d.size, dtype=np.uint64).astype(np.uint32)
    a[:9] = [0, 1, 0xffffffff, 0xfffffffe, 25599, 25600, 25601, 0x80000000, 65535]
    a[-9:] = 0xffffffff
    out = np.zeros_like(a)
    up = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint32))  # noqa: E731
    capi.check(capi.lib().akr_host_fastmod(a.size, up(a), up(d), up(out)))
    assert np.array_equal(out, a % d)


def test_unorm16_by_reciprocal_is_the_division():
    """dpath.h unorm16 on the device: q = v y, q + (v - 65535 q) y with y = RN(1 / 65535) (two fma) -- the correctly rounded
    v / 65535 for all 65536 values (the fma is emulated exactly in float64: every product and sum here fits its 53 bits)."""
    v = np.arange(65536, dtype=np.float64)
    want = (v.astype(np.float32) / np.float32(65535.0)).astype(np.float32)
    y = np.float32(1.5259021893143654e-05)
    assert y == np.float32(1.0) / np.float32(65535.0)
    fv = v.astype(np.float32)
    q = (fv * y).astype(np.float32)
    r = (np.float64(-65535.0) * q.astype(np.float64) + fv.astype(np.float64)).astype(np.float32)
    got = (r.astype(np.float64) * np.float64(y) + q.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
