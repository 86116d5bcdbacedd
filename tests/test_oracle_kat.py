"""Known-answer and property tests that pin the CPU oracle: published vectors of the third-party
algorithms the reference builds on, and the reference's own property tests restated."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle

u64p = C.POINTER(C.c_uint64)


def test_pcg32_reference_vectors(oracle_lib):
    # pcg-c-basic demo: pcg32_srandom_r(&rng, 42u, 54u) -> first six outputs. The reference's
    # Pcg32Var::set_seq_offset(seq, seed) is pcg32_srandom_r(initstate = seed, initseq = seq) (sampler/mod.rs:88-94).
    st, inc = C.c_uint64(), C.c_uint64()
    # new_seq_offset(54, 42): reproduce through the raw generator
    out = np.zeros(1, dtype=np.uint32)
    state, incv = 0, (54 << 1) | 1
    oracle_lib.or_kat_pcg32(state, incv, 1, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    state = (0 * 0x5851F42D4C957F2D + incv) & (2**64 - 1)
    state = (state + 42) & (2**64 - 1)
    out = np.zeros(7, dtype=np.uint32)
    oracle_lib.or_kat_pcg32(state, incv, 7, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert [hex(v) for v in out[1:]] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_chacha_core_rfc7539(oracle_lib):
    key = np.arange(32, dtype=np.uint8).view("<u4").copy()
    out = np.zeros(16, dtype=np.uint32)
    counter = 1 | (0x09000000 << 32)
    stream = 0x4A000000
    oracle_lib.or_kat_chacha_block(key.ctypes.data_as(C.POINTER(C.c_uint32)), counter, stream, 20, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    expect = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
              0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
    assert list(out) == expect


@pytest.mark.parametrize("rounds,first16", [
    (20, "76b8e0ada0f13d90405d6ae55386bd28"),
    (12, "9bf49a6a0755f953811fce125f2683d5"),
    (8, "3e00ef2f895f40d67f5bb8e81f09a5a1"),
])
def test_chacha_zero_key_keystreams(oracle_lib, rounds, first16):
    key = np.zeros(8, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    oracle_lib.or_kat_chacha_block(key.ctypes.data_as(C.POINTER(C.c_uint32)), 0, 0, rounds, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert out.astype("<u4").tobytes()[:16].hex() == first16


def test_stdrng_structure(oracle_lib):
    # seed_from_u64 = PCG32 expansion of the u64 into the key; the stream is ChaCha12 blocks 0,1,... as u64 pairs
    def key_from_u64(state):
        key = []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & (2**64 - 1)
            xs = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
            rot = state >> 59
            key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF)
        return np.array(key, dtype=np.uint32)
    for seed in (0, 1, 0xDEADBEEF12345678):
        got = np.zeros(20, dtype=np.uint64)
        oracle_lib.or_kat_stdrng_u64(seed, 20, got.ctypes.data_as(u64p))
        key = key_from_u64(seed)
        words = []
        for blk in range(3):
            out = np.zeros(16, dtype=np.uint32)
            oracle_lib.or_kat_chacha_block(key.ctypes.data_as(C.POINTER(C.c_uint32)), blk, 0, 12, out.ctypes.data_as(C.POINTER(C.c_uint32)))
            words += list(out)
        exp = [int(words[2 * i]) | (int(words[2 * i + 1]) << 32) for i in range(20)]
        assert [int(v) for v in got] == exp


def _xxhash32_4_py(p):
    M = 0xFFFFFFFF
    P2, P3, P4, P5 = 2246822519, 3266489917, 668265263, 374761393
    rot = lambda h: ((h << 17) | (h >> 15)) & M
    h = (p[3] + P5 + p[0] * P3) & M
    h = (P4 * rot(h)) & M
    h = (h + p[1] * P3) & M
    h = (P4 * rot(h)) & M
    h = (h + p[2] * P3) & M
    h = (P4 * rot(h)) & M
    h = (P2 * (h ^ (h >> 15))) & M
    h = (P3 * (h ^ (h >> 13))) & M
    return h ^ (h >> 16)


def test_xxhash32_4_and_mix_bits(oracle_lib):
    rng = np.random.default_rng(0)
    for _ in range(200):
        p = [int(v) for v in rng.integers(0, 2**32, 4)]
        assert oracle_lib.or_kat_xxhash32_4(*p) == _xxhash32_4_py(p)
    def mix(v):
        M = 2**64 - 1
        v ^= v >> 31; v = (v * 0x7FB5D329728EA185) & M
        v ^= v >> 27; v = (v * 0x81DADEF4BC2DD44D) & M
        v ^= v >> 33
        return v
    for v in (0, 1, 12345678901234567, 2**64 - 1):
        assert oracle_lib.or_kat_mix_bits(v) == mix(v)


def test_pcg_advance_matches_reference_loop(oracle_lib):
    # sampler/mod.rs:115-131 re-implemented in Python integers
    M = 2**64 - 1
    def advance(state, inc, delta):
        cur_mult, cur_plus, acc_mult, acc_plus = 0x5851F42D4C957F2D, inc, 1, 0
        delta &= M
        while delta > 0:
            if delta & 1:
                acc_mult = (acc_mult * cur_mult) & M
                acc_plus = (acc_plus + cur_mult + cur_plus) & M
            cur_plus = ((cur_mult + 1) * cur_plus) & M
            cur_mult = (cur_mult * cur_mult) & M
            delta >>= 1
        return (acc_mult * state + acc_plus) & M
    rng = np.random.default_rng(5)
    for delta in (16384, -7, -1234, 1, 0, 99999):
        s0, inc = int(rng.integers(0, 2**63)), int(rng.integers(0, 2**62)) * 2 + 1
        st = C.c_uint64(s0)
        oracle_lib.or_kat_pcg32_advance(C.byref(st), inc, delta)
        assert st.value == advance(s0, inc, delta)


def test_next_1d_range(oracle_lib):
    st = C.c_uint64(123)
    vals = [oracle_lib.or_kat_next_1d(C.byref(st), 7) for _ in range(20000)]
    assert min(vals) >= 0.0 and max(vals) <= 1.0
    assert abs(np.mean(vals) - 0.5) < 0.01


def test_alias_table_mass(oracle_lib):
    """The reference's own test (util/distribution.rs:116-146): per-bin mass reconstructed from (t, j)."""
    rng = np.random.default_rng(11)
    w = rng.random(100).astype(np.float32)
    j = np.zeros(100, dtype=np.uint32); t = np.zeros(100, dtype=np.float32); pdf = np.zeros(100, dtype=np.float32)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    oracle_lib.or_kat_alias_build(w.ctypes.data_as(fp), 100, j.ctypes.data_as(up), t.ctypes.data_as(fp), pdf.ctypes.data_as(fp))
    mass = np.zeros(100)
    for i in range(100):
        mass[i] += t[i] / 100
        mass[j[i]] += (1 - t[i]) / 100
    assert np.max(np.abs(mass - w / w.sum())) < 1e-3
    assert np.allclose(pdf, w / w.sum(), atol=1e-6)


def test_alias_table_small_known(oracle_lib):
    w = np.array([1, 2, 3, 4], dtype=np.float32)
    j = np.zeros(4, dtype=np.uint32); t = np.zeros(4, dtype=np.float32); pdf = np.zeros(4, dtype=np.float32)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    oracle_lib.or_kat_alias_build(w.ctypes.data_as(fp), 4, j.ctypes.data_as(up), t.ctypes.data_as(fp), pdf.ctypes.data_as(fp))
    assert np.allclose(pdf, [0.1, 0.2, 0.3, 0.4])
    mass = np.zeros(4)
    for i in range(4):
        mass[i] += t[i] / 4; mass[j[i]] += (1 - t[i]) / 4
    assert np.allclose(mass, [0.1, 0.2, 0.3, 0.4], atol=1e-6)


def test_elementary_functions_accuracy(oracle_lib):
    xs = np.linspace(0, 2 * np.pi, 20001).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for x in xs[::7]:
        oracle_lib.or_kat_sincos(float(x), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
    assert worst < 2.5e-7
    for x in np.geomspace(1e-10, 1.0, 3000):
        x = float(np.float32(x))
        assert abs(oracle_lib.or_kat_log(x) - np.log(x)) <= 2e-7 * max(1.0, abs(np.log(x)))
    assert oracle_lib.or_kat_log(0.0) == -np.inf


def test_sampling_warps(oracle_lib):
    rng = np.random.default_rng(2)
    fp = C.POINTER(C.c_float)
    out3 = np.zeros(3, dtype=np.float32); out2 = np.zeros(2, dtype=np.float32)
    zs = []
    for _ in range(4000):
        u = rng.random(2)
        oracle_lib.or_kat_cos_sample_hemisphere(u[0], u[1], out3.ctypes.data_as(fp))
        assert out3[2] >= 0 and abs(np.linalg.norm(out3) - 1) < 1e-5
        zs.append(out3[2])
        oracle_lib.or_kat_uniform_sample_triangle(u[0], u[1], out2.ctypes.data_as(fp))
        assert out2[0] >= 0 and out2[1] >= 0 and out2[0] + out2[1] <= 1 + 1e-6
    assert abs(np.mean(zs) - 2 / 3) < 0.02  # E[cos] under a cosine-weighted density
