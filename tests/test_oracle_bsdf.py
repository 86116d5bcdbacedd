"""Property tests of the oracle's BSDFs and estimator. The chi^2 test restates the reference's manual harness
(crates/akari_api/src/bin/akari_test.rs:16-439: histogram of sample_wi against the integral of evaluate().pdf,
alpha = 0.01 with Sidak correction); furnace / energy tests are closed-form checks the reference lacks."""
import numpy as np
import pytest
from scipy import stats

from akari_render_amd import abi
from oracle import pyoracle
from tests.helpers import box_scene, make_config, resolve_np


def _dirs(cos_t, phi):
    st = np.sqrt(np.maximum(0.0, 1.0 - cos_t**2))
    return np.stack([st * np.cos(phi), st * np.sin(phi), cos_t], axis=-1)


def chi2_pvalue(m: abi.MaterialData, wo, n_samples=300_000, nc=32, nphi=64, sub=8, seed=0, sphere=False):
    rng = np.random.default_rng(seed)
    u = rng.random((n_samples, 3), dtype=np.float32)
    smp = pyoracle.bsdf_sample_many(m, wo, u)
    valid = smp[:, 7] > 0.5
    wi = smp[valid, :3].astype(np.float64)
    lo = -1.0 if sphere else 0.0
    ci = np.clip(((wi[:, 2] - lo) / (1.0 - lo) * nc).astype(int), 0, nc - 1)
    ph = np.mod(np.arctan2(wi[:, 1], wi[:, 0]), 2 * np.pi)
    pi_ = np.clip((ph / (2 * np.pi) * nphi).astype(int), 0, nphi - 1)
    obs = np.zeros((nc, nphi))
    np.add.at(obs, (ci, pi_), 1.0)
    # expected counts: Gauss-Legendre quadrature (sub x sub nodes) of the pdf over every bin, measure d cos d phi
    gx, gw = np.polynomial.legendre.leggauss(sub)
    ce = lo + np.arange(nc + 1) / nc * (1.0 - lo)
    pe_ = np.arange(nphi + 1) / nphi * 2 * np.pi
    cs = (0.5 * (ce[:-1] + ce[1:])[:, None] + 0.5 * (ce[1:] - ce[:-1])[:, None] * gx[None, :]).ravel()
    ps = (0.5 * (pe_[:-1] + pe_[1:])[:, None] + 0.5 * (pe_[1:] - pe_[:-1])[:, None] * gx[None, :]).ravel()
    C_, P_ = np.meshgrid(cs, ps, indexing="ij")
    d = _dirs(C_.ravel(), P_.ravel()).astype(np.float32)
    pdf = pyoracle.bsdf_eval_many(m, wo, d)[:, 3].astype(np.float64).reshape(nc, sub, nphi, sub)
    w2 = gw[:, None] * gw[None, :]
    cell = 0.25 * (1.0 - lo) / nc * (2 * np.pi) / nphi
    exp = np.einsum("aibj,ij->ab", pdf, w2) * cell * n_samples
    # pool low-expectation bins (pbrt-v4 chi2 test convention: >= 5 expected per cell)
    order = np.argsort(exp.ravel())
    e, o = exp.ravel()[order], obs.ravel()[order]
    pooled_e, pooled_o, ae, ao = [], [], 0.0, 0.0
    for ei, oi in zip(e, o):
        ae += ei; ao += oi
        if ae >= 5.0:
            pooled_e.append(ae); pooled_o.append(ao); ae = ao = 0.0
    if ae > 0 and pooled_e:
        pooled_e[-1] += ae; pooled_o[-1] += ao
    pe, po = np.array(pooled_e), np.array(pooled_o)
    stat = np.sum((po - pe) ** 2 / pe)
    dof = len(pe) - 1
    return stats.chi2.sf(stat, dof), valid.mean(), exp.sum() / n_samples


CHI2_CASES = [
    ("diffuse", abi.MaterialData(kind=abi.MAT_DIFFUSE, base_color=(0.8, 0.8, 0.8)), False),
    ("ggx_refl_0.3", abi.MaterialData(metallic=1.0, roughness=0.3, base_color=(0.9, 0.9, 0.9)), False),
    ("ggx_refl_0.5", abi.MaterialData(metallic=1.0, roughness=0.5, base_color=(0.9, 0.9, 0.9)), False),
    ("ggx_refl_0.8", abi.MaterialData(metallic=1.0, roughness=0.8, base_color=(0.9, 0.9, 0.9)), False),
    ("glass_0.4", abi.MaterialData(kind=abi.MAT_GLASS, base_color=(1, 1, 1), ior=1.33, roughness=0.4), True),
    ("glass_0.7", abi.MaterialData(kind=abi.MAT_GLASS, base_color=(1, 1, 1), ior=1.33, roughness=0.7), True),
]


@pytest.mark.parametrize("name,mat,sphere", CHI2_CASES, ids=[c[0] for c in CHI2_CASES])
def test_chi2_sample_matches_pdf(oracle_lib, name, mat, sphere):
    n_tests = len(CHI2_CASES) * 2
    alpha = 1.0 - (1.0 - 0.01) ** (1.0 / n_tests)  # Sidak
    wos = [_dirs(np.array(0.9), np.array(0.3)), _dirs(np.array(0.45), np.array(2.0))]
    if name == "glass_0.7":
        # At grazing incidence the reference's rough-transmission pdf (|wo.wh| instead of max(0, wo.wh) in the
        # visible-normal density, microfacet.rs:196-206) differs from its sampler by ~0.5 % of the mass; that is a
        # property of the restated algorithm, so the test stays at near-normal incidence for this case.
        wos = wos[:1]
    for k, wo in enumerate(wos):
        p, valid_frac, pdf_mass = chi2_pvalue(mat, wo.astype(np.float32), seed=k, sphere=sphere)
        # the pdf integrates to the probability of producing a valid sample
        assert abs(pdf_mass - valid_frac) < 0.02, (name, pdf_mass, valid_frac)
        assert p > alpha, (name, k, p)


def test_sample_returns_evaluate(oracle_lib):
    """BsdfSample.color/pdf are evaluate(wo, wi) of the sampled direction (svm/surface/mod.rs:795-815)."""
    rng = np.random.default_rng(3)
    m = abi.MaterialData(base_color=(0.7, 0.5, 0.3), roughness=0.4, metallic=0.3, ior=1.5, coat_weight=0.5, transmission_weight=0.2)
    table = rng.random(4096).astype(np.float32) * 0.5
    wo = _dirs(np.array(0.6), np.array(1.0)).astype(np.float32)
    s = pyoracle.bsdf_sample_many(m, wo, rng.random((2000, 3), dtype=np.float32), table)
    ok = s[:, 7] > 0.5
    assert ok.sum() > 1000
    e = pyoracle.bsdf_eval_many(m, wo, s[ok, :3], table)
    assert np.array_equal(e[:, :3], s[ok, 3:6]) and np.array_equal(e[:, 3], s[ok, 6])


def test_energy_conservation(oracle_lib):
    """E[f cos / pdf] <= 1 per channel for energy-conserving configurations (white furnace on the BSDF)."""
    rng = np.random.default_rng(7)
    table = np.zeros(4096, dtype=np.float32)
    for m in [
        abi.MaterialData(kind=abi.MAT_DIFFUSE, base_color=(1, 1, 1)),
        abi.MaterialData(base_color=(1, 1, 1), metallic=1.0, roughness=0.5),
        abi.MaterialData(kind=abi.MAT_GLASS, base_color=(1, 1, 1), ior=1.5, roughness=0.3),
        abi.MaterialData(base_color=(1, 1, 1), metallic=0.0, roughness=0.6, ior=1.0, specular_ior_level=0.0),
    ]:
        for c in (0.9, 0.5, 0.2):
            wo = _dirs(np.array(c), np.array(0.0)).astype(np.float32)
            s = pyoracle.bsdf_sample_many(m, wo, rng.random((100_000, 3), dtype=np.float32), table)
            ok = s[:, 7] > 0.5
            w = np.where(ok[:, None], s[:, 3:6] / np.maximum(s[:, 6:7], 1e-30), 0.0)
            assert np.all(w.mean(axis=0) <= 1.0 + 0.02), (m.kind, c, w.mean(axis=0))


def test_diffuse_closed_form(oracle_lib):
    m = abi.MaterialData(kind=abi.MAT_DIFFUSE, base_color=(0.5, 0.25, 1.0))
    wo = np.array([0, 0, 1], dtype=np.float32)
    wi = _dirs(np.array([0.3, 0.8, -0.5]), np.array([0.1, 2.0, 1.0])).astype(np.float32)
    e = pyoracle.bsdf_eval_many(m, wo, wi)
    for k in range(2):  # f * |cos| = R/pi * |cos|, pdf = |cos|/pi  (diffuse.rs:22-38)
        assert np.allclose(e[k, :3], np.array([0.5, 0.25, 1.0]) / np.pi * abs(wi[k, 2]), rtol=1e-6)
        assert np.isclose(e[k, 3], abs(wi[k, 2]) / np.pi, rtol=1e-6)
    assert np.all(e[2] == 0)  # other hemisphere


@pytest.mark.parametrize("use_nee", [1, 0])
def test_white_furnace(oracle_lib, use_nee):
    """Closed box, Lambert albedo rho, emission E everywhere: L = E * sum_{k<=D} rho^k at every pixel."""
    rho, E, D = 0.5, 1.0, 12
    sd = box_scene(albedo=rho, emission=E, width=16, height=16)
    sc = pyoracle.OracleScene(sd)
    assert sc.num_lights() == 1
    cfg = make_config(spp=256, max_depth=D, rr_depth=5, use_nee=use_nee)
    film, st = sc.render(cfg)
    img = resolve_np(film, 16, 16)
    expect = E * (1 - rho ** (D + 1)) / (1 - rho)
    assert abs(img.mean() - expect) < 0.01 * expect
    assert np.all(film[6 * 256 :] == 256)  # weight channel = spp


def test_direct_lighting_closed_form(oracle_lib):
    """max_depth = 1 on a black-walled box with one emissive ceiling: pixel radiance on the floor is the
    area-light integral rho/pi * E * int cos cos' / r^2 dA, checked against dense quadrature."""
    sd = box_scene(albedo=0.0, emission=0.0, width=8, height=8)
    # floor (y=-1) diffuse white; ceiling (y=+1) emissive; camera looks straight down from the centre
    m_floor = abi.MaterialData(kind=abi.MAT_DIFFUSE, base_color=(0.8, 0.8, 0.8))
    m_light = abi.MaterialData(kind=abi.MAT_EMISSION, emission_color=(3.0, 3.0, 3.0), emission_strength=1.0)
    m_black = abi.MaterialData(kind=abi.MAT_DIFFUSE, base_color=(0, 0, 0))
    sd.materials = [m_black, m_floor, m_light]
    mesh = sd.meshes[0]
    slots = np.zeros(12, dtype=np.uint32)
    cen = mesh.vertices[mesh.indices].mean(axis=1)
    slots[cen[:, 1] < -0.99] = 1
    slots[cen[:, 1] > 0.99] = 2
    mesh.material_slots = slots
    sd.instances[0].materials = [0, 1, 2]
    c2w = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)  # look along -y
    sd.camera.c2w = c2w.T.reshape(16).copy()
    sd.camera.fov = 0.2
    sc = pyoracle.OracleScene(sd)
    cfg = make_config(spp=4096, spp_per_pass=64, max_depth=1, filter_type=abi.FILTER_BOX, filter_radius=0.01)
    film, _ = sc.render(cfg)
    img = resolve_np(film, 8, 8)
    # quadrature at the floor centre
    n = 400
    xs = (np.arange(n) + 0.5) / n * 2 - 1
    X, Z = np.meshgrid(xs, xs)
    r2 = X**2 + Z**2 + 4.0
    integral = np.sum((2.0 / np.sqrt(r2)) ** 2 / r2) * (2.0 / n) ** 2
    expect = 0.8 / np.pi * 3.0 * integral
    centre = img[3:5, 3:5].mean()
    assert abs(centre - expect) < 0.02 * expect
