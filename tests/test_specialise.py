"""Per-scene kernels (csrc/host/specialise.cpp): the text generated from a scene's shader graphs must compute what the
interpreter computes, bit for bit. CPU side of the proof: the generated header is compiled FOR THE HOST (same headers, same flags)
and run next to the library's interpreter on the same materials and uv points; fold_inputs_fed against fold_inputs over all masks;
hiprtc cross-compiles the kernel for gfx950 without a device. The GPU side (films identical under both settings) is
tests/test_gpu_specialise.py."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from akari_render_amd import abi, build as B, capi
from tests.helpers import textured_room

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

HOST_SRC = r'''
#define AKR_SPEC_GRAPHS 1
#include "device/dtex.h"
using namespace akr;
extern "C" __attribute__((visibility("default"))) void spec_host_eval(const void* nodes, const void* images, const void* texels, const void* mat_inputs, const void* materials,
                               uint32_t material, uint32_t n, const float* uv, uint32_t* out64, float* alpha, float* emission3) {
    const TexScene ts{(const DNode*)nodes, (const DImage*)images, (const uint32_t*)texels, (const MatInputs*)mat_inputs, 0, 0};
    const DMaterial& folded = ((const DMaterial*)materials)[material];
    for (uint32_t i = 0; i < n; i++) {
        const vec2 p = mk2(uv[2 * i], uv[2 * i + 1]);
        DMaterial m = folded;
        material_at(ts, material, p, m);
        __builtin_memcpy(out64 + 64ull * i, &m, sizeof m);
        const bool tex = (folded.flags & MF_TEXTURED) != 0;
        alpha[i] = tex ? material_alpha_at(ts, folded, material, p) : folded.base_alpha;
        const vec3 e = tex ? material_emission_inputs_at(ts, folded, material, p) : folded.emission;
        emission3[3 * i] = e.x; emission3[3 * i + 1] = e.y; emission3[3 * i + 2] = e.z;
    }
}
// fold_inputs_fed(kind, fed, m, d) on top of fold_inputs(constants) against fold_inputs(m): 1 = identical records
extern "C" __attribute__((visibility("default"))) int fold_fed_matches(const float* consts26, const float* here26, uint32_t fed) {
    MatInputs c, h;
    __builtin_memcpy(&c, consts26, sizeof c);
    __builtin_memcpy(&h, here26, sizeof h);
    // `here` differs from the constants only in the fed inputs
    const uint32_t off[15] = {1, 5, 6, 7, 8, 9, 12, 13, 14, 15, 16, 19, 22, 23, 26};  // word offsets of the 14 inputs in MatInputs (base_color also owns base_alpha, word 4)
    uint32_t* cw = (uint32_t*)&c; const uint32_t* hw = (const uint32_t*)&h;
    MatInputs mixed = c;
    uint32_t* mw = (uint32_t*)&mixed;
    for (int k = 0; k < 14; k++)
        if (fed & (1u << k)) {
            for (uint32_t w = off[k]; w < off[k + 1]; w++) mw[w] = hw[w];
            if (k == 0) mw[4] = hw[4];
        }
    (void)cw;
    DMaterial full, part;
    __builtin_memset(&full, 0, sizeof full);
    __builtin_memset(&part, 0, sizeof part);
    if (!fold_inputs(mixed, full)) return -1;
    fold_inputs(c, part);
    fold_inputs_fed(c.kind, fed, mixed, part);
    return __builtin_memcmp(&full, &part, sizeof full) == 0 ? 1 : 0;
}
'''


def build_host_module(tmp_path, header: str, tag: str):
    d = tmp_path / tag
    d.mkdir()
    (d / "akr_scene_spec.h").write_text(header)
    (d / "host.hip").write_text(HOST_SRC)
    so = d / "libspec_host.so"
    cmd = ["/opt/rocm/bin/hipcc"] + [f for f in B.FLAGS if not f.startswith("--offload-arch")] + ["--cuda-host-only", "-shared", str(d / "host.hip"), "-I", B.CSRC, "-I", str(d),
                                                                                                 "-o", str(so)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-3000:]
    L = C.CDLL(str(so))
    L.spec_host_eval.restype = None
    L.fold_fed_matches.restype = C.c_int
    return L


def check_scene_on_host(L, sc: capi.Scene, n_points=64, seed=0):
    nodes, images = sc.array(capi.ARRAY_TEX_NODES, np.uint8), sc.array(capi.ARRAY_TEX_IMAGES, np.uint8)
    texels, inputs = sc.array(capi.ARRAY_TEX_TEXELS, np.uint32), sc.array(capi.ARRAY_MAT_INPUTS, np.uint8)
    mats = sc.array(capi.ARRAY_MATERIALS, np.uint32).reshape(-1, 64)
    rng = np.random.default_rng(seed)
    uv = np.concatenate([rng.uniform(-2.5, 3.5, size=(n_points - 4, 2)), [[0.0, 0.0], [1.0, 1.0], [0.5, -0.0], [np.inf, np.nan]]]).astype(np.float32)
    n_tex = 0
    for mi in range(mats.shape[0]):
        want, want_a, want_e = sc.material_folded_host(mi, uv)
        got, got_a, got_e = np.zeros_like(want), np.zeros_like(want_a), np.zeros_like(want_e)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        L.spec_host_eval(vp(nodes), vp(images), vp(texels), vp(inputs), vp(mats), C.c_uint32(mi), C.c_uint32(uv.shape[0]), vp(uv), vp(got), vp(got_a), vp(got_e))
        assert np.array_equal(want, got), f"material {mi}: folded record differs in words {np.argwhere(want != got)[:8].tolist()}"
        assert np.array_equal(want_a.view(np.uint32), got_a.view(np.uint32)), f"material {mi}: alpha differs"
        assert np.array_equal(want_e.view(np.uint32), got_e.view(np.uint32)), f"material {mi}: emission differs"
        n_tex += int((mats[mi, 1] & 0x100) != 0)
    return n_tex


def test_generated_code_is_the_interpreter_on_the_textured_room(tmp_path):
    sc = capi.Scene(None, textured_room(32, 32, alpha_cutout=True))
    src = sc.spec_source()
    assert "spec_material_at" in src and "case 4u" in src and "kSpecAbsent = 0xb" in src  # five shader kinds; no coat, transmission, glass
    L = build_host_module(tmp_path, src, "room")
    assert check_scene_on_host(L, sc) == 5


def test_materials_of_one_shape_share_a_kind_and_read_their_constants(tmp_path):
    """Two materials with the same graph shape but different constants and images (same format / filter / address): one case,
    the differing constants come from the node records, the image from its header."""
    sd = textured_room(32, 32)
    N = abi.NodeData
    rng = np.random.default_rng(5)
    sd.images.append(abi.ImageData(rng.integers(0, 256, size=(5, 7, 4), dtype=np.uint8), abi.TEX_FILTER_LINEAR, abi.TEX_REPEAT))
    def wall(scale, image):
        return abi.GraphData([N(abi.NODE_TEXCOORDS), N(abi.NODE_EXTRACT, (0, abi.FIELD_UV)), N(abi.NODE_CONST, (), (0.125, -0.25, 0.0)), N(abi.NODE_CONST, (), scale),
                              N(abi.NODE_MAPPING, (1, 2, 3, abi.MAPPING_POINT)), N(abi.NODE_IMAGE, (image, 4, 1)), N(abi.NODE_SPECTRAL_UPLIFT, (5,))], {"base_color": 6})
    sd.materials[1].graph = wall((1.5, 2.0, 1.0), 0)
    sd.materials[3].graph = wall((0.5, 3.0, 1.0), len(sd.images) - 1)
    sc = capi.Scene(None, sd)
    src = sc.spec_source()
    assert "materials 1 3" in src and "nd[3].k[0]" in src and "ts.images[nd[5].arg[0]]" in src
    assert "u2f(0x3e000000u)" in src  # the location constant both share is a literal
    L = build_host_module(tmp_path, src, "shared")
    assert check_scene_on_host(L, sc) == 5


@pytest.mark.parametrize("first", [0, 40, 80])
def test_generated_code_is_the_interpreter_on_random_graphs(tmp_path, first):
    import soak

    done = 0
    for seed in range(900000 + first, 900000 + first + 40):
        sd, _cfg = soak.rand_scene(seed, textures=True)
        try:
            sc = capi.Scene(None, sd)
        except capi.AkariError as e:
            assert e.code == capi.ERR_UNSUPPORTED, str(e)
            continue
        src = sc.spec_source()
        if not src:
            continue
        L = build_host_module(tmp_path, src, f"s{seed}")
        done += 1 if check_scene_on_host(L, sc, n_points=24, seed=seed) else 0
        if done >= 6:
            break
    assert done >= 4


def test_fold_inputs_fed_is_fold_inputs(tmp_path):
    sc = capi.Scene(None, textured_room(16, 16))
    L = build_host_module(tmp_path, sc.spec_source(), "fold")
    import soak

    rng = np.random.default_rng(1)
    n = 0
    for _ in range(400):
        a, b = soak.rand_material(rng, emissive=rng.random() < 0.3), soak.rand_material(rng, emissive=rng.random() < 0.3)
        b.kind = a.kind
        a.colorspaces = b.colorspaces = 0
        if rng.random() < 0.2:
            b.specular_tint = (float("inf"), 0.0, float("nan"))  # the non-finite tint rule of fold_inputs
        if rng.random() < 0.2:
            b.normal = (0.0, 0.0, 0.0)
        ca = np.frombuffer(bytes(a.to_struct()), dtype=np.float32).copy()
        cb = np.frombuffer(bytes(b.to_struct()), dtype=np.float32).copy()
        for fed in [int(x) for x in rng.integers(0, 1 << 14, size=12)] + [0, (1 << 14) - 1, 1, 1 << 13]:
            r = L.fold_fed_matches(ca.ctypes.data_as(C.c_void_p), cb.ctypes.data_as(C.c_void_p), C.c_uint32(fed))
            assert r == 1, (a, b, hex(fed))
            n += 1
    assert n == 400 * 16


def test_kernel_compiles_for_gfx950_without_a_device():
    for n_floor, bvh in ((1, False), (3, True)):
        sc = capi.Scene(None, textured_room(32, 32, n_floor=n_floor, alpha_cutout=True))
        for waves in (3, 4):
            assert sc.spec_compile(bvh=bvh, pmj=False, stage=True, defer=True, min_waves=waves) > 20000
    assert sc.spec_compile(bvh=True, pmj=True, stage=False, defer=False) > 20000


def test_kernel_of_a_kept_scene_compiles():
    """Round 6: scenes kept as meshes + instances get per-scene kernels too (pt_pass_body<.., INST> wrapped by the same generated text)."""
    from tests.helpers import instanced_scene
    with capi.options(instancing=1):
        sc = capi.Scene(None, instanced_scene(textured=True, alpha=True))
    assert sc.info().uses_bvh == 2 and sc.spec_source() != ""
    assert sc.spec_compile(bvh=True, pmj=False, stage=False, defer=False, inst=True) > 20000


def test_no_per_scene_code_without_textures(cbox_path):
    sc = capi.Scene(None, cbox_path, 32, 32)
    assert sc.spec_source() == ""
    with pytest.raises(capi.AkariError) as e:
        sc.spec_compile()
    assert e.value.code == capi.ERR_UNSUPPORTED
