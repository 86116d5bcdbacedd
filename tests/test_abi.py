"""The C-ABI library loads without a GPU and exports every symbol include/akari_hip.h declares (and, in the test build, every
test hook of include/akari_hip_test.h)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from akari_render_amd import abi, capi


def _declared(root, header="akari_hip.h", macro="AKR_API"):
    text = open(os.path.join(root, "include", header)).read()
    return sorted(set(re.findall(macro + r"\s+[\w\s\*]+?\b(akr_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported(hip_lib, root):
    """Two symbol sets: the drop-in boundary (include/akari_hip.h, what INTEGRATION.md binds) and the test hooks
    (include/akari_hip_test.h, only in a library built with -DAKR_TEST_HOOKS=1 -- the in-tree test build)."""
    names = _declared(root)
    assert len(names) >= 40
    for n in names:
        assert hasattr(hip_lib, n), f"libakari_hip.so does not export {n}"
    # and the binding lists exactly the header's symbols
    assert sorted(capi.EXPORTS) == names
    hooks = _declared(root, "akari_hip_test.h", "AKR_TEST_API")
    assert sorted(capi.TEST_EXPORTS) == hooks and len(hooks) >= 20
    assert not set(hooks) & set(names)
    # nothing that looks like a hook is left in the public header (the spec compile entry points are the JIT's, used by akari-cli)
    assert [n for n in names if n.startswith("akr_probe_") or (n.startswith("akr_host_") and "spec_compile" not in n)] == []
    have = [hasattr(hip_lib, n) for n in hooks]
    assert all(have) or not any(have), "the library exports some test hooks and not others"
    from akari_render_amd import build
    assert all(have) == build.TEST_HOOKS


def test_a_shipping_build_leaves_the_hooks_out(root):
    """api_probe.cpp compiles to nothing without AKR_TEST_HOOKS (what AKR_SHIP=1 builds): checked on the preprocessed source."""
    import subprocess
    src = os.path.join(root, "akari_render_amd", "csrc", "host", "api_probe.cpp")
    out = subprocess.run(["g++", "-E", "-P", "-x", "c++", src], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stderr[-500:]
    assert "akr_" not in out.stdout


def test_struct_sizes_match_header(hip_lib):
    # akr_pt_config_default writes sizeof(akr_pt_config) bytes: guard bytes after the ctypes struct must survive
    class Guarded(C.Structure):
        _fields_ = [("cfg", abi.PtConfig), ("guard", C.c_uint8 * 64)]
    g = Guarded()
    for i in range(64):
        g.guard[i] = 0xAB
    assert hip_lib.akr_pt_config_default(C.byref(g.cfg)) == 0
    assert all(b == 0xAB for b in g.guard)
    d = abi.PtConfig.default()
    assert bytes(g.cfg) == bytes(d)
    assert C.sizeof(abi.MaterialDesc) == 4 * 26 and C.sizeof(abi.PtConfig) == 88
    # akr_struct_size: what the library was built with, per struct (capi.lib() compares every one with abi.py when it loads)
    hip_lib.akr_struct_size.restype = C.c_uint32
    assert hip_lib.akr_struct_size(6) == 88 and hip_lib.akr_struct_size(3) == 104 and hip_lib.akr_struct_size(0) == 0 and hip_lib.akr_struct_size(99) == 0
    assert hip_lib.akr_struct_size(8) == C.sizeof(abi.SceneInfo) and hip_lib.akr_struct_size(9) == C.sizeof(abi.KernelInfo)
    assert hip_lib.akr_version().startswith(b"akari_hip 0.3.")


def test_no_cpu_fallback(hip_lib):
    """Without a GPU the library must say so; with one this test is a no-op."""
    h = C.c_void_p()
    rc = hip_lib.akr_context_create(0, C.byref(h))
    if rc == 0:
        hip_lib.akr_context_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == capi.ERR_NO_DEVICE
    assert b"no CPU path" in hip_lib.akr_last_error()


def test_error_reporting(hip_lib, tmp_path):
    h = C.c_void_p()
    assert hip_lib.akr_scene_load(None, b"/nonexistent/scene.json", 0, 0, C.byref(h)) == capi.ERR_IO
    bad = tmp_path / "bad.json"
    bad.write_text("{ not json")
    assert hip_lib.akr_scene_load(None, str(bad).encode(), 0, 0, C.byref(h)) == capi.ERR_PARSE
    assert hip_lib.akr_scene_create(None, None, C.byref(h)) == capi.ERR_INVALID_ARGUMENT
    cfg = abi.PtConfig()
    assert hip_lib.akr_pt_config_from_json(b'{"method": {"type": "mcmc"}}', C.byref(cfg), None, 0) == capi.ERR_UNSUPPORTED
    assert hip_lib.akr_pt_config_from_json(b'{"sampler": {"type": "sobol", "seed": 7}}', C.byref(cfg), None, 0) == 0
    assert cfg.sampler_type == abi.SAMPLER_SOBOL and cfg.sampler_seed == 7
    assert hip_lib.akr_pt_config_from_json(b'{"sampler": {"type": "halton", "seed": 0}}', C.byref(cfg), None, 0) == capi.ERR_PARSE
    # ColorPipeline (color.rs:663-676): srgb | aces for both members; spectral is todo!() in the reference too
    assert hip_lib.akr_pt_config_from_json(b'{"color": {"color_repr": {"type": "rgb"}, "rgb_colorspace": "aces"}}', C.byref(cfg), None, 0) == 0
    assert cfg.color == abi.COLOR_RGB_ACESCG
    assert hip_lib.akr_pt_config_from_json(b'{"color": {"color_repr": {"type": "rgb", "colorspace": "aces"}, "rgb_colorspace": "aces"}}', C.byref(cfg), None, 0) == 0
    assert cfg.color == abi.COLOR_RGB_ACESCG | abi.COLOR_REPR_ACESCG
    assert hip_lib.akr_pt_config_from_json(b'{"color": {"color_repr": "rgb_aces"}}', C.byref(cfg), None, 0) == 0 and cfg.color == abi.COLOR_REPR_ACESCG
    assert hip_lib.akr_pt_config_from_json(b'{"color": {"color_repr": {"type": "spectral"}}}', C.byref(cfg), None, 0) == capi.ERR_UNSUPPORTED
    assert hip_lib.akr_pt_config_from_json(b'{"color": {"rgb_colorspace": "xyz"}}', C.byref(cfg), None, 0) == capi.ERR_PARSE
    assert hip_lib.akr_pt_config_from_json(b'{"method": {"type": "aov"}, "color": {"rgb_colorspace": "aces"}}', C.byref(cfg), None, 0) == capi.ERR_UNSUPPORTED
    assert hip_lib.akr_pt_config_from_json(b'{"color": {"rgb_colorspace": "srgb"}, "film": {"color": "srgb"}}', C.byref(cfg), None, 0) == 0
    assert hip_lib.akr_pt_config_from_json(b'{"film": {"color": "xyz"}}', C.byref(cfg), None, 0) == capi.ERR_UNSUPPORTED
    assert len(hip_lib.akr_last_error()) > 0


def test_method_json_defaults_and_overrides(hip_lib, root):
    cfg, out = capi.config_from_json("{}")
    assert bytes(cfg) == bytes(abi.PtConfig.default())
    assert out == "out.exr"  # FilmConfig::default (akari_integrator/src/lib.rs:82-90)
    text = open(os.path.join(root, "scenes", "cbox", "pt.json")).read().replace("pmj02bn", "independent")
    cfg, out = capi.config_from_json(text)
    assert (cfg.spp, cfg.max_depth, cfg.rr_depth, cfg.spp_per_pass) == (4096, 12, 5, 64)
    assert (cfg.use_nee, cfg.force_diffuse, cfg.indirect_only) == (1, 0, 0)
    assert cfg.filter_type == abi.FILTER_GAUSSIAN and abs(cfg.filter_radius - 1.5) < 1e-7
    assert out == "output/pt.exr"
    cfg, _ = capi.config_from_json('[{"method": {"type": "pt", "spp": 3, "pixel_offset": [1, -2], "debug_depth": 2}, "film": {"filter": {"type": "box", "radius": 0.5}}}]')
    assert cfg.spp == 3 and list(cfg.pixel_offset) == [1, -2] and cfg.debug_depth == 2 and cfg.filter_type == abi.FILTER_BOX


def test_aov_method_json_and_config(hip_lib):
    """Method::NormalVis ("type": "aov", aov.rs:9-39): accepted by the render-task parser, rejected by the pt-config entry point."""
    c = abi.AovConfig()
    assert hip_lib.akr_aov_config_default(C.byref(c)) == 0
    assert bytes(c) == bytes(abi.AovConfig.default()) and (c.spp, c.aov, c.remap) == (256, abi.AOV_NS, 1)
    cfg = abi.PtConfig()
    assert hip_lib.akr_pt_config_from_json(b'{"method": {"type": "aov", "aov": "ng"}}', C.byref(cfg), None, 0) == capi.ERR_UNSUPPORTED
    assert C.sizeof(abi.AovConfig) == 56


def test_json_nesting_limit(hip_lib):
    """A file of two million brackets is refused (serde_json's 128-level recursion limit), not a stack overflow."""
    cfg = abi.PtConfig()
    for doc in (b"[" * 2_000_000, b'{"a":' * 2_000_000, b"[" * 129 + b"]" * 129):
        assert hip_lib.akr_pt_config_from_json(doc, C.byref(cfg), None, 0) == capi.ERR_PARSE
    assert hip_lib.akr_pt_config_from_json(b'{"x": ' + b"[" * 100 + b"]" * 100 + b"}", C.byref(cfg), None, 0) == 0
