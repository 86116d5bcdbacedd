"""ctypes binding of libakari_hip.so (the C ABI in include/akari_hip.h).

This is the only way Python reaches the path tracer: there is no Python or PyTorch fallback. If the shared
library is missing it is built with hipcc (akari_render_amd/build.py); if there is no GPU, `Context()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import abi
from .build import LIB, build

AKR_OK = 0
FILM_PLANE_RGB, FILM_PLANE_SPLAT, FILM_PLANE_WEIGHT, FILM_PLANES_PT, FILM_PLANES_ALL = 1, 2, 4, 5, 7
ERR_INVALID_ARGUMENT, ERR_HIP, ERR_NO_DEVICE, ERR_IO, ERR_PARSE, ERR_UNSUPPORTED, ERR_OOM, ERR_RENDER = -1, -2, -3, -4, -5, -6, -7, -8

(ARRAY_WOOP, ARRAY_TRI_GID, ARRAY_SHADE, ARRAY_INSTANCES, ARRAY_MATERIALS, ARRAY_BVH_NODES, ARRAY_LIGHT_ENTRIES,
 ARRAY_LIGHT_PDF, ARRAY_AREA_ENTRIES, ARRAY_AREA_PDF, ARRAY_INST_TRI_OFFSET, ARRAY_R2C, ARRAY_C2W,
 ARRAY_TEX_NODES, ARRAY_TEX_IMAGES, ARRAY_TEX_TEXELS, ARRAY_MAT_INPUTS,
 ARRAY_INST_LEAVES, ARRAY_MESH_TRIS, ARRAY_MESH_POS, ARRAY_MESH_META, ARRAY_MESH_NORMALS) = range(22)

# every symbol include/akari_hip.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    "akr_last_error", "akr_version", "akr_struct_size", "akr_option_set", "akr_option_get", "akr_context_create",
    "akr_context_destroy", "akr_context_synchronize", "akr_context_device_info", "akr_scene_create", "akr_scene_load",
    "akr_scene_destroy", "akr_scene_set_resolution", "akr_scene_get_info", "akr_scene_get_light", "akr_scene_get_ggx_table",
    "akr_scene_get_desc_counts", "akr_scene_get_mesh", "akr_scene_get_instance", "akr_scene_get_material", "akr_scene_get_camera",
    "akr_scene_get_array", "akr_scene_get_image_count", "akr_scene_get_image", "akr_scene_get_material_graph", "akr_film_create",
    "akr_film_wrap", "akr_film_destroy", "akr_film_clear", "akr_film_read", "akr_film_write", "akr_film_resolve",
    "akr_film_device_ptr", "akr_pt_config_default", "akr_pt_config_from_json", "akr_pt_render", "akr_pt_begin", "akr_pt_passes",
    "akr_pt_end", "akr_pt_get_stats", "akr_render_task", "akr_image_write", "akr_aov_config_default", "akr_aov_render",
    "akr_gpt_config_default", "akr_gpt_render", "akr_gpt_begin", "akr_gpt_sample", "akr_gpt_sums", "akr_gpt_sums_read",
    "akr_gpt_sums_write", "akr_gpt_finish", "akr_gpt_abort", "akr_gpt_reduce", "akr_mcmc_config_default", "akr_mcmc_render",
    "akr_film_set_splat_scale", "akr_film_get_splat_scale", "akr_pt_read_sampler_states", "akr_context_device_ordinal",
    "akr_device_count", "akr_comm_unique_id", "akr_comm_create", "akr_comm_wrap", "akr_comm_destroy", "akr_film_reduce",
    "akr_pt_kernel_info", "akr_scene_spec_source", "akr_host_spec_compile", "akr_host_spec_compile_text",
    "akr_film_reduce_planes", "akr_mcmc_render_shard", "akr_mcmc_combine_host", "akr_mcmc_combine",
]
# include/akari_hip_test.h: the test hooks (compiled into the in-tree test build, absent from a build with AKR_SHIP=1)
TEST_EXPORTS = [
    "akr_probe_material_inputs_host", "akr_host_stdrng_u64", "akr_host_chacha_block", "akr_host_pcg32_states",
    "akr_host_pcg_start", "akr_host_alias_table", "akr_probe_math", "akr_probe_bsdf", "akr_probe_intersect",
    "akr_probe_surface_interaction", "akr_probe_material_inputs", "akr_host_decode_png", "akr_host_decode_jpeg",
    "akr_host_decode_exr", "akr_host_decode_tiff", "akr_host_decode_dds", "akr_host_pmj02bn_tables",
    "akr_probe_material_folded_host", "akr_host_sobol_dim1", "akr_host_fastmod", "akr_host_tri_pretest",
]


class RenderSession(C.Structure):
    """akr_render_session = RenderSession of the reference (akari_integrator/src/lib.rs:8-23) minus the GUI channel."""

    _fields_ = [("save_intermediate", C.c_int32), ("save_stats", C.c_int32), ("name", C.c_char_p),
                ("override_sampler_independent", C.c_int32), ("verbose", C.c_int32)]


class AkariError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"akari_hip error {code}: {msg}")
        self.code = code


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Loads (building first if needed) libakari_hip.so and declares the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    # AKR_HIP_LIB: an A/B build of the same library (build.build_variant) for measurements; the product is LIB
    L = C.CDLL(os.environ.get("AKR_HIP_LIB") or LIB)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float
    fp, up, u64p, vpp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)
    L.akr_last_error.restype = C.c_char_p
    L.akr_version.restype = C.c_char_p
    # the library's structs against these bindings' (a stale libakari_hip.so or a stale abi.py fails here, not inside a render)
    L.akr_struct_size.restype = C.c_uint32
    L.akr_struct_size.argtypes = [C.c_int32]
    for sid, cls in enumerate((abi.MeshDesc, abi.InstanceDesc, abi.MaterialDesc, abi.CameraDesc, abi.SceneDesc, abi.PtConfig, abi.PtStats, abi.SceneInfo,
                               abi.KernelInfo, abi.AovConfig, abi.GptConfig, abi.McmcConfig, abi.McmcResult, abi.McmcPartial), start=1):
        if L.akr_struct_size(sid) != C.sizeof(cls):
            raise ImportError("libakari_hip.so and akari_render_amd/abi.py disagree on sizeof(%s): %d vs %d" % (cls.__name__, L.akr_struct_size(sid), C.sizeof(cls)))

    def proto(name, *args):
        if name in TEST_EXPORTS and not hasattr(L, name):
            return  # a library built without the test hooks (AKR_SHIP=1)
        fn = getattr(L, name)
        fn.restype = i32
        fn.argtypes = list(args)

    proto("akr_context_create", i32, vpp)
    proto("akr_context_destroy", vp)
    proto("akr_context_synchronize", vp)
    proto("akr_context_device_info", vp, C.c_char_p, u32, up, u64p)
    proto("akr_scene_create", vp, C.POINTER(abi.SceneDesc), vpp)
    proto("akr_scene_load", vp, C.c_char_p, u32, u32, vpp)
    proto("akr_scene_destroy", vp)
    proto("akr_scene_set_resolution", vp, u32, u32)
    proto("akr_scene_get_info", vp, C.POINTER(abi.SceneInfo))
    proto("akr_scene_get_light", vp, u32, up, fp, fp)
    proto("akr_scene_get_ggx_table", vp, fp)
    proto("akr_scene_get_desc_counts", vp, up, up, up)
    proto("akr_scene_get_mesh", vp, u32, C.POINTER(abi.MeshDesc))
    proto("akr_scene_get_instance", vp, u32, C.POINTER(abi.InstanceDesc))
    proto("akr_scene_get_material", vp, u32, C.POINTER(abi.MaterialDesc))
    proto("akr_scene_get_camera", vp, C.POINTER(abi.CameraDesc))
    proto("akr_scene_get_array", vp, i32, vpp, u64p)
    proto("akr_scene_get_image_count", vp, up)
    proto("akr_scene_get_image", vp, u32, C.POINTER(abi.ImageDesc))
    proto("akr_scene_get_material_graph", vp, u32, C.POINTER(abi.MaterialGraph))
    proto("akr_film_create", vp, u32, u32, vpp)
    proto("akr_film_wrap", vp, u32, u32, vp, vpp)
    proto("akr_film_destroy", vp)
    proto("akr_film_clear", vp)
    proto("akr_film_read", vp, fp)
    proto("akr_film_write", vp, fp)
    proto("akr_film_resolve", vp, fp)
    proto("akr_film_device_ptr", vp, vpp, u64p)
    proto("akr_pt_kernel_info", vp, C.POINTER(abi.KernelInfo))
    proto("akr_scene_spec_source", vp, C.c_char_p, u64, u64p)
    proto("akr_host_spec_compile", vp, u32, u32, C.c_char_p, u64p, C.c_char_p, u32)
    proto("akr_probe_material_folded_host", vp, u32, u32, fp, up, fp, fp)
    proto("akr_host_sobol_dim1", u32, up, up, up)
    proto("akr_host_fastmod", u32, up, up, up)
    proto("akr_host_tri_pretest", u32, fp, fp, f32, up, up, fp)
    proto("akr_host_spec_compile_text", C.c_char_p, u32, u32, C.c_char_p, C.c_char_p)
    proto("akr_context_device_ordinal", vp, C.POINTER(C.c_int32))
    proto("akr_device_count", C.POINTER(C.c_int32))
    proto("akr_option_set", C.c_char_p, i32)
    proto("akr_option_get", C.c_char_p, C.POINTER(C.c_int32))
    proto("akr_comm_unique_id", C.POINTER(C.c_uint8))
    proto("akr_comm_create", vp, C.POINTER(C.c_uint8), i32, i32, vpp)
    proto("akr_comm_wrap", vp, vp, i32, i32, vpp)
    proto("akr_comm_destroy", vp)
    proto("akr_film_reduce", vp, vp, i32, i32)
    proto("akr_film_reduce_planes", vp, vp, i32, i32, u32)
    proto("akr_mcmc_render_shard", vp, vp, C.POINTER(abi.McmcConfig), u32, u32, vp, C.POINTER(abi.McmcPartial), up, C.POINTER(abi.PtStats))
    proto("akr_mcmc_combine_host", vp, C.POINTER(abi.McmcPartial), u32, C.POINTER(abi.McmcResult))
    proto("akr_mcmc_combine", vp, vp, i32, C.POINTER(abi.McmcPartial), C.POINTER(abi.McmcResult))
    proto("akr_pt_config_default", C.POINTER(abi.PtConfig))
    proto("akr_aov_config_default", C.POINTER(abi.AovConfig))
    proto("akr_aov_render", vp, vp, C.POINTER(abi.AovConfig), vp, C.POINTER(abi.PtStats))
    proto("akr_gpt_config_default", C.POINTER(abi.GptConfig))
    proto("akr_gpt_render", vp, vp, C.POINTER(abi.GptConfig), vp, fp, C.POINTER(abi.PtStats))
    proto("akr_gpt_begin", vp, vp, C.POINTER(abi.GptConfig), vp, vp, vpp)
    proto("akr_gpt_sample", vp, u32, i32)
    proto("akr_gpt_sums", vp, C.POINTER(fp), u64p)
    proto("akr_gpt_sums_read", vp, fp)
    proto("akr_gpt_sums_write", vp, fp)
    proto("akr_gpt_finish", vp, fp, C.POINTER(abi.PtStats))
    proto("akr_gpt_abort", vp, C.POINTER(abi.PtStats))
    proto("akr_gpt_reduce", vp, vp, i32, i32)
    proto("akr_mcmc_config_default", C.POINTER(abi.McmcConfig))
    proto("akr_mcmc_render", vp, vp, C.POINTER(abi.McmcConfig), vp, C.POINTER(abi.McmcResult), up, C.POINTER(abi.PtStats))
    proto("akr_film_set_splat_scale", vp, C.c_float)
    proto("akr_film_get_splat_scale", vp, fp)
    proto("akr_pt_config_from_json", C.c_char_p, C.POINTER(abi.PtConfig), C.c_char_p, u32)
    proto("akr_pt_render", vp, vp, C.POINTER(abi.PtConfig), vp, C.POINTER(abi.PtStats))
    proto("akr_pt_begin", vp, vp, C.POINTER(abi.PtConfig), vp, vpp)
    proto("akr_pt_passes", vp, u32, i32, up)
    proto("akr_pt_end", vp, C.POINTER(abi.PtStats))
    proto("akr_pt_get_stats", vp, C.POINTER(abi.PtStats))
    proto("akr_render_task", vp, vp, C.c_char_p, C.POINTER(RenderSession), C.POINTER(abi.PtStats))
    proto("akr_image_write", C.c_char_p, fp, u32, u32)
    proto("akr_pt_read_sampler_states", vp, u64p)
    proto("akr_host_stdrng_u64", u64, u32, u64p)
    proto("akr_host_chacha_block", up, u64, u64, i32, up)
    proto("akr_host_pcg32_states", u64, u64, u64p)
    proto("akr_host_pcg_start", u64p, u64)
    proto("akr_host_alias_table", fp, u32, up, fp, fp)
    proto("akr_probe_math", vp, u32, fp, fp, fp, fp)
    proto("akr_probe_bsdf", vp, C.POINTER(abi.MaterialDesc), fp, i32, fp, u32, fp, fp)
    proto("akr_probe_intersect", vp, vp, u32, fp, up, fp)
    proto("akr_probe_surface_interaction", vp, vp, u32, up, fp, fp)
    proto("akr_probe_material_inputs", vp, vp, u32, u32, fp, fp)
    proto("akr_probe_material_inputs_host", vp, u32, u32, u32, fp, fp)
    proto("akr_host_decode_png", C.c_char_p, u64, up, up, C.POINTER(C.c_uint8), u64)
    proto("akr_host_decode_jpeg", C.c_char_p, u64, up, up, C.POINTER(C.c_uint8), u64)
    proto("akr_host_decode_exr", C.c_char_p, u64, up, up, fp, u64)
    proto("akr_host_decode_tiff", C.c_char_p, u64, up, up, C.POINTER(C.c_uint8), u64)
    proto("akr_host_decode_dds", C.c_char_p, u64, up, up, C.POINTER(C.c_uint8), u64)
    proto("akr_host_pmj02bn_tables", up, C.POINTER(C.c_uint16))
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != AKR_OK:
        raise AkariError(rc, lib().akr_last_error().decode("utf-8", "replace"))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class Context:
    """akr_context: one HIP device + stream. Raises AkariError(ERR_NO_DEVICE) without a GPU."""

    def __init__(self, device: int = 0):
        self.h = C.c_void_p()
        check(lib().akr_context_create(device, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().akr_context_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib().akr_context_synchronize(self.h))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_uint32(), C.c_uint64()
        check(lib().akr_context_device_info(self.h, name, 256, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}


class Scene:
    """akr_scene. ctx=None compiles on the host only (inspectable, not renderable)."""

    def __init__(self, ctx: Optional[Context], source, width: int = 0, height: int = 0):
        self.ctx = ctx
        self.h = C.c_void_p()
        ch = ctx.h if ctx is not None else C.c_void_p()
        if isinstance(source, (str, os.PathLike)):
            check(lib().akr_scene_load(ch, os.fspath(source).encode(), width, height, C.byref(self.h)))
        else:
            desc, keep = source.to_desc()
            if width and height:
                desc.camera.width, desc.camera.height = width, height
            check(lib().akr_scene_create(ch, C.byref(desc), C.byref(self.h)))
            del keep

    def close(self):
        if self.h:
            lib().akr_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> abi.SceneInfo:
        i = abi.SceneInfo()
        check(lib().akr_scene_get_info(self.h, C.byref(i)))
        return i

    def spec_source(self) -> str:
        """akr_scene_spec_source: the kernel text generated for the scene's shader kinds ("" without texture-fed materials)."""
        n = C.c_uint64()
        check(lib().akr_scene_spec_source(self.h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value + 1)
        check(lib().akr_scene_spec_source(self.h, buf, n.value + 1, C.byref(n)))
        return buf.value.decode()

    def spec_compile(self, bvh: bool = False, pmj: bool = False, stage: bool = True, defer: bool = False, min_waves: int = 3, arch: str = "gfx950", inst: bool = False) -> int:
        """akr_host_spec_compile: hiprtc-compiles the scene's per-scene kernel (no device needed); returns the code object's size."""
        nbytes = C.c_uint64()
        log = C.create_string_buffer(4096)
        flags = (1 if bvh else 0) | (2 if pmj else 0) | (4 if stage else 0) | (8 if defer else 0) | (16 if inst else 0)
        check(lib().akr_host_spec_compile(self.h, flags, min_waves, arch.encode(), C.byref(nbytes), log, 4096))
        return nbytes.value

    def material_folded_host(self, material: int, uv: np.ndarray):
        """akr_probe_material_folded_host: (folded records u32[n, 64], alpha f32[n], emission f32[n, 3]) of the interpreter, on the host."""
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        n = uv.shape[0]
        out, alpha, em = np.zeros((n, 64), np.uint32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
        check(lib().akr_probe_material_folded_host(self.h, material, n, _fp(uv), out.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(alpha), _fp(em)))
        return out, alpha, em

    def set_resolution(self, w: int, h: int):
        check(lib().akr_scene_set_resolution(self.h, w, h))

    def light(self, i: int):
        inst, power, pdf = C.c_uint32(), C.c_float(), C.c_float()
        check(lib().akr_scene_get_light(self.h, i, C.byref(inst), C.byref(power), C.byref(pdf)))
        return inst.value, power.value, pdf.value

    def ggx_table(self) -> np.ndarray:
        t = np.zeros(4096, dtype=np.float32)
        check(lib().akr_scene_get_ggx_table(self.h, _fp(t)))
        return t

    def array(self, which: int, dtype) -> np.ndarray:
        p, n = C.c_void_p(), C.c_uint64()
        check(lib().akr_scene_get_array(self.h, which, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * n.value).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).copy()

    def to_scene_data(self) -> abi.SceneData:
        """The flattened description the library holds (what akr_scene_load produced)."""
        nm, ni, nmat = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().akr_scene_get_desc_counts(self.h, C.byref(nm), C.byref(ni), C.byref(nmat)))
        meshes, instances, materials = [], [], []

        def arr(ptr, n, dtype):
            if not ptr or n == 0:
                return None
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

        for i in range(nm.value):
            m = abi.MeshDesc()
            check(lib().akr_scene_get_mesh(self.h, i, C.byref(m)))
            nt = m.n_triangles
            uv = arr(m.uvs, 6 * nt, np.float32)
            nr = arr(m.normals, 9 * nt, np.float32)
            tg = arr(m.tangents, 9 * nt, np.float32)
            meshes.append(abi.MeshData(
                vertices=arr(m.vertices, 3 * m.n_vertices, np.float32).reshape(-1, 3),
                indices=arr(m.indices, 3 * nt, np.uint32).reshape(-1, 3),
                uvs=None if uv is None else uv.reshape(-1, 3, 2),
                normals=None if nr is None else nr.reshape(-1, 3, 3),
                tangents=None if tg is None else tg.reshape(-1, 3, 3),
                material_slots=arr(m.material_slots, nt, np.uint32)))
        for i in range(ni.value):
            d = abi.InstanceDesc()
            check(lib().akr_scene_get_instance(self.h, i, C.byref(d)))
            instances.append(abi.InstanceData(d.mesh, [int(d.materials[k]) for k in range(d.n_materials)],
                                              np.array(list(d.transform), dtype=np.float32)))
        for i in range(nmat.value):
            d = abi.MaterialDesc()
            check(lib().akr_scene_get_material(self.h, i, C.byref(d)))
            md = abi.MaterialData()
            md.kind = d.kind & abi.MAT_KIND_MASK
            md.colorspaces = d.kind & 0xF00
            for name in ("base_color", "specular_tint", "coat_tint", "emission_color", "normal"):
                setattr(md, name, tuple(float(x) for x in getattr(d, name)))
            for name in ("base_alpha", "metallic", "roughness", "ior", "specular_ior_level", "transmission_weight",
                         "coat_weight", "coat_roughness", "coat_ior", "emission_strength"):
                setattr(md, name, float(getattr(d, name)))
            g = abi.MaterialGraph()
            check(lib().akr_scene_get_material_graph(self.h, i, C.byref(g)))
            if g.n_nodes:
                nodes = [abi.NodeData(int(g.nodes[j].op), tuple(int(a) for a in g.nodes[j].arg), tuple(float(x) for x in g.nodes[j].k))
                         for j in range(g.n_nodes)]
                inputs = {abi.INPUT_NAMES[k]: int(g.input[k]) for k in range(abi.IN_COUNT) if g.input[k] != abi.NODE_NONE}
                md.graph = abi.GraphData(nodes, inputs)
            materials.append(md)
        images = []
        nimg = C.c_uint32()
        check(lib().akr_scene_get_image_count(self.h, C.byref(nimg)))
        for i in range(nimg.value):
            d = abi.ImageDesc()
            check(lib().akr_scene_get_image(self.h, i, C.byref(d)))
            n = d.width * d.height * 4
            if d.format == abi.IMAGE_RGBA8:
                t = np.frombuffer((C.c_uint8 * n).from_address(d.texels), dtype=np.uint8).copy()
            else:
                t = np.frombuffer((C.c_float * n).from_address(d.texels), dtype=np.float32).copy()
            images.append(abi.ImageData(t.reshape(d.height, d.width, 4), d.filter, d.address))
        c = abi.CameraDesc()
        check(lib().akr_scene_get_camera(self.h, C.byref(c)))
        cam = abi.CameraData(np.array(list(c.c2w), dtype=np.float32), float(c.fov), c.width, c.height)
        return abi.SceneData(meshes, instances, materials, cam, images=images)


def set_option(name: str, value: int) -> None:
    """akr_option_set: process-wide tuning switches / test hooks ("force_bvh", "bvh_balanced", "defer_metal", "defer_on", "wavefront", "simple_kernels")."""
    check(lib().akr_option_set(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_int32()
    check(lib().akr_option_get(name.encode(), C.byref(v)))
    return v.value


class options:
    """with capi.options(force_bvh=1, wavefront=1): ...   -- sets the options, restores the previous values on exit."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def last_error() -> str:
    """akr_last_error(): the calling thread's last failure message ("" if none)."""
    return lib().akr_last_error().decode("utf-8", "replace")


def device_count() -> int:
    n = C.c_int32()
    check(lib().akr_device_count(C.byref(n)))
    return n.value


def comm_unique_id() -> bytes:
    """akr_comm_unique_id: the 128 bytes rank 0 hands to the other ranks (ncclGetUniqueId)."""
    buf = (C.c_uint8 * 128)()
    check(lib().akr_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """akr_comm: an RCCL communicator of one process per GPU, for akr_film_reduce."""

    def __init__(self, ctx: "Context", unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        self.h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(lib().akr_comm_create(ctx.h, buf, rank, world, C.byref(self.h)))
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "h", None):
            lib().akr_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def mcmc_combine(self, film: "Film", partial, root: int = 0) -> dict:
        """akr_mcmc_combine: film reduce (all planes) + all-reduce of the normalisation sums over RCCL."""
        res = abi.McmcResult()
        check(lib().akr_mcmc_combine(film.h, self.h, root, C.byref(partial), C.byref(res)))
        return {k: getattr(res, k) for k, _ in abi.McmcResult._fields_ if k != "_pad"}

    def reduce_film(self, film: "Film", root: int = 0, blocking: bool = True, planes: int = 7):
        """Sum of the ranks' films in place, onto `root` (or onto every rank with root = -1). planes: FILM_PLANES_PT (rgb + weight:
        what a pt / aov film holds, 4 N floats) or FILM_PLANES_ALL (7 N; gpt / mcmc_opt splats)."""
        check(lib().akr_film_reduce_planes(film.h, self.h, root, 1 if blocking else 0, planes))


class Film:
    """akr_film: f32[7*W*H] on the device in the reference layout [rgb*N | splat*N | weight*N]."""

    def __init__(self, ctx: Context, width: int, height: int, device_ptr: Optional[int] = None):
        self.ctx, self.width, self.height = ctx, width, height
        self.h = C.c_void_p()
        if device_ptr is None:
            check(lib().akr_film_create(ctx.h, width, height, C.byref(self.h)))
        else:
            check(lib().akr_film_wrap(ctx.h, width, height, C.c_void_p(device_ptr), C.byref(self.h)))

    def close(self):
        if self.h:
            lib().akr_film_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        check(lib().akr_film_clear(self.h))

    def read(self) -> np.ndarray:
        out = np.zeros(7 * self.width * self.height, dtype=np.float32)
        check(lib().akr_film_read(self.h, _fp(out)))
        return out

    def write(self, data: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.float32)
        assert data.size == 7 * self.width * self.height
        check(lib().akr_film_write(self.h, _fp(data)))

    def resolve(self) -> np.ndarray:
        out = np.zeros(3 * self.width * self.height, dtype=np.float32)
        check(lib().akr_film_resolve(self.h, _fp(out)))
        return out.reshape(self.height, self.width, 3)

    @property
    def splat_scale(self) -> float:
        v = C.c_float()
        check(lib().akr_film_get_splat_scale(self.h, C.byref(v)))
        return v.value

    @splat_scale.setter
    def splat_scale(self, scale: float):
        check(lib().akr_film_set_splat_scale(self.h, scale))

    def device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        check(lib().akr_film_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value


class PtSession:
    """akr_pt_begin / akr_pt_passes / akr_pt_end (the reference's `while cnt < spp` loop, pt.rs:1126-1149)."""

    def __init__(self, ctx: Context, scene: Scene, cfg: abi.PtConfig, film: Film):
        self.ctx = ctx
        self.h = C.c_void_p()
        self._cfg = cfg.copy()
        check(lib().akr_pt_begin(ctx.h, scene.h, C.byref(self._cfg), film.h, C.byref(self.h)))

    def passes(self, n: int = 1, blocking: bool = False) -> int:
        done = C.c_uint32()
        check(lib().akr_pt_passes(self.h, n, 1 if blocking else 0, C.byref(done)))
        return done.value

    def sampler_states(self, n_pixels: int) -> np.ndarray:
        st = np.zeros(2 * n_pixels, dtype=np.uint64)
        check(lib().akr_pt_read_sampler_states(self.h, st.ctypes.data_as(C.POINTER(C.c_uint64))))
        return st

    def stats(self) -> dict:
        st = abi.PtStats()
        check(lib().akr_pt_get_stats(self.h, C.byref(st)))
        return st.as_dict()

    def kernel_info(self) -> dict:
        """akr_pt_kernel_info: per-scene kernel (hiprtc) or interpreter, compile / load times, why."""
        ki = abi.KernelInfo()
        ki.struct_size = C.sizeof(abi.KernelInfo)
        check(lib().akr_pt_kernel_info(self.h, C.byref(ki)))
        return ki.as_dict()

    def end(self) -> dict:
        st = abi.PtStats()
        h, self.h = self.h, C.c_void_p()
        check(lib().akr_pt_end(h, C.byref(st)))
        return st.as_dict()

    def __del__(self):
        try:
            if self.h:
                lib().akr_pt_end(self.h, None)
        except Exception:
            pass


def pt_render(ctx: Context, scene: Scene, cfg: abi.PtConfig, film: Film) -> dict:
    """pt::render (pt.rs:1161-1172): all passes, blocking; returns the device counters."""
    st = abi.PtStats()
    c = cfg.copy()
    check(lib().akr_pt_render(ctx.h, scene.h, C.byref(c), film.h, C.byref(st)))
    return st.as_dict()


def render_task(ctx: Context, scene: Scene, method_json: str, name: Optional[str] = None, save_intermediate: bool = False,
                save_stats: bool = False, override_sampler_independent: bool = False, verbose: bool = False) -> dict:
    """akari_integrator::render (lib.rs:111-207): every task of the method file, film.out written at the end."""
    ses = RenderSession(1 if save_intermediate else 0, 1 if save_stats else 0, name.encode() if name else None,
                        1 if override_sampler_independent else 0, 1 if verbose else 0)
    st = abi.PtStats()
    check(lib().akr_render_task(ctx.h, scene.h, method_json.encode(), C.byref(ses), C.byref(st)))
    return st.as_dict()


def image_write(path: str, rgb: np.ndarray) -> None:
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[0], rgb.shape[1]
    check(lib().akr_image_write(os.fspath(path).encode(), _fp(rgb), w, h))


def config_from_json(text: str):
    cfg = abi.PtConfig()
    out = C.create_string_buffer(1024)
    check(lib().akr_pt_config_from_json(text.encode(), C.byref(cfg), out, 1024))
    return cfg, out.value.decode()


# ---- host-side known-answer hooks -------------------------------------------------------------------------
def host_stdrng_u64(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint64)
    check(lib().akr_host_stdrng_u64(seed, n, out.ctypes.data_as(C.POINTER(C.c_uint64))))
    return out


def host_chacha_block(key, counter: int, stream: int, rounds: int) -> np.ndarray:
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    check(lib().akr_host_chacha_block(_up(key), counter, stream, rounds, _up(out)))
    return out


def host_pcg32_states(seed: int, n: int) -> np.ndarray:
    out = np.zeros(2 * n, dtype=np.uint64)
    check(lib().akr_host_pcg32_states(seed, n, out.ctypes.data_as(C.POINTER(C.c_uint64))))
    return out


def host_pcg_start(state: int, inc: int) -> int:
    s = C.c_uint64(state)
    check(lib().akr_host_pcg_start(C.byref(s), inc))
    return s.value


def host_tri_pretest(rays8: np.ndarray, tris9: np.ndarray, plane_shift: float = 0.0):
    """(may_hit, exact_accept, exact_t) per (ray, triangle) pair: dinst.h tri_may_hit and the exact Woop test on the host."""
    rays8 = np.ascontiguousarray(rays8, dtype=np.float32).reshape(-1, 8)
    tris9 = np.ascontiguousarray(tris9, dtype=np.float32).reshape(-1, 9)
    n = rays8.shape[0]
    may, exact, t = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.float32)
    check(lib().akr_host_tri_pretest(n, _fp(rays8), _fp(tris9), C.c_float(plane_shift), _up(may), _up(exact), _fp(t)))
    return may.astype(bool), exact.astype(bool), t


def host_alias_table(weights):
    w = np.ascontiguousarray(weights, dtype=np.float32)
    j = np.zeros(w.size, dtype=np.uint32)
    t = np.zeros(w.size, dtype=np.float32)
    pdf = np.zeros(w.size, dtype=np.float32)
    check(lib().akr_host_alias_table(_fp(w), w.size, _up(j), _fp(t), _fp(pdf)))
    return j, t, pdf


# ---- device probes ------------------------------------------------------------------------------------------
def probe_math(ctx: Context, x: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s, c, l = (np.zeros_like(x) for _ in range(3))
    check(lib().akr_probe_math(ctx.h, x.size, _fp(x), _fp(s), _fp(c), _fp(l)))
    return s, c, l


def probe_bsdf(ctx: Context, m: abi.MaterialData, mode: int, wo, data: np.ndarray, table: Optional[np.ndarray] = None) -> np.ndarray:
    ms = m.to_struct()
    wo = np.ascontiguousarray(wo, dtype=np.float32)
    data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((data.shape[0], 4 if mode == 0 else 8), dtype=np.float32)
    tp = _fp(np.ascontiguousarray(table, dtype=np.float32)) if table is not None else C.POINTER(C.c_float)()
    check(lib().akr_probe_bsdf(ctx.h, C.byref(ms), tp, mode, _fp(wo), data.shape[0], _fp(data), _fp(out)))
    return out


def probe_intersect(ctx: Context, scene: Scene, rays: np.ndarray):
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    out = np.zeros((rays.shape[0], 3), dtype=np.uint32)
    bary = np.zeros((rays.shape[0], 2), dtype=np.float32)
    check(lib().akr_probe_intersect(ctx.h, scene.h, rays.shape[0], _fp(rays), _up(out), _fp(bary)))
    return out, bary


def probe_surface_interaction(ctx: Context, scene: Scene, inst_prim: np.ndarray, bary: np.ndarray) -> np.ndarray:
    ip = np.ascontiguousarray(inst_prim, dtype=np.uint32).reshape(-1, 2)
    b = np.ascontiguousarray(bary, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((ip.shape[0], 19), dtype=np.float32)
    check(lib().akr_probe_surface_interaction(ctx.h, scene.h, ip.shape[0], _up(ip), _fp(b), _fp(out)))
    return out


def probe_material_inputs(ctx: Optional[Context], scene: Scene, material: int, uv: np.ndarray) -> np.ndarray:
    """Evaluated inputs (akr_material_desc words, as float32 view) of `material` at uv points; ctx=None runs the host build
    of the same code."""
    u = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((u.shape[0], 26), dtype=np.float32)
    check(lib().akr_probe_material_inputs(ctx.h if ctx is not None else C.c_void_p(), scene.h, material, u.shape[0], _fp(u), _fp(out)))
    return out


def probe_material_inputs_host(scene: Scene, material: int, uv: np.ndarray, color: int = 0) -> np.ndarray:
    """Evaluated inputs of `material` at uv points under the colour pipeline `color`, on the host (no GPU)."""
    u = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((u.shape[0], 26), dtype=np.float32)
    check(lib().akr_probe_material_inputs_host(scene.h, material, color, u.shape[0], _fp(u), _fp(out)))
    return out


def _host_decode(fn, data: bytes) -> np.ndarray:
    w, h = C.c_uint32(), C.c_uint32()
    check(fn(data, len(data), C.byref(w), C.byref(h), None, 0))
    out = np.zeros((h.value, w.value, 4), dtype=np.uint8)
    check(fn(data, len(data), C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size))
    return out


def host_decode_png(data: bytes) -> np.ndarray:
    """PNG -> (H, W, 4) uint8 in file order (top row first), the image crate's to_rgba8 conventions."""
    return _host_decode(lib().akr_host_decode_png, data)


def host_decode_jpeg(data: bytes) -> np.ndarray:
    """JPEG -> (H, W, 4) uint8 in file order."""
    return _host_decode(lib().akr_host_decode_jpeg, data)


def host_decode_tiff(data: bytes) -> np.ndarray:
    """TIFF -> (H, W, 4) uint8 in file order."""
    return _host_decode(lib().akr_host_decode_tiff, data)


def host_decode_dds(data: bytes) -> np.ndarray:
    """DDS (DXT1 / DXT3 / DXT5) -> (H, W, 4) uint8 in file order."""
    return _host_decode(lib().akr_host_decode_dds, data)


def aov_render(ctx: Context, scene: Scene, cfg: abi.AovConfig, film: Film) -> dict:
    """akr_aov_render: the `aov` integrator (akari_integrator/src/aov.rs)."""
    st = abi.PtStats()
    check(lib().akr_aov_render(ctx.h, scene.h, C.byref(cfg), film.h, C.byref(st)))
    return st.as_dict()


def gpt_render(ctx: Context, scene: Scene, cfg: abi.GptConfig, film: Film, want_aux: bool = False):
    """akr_gpt_render: the `gpt` integrator (akari_integrator/src/gpt.rs). Returns the counters, or (counters, (primal sums
    (H, W, 3), Gx sums (H+1, W+1, 3), Gy sums)) with want_aux (reconstruction != none)."""
    st = abi.PtStats()
    w, h = film.width, film.height
    n, ng = w * h, (w + 1) * (h + 1)
    aux = np.zeros(3 * n + 6 * ng, dtype=np.float32) if want_aux else None
    check(lib().akr_gpt_render(ctx.h, scene.h, C.byref(cfg), film.h, _fp(aux) if want_aux else None, C.byref(st)))
    if not want_aux:
        return st.as_dict()
    return st.as_dict(), (aux[:3 * n].reshape(h, w, 3), aux[3 * n:3 * n + 3 * ng].reshape(h + 1, w + 1, 3), aux[3 * n + 3 * ng:].reshape(h + 1, w + 1, 3))


class Shard(C.Structure):
    """akr_shard"""

    _fields_ = [("shard_rank", C.c_uint32), ("shard_count", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32)]


class GptSession:
    """akr_gpt_begin / _sample / _reduce / _finish: a gpt render in steps (one rank's share of a sharded render, or the whole frame)."""

    def __init__(self, ctx: Context, scene: Scene, cfg: abi.GptConfig, film: Film, rank: int = 0, world: int = 1, tile_w: int = 32, tile_h: int = 32):
        self.ctx, self.scene, self.film, self.cfg = ctx, scene, film, cfg
        self.h = C.c_void_p()
        sh = Shard(rank, world, tile_w, tile_h)
        check(lib().akr_gpt_begin(ctx.h, scene.h, C.byref(cfg), C.byref(sh) if world > 1 else None, film.h, C.byref(self.h)))

    def sample(self, n: int = 0, blocking: bool = True):
        check(lib().akr_gpt_sample(self.h, n, 1 if blocking else 0))

    def n_sums(self) -> int:
        p, n = C.POINTER(C.c_float)(), C.c_uint64()
        check(lib().akr_gpt_sums(self.h, C.byref(p), C.byref(n)))
        return n.value

    def read_sums(self) -> np.ndarray:
        out = np.zeros(self.n_sums(), dtype=np.float32)
        if out.size:
            check(lib().akr_gpt_sums_read(self.h, _fp(out)))
        return out

    def write_sums(self, a: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.size == self.n_sums()
        if a.size:
            check(lib().akr_gpt_sums_write(self.h, _fp(a)))

    def reduce(self, comm: "Comm", root: int = 0, blocking: bool = True):
        check(lib().akr_gpt_reduce(self.h, comm.h, root, 1 if blocking else 0))

    def finish(self, want_aux: bool = False):
        st = abi.PtStats()
        w, h = self.film.width, self.film.height
        n, ng = w * h, (w + 1) * (h + 1)
        aux = np.zeros(3 * n + 6 * ng, dtype=np.float32) if want_aux else None
        hh, self.h = self.h, C.c_void_p()
        check(lib().akr_gpt_finish(hh, _fp(aux) if want_aux else None, C.byref(st)))
        return (st.as_dict(), aux) if want_aux else st.as_dict()

    def abort(self) -> dict:
        """akr_gpt_abort: frees the session without reconstructing (non-root ranks of a reduced render; abandoned renders)."""
        st = abi.PtStats()
        hh, self.h = self.h, C.c_void_p()
        check(lib().akr_gpt_abort(hh, C.byref(st)))
        return st.as_dict()

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            lib().akr_gpt_abort(self.h, None)  # never reconstruct implicitly: a forgotten session must not write the film
            self.h = C.c_void_p()


def mcmc_render(ctx: Context, scene: Scene, cfg: abi.McmcConfig, film: Film):
    """akr_mcmc_render: the `mcmc_opt` integrator. Returns (counters, result dict, chain states as a structured array)."""
    st, res = abi.PtStats(), abi.McmcResult()
    chains = np.zeros(cfg.n_chains, dtype=abi.MARKOV_STATE_DTYPE)
    check(lib().akr_mcmc_render(ctx.h, scene.h, C.byref(cfg), film.h, C.byref(res), chains.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(st)))
    return st.as_dict(), {k: getattr(res, k) for k, _ in abi.McmcResult._fields_ if k != "_pad"}, chains


def mcmc_render_shard(ctx: Context, scene: Scene, cfg: abi.McmcConfig, film: Film, rank: int, world: int):
    """akr_mcmc_render_shard: rank `rank` of `world` -- its chains, its tiles of the direct pass. Returns (counters, partial sums, chain
    states: n_chains records, this rank's filled)."""
    st, part = abi.PtStats(), abi.McmcPartial()
    chains = np.zeros(cfg.n_chains, dtype=abi.MARKOV_STATE_DTYPE)
    check(lib().akr_mcmc_render_shard(ctx.h, scene.h, C.byref(cfg), rank, world, film.h, C.byref(part), chains.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(st)))
    return st.as_dict(), part, chains


def mcmc_combine_host(film: Optional[Film], partials) -> dict:
    """akr_mcmc_combine_host: the normalisation from the ranks' partial sums (sets the film's splat scale when a film is given)."""
    arr = (abi.McmcPartial * len(partials))(*partials)
    res = abi.McmcResult()
    check(lib().akr_mcmc_combine_host(film.h if film is not None else None, arr, len(partials), C.byref(res)))
    return {k: getattr(res, k) for k, _ in abi.McmcResult._fields_ if k != "_pad"}


def host_decode_exr(data: bytes) -> np.ndarray:
    """OpenEXR -> (H, W, 4) float32 in file order."""
    w, h = C.c_uint32(), C.c_uint32()
    check(lib().akr_host_decode_exr(data, len(data), C.byref(w), C.byref(h), None, 0))
    out = np.zeros((h.value, w.value, 4), dtype=np.float32)
    check(lib().akr_host_decode_exr(data, len(data), C.byref(w), C.byref(h), _fp(out), out.size))
    return out


def host_pmj02bn_tables():
    """(sets u32[5, 65536, 2], bluenoise u16[48, 128, 128]) exactly as the library's pmj02bn sampler reads them."""
    sets = np.zeros((5, 65536, 2), dtype=np.uint32)
    bn = np.zeros((48, 128, 128), dtype=np.uint16)
    check(lib().akr_host_pmj02bn_tables(_up(sets), bn.ctypes.data_as(C.POINTER(C.c_uint16))))
    return sets, bn
