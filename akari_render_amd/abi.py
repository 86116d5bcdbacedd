"""ctypes mirror of include/akari_hip.h (struct layouts only; no library is loaded here).

`SceneData` is a plain numpy container for a flattened scene; `SceneData.to_desc()` builds the
`akr_scene_desc` the C ABI takes (and keeps the numpy arrays alive for as long as the desc lives).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

c_f32p = C.POINTER(C.c_float)
c_u32p = C.POINTER(C.c_uint32)


class _Struct(C.Structure):
    """ctypes.Structure that refuses attributes which are not fields (a misspelt field would otherwise be a silent no-op)."""

    def __setattr__(self, name, value):
        if not any(f[0] == name for f in self._fields_):
            raise AttributeError("%s has no field %r" % (type(self).__name__, name))
        super().__setattr__(name, value)


class MeshDesc(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_uint32),
        ("n_triangles", C.c_uint32),
        ("vertices", c_f32p),
        ("indices", c_u32p),
        ("uvs", c_f32p),
        ("normals", c_f32p),
        ("tangents", c_f32p),
        ("material_slots", c_u32p),
    ]


class InstanceDesc(C.Structure):
    _fields_ = [
        ("mesh", C.c_uint32),
        ("n_materials", C.c_uint32),
        ("materials", c_u32p),
        ("transform", C.c_float * 16),
    ]


MAT_PRINCIPLED, MAT_DIFFUSE, MAT_GLASS, MAT_EMISSION = 0, 1, 2, 3
MAT_KIND_MASK = 0xFF
MAT_CS_BASE_COLOR, MAT_CS_SPECULAR_TINT, MAT_CS_COAT_TINT, MAT_CS_EMISSION_COLOR = 0x100, 0x200, 0x400, 0x800  # input given in ACEScg
COLOR_REPR_ACESCG, COLOR_RGB_ACESCG = 1, 2  # akr_pt_config.color bits (ColorPipeline, color.rs:663-676)


class MaterialDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("base_color", C.c_float * 3),
        ("base_alpha", C.c_float),
        ("metallic", C.c_float),
        ("roughness", C.c_float),
        ("ior", C.c_float),
        ("specular_ior_level", C.c_float),
        ("specular_tint", C.c_float * 3),
        ("transmission_weight", C.c_float),
        ("coat_weight", C.c_float),
        ("coat_roughness", C.c_float),
        ("coat_ior", C.c_float),
        ("coat_tint", C.c_float * 3),
        ("emission_color", C.c_float * 3),
        ("emission_strength", C.c_float),
        ("normal", C.c_float * 3),
    ]


# shader graphs / textures (akr_node_op, akr_material_input, akr_image_desc)
NODE_NONE = 0xFFFFFFFF
(NODE_CONST, NODE_RGB, NODE_TEXCOORDS, NODE_IMAGE, NODE_MAPPING, NODE_CHECKERBOARD, NODE_SPECTRAL_UPLIFT, NODE_SEPARATE_COLOR,
 NODE_EXTRACT, NODE_NORMAL_MAP) = range(10)
MAPPING_POINT, MAPPING_TEXTURE = 0, 1
FIELD_RED, FIELD_GREEN, FIELD_BLUE, FIELD_UV = 0, 1, 2, 3
INPUT_NAMES = ("base_color", "metallic", "roughness", "ior", "specular_ior_level", "specular_tint", "transmission_weight",
               "coat_weight", "coat_roughness", "coat_ior", "coat_tint", "emission_color", "emission_strength", "normal")
IN_COUNT = len(INPUT_NAMES)
IMAGE_RGBA8, IMAGE_RGBA32F = 0, 1
TEX_FILTER_NEAREST, TEX_FILTER_LINEAR = 0, 1
TEX_REPEAT, TEX_CLIP, TEX_MIRROR, TEX_EXTEND = 0, 1, 2, 3


class ShaderNode(C.Structure):
    _fields_ = [("op", C.c_uint32), ("arg", C.c_uint32 * 4), ("k", C.c_float * 3)]


class MaterialGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("_pad", C.c_uint32), ("nodes", C.POINTER(ShaderNode)), ("input", C.c_uint32 * IN_COUNT)]


class ImageDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("filter", C.c_uint32),
                ("address", C.c_uint32), ("_pad", C.c_uint32), ("texels", C.c_void_p)]


class CameraDesc(C.Structure):
    _fields_ = [("c2w", C.c_float * 16), ("fov", C.c_float), ("width", C.c_uint32), ("height", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_meshes", C.c_uint32),
        ("n_instances", C.c_uint32),
        ("n_materials", C.c_uint32),
        ("_pad", C.c_uint32),
        ("meshes", C.POINTER(MeshDesc)),
        ("instances", C.POINTER(InstanceDesc)),
        ("materials", C.POINTER(MaterialDesc)),
        ("camera", CameraDesc),
        ("ggx_dielectric_table", c_f32p),
        ("n_images", C.c_uint32),
        ("_pad2", C.c_uint32),
        ("images", C.POINTER(ImageDesc)),
        ("material_graphs", C.POINTER(MaterialGraph)),
    ]


FILTER_BOX, FILTER_GAUSSIAN = 0, 1
SAMPLER_INDEPENDENT, SAMPLER_PMJ02BN, SAMPLER_SOBOL = 0, 1, 2


class PtConfig(_Struct):
    """akr_pt_config = pt::Config (pt.rs:916-944) + filter + sampler + shard."""

    _fields_ = [
        ("spp", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("spp_per_pass", C.c_uint32),
        ("rr_depth", C.c_uint32),
        ("use_nee", C.c_uint32),
        ("indirect_only", C.c_uint32),
        ("force_diffuse", C.c_uint32),
        ("pixel_offset", C.c_int32 * 2),
        ("debug_depth", C.c_int32),
        ("filter_type", C.c_uint32),
        ("filter_radius", C.c_float),
        ("sampler_type", C.c_uint32),
        ("color", C.c_uint32),   # akr_color_pipeline_bits (COLOR_REPR_ACESCG | COLOR_RGB_ACESCG); 0 = sRGB / sRGB
        ("sampler_seed", C.c_uint64),
        ("shard_rank", C.c_uint32),
        ("shard_count", C.c_uint32),
        ("tile_w", C.c_uint32),
        ("tile_h", C.c_uint32),
        # samples [sample_begin, sample_begin + sample_count) of the spp of the whole render; count 0 = all. Index-based samplers only.
        ("sample_begin", C.c_uint32),
        ("sample_count", C.c_uint32),
    ]

    @staticmethod
    def default() -> "PtConfig":
        """pt::Config::default() (pt.rs:930-944), PixelFilter::default() (film.rs:50-54),
        SamplerConfig::default() (sampler/mod.rs:290-294)."""
        c = PtConfig()
        c.spp, c.max_depth, c.spp_per_pass, c.rr_depth = 256, 7, 64, 5
        c.use_nee, c.indirect_only, c.force_diffuse = 1, 0, 0
        c.pixel_offset[0] = c.pixel_offset[1] = 0
        c.debug_depth = -1
        c.filter_type, c.filter_radius = FILTER_GAUSSIAN, 1.5
        c.sampler_type, c.sampler_seed = SAMPLER_INDEPENDENT, 0
        c.shard_rank, c.shard_count, c.tile_w, c.tile_h = 0, 1, 32, 32
        return c

    def copy(self) -> "PtConfig":
        c = PtConfig()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(PtConfig))
        return c


AOV_NS, AOV_NG, AOV_TANGENT, AOV_BITANGENT, AOV_ALBEDO, AOV_ROUGHNESS = range(6)
AOV_NAMES = ("ns", "ng", "tangent", "bitangent", "albedo", "roughness")


class AovConfig(_Struct):
    """akr_aov_config = aov::Config (aov.rs:23-39) + filter + sampler + shard."""

    _fields_ = [
        ("spp", C.c_uint32),
        ("aov", C.c_uint32),
        ("remap", C.c_uint32),
        ("filter_type", C.c_uint32),
        ("filter_radius", C.c_float),
        ("sampler_type", C.c_uint32),
        ("sampler_seed", C.c_uint64),
        ("shard_rank", C.c_uint32),
        ("shard_count", C.c_uint32),
        ("tile_w", C.c_uint32),
        ("tile_h", C.c_uint32),
        ("color", C.c_uint32),
        ("_pad", C.c_uint32),
    ]

    @staticmethod
    def default() -> "AovConfig":
        c = AovConfig()
        c.spp, c.aov, c.remap = 256, AOV_NS, 1
        c.filter_type, c.filter_radius = FILTER_GAUSSIAN, 1.5
        c.sampler_type, c.sampler_seed = SAMPLER_INDEPENDENT, 0
        c.shard_rank, c.shard_count, c.tile_w, c.tile_h = 0, 1, 32, 32
        return c


GPT_RECON_NONE, GPT_RECON_UNIFORM, GPT_RECON_WEIGHTED = 0, 1, 2
GPT_RECON_NAMES = ("none", "uniform", "weighted")


class GptConfig(_Struct):
    """akr_gpt_config = gpt::Config (gpt.rs:32-65) + filter + sampler."""

    _fields_ = [
        ("spp", C.c_uint32), ("max_depth", C.c_uint32), ("rr_depth", C.c_uint32), ("spp_per_pass", C.c_uint32),
        ("use_nee", C.c_uint32), ("indirect_only", C.c_uint32), ("reconnect", C.c_uint32), ("stride", C.c_uint32),
        ("separate_weights", C.c_uint32), ("reconstruction", C.c_uint32), ("reconstruction_iter", C.c_uint32),
        ("filter_type", C.c_uint32),
        ("filter_radius", C.c_float),
        ("sampler_type", C.c_uint32),
        ("sampler_seed", C.c_uint64),
        ("seed", C.c_uint64),
        ("color", C.c_uint32),
        ("_pad", C.c_uint32),
    ]

    @staticmethod
    def default() -> "GptConfig":
        c = GptConfig()
        c.spp, c.max_depth, c.rr_depth, c.spp_per_pass = 256, 7, 5, 64
        c.use_nee, c.indirect_only, c.reconnect, c.stride = 1, 0, 1, 1
        c.separate_weights, c.reconstruction, c.reconstruction_iter = 0, GPT_RECON_NONE, 30
        c.filter_type, c.filter_radius = FILTER_GAUSSIAN, 1.5
        c.sampler_type, c.sampler_seed, c.seed = SAMPLER_INDEPENDENT, 0, 0
        return c


class McmcConfig(_Struct):
    """akr_mcmc_config = mcmc::Config + Method::Kelemen (mcmc.rs:8-80) + filter + sampler of the direct pass."""

    _fields_ = [
        ("spp", C.c_uint32), ("max_depth", C.c_uint32), ("rr_depth", C.c_uint32), ("spp_per_pass", C.c_uint32),
        ("use_nee", C.c_uint32), ("mcmc_depth", C.c_uint32), ("n_chains", C.c_uint32), ("n_bootstrap", C.c_uint32),
        ("direct_spp", C.c_int32), ("exponential_mutation", C.c_uint32),
        ("small_sigma", C.c_float), ("large_step_prob", C.c_float), ("image_mutation_prob", C.c_float), ("image_mutation_size", C.c_float),
        ("adaptive", C.c_uint32), ("wis", C.c_uint32),
        ("seed", C.c_uint64),
        ("filter_type", C.c_uint32), ("filter_radius", C.c_float), ("sampler_type", C.c_uint32), ("color", C.c_uint32),
        ("sampler_seed", C.c_uint64),
    ]

    @staticmethod
    def default() -> "McmcConfig":
        c = McmcConfig()
        c.spp, c.max_depth, c.rr_depth, c.spp_per_pass, c.use_nee = 256, 7, 5, 64, 1
        c.mcmc_depth, c.n_chains, c.n_bootstrap, c.direct_spp = 0xFFFFFFFF, 512, 100000, 64
        c.exponential_mutation, c.small_sigma, c.large_step_prob, c.image_mutation_prob, c.image_mutation_size = 1, 0.01, 0.1, 0.0, 0.0
        c.filter_type, c.filter_radius = FILTER_GAUSSIAN, 1.5
        c.sampler_type, c.sampler_seed = SAMPLER_INDEPENDENT, 0
        return c


class McmcResult(C.Structure):
    _fields_ = [("normalization", C.c_double), ("acceptance_rate", C.c_double), ("splat_scale", C.c_float), ("contribution", C.c_float),
                ("n_mutations", C.c_uint64), ("sample_dimension", C.c_uint32), ("_pad", C.c_uint32)]


class McmcPartial(C.Structure):
    """akr_mcmc_partial: one rank's share of the mcmc_opt normalisation (akr_mcmc_render_shard -> akr_mcmc_combine)."""

    _fields_ = [("bootstrap_sum", C.c_double), ("b_sum", C.c_double), ("n_bootstrap", C.c_uint64), ("b_cnt", C.c_uint64), ("n_accepted", C.c_uint64),
                ("n_mutations", C.c_uint64), ("n_executed", C.c_uint64), ("spp", C.c_uint32), ("contribution", C.c_float)]


MARKOV_STATE_DTYPE = [("cur_pixel", "<u4", (2,)), ("chain_id", "<u4"), ("cur_f", "<f4"), ("b", "<f4"), ("b_cnt", "<u4"), ("n_accepted", "<u4"),
                      ("n_mutations", "<u4"), ("cur_iter", "<u4"), ("last_large_iter", "<u4")]


class PtStats(C.Structure):
    _fields_ = [
        ("n_samples", C.c_uint64),
        ("n_closest", C.c_uint64),
        ("n_shadow", C.c_uint64),
        ("n_shaded", C.c_uint64),
        ("n_node_visits", C.c_uint64),
        ("n_tri_tests", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("n_launches", C.c_uint32),
        ("_pad", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("_")}


class KernelInfo(C.Structure):
    """akr_kernel_info (include/akari_hip.h): which kernel a pt session launches -- per-scene (hiprtc) or the interpreter."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("specialised", C.c_uint32),
        ("cache_hit", C.c_uint32),
        ("n_shader_kinds", C.c_uint32),
        ("absent_mask", C.c_uint32),
        ("min_waves", C.c_uint32),
        ("vgprs", C.c_uint32),
        ("scratch_bytes", C.c_uint32),
        ("kernel_flags", C.c_uint32),
        ("_pad", C.c_uint32),
        ("compile_ms", C.c_double),
        ("load_ms", C.c_double),
        ("status", C.c_char * 256),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("struct_size", "status", "_pad")}
        d["status"] = self.status.decode(errors="replace")
        return d


class SceneInfo(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("n_instances", C.c_uint32),
        ("n_triangles", C.c_uint32),
        ("n_materials", C.c_uint32),
        ("n_lights", C.c_uint32),
        ("n_bvh_nodes", C.c_uint32),
        ("uses_bvh", C.c_uint32),
        ("device_bytes", C.c_uint64),
        ("node_bytes", C.c_uint32),
        ("node_stride_bytes", C.c_uint32),
        ("tri_bytes", C.c_uint32),
        ("bvh_depth", C.c_uint32),
    ]


# ------------------------------------------------------------------------------------------------------
@dataclass
class MeshData:
    vertices: np.ndarray  # (nv,3) f32
    indices: np.ndarray  # (nt,3) u32
    uvs: Optional[np.ndarray] = None  # (nt,3,2) f32
    normals: Optional[np.ndarray] = None  # (nt,3,3) f32
    tangents: Optional[np.ndarray] = None  # (nt,3,3) f32
    material_slots: Optional[np.ndarray] = None  # (nt,) u32


@dataclass
class InstanceData:
    mesh: int
    materials: List[int]
    transform: np.ndarray  # (16,) f32 column-major


@dataclass
class MaterialData:
    kind: int = MAT_PRINCIPLED
    base_color: tuple = (0.8, 0.8, 0.8)
    base_alpha: float = 1.0
    metallic: float = 0.0
    roughness: float = 0.5
    ior: float = 1.45
    specular_ior_level: float = 0.5
    specular_tint: tuple = (1.0, 1.0, 1.0)
    transmission_weight: float = 0.0
    coat_weight: float = 0.0
    coat_roughness: float = 0.03
    coat_ior: float = 1.5
    coat_tint: tuple = (1.0, 1.0, 1.0)
    emission_color: tuple = (0.0, 0.0, 0.0)
    emission_strength: float = 0.0
    normal: tuple = (0.0, 0.0, 0.0)
    graph: Optional["GraphData"] = None
    colorspaces: int = 0  # MAT_CS_* bits: which constant colour inputs are given in ACEScg

    def to_struct(self) -> MaterialDesc:
        m = MaterialDesc()
        m.kind = self.kind | self.colorspaces
        for name in ("base_color", "specular_tint", "coat_tint", "emission_color", "normal"):
            v = np.asarray(getattr(self, name), dtype=np.float32)
            arr = getattr(m, name)
            for i in range(3):
                arr[i] = float(v[i])
        for name in (
            "base_alpha",
            "metallic",
            "roughness",
            "ior",
            "specular_ior_level",
            "transmission_weight",
            "coat_weight",
            "coat_roughness",
            "coat_ior",
            "emission_strength",
        ):
            setattr(m, name, float(np.float32(getattr(self, name))))
        return m


@dataclass
class NodeData:
    """One akr_shader_node: op, up to four arguments (node indices / immediates, NODE_NONE = absent), three constants."""
    op: int
    args: tuple = ()
    k: tuple = (0.0, 0.0, 0.0)


@dataclass
class GraphData:
    """Node list of a material + which node feeds which input (by name, see INPUT_NAMES)."""
    nodes: List[NodeData] = field(default_factory=list)
    inputs: dict = field(default_factory=dict)  # input name -> node index


@dataclass
class ImageData:
    texels: np.ndarray  # (H, W, 4) uint8 or float32; row 0 is v = 0
    filter: int = TEX_FILTER_LINEAR
    address: int = TEX_REPEAT


@dataclass
class CameraData:
    c2w: np.ndarray  # (16,) f32 column-major
    fov: float  # radians
    width: int
    height: int


@dataclass
class SceneData:
    meshes: List[MeshData]
    instances: List[InstanceData]
    materials: List[MaterialData]
    camera: CameraData
    ggx_table: Optional[np.ndarray] = None  # (4096,) f32
    instance_names: List[str] = field(default_factory=list)
    material_names: List[str] = field(default_factory=list)
    images: List["ImageData"] = field(default_factory=list)

    def n_triangles(self) -> int:
        return sum(self.meshes[i.mesh].indices.shape[0] for i in self.instances)

    def to_desc(self):
        """Returns (SceneDesc, keepalive). The desc points into numpy arrays held by `keepalive`."""
        keep = []

        def fptr(a, shape_tail=None):
            if a is None:
                return c_f32p()
            a = np.ascontiguousarray(a, dtype=np.float32)
            keep.append(a)
            return a.ctypes.data_as(c_f32p)

        def uptr(a):
            if a is None:
                return c_u32p()
            a = np.ascontiguousarray(a, dtype=np.uint32)
            keep.append(a)
            return a.ctypes.data_as(c_u32p)

        meshes = (MeshDesc * max(1, len(self.meshes)))()
        for i, m in enumerate(self.meshes):
            md = meshes[i]
            md.n_vertices = int(np.asarray(m.vertices).reshape(-1, 3).shape[0])
            md.n_triangles = int(np.asarray(m.indices).reshape(-1, 3).shape[0])
            md.vertices = fptr(m.vertices)
            md.indices = uptr(m.indices)
            md.uvs = fptr(m.uvs)
            md.normals = fptr(m.normals)
            md.tangents = fptr(m.tangents)
            md.material_slots = uptr(m.material_slots)
        insts = (InstanceDesc * max(1, len(self.instances)))()
        for i, inst in enumerate(self.instances):
            d = insts[i]
            d.mesh = inst.mesh
            d.n_materials = len(inst.materials)
            d.materials = uptr(np.asarray(inst.materials, dtype=np.uint32))
            t = np.asarray(inst.transform, dtype=np.float32).reshape(16)
            for k in range(16):
                d.transform[k] = float(t[k])
        mats = (MaterialDesc * max(1, len(self.materials)))()
        for i, m in enumerate(self.materials):
            mats[i] = m.to_struct()
        desc = SceneDesc()
        desc.n_meshes, desc.n_instances, desc.n_materials = len(self.meshes), len(self.instances), len(self.materials)
        desc.meshes = C.cast(meshes, C.POINTER(MeshDesc))
        desc.instances = C.cast(insts, C.POINTER(InstanceDesc))
        desc.materials = C.cast(mats, C.POINTER(MaterialDesc))
        c2w = np.asarray(self.camera.c2w, dtype=np.float32).reshape(16)
        for k in range(16):
            desc.camera.c2w[k] = float(c2w[k])
        desc.camera.fov = float(np.float32(self.camera.fov))
        desc.camera.width, desc.camera.height = int(self.camera.width), int(self.camera.height)
        desc.ggx_dielectric_table = fptr(self.ggx_table)
        keep += [meshes, insts, mats]
        if self.images:
            imgs = (ImageDesc * len(self.images))()
            for i, im in enumerate(self.images):
                t = np.asarray(im.texels)
                assert t.ndim == 3 and t.shape[2] == 4 and t.dtype in (np.uint8, np.float32), "image texels: (H, W, 4) uint8 / float32"
                t = np.ascontiguousarray(t)
                keep.append(t)
                imgs[i].height, imgs[i].width = int(t.shape[0]), int(t.shape[1])
                imgs[i].format = IMAGE_RGBA8 if t.dtype == np.uint8 else IMAGE_RGBA32F
                imgs[i].filter, imgs[i].address = int(im.filter), int(im.address)
                imgs[i].texels = t.ctypes.data
            desc.n_images = len(self.images)
            desc.images = C.cast(imgs, C.POINTER(ImageDesc))
            keep.append(imgs)
        if any(m.graph is not None for m in self.materials):
            graphs = (MaterialGraph * len(self.materials))()
            for i, m in enumerate(self.materials):
                g = graphs[i]
                for k in range(IN_COUNT):
                    g.input[k] = NODE_NONE
                if m.graph is None or not m.graph.nodes:
                    continue
                nodes = (ShaderNode * len(m.graph.nodes))()
                for j, nd in enumerate(m.graph.nodes):
                    nodes[j].op = int(nd.op)
                    for a in range(4):
                        nodes[j].arg[a] = int(nd.args[a]) if a < len(nd.args) and nd.args[a] is not None else NODE_NONE
                    for a in range(3):
                        nodes[j].k[a] = float(np.float32(nd.k[a])) if a < len(nd.k) else 0.0
                g.n_nodes = len(m.graph.nodes)
                g.nodes = C.cast(nodes, C.POINTER(ShaderNode))
                for name, node in m.graph.inputs.items():
                    g.input[INPUT_NAMES.index(name)] = int(node)
                keep.append(nodes)
            desc.material_graphs = C.cast(graphs, C.POINTER(MaterialGraph))
            keep.append(graphs)
        return desc, keep
