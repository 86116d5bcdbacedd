"""ctypes binding of oracle/liborakr.so (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from akari_render_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class OrStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_samples", "n_closest", "n_shadow", "n_shaded", "n_tri_tests")]


def build(force: bool = False) -> None:
    """Compiles the oracle with gcc (oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return "fma" in fl and "avx2" in fl
    except OSError:
        pass
    return False


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    name = "liborakr.so" if _cpu_has_fma() else "liborakr_generic.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int32
    fp, up, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.or_scene_create.restype = vp
    L.or_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
    L.or_scene_destroy.argtypes = [vp]
    L.or_pt_render.restype = i32
    L.or_pt_render.argtypes = [vp, C.POINTER(abi.PtConfig), fp, u64p, u32, C.POINTER(OrStats)]
    L.or_film_resolve.argtypes = [fp, u32, u32, fp]
    L.or_film_resolve_scaled.argtypes = [fp, u32, u32, f32, fp]
    L.or_scene_shared_plane_rows.restype = u32
    L.or_scene_shared_plane_rows.argtypes = [vp]
    L.or_set_share_plane_rows.argtypes = [i32]
    L.or_scene_set_color.argtypes = [vp, u32]
    L.or_kat_sobol_2d.argtypes = [u32, u32, u32, u32, u32, u32, fp]
    L.or_scene_build_bvh.restype = u32
    L.or_scene_build_bvh.argtypes = [vp]
    L.or_scene_free_bvh.argtypes = [vp]
    L.or_scene_intersect_many.argtypes = [vp, u32, fp, i32, up, fp, u32]
    L.or_mt_f64_many.argtypes = [vp, u32, fp, up, up, C.POINTER(C.c_double), u32]
    L.or_scene_world_vertices.argtypes = [vp, fp]
    L.or_mcmc_render.restype = i32
    L.or_mcmc_render.argtypes = [vp, C.POINTER(abi.McmcConfig), fp, C.POINTER(C.c_double), up, u32]
    L.or_gpt_render.restype = i32
    L.or_gpt_render.argtypes = [vp, C.POINTER(abi.GptConfig), fp, fp, u32]
    L.or_init_pcg32_buffer_with_seed.argtypes = [u64, u64, u64p]
    L.or_ggx_dielectric_table_entry.restype = f32
    L.or_ggx_dielectric_table_entry.argtypes = [u32, u32, u32, u32]
    L.or_kat_pcg32.argtypes = [u64, u64, u32, up]
    L.or_kat_pcg32_advance.argtypes = [u64p, u64, C.c_int64]
    L.or_kat_pcg32_new_seq.argtypes = [u64, u64p, u64p]
    L.or_kat_next_1d.restype = f32
    L.or_kat_next_1d.argtypes = [u64p, u64]
    L.or_kat_chacha_block.argtypes = [up, u64, u64, i32, up]
    L.or_kat_stdrng_u64.argtypes = [u64, u32, u64p]
    L.or_kat_xxhash32_4.restype = u32
    L.or_kat_xxhash32_4.argtypes = [u32, u32, u32, u32]
    L.or_kat_mix_bits.restype = u64
    L.or_kat_mix_bits.argtypes = [u64]
    L.or_kat_sincos.argtypes = [f32, fp, fp]
    L.or_kat_log.restype = f32
    L.or_kat_log.argtypes = [f32]
    L.or_kat_exp.restype = f32
    L.or_kat_exp.argtypes = [f32]
    L.or_kat_pow.restype = f32
    L.or_kat_pow.argtypes = [f32, f32]
    L.or_material_inputs.argtypes = [vp, u32, u32, fp, fp]
    L.or_tex_sample_many.argtypes = [C.POINTER(abi.ImageDesc), u32, fp, fp]
    L.or_kat_alias_build.argtypes = [fp, u32, up, fp, fp]
    L.or_kat_offset_ray_origin.argtypes = [fp, fp, fp]
    L.or_kat_uniform_sample_triangle.argtypes = [f32, f32, fp]
    L.or_kat_cos_sample_hemisphere.argtypes = [f32, f32, fp]
    L.or_scene_num_lights.restype = u32
    L.or_scene_num_lights.argtypes = [vp]
    L.or_scene_num_triangles.restype = u32
    L.or_scene_num_triangles.argtypes = [vp]
    L.or_scene_light_info.argtypes = [vp, u32, up, fp, fp]
    L.or_scene_camera.argtypes = [vp, fp, fp, C.POINTER(i32)]
    L.or_scene_surface_interaction.argtypes = [vp, u32, u32, f32, f32, fp]
    L.or_scene_intersect.restype = i32
    L.or_scene_intersect.argtypes = [vp, fp, fp, f32, f32, up, up, fp]
    L.or_bsdf_probe_many.argtypes = [C.POINTER(abi.MaterialDesc), fp, fp, u32, fp, fp]
    L.or_bsdf_eval_many.argtypes = [C.POINTER(abi.MaterialDesc), fp, fp, u32, fp, fp]
    L.or_set_pmj_tables.argtypes = [up, C.POINTER(C.c_uint16)]
    L.or_aov_render.restype = i32
    L.or_aov_render.argtypes = [vp, C.POINTER(abi.AovConfig), fp, u32, u64p]
    for n in ("or_sizeof_material", "or_sizeof_config", "or_sizeof_scene_desc", "or_sizeof_aov_config"):
        getattr(L, n).restype = u32
    assert L.or_sizeof_material() == C.sizeof(abi.MaterialDesc)
    assert L.or_sizeof_config() == C.sizeof(abi.PtConfig)
    assert L.or_sizeof_scene_desc() == C.sizeof(abi.SceneDesc)
    assert L.or_sizeof_aov_config() == C.sizeof(abi.AovConfig)
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleScene:
    def __init__(self, scene: abi.SceneData, bvh: bool = False, share_plane_rows: bool = True):
        """bvh: build the oracle-side hierarchy (or_accel.h; same hits as the exhaustive loop, checked in
        tests/test_oracle_accel.py) -- for the 1 M / 10 M-triangle configurations. share_plane_rows = False: every triangle
        keeps the plane row of its own vertices (the records as they were before the coplanar-neighbour rule)."""
        self.data = scene
        desc, self._keep = scene.to_desc()
        self._desc = desc
        lib().or_set_share_plane_rows(1 if share_plane_rows else 0)
        try:
            self.h = lib().or_scene_create(C.byref(desc))
        finally:
            lib().or_set_share_plane_rows(1)
        self.width, self.height = scene.camera.width, scene.camera.height
        self.n_bvh_nodes = lib().or_scene_build_bvh(self.h) if bvh else 0

    def intersect_many(self, rays: np.ndarray, any_hit: bool = False, n_threads: int = 0):
        """rays (n, 8) = o, d, tmin, tmax -> (hit/inst/prim u32 (n, 3), t/u/v f32 (n, 3)); through or_trace."""
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros((r.shape[0], 3), dtype=np.uint32)
        tuv = np.zeros((r.shape[0], 3), dtype=np.float32)
        lib().or_scene_intersect_many(self.h, r.shape[0], _fp(r), 1 if any_hit else 0, out.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(tuv),
                                      n_threads if n_threads > 0 else (os.cpu_count() or 1))
        return out, tuv

    def mt_f64(self, rays: np.ndarray, gids: np.ndarray | None = None, n_threads: int = 0):
        """Independent f64 Moeller-Trumbore from the f32 world-space vertices. gids None: closest hit over all triangles ->
        (gid u32 (n,), (t, u, v, margin) f64 (n, 4)); gids given: the solve for that triangle, no range test."""
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros((r.shape[0], 4), dtype=np.float64)
        og = np.zeros(r.shape[0], dtype=np.uint32)
        gp = C.POINTER(C.c_uint32)()
        if gids is not None:
            gids = np.ascontiguousarray(gids, dtype=np.uint32)
            gp = gids.ctypes.data_as(C.POINTER(C.c_uint32))
        lib().or_mt_f64_many(self.h, r.shape[0], _fp(r), gp, og.ctypes.data_as(C.POINTER(C.c_uint32)), out.ctypes.data_as(C.POINTER(C.c_double)),
                             n_threads if n_threads > 0 else (os.cpu_count() or 1))
        return (og if gids is None else gids), out

    def world_vertices(self) -> np.ndarray:
        out = np.zeros((lib().or_scene_num_triangles(self.h), 3, 3), dtype=np.float32)
        lib().or_scene_world_vertices(self.h, _fp(out))
        return out

    def tri_offsets(self) -> np.ndarray:
        """first global triangle id of every instance"""
        n = [self.data.meshes[i.mesh].indices.reshape(-1, 3).shape[0] for i in self.data.instances]
        return np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.uint32)

    def close(self):
        if self.h:
            lib().or_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def shared_plane_rows(self) -> int:
        return lib().or_scene_shared_plane_rows(self.h)

    def num_lights(self):
        return lib().or_scene_num_lights(self.h)

    def light_info(self, i):
        inst, power, pdf = C.c_uint32(), C.c_float(), C.c_float()
        lib().or_scene_light_info(self.h, i, C.byref(inst), C.byref(power), C.byref(pdf))
        return inst.value, power.value, pdf.value

    def aov_render(self, cfg: abi.AovConfig, n_threads: int = 0):
        """The aov integrator (aov.rs): returns (film f32[7*N], camera rays traced)."""
        film = np.zeros(7 * self.width * self.height, dtype=np.float32)
        n = C.c_uint64()
        rc = lib().or_aov_render(self.h, C.byref(cfg), _fp(film), n_threads if n_threads > 0 else (os.cpu_count() or 1), C.byref(n))
        assert rc == 0
        return film, n.value

    def gpt_render(self, cfg: abi.GptConfig, n_threads: int = 0):
        """The gpt integrator (gpt.rs): returns (film f32[7*N], (primal, gx, gy) sums or None)."""
        w, h = self.width, self.height
        n, ng = w * h, (w + 1) * (h + 1)
        film = np.zeros(7 * n, dtype=np.float32)
        aux = np.zeros(3 * n + 6 * ng, dtype=np.float32)
        rc = lib().or_gpt_render(self.h, C.byref(cfg), _fp(film), _fp(aux), n_threads if n_threads > 0 else (os.cpu_count() or 1))
        assert rc == 0, "or_gpt_render rejected the configuration"
        if cfg.reconstruction == 0:
            return film, None
        return film, (aux[:3 * n].reshape(h, w, 3), aux[3 * n:3 * n + 3 * ng].reshape(h + 1, w + 1, 3), aux[3 * n + 3 * ng:].reshape(h + 1, w + 1, 3))

    def mcmc_render(self, cfg: abi.McmcConfig, n_threads: int = 0):
        """The mcmc_opt integrator: (film f32[7*N], {normalization, acceptance_rate, splat_scale, contribution}, chain states)."""
        film = np.zeros(7 * self.width * self.height, dtype=np.float32)
        res = np.zeros(4, dtype=np.float64)
        chains = np.zeros(cfg.n_chains, dtype=abi.MARKOV_STATE_DTYPE)
        rc = lib().or_mcmc_render(self.h, C.byref(cfg), _fp(film), res.ctypes.data_as(C.POINTER(C.c_double)), chains.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  n_threads if n_threads > 0 else (os.cpu_count() or 1))
        assert rc == 0, f"or_mcmc_render failed ({rc})"
        return film, {"normalization": res[0], "acceptance_rate": res[1], "splat_scale": np.float32(res[2]), "contribution": np.float32(res[3])}, chains

    def material_inputs(self, material: int, uv, color: int = 0) -> np.ndarray:
        """Evaluated inputs (akr_material_desc words) of `material` at uv points under the ColorPipeline bits `color`."""
        lib().or_scene_set_color(self.h, color)
        u = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((u.shape[0], 26), dtype=np.float32)
        lib().or_material_inputs(self.h, material, u.shape[0], _fp(u), _fp(out))
        return out

    def surface_interaction(self, inst, prim, u, v):
        out = np.zeros(19, dtype=np.float32)
        lib().or_scene_surface_interaction(self.h, inst, prim, u, v, _fp(out))
        return out

    def intersect(self, o, d, tmin=0.0, tmax=1e20):
        o = np.asarray(o, dtype=np.float32)
        d = np.asarray(d, dtype=np.float32)
        inst, prim = C.c_uint32(), C.c_uint32()
        bary = np.zeros(2, dtype=np.float32)
        hit = lib().or_scene_intersect(self.h, _fp(o), _fp(d), tmin, tmax, C.byref(inst), C.byref(prim), _fp(bary))
        return bool(hit), inst.value, prim.value, bary

    def render(self, cfg: abi.PtConfig, n_threads: int = 0, film: np.ndarray | None = None, states: np.ndarray | None = None):
        """Returns (film f32[7*N] in the reference layout, stats dict)."""
        N = self.width * self.height
        if film is None:
            film = np.zeros(7 * N, dtype=np.float32)
        if n_threads <= 0:
            n_threads = os.cpu_count() or 1
        st = OrStats()
        sp = states.ctypes.data_as(C.POINTER(C.c_uint64)) if states is not None else C.POINTER(C.c_uint64)()
        rc = lib().or_pt_render(self.h, C.byref(cfg), _fp(film), sp, n_threads, C.byref(st))
        assert rc == 0
        return film, {k: getattr(st, k) for k, _ in OrStats._fields_}


def resolve(film: np.ndarray, width: int, height: int, splat_scale: float = 1.0) -> np.ndarray:
    out = np.zeros(3 * width * height, dtype=np.float32)
    lib().or_film_resolve_scaled(_fp(film), width, height, C.c_float(splat_scale), _fp(out))
    return out.reshape(height, width, 3)


def init_pcg32_states(count: int, seed: int) -> np.ndarray:
    st = np.zeros(2 * count, dtype=np.uint64)
    lib().or_init_pcg32_buffer_with_seed(count, seed, st.ctypes.data_as(C.POINTER(C.c_uint64)))
    return st


def bsdf_sample_many(m: abi.MaterialData, wo, u: np.ndarray, table: np.ndarray | None = None) -> np.ndarray:
    ms = m.to_struct()
    wo = np.asarray(wo, dtype=np.float32)
    u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((u.shape[0], 8), dtype=np.float32)
    tp = _fp(np.ascontiguousarray(table, dtype=np.float32)) if table is not None else C.POINTER(C.c_float)()
    lib().or_bsdf_probe_many(C.byref(ms), tp, _fp(wo), u.shape[0], _fp(u), _fp(out))
    return out


def bsdf_eval_many(m: abi.MaterialData, wo, wi: np.ndarray, table: np.ndarray | None = None) -> np.ndarray:
    ms = m.to_struct()
    wo = np.asarray(wo, dtype=np.float32)
    wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((wi.shape[0], 4), dtype=np.float32)
    tp = _fp(np.ascontiguousarray(table, dtype=np.float32)) if table is not None else C.POINTER(C.c_float)()
    lib().or_bsdf_eval_many(C.byref(ms), tp, _fp(wo), wi.shape[0], _fp(wi), _fp(out))
    return out


def tex_sample(image: abi.ImageData, uv) -> np.ndarray:
    """or_tex_sample of one image at uv points -> (n, 4) float32."""
    t = np.ascontiguousarray(image.texels)
    d = abi.ImageDesc()
    d.height, d.width = t.shape[0], t.shape[1]
    d.format = abi.IMAGE_RGBA8 if t.dtype == np.uint8 else abi.IMAGE_RGBA32F
    d.filter, d.address = image.filter, image.address
    d.texels = t.ctypes.data
    u = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((u.shape[0], 4), dtype=np.float32)
    lib().or_tex_sample_many(C.byref(d), u.shape[0], _fp(u), _fp(out))
    return out


_pmj_keep = None


def set_pmj_tables(sets: np.ndarray, bluenoise: np.ndarray) -> None:
    """Hands the pmj02bn tables to the oracle (it has none of its own: the reference's are absent from its tree)."""
    global _pmj_keep
    sets = np.ascontiguousarray(sets, dtype=np.uint32)
    bluenoise = np.ascontiguousarray(bluenoise, dtype=np.uint16)
    _pmj_keep = (sets, bluenoise)
    lib().or_set_pmj_tables(sets.ctypes.data_as(C.POINTER(C.c_uint32)), bluenoise.ctypes.data_as(C.POINTER(C.c_uint16)))
