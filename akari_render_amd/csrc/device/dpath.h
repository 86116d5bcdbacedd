// dpath.h -- per-path state and the "one path vertex" step shared by the two schedules of the path tracer:
// the persistent-lane megakernel (pt_kernels.hip: k_pt_pass) and the wavefront pipeline (wf_kernels.hip).
// Everything here follows crates/akari_integrator/src/pt.rs:95-323,329-900 (shift_mapping = None), camera/mod.rs,
// film.rs, sampler/mod.rs, light/{mod,area}.rs of the reference; file:line cited per function.
#pragma once
#include "dinst_trav.h"
#include "drng.h"
#include "../kernels.h"

#ifndef AKR_TEX_LEAN
#define AKR_TEX_LEAN 0  // 1 = the kernels of scenes with textures recompute the normal-map frame where it is read too (dbsdf.h: lean)
#endif

namespace akr {

// ----------------------------------------------------------------------------------------------------------
// work distribution: item index -> pixel. Items enumerate the pixels of the tiles this rank owns
// (kernels.h tile_owner: a tile's Morton code modulo the ranks; the session's owned_tiles lists them), tile by tile, and inside a tile in 8x8 blocks so that one wave
// covers an 8x8 pixel square (coherent primary rays, one film cache line per row segment).
AKR_D bool item_to_pixel(const PtParams& p, uint32_t item, uint32_t& px, uint32_t& py) {
    const uint32_t tile_px = p.tile_w * p.tile_h;
    uint32_t j = item / tile_px, within = item - j * tile_px;
    uint32_t tile = j;  // one rank: every tile, row by row
    if (p.shard_count > 1) {
        if (item >= p.n_items) return false;
        tile = p.owned_tiles[j];
    }
    if (tile >= p.tiles_x * p.tiles_y) return false;
    uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    uint32_t block = within >> 6, lane = within & 63u;
    uint32_t bpr = p.tile_w >> 3;  // 8x8 blocks per tile row
    uint32_t by = block / bpr, bx = block - by * bpr;
    px = tx * p.tile_w + bx * 8 + (lane & 7u);
    py = ty * p.tile_h + by * 8 + (lane >> 3);
    return px < p.width && py < p.height;
}

// film.rs:32-49
AKR_D vec2 filter_sample(const PtParams& p, vec2 u) {
    if (p.filter_type == 0) return mk2((u.x - 0.5f) * p.filter_radius, (u.y - 0.5f) * p.filter_radius);
    float width = p.filter_radius;
    float sigma = width / 3.0f;
    float r = __builtin_sqrtf(-2.0f * log_f(u.x));
    float theta = 2.0f * kPi * u.y;
    float sn, cs;
    sincos_f(theta, sn, cs);
    vec2 off = mk2((r * cs) * sigma, (r * sn) * sigma);
    return mk2(clamp_f(off.x, -width, width), clamp_f(off.y, -width, width));
}

// The sampler of one pixel. Independent (sampler/mod.rs:161-217): PCG32 state + dimensions drawn so far. Pmj02Bn
// (sampler/mod.rs:329-700, Pmj02BnState): pcg.state = sample index (u32::MAX before the first start()), pcg.inc =
// x | y << 32 of the pixel the sampler was created for, dim = dimension counter; seed / spp / w and the tables come from
// PtParams. Which one is a template parameter of the kernels (PMJ): the PCG path carries none of the table code.
struct Sampler {
    Pcg32 pcg;
    uint32_t dim;
};
constexpr uint32_t kPmjSets = 5, kPmjSamples = 65536, kBlueNoiseTextures = 48, kBlueNoiseRes = 128;
// permute_element (sampler/mod.rs:473-507; Kensler's hashed permutation of [0, l)). The last line is (i + p) % l: with l a power
// of two (w = l - 1) that is a mask; otherwise the remainder by the launch's precomputed constant (drng.h fastmod_u32, magic =
// fastmod_magic(l) from PtParams) -- the same number either way.
AKR_D uint32_t permute_element(uint32_t i, uint32_t l, uint32_t w, uint32_t p, uint64_t magic) {
    do {
        i ^= p;
        i *= 0xe170893du;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8;
        i *= 0x0929eb3fu;
        i ^= p >> 23;
        i ^= (i & w) >> 1;
        i *= 1u | p >> 27;
        i *= 0x6935fa69u;
        i ^= (i & w) >> 11;
        i *= 0x74dcb303u;
        i ^= (i & w) >> 2;
        i *= 0x9e501cc3u;
        i ^= (i & w) >> 2;
        i *= 0xc860a3dfu;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    if (l == w + 1u) return (i + p) & w;  // wave-uniform
    return fastmod_u32(i + p, magic, l);
}
// unorm16 -> float: v / 65535 correctly rounded, by the reciprocal known in advance (q = v y, q + (v - 65535 q) y with y = RN(1 / 65535):
// two fma instead of an IEEE division; identical for all 65536 values, tests/test_pmj02bn.py)
AKR_HD float unorm16(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = 1.5259021893143654e-05f, fv = (float)v;
    const float q = fv * y;
    return __builtin_fmaf(__builtin_fmaf(-65535.0f, q, fv), y, q);
#else
    return (float)v / 65535.0f;
#endif
}
// bluenoise(tex_index, p): uv = p.yx() % 128 of texture tex_index % 48 (sampler/mod.rs:545-553), unorm16 -> float
// k_pt_pass keeps a lane's pixel for the whole launch, and a pixel reads ONE texel of each of the 48 arrays: when the launch has
// room (PtParams.bn_offset != 0, launch_pt_pass) the lane's 48 values sit in a column of LDS (pmj_bluenoise_stage below) and a
// lookup is a ds_read_u16 instead of a gather from a 1.5 MB table.
AKR_D float pmj_bluenoise(const PtParams& p, uint32_t tex, uint32_t px, uint32_t py) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (p.bn_offset != 0) {  // wave-uniform
        extern __shared__ __attribute__((aligned(16))) uint32_t akr_dynamic_lds[];
        const uint16_t* bn = reinterpret_cast<const uint16_t*>(akr_dynamic_lds + p.bn_offset);
        return unorm16(bn[(tex % kBlueNoiseTextures) * 256u + threadIdx.x]);
    }
#endif
    uint32_t tx = py % kBlueNoiseRes, ty = px % kBlueNoiseRes;  // uv = (p.y, p.x)
    uint16_t v = p.bluenoise[((size_t)(tex % kBlueNoiseTextures) * kBlueNoiseRes + ty) * kBlueNoiseRes + tx];
    return unorm16(v);
}
// the lane's column of blue-noise values (see pmj_bluenoise): every lane writes and later reads only its own entries -- no barrier
AKR_D void pmj_bluenoise_stage(const PtParams& p, uint32_t px, uint32_t py) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint32_t akr_dynamic_lds[];
    uint16_t* bn = reinterpret_cast<uint16_t*>(akr_dynamic_lds + p.bn_offset);
    const uint32_t tx = py % kBlueNoiseRes, ty = px % kBlueNoiseRes;
    for (uint32_t t = 0; t < kBlueNoiseTextures; t++) bn[t * 256u + threadIdx.x] = p.bluenoise[((size_t)t * kBlueNoiseRes + ty) * kBlueNoiseRes + tx];
#endif
}
constexpr float kOneMinusEpsilon = 0.99999994f;
template <bool PMJ>
AKR_D float next_1d(const PtParams& p, Sampler& s) {
    if (!PMJ) {
        s.dim += 1;
        return pcg_next_1d(s.pcg);
    }
    // Pmj02BnSampler::next_1d (sampler/mod.rs:555-580)
    const uint32_t px = (uint32_t)s.pcg.inc, py = (uint32_t)(s.pcg.inc >> 32), sample_index = (uint32_t)s.pcg.state;
    uint32_t hash = xxhash32_4(px, py, s.dim, p.smp_seed);
    uint32_t index = permute_element(sample_index, p.smp_spp, p.smp_w, hash, p.smp_mod_magic);
    if (p.sampler == 2u) {  // sobol: the permuted index through the scrambled radical inverse (wave-uniform branch)
        // owen_scramble(reverse_bits32(index), seed): the reversal of the argument and the scramble's first one cancel
        const uint32_t v = owen_scramble_of_reversed(index, xxhash32_4(py, px, s.dim, ~p.smp_seed));
        s.dim += 1;
        return min_f((float)v * 2.3283064365386963e-10f, kOneMinusEpsilon);
    }
    float delta = pmj_bluenoise(p, s.dim, px, py);
    s.dim += 1;
    return min_f(((float)index + delta) / (float)p.smp_spp, kOneMinusEpsilon);
}
template <bool PMJ>
AKR_D vec2 next_2d(const PtParams& p, Sampler& s) {
    if (!PMJ) {
        float a = next_1d<PMJ>(p, s);
        float b = next_1d<PMJ>(p, s);
        return mk2(a, b);
    }
    // Pmj02BnSampler::next_2d (sampler/mod.rs:582-630)
    const uint32_t px = (uint32_t)s.pcg.inc, py = (uint32_t)(s.pcg.inc >> 32);
    uint32_t index = (uint32_t)s.pcg.state;
    const uint32_t dim = s.dim, pmj_instance = dim / 2;
    if (p.sampler == 2u) {  // sobol: every dimension pair is the (0,2)-sequence under its own index permutation and scramble
        const uint32_t hash = xxhash32_4(px, py, dim, p.smp_seed);
        const uint32_t i = permute_element(index, p.smp_spp, p.smp_w, hash, p.smp_mod_magic);
        // owen_scramble(reverse_bits32(i), .) and owen_scramble(sobol_dim1(i), .) with the cancelling reversals left out
        const uint32_t vx = owen_scramble_of_reversed(i, xxhash32_4(py, px, dim, ~p.smp_seed));
        const uint32_t vy = owen_scramble_of_reversed(sobol_dim1_reversed(i), xxhash32_4(py, px, dim + 1u, ~p.smp_seed));
        s.dim += 2;
        return mk2(min_f((float)vx * 2.3283064365386963e-10f, kOneMinusEpsilon), min_f((float)vy * 2.3283064365386963e-10f, kOneMinusEpsilon));
    }
    if (pmj_instance >= kPmjSets) index = permute_element(index, p.smp_spp, p.smp_w, xxhash32_4(px, py, dim, p.smp_seed), p.smp_mod_magic);
    const uint32_t* smp = p.pmj_sets + 2 * ((size_t)kPmjSamples * (pmj_instance % kPmjSets) + (index % kPmjSamples));
    vec2 u = mk2((float)smp[0] * 2.3283064365386963e-10f, (float)smp[1] * 2.3283064365386963e-10f);
    float dx = pmj_bluenoise(p, dim, px, py), dy = pmj_bluenoise(p, dim + 1, px, py);
    u = mk2(u.x + dx, u.y + dy);
    s.dim += 2;
    u = mk2(u.x - __builtin_floorf(u.x), u.y - __builtin_floorf(u.y));
    return mk2(min_f(u.x, kOneMinusEpsilon), min_f(u.y, kOneMinusEpsilon));
}
template <bool PMJ>
AKR_D vec3 next_3d(const PtParams& p, Sampler& s) {  // trait default: (next_1d, next_2d), sampler/mod.rs:22-27
    float a = next_1d<PMJ>(p, s);
    vec2 b = next_2d<PMJ>(p, s);
    return mk3(a, b.x, b.y);
}
// sampler.start() (sampler/mod.rs:199-203 / 650-663)
template <bool PMJ>
AKR_D void sampler_start(const PtParams& p, Sampler& s) {
    if (!PMJ) {
        pcg_start(s.pcg, p.start);
    } else {
        s.dim = 4;
        uint32_t idx = (uint32_t)s.pcg.state;
        s.pcg.state = idx == 0xffffffffu ? 0u : idx + 1u;
    }
}
// end of a pass: Drop of the sampler (sampler/mod.rs:168-177 / 633-640) and its re-creation from the stored state
template <bool PMJ>
AKR_D void sampler_end_pass(const PtParams& p, Sampler& s) {
    if (!PMJ) {
        pcg_advance(s.pcg, -(int64_t)s.dim);
        s.dim = 0;
    }  // pmj02bn: the state is stored as it is; dim is reset by the next start()
}

// camera/mod.rs:70-103
AKR_D void generate_ray_from(const PtParams& p, uint32_t px, uint32_t py, vec2 u_filter, vec3& o, vec3& d) {
    vec2 fpixel = mk2((float)px + 0.5f, (float)py + 0.5f);
    vec2 offset = filter_sample(p, u_filter);
    vec2 pf = mk2(fpixel.x + offset.x, fpixel.y + offset.y);
    const float* m = p.r2c;
    float qx = ((m[0] * pf.x + m[4] * pf.y) + m[8] * 0.0f) + m[12] * 1.0f;
    float qy = ((m[1] * pf.x + m[5] * pf.y) + m[9] * 0.0f) + m[13] * 1.0f;
    float qz = ((m[2] * pf.x + m[6] * pf.y) + m[10] * 0.0f) + m[14] * 1.0f;
    float qw = ((m[3] * pf.x + m[7] * pf.y) + m[11] * 0.0f) + m[15] * 1.0f;
    d = normalize(div_s(mk3(qx, qy, qz), qw));
    o = mk3(0, 0, 0);
    if (!p.c2w_identity) {
        const float* c = p.c2w;
        o = div_s(mk3(c[12], c[13], c[14]), c[15]);
        d = mk3((c[0] * d.x + c[4] * d.y) + c[8] * d.z, (c[1] * d.x + c[5] * d.y) + c[9] * d.z,
                (c[2] * d.x + c[6] * d.y) + c[10] * d.z);
    }
}
template <bool PMJ>
AKR_D void generate_ray(const PtParams& p, uint32_t px, uint32_t py, Sampler& smp, vec3& o, vec3& d) {
    generate_ray_from(p, px, py, next_2d<PMJ>(p, smp), o, d);
}

AKR_D float mis_weight(float a, float b) {  // pt.rs:962-973 with power = 1
    float pa = 1.0f * a, pb = 1.0f * b;
    return pa / (pa + pb);
}

// emission of the material at a surface point (AreaLightExpr::emission, light/area.rs:19-31): Principled returns
// its emission constant (principled.rs:267-274), an Emission node likewise, everything else is black.
AKR_D vec3 material_emission(const DMaterial& m) {
    return (m.kind == MAT_PRINCIPLED || m.kind == MAT_EMISSION) ? m.emission : mk3(0, 0, 0);
}
// the same at a point of a material whose emission inputs may be texture-fed (TEX kernels only)
template <bool TEX>
AKR_D vec3 material_emission_at(const DScene& sc, uint32_t material, vec2 uv) {
    const DMaterial& m = sc.materials[material];
    if (TEX) {
        if ((m.flags & MF_TEXTURED) && (m.tex_input[IN_EMISSION_COLOR] != kNodeNone || m.tex_input[IN_EMISSION_STRENGTH] != kNodeNone)) {
            if (!(m.kind == MAT_PRINCIPLED || m.kind == MAT_EMISSION)) return mk3(0, 0, 0);
            return material_emission_inputs_at(sc.tex, m, material, uv);
        }
    }
    return material_emission(m);
}

struct LightSample {
    vec3 li, wi;
    float pdf;
    vec3 ro;
    float tmax;
    uint32_t ex1;
    bool valid;
};
// LightAggregate::sample_direct (light/mod.rs:115-132) + AreaLight::sample_direct (light/area.rs:51-107)
template <bool TEX, bool INST = false>
AKR_D LightSample sample_direct(const DScene& sc, vec3 pn_p, vec3 pn_n, float u_select, vec2 u_sample) {
    LightSample s;
    s.li = mk3(0, 0, 0);
    s.wi = mk3(0, 0, 0);
    s.pdf = 0.0f;
    s.ro = mk3(0, 0, 0);
    s.tmax = 0.0f;
    s.ex1 = kInvalid;
    s.valid = false;
    if (sc.n_lights == 0) return s;
    float light_choice_pdf, u_sel2, pdf_prim, u_unused;
    uint32_t light = alias_sample_and_remap(sc.light_alias, sc.n_lights, u_select, light_choice_pdf, u_sel2);
    const LightRec L = sc.lights[light];
    uint32_t prim = alias_sample_and_remap(sc.area_alias + L.tri_offset, L.n_tris, u_sel2, pdf_prim, u_unused);
    uint32_t gid = L.first_gid + prim;
    vec2 bary = uniform_sample_triangle(u_sample);
    SurfacePoint y = surface_interaction_any<INST>(sc, gid, bary);
    vec3 wi = y.p - pn_p;
    if (length2(wi) == 0.0f) return s;
    float dist2 = length2(wi);
    wi = div_s(wi, __builtin_sqrtf(dist2));
    vec3 emission = material_emission_at<TEX>(sc, y.material, y.uv);
    s.li = dot(wi, y.ng) < 0.0f ? emission : mk3(0, 0, 0);
    float cos_theta_i = abs_f(dot(y.ng, wi));
    float pdf = pdf_prim / y.prim_area * dist2 / cos_theta_i;
    s.ro = offset_ray_origin(pn_p, face_forward(pn_n, wi));
    float dist = __builtin_sqrtf(dist2);
    s.tmax = dist * (1.0f - 1e-3f);
    s.ex1 = gid;
    s.wi = wi;
    s.valid = is_finite(pdf);
    s.pdf = pdf * light_choice_pdf;
    return s;
}
// LightAggregate::pdf_direct (light/mod.rs:134-147) + AreaLight::pdf_direct (light/area.rs:109-130)
AKR_D float pdf_direct(const DScene& sc, const SurfacePoint& si, uint32_t gid, vec3 pn_p) {
    uint32_t light = (uint32_t)si.light;
    float light_choice_pdf = sc.light_pdf[light];
    const LightRec L = sc.lights[light];
    uint32_t prim = gid - L.first_gid;
    float prim_pdf = sc.area_pdf[L.tri_offset + prim];
    vec3 wi = si.p - pn_p;
    float dist2 = length2(wi);
    wi = div_s(wi, __builtin_sqrtf(dist2));
    float pdf = prim_pdf / si.prim_area * dist2 / max_f(abs_f(dot(si.ng, wi)), 1e-6f);
    return light_choice_pdf * pdf;
}

// The tables the shading phase gathers from, copied to LDS once per workgroup; `staged` (a copy of `p`) gets pointers to the
// copies. The shading phase is a chain of dependent gathers; from LDS each link costs a fraction of an L1 hit through the
// texture path, let alone of an L2 / HBM round trip, and a workgroup of the path tracer lives for a whole launch (16 passes x
// 64 spp). Exhaustive path (BVH = false: small scene; the host sends a scene down this path only if everything fits,
// scene_build.cpp): shading records, normals, instance transforms, materials, light tables -- 11.5 KB for the cbox.
// BVH path: instance transforms, materials and light tables, behind the traversal stacks, when the host found that they fit
// (PtParams.stage_total != 0); the per-triangle records stay in HBM. Must be called by every thread of the workgroup.
template <bool BVH, bool TEX = false, bool GGX = false>
AKR_D void stage_scene_tables(const PtParams& p, uint32_t* lds, PtParams& staged) {
    const void* src[13] = {p.sc.shade,      p.sc.normals, p.sc.inst,      p.sc.materials, p.sc.light_alias, p.sc.area_alias,
                           p.sc.lights,     p.sc.light_pdf, p.sc.area_pdf, p.sc.tex.nodes, p.sc.tex.images, p.sc.tex.mat_inputs,
                           p.sc.ggx_table};
    uint32_t* dst[13];
    uint32_t off = BVH ? p.sc.bvh_stack_depth * 256u : 0u;  // in words, behind the stacks
#pragma unroll
    for (int e = BVH ? 2 : 0; e < 13; e++) {
        if ((e >= 9 && e < 12 && !TEX) || (e == 12 && !GGX)) continue;
        const uint32_t n = p.stage_bytes[e] >> 2;
        const uint32_t* g = (const uint32_t*)src[e];
        uint32_t* l = lds + off;
        for (uint32_t i = threadIdx.x; i < n; i += 256u) l[i] = g[i];
        dst[e] = l;  // unconditionally an LDS address: the compiler then reads the tables with ds_read, not flat loads
        off += ((p.stage_bytes[e] + 15u) & ~15u) >> 2;
    }
    __syncthreads();
    if (!BVH) {
        staged.sc.shade = (const float4*)dst[0];
        staged.sc.normals = (const float4*)dst[1];  // non-null even without normals: only gates reading the flags in shade row 7
    }
    staged.sc.inst = (const float4*)dst[2];
    staged.sc.materials = (const DMaterial*)dst[3];
    staged.sc.light_alias = (const AliasPacked*)dst[4];
    staged.sc.area_alias = (const AliasPacked*)dst[5];
    staged.sc.lights = (const LightRec*)dst[6];
    staged.sc.light_pdf = (const float*)dst[7];
    staged.sc.area_pdf = (const float*)dst[8];
    // Texture-fed materials: a textured hit walks its node list (32 B per node), reads image headers and the raw input record
    // in dependent chains; from LDS each link is a ds_read instead of an L1 / L2 round trip. Texels stay in HBM.
    // The host stages either all of it or nothing (api.cpp fill_params, scene_build.cpp), so a TEX kernel that stages at all
    // reads the texture tables through LDS addresses unconditionally.
    // The 16 KB albedo table of the specular layer / coat (three to five trilinear lookups of 8 gathers each per shaded vertex),
    // for the kernels of scenes with textures, whose L1 is busy with texels: textured room 822 -> 861 Msamples/s. The plain
    // full-graph kernel is better off with the table in L1 (C3: 1887 with, 1851 without L1 -- LDS bank conflicts of the random
    // gathers). The host leaves stage_bytes[12] at 0 when the workgroup's LDS budget is spent; the table then stays where it is.
    if (GGX && p.stage_bytes[12] != 0) staged.sc.ggx_table = (const float*)dst[12];
    if (TEX) {
        staged.sc.tex.nodes = (const DNode*)dst[9];
        staged.sc.tex.images = (const DImage*)dst[10];
        staged.sc.tex.mat_inputs = (const MatInputs*)dst[11];
    }
}

// Per-thread intersection context: the LDS stack slot of this lane and the traversal counters.
struct TraceCtx {
    uint32_t* stack;
    TraceCounters cnt;
};
template <bool BVH, bool ANY_HIT>
AKR_D bool trace(const PtParams& p, TraceCtx& tc, vec3 o, vec3 d, float tmin, float tmax, uint32_t ex0, uint32_t ex1, Hit& hit) {
    if (BVH) return trace_bvh<ANY_HIT>(p.sc, o, d, tmin, tmax, ex0, ex1, hit, tc.stack, tc.cnt);
    return trace_exhaustive<ANY_HIT>(p.sc, o, d, tmin, tmax, ex0, ex1, hit);
}

AKR_D uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}


// ----------------------------------------------------------------------------------------------------------
// One lane's path-tracing state (PathTracerBase, pt.rs:28-57, plus the bookkeeping of the kernel closure
// pt.rs:1077-1102). The megakernel keeps it in registers; the wavefront pipeline streams it through HBM.
struct PathRegs {
    vec3 ro, rd;            // next closest-hit ray
    uint32_t ray_ex0;
    vec3 radiance, beta, base;
    uint32_t depth;
    float prev_bsdf_pdf;
    // shadow ray of the vertex shaded last, traced together with the next closest-hit ray
    vec3 s_o, s_d, s_contrib;
    float s_tmax;
    uint32_t s_ex0, s_ex1;
    bool has_ray, has_shadow, s_add, s_depth1, finalize, lane_done, active;
    uint32_t samples_done, pass_idx, cur_spp;
    Sampler smp;
    vec3 film_rgb;
    float film_w;
    uint32_t c_samples, c_closest, c_shadow, c_shaded;
    bool carry;  // BVH kernels: the lane's rays of the last intersection phase are still being traced (pt_kernels.hip: AKR_PT_STRAGGLERS)
    // a vertex whose shading was put off by one iteration (pt_kernels.hip: conductor hits are shaded on even iterations only)
    bool deferred;
    uint32_t d_gid;
    float d_u, d_v;
};

// PARK: the part of a lane's state that the shading code never touches, kept out of the register file while a vertex is shaded.
// The full-graph kernels need ~260 registers at their peak (the Principled closure tree) and run at 4 waves per SIMD, i.e. with
// 128; what does not fit the compiler spills to scratch memory -- vector-memory round trips through L1 / L2, 300 B per sample of
// fabric traffic on C3 and a quarter of the wave cycles waiting. Scratch is the only place the compiler knows; the workgroup's
// LDS has room for a column of kParkSlots words per lane (slot s of lane i at word s * 256 + i: one bank per lane, no
// conflicts). path_step<.., PARK> writes the cold fields there before it shades a vertex and reads them back after: two LDS
// instructions per field and iteration instead of a scratch store and load, and the registers are free in between. The pixel
// of the lane (pix, sx, sy: constant for the launch) lives there for the whole launch.
enum : uint32_t { PK_PIX = 0, PK_SX, PK_SY, PK_FILM, PK_FILM_W = PK_FILM + 3, PK_CNT, PK_SPP = PK_CNT + 3, PK_DEFER = PK_SPP + 3, PK_END = PK_DEFER + 3 };
static_assert(PK_END <= kParkSlots, "park column too small");
AKR_D void park_put(uint32_t* park, uint32_t slot, uint32_t v) { park[slot * 256u] = v; }
AKR_D uint32_t park_get(const uint32_t* park, uint32_t slot) { return park[slot * 256u]; }

template <bool PMJ = false>
AKR_D void path_regs_init(PathRegs& r, const PtParams& p, bool active, uint32_t pix, uint32_t sx, uint32_t sy) {
    const size_t N = (size_t)p.width * p.height;
    r.ro = mk3(0, 0, 0); r.rd = mk3(0, 0, 1); r.ray_ex0 = kInvalid;
    r.radiance = mk3(0, 0, 0); r.beta = mk3(1, 1, 1); r.base = mk3(0, 0, 0);
    r.depth = 0; r.prev_bsdf_pdf = 0.0f;
    r.s_o = mk3(0, 0, 0); r.s_d = mk3(0, 0, 1); r.s_contrib = mk3(0, 0, 0);
    r.s_tmax = -1.0f; r.s_ex0 = kInvalid; r.s_ex1 = kInvalid;
    r.active = active; r.has_ray = active; r.has_shadow = false; r.s_add = false; r.s_depth1 = false;
    r.finalize = false; r.lane_done = false;
    r.deferred = false; r.d_gid = kInvalid; r.d_u = 0.0f; r.d_v = 0.0f;
    r.carry = false;
    r.samples_done = 0; r.pass_idx = 0; r.c_samples = 0;
    r.cur_spp = (p.n_passes == 1) ? p.last_pass_spp : p.pass_spp;
    r.c_closest = 0; r.c_shadow = 0; r.c_shaded = 0;
    r.smp.pcg = Pcg32{0, 1}; r.smp.dim = 0;
    r.film_rgb = mk3(0, 0, 0); r.film_w = 0.0f;
    if (active) {
        r.smp.pcg = p.states[pix];  // SamplerCreator::create, sampler/mod.rs:317-327
        r.film_rgb = mk3(p.film[3 * (size_t)pix + 0], p.film[3 * (size_t)pix + 1], p.film[3 * (size_t)pix + 2]);
        r.film_w = p.film[6 * N + pix];
        sampler_start<PMJ>(p, r.smp);  // sampler.start()
        generate_ray<PMJ>(p, sx, sy, r.smp, r.ro, r.rd);
    }
}

// shifted pixel (pt.rs:1084-1088)
AKR_D void shifted_pixel(const PtParams& p, uint32_t px, uint32_t py, uint32_t& sx, uint32_t& sy) {
    int32_t sxi = (int32_t)px + p.pixel_offset[0], syi = (int32_t)py + p.pixel_offset[1];
    sxi = sxi < 0 ? 0 : (sxi > (int32_t)p.width - 1 ? (int32_t)p.width - 1 : sxi);
    syi = syi < 0 ? 0 : (syi > (int32_t)p.height - 1 ? (int32_t)p.height - 1 : syi);
    sx = (uint32_t)sxi;
    sy = (uint32_t)syi;
}

// Everything between two intersection phases for one lane: resolve the shadow ray traced together with `hit`, finish
// the previous sample if it ended, shade the vertex found by the closest-hit ray (emission + MIS, light sample, BSDF
// evaluate + sample, Russian roulette), and prepare the next pair of rays or the next camera ray.
// FD: 1 / 0 = force_diffuse known at compile time (the reference's JIT also specialises the kernel on it: the branch
// at pt.rs:268 is taken while tracing the kernel, so a force_diffuse kernel contains no Principled code); -1 = read
// p.force_diffuse at run time.
template <int FD = -1, bool TEX = false, bool PMJ = false, int PARK = 0, uint32_t ABSENT = 0, bool INST = false>  // PARK: 0 no, 1 yes, 2 yes without the DEFER fields; ABSENT: dbsdf.h AB_*; INST: meshes + instances (dinst_trav.h)
AKR_D void path_step(const PtParams& p, PathRegs& r, const Hit& hit, bool found, bool occluded, uint32_t pix_in, uint32_t sx_in, uint32_t sy_in,
                     uint32_t* park = nullptr) {
    const bool force_diffuse = FD < 0 ? (p.force_diffuse != 0) : (FD != 0);
    // PARK: the lane's pixel comes from its LDS column where it is needed (the arguments are ignored)
    auto pix_of = [&]() { return PARK ? park_get(park, PK_PIX) : pix_in; };
    const DScene& sc = p.sc;
    const size_t N = (size_t)p.width * p.height;
    // ---- resolve the shadow ray (pt.rs:504-513) ----
    if (r.has_shadow) {
        if (!occluded && r.s_add) r.radiance = r.radiance + r.s_contrib;
        if (r.s_depth1) r.base = r.radiance;
        r.has_shadow = false;
    }
    // ---- finish the sample whose last vertex was shaded in the previous step ----
    if (r.finalize) {
        // pt.rs:871-876 (clamp_indirect = 1000), then film.add_sample with weight 1 (film.rs:196-229)
        vec3 ind = r.radiance - r.base;
        ind = mk3(clamp_f(ind.x, 0.0f, 1000.0f), clamp_f(ind.y, 0.0f, 1000.0f), clamp_f(ind.z, 0.0f, 1000.0f));
        vec3 L = r.base + ind;
        if (is_nan(L.x) || is_nan(L.y) || is_nan(L.z)) L = mk3(0, 0, 0);
        if (p.color & COLOR_REPR_ACES) L = cs_convert(L, true, false);  // the film is sRGB: color.to_rgb(SRgb), film.rs:218, color.rs:262-275
        r.film_rgb = mk3(r.film_rgb.x + L.x * 1.0f, r.film_rgb.y + L.y * 1.0f, r.film_rgb.z + L.z * 1.0f);
        r.film_w = r.film_w + 1.0f;
        r.radiance = mk3(0, 0, 0);
        r.beta = mk3(1, 1, 1);
        r.base = mk3(0, 0, 0);
        r.depth = 0;
        r.prev_bsdf_pdf = 0.0f;
        r.finalize = false;
        if (r.lane_done) {
            r.active = false;
            const uint32_t pix = pix_of();
            p.states[pix] = r.smp.pcg;
            p.film[3 * (size_t)pix + 0] = r.film_rgb.x;
            p.film[3 * (size_t)pix + 1] = r.film_rgb.y;
            p.film[3 * (size_t)pix + 2] = r.film_rgb.z;
            p.film[6 * N + pix] = r.film_w;
        }
    }
    // ---- shade the vertex the closest-hit ray found ----
    if (r.active && r.has_ray) {
        bool terminated = false;
        if (PARK) {  // nothing below reads these before the sample-end bookkeeping
            park_put(park, PK_FILM + 0, f2u(r.film_rgb.x)); park_put(park, PK_FILM + 1, f2u(r.film_rgb.y)); park_put(park, PK_FILM + 2, f2u(r.film_rgb.z));
            park_put(park, PK_FILM_W, f2u(r.film_w));
            park_put(park, PK_CNT + 0, r.c_samples); park_put(park, PK_CNT + 1, r.c_closest); park_put(park, PK_CNT + 2, r.c_shadow);
            park_put(park, PK_SPP + 0, r.samples_done); park_put(park, PK_SPP + 1, r.pass_idx); park_put(park, PK_SPP + 2, r.cur_spp);
            if (PARK == 1) { park_put(park, PK_DEFER + 0, r.d_gid); park_put(park, PK_DEFER + 1, f2u(r.d_u)); park_put(park, PK_DEFER + 2, f2u(r.d_v)); }
        }
        if (!found) {
            terminated = true;  // pt.rs:381-396 (hit_envmap adds zero)
        } else {
            SurfacePoint si = surface_interaction_any<INST>(sc, hit.gid, mk2(hit.u, hit.v));
            vec3 wo = -r.rd;
            // Everything that reads the material, as a function of where the record lives: the folded record in HBM, or --
            // TEX kernels, texture-fed inputs -- a per-hit record (graph evaluated at si.uv, folded here). Two instantiations
            // in the TEX kernels: lanes on constant materials keep the register-only path and no 256-byte private copy.
            auto shade_vertex = [&](const DMaterial& mat) {
            {  // handle_surface_light, pt.rs:230-258
                vec3 direct = mk3(0, 0, 0);
                float w = 0.0f;
                if (si.light >= 0 && (!p.indirect_only || r.depth > 1)) {
                    vec3 emission = material_emission(mat);
                    direct = dot(si.ng, r.rd) < 0.0f ? emission : mk3(0, 0, 0);
                    if (r.depth == 0 || !p.use_nee)
                        w = 1.0f;
                    else
                        w = mis_weight(r.prev_bsdf_pdf, pdf_direct(sc, si, hit.gid, r.ro));
                }
                if (p.debug_depth < 0 || r.depth == (uint32_t)p.debug_depth) r.radiance = r.radiance + r.beta * (direct * w);
            }
            if (r.depth == 0) r.base = r.radiance;
            if (r.depth >= p.max_depth) {
                terminated = true;
            } else {
                r.depth += 1;
                r.c_shaded++;
                vec3 u_direct = next_3d<PMJ>(p, r.smp);
                LightSample dl;
                dl.valid = false;
                if (p.use_nee && (!p.indirect_only || r.depth > 1))
                    dl = sample_direct<TEX, INST>(sc, si.p, si.ng, u_direct.x, mk2(u_direct.y, u_direct.z));
                vec3 u_bsdf = next_3d<PMJ>(p, r.smp);
                // sample_surface_and_shade_direct, pt.rs:297-323
                ShadePoint sp;
                shade_point_init(sp, mat, si.frame, si.ng, force_diffuse, /*lean=*/!TEX || AKR_TEX_LEAN != 0, /*absent=*/ABSENT);
#ifndef AKR_NO_WO_CACHE  // (A/B switch of tools/r2_ab.sh)
                if (FD != 1) shade_point_cache_wo(sp, mat, sc.ggx_table, wo);
#endif
                if (dl.valid) {
                    BsdfEval e = shade_evaluate(sp, mat, sc.ggx_table, wo, dl.wi);
                    float w = mis_weight(dl.pdf, e.pdf);
                    vec3 direct = div_s((dl.li * e.f) * w, dl.pdf);
                    // the shadow ray is traced in the next intersection phase; what it would add is fixed now
                    // (radiance += beta * direct with the beta of THIS vertex, pt.rs:134-138,508)
                    r.s_contrib = r.beta * direct;
                    r.s_add = p.debug_depth < 0 || r.depth == (uint32_t)p.debug_depth;
                    r.s_depth1 = r.depth == 1;
                    r.s_o = dl.ro;
                    r.s_d = dl.wi;
                    r.s_tmax = dl.tmax;
                    r.s_ex0 = hit.gid;
                    r.s_ex1 = dl.ex1;
                    r.has_shadow = true;
                }
                BsdfSample bs = shade_sample(sp, mat, sc.ggx_table, wo, u_bsdf.x, mk2(u_bsdf.y, u_bsdf.z));
                r.beta = r.beta * div_s(bs.color, bs.pdf);  // pt.rs:783
                if (bs.pdf <= 0.0f || !bs.valid || min3(bs.color) < 0.0f) {
                    terminated = true;  // pt.rs:832-842
                } else {
                    bool cont = true;
                    if (r.depth > p.rr_depth) {  // pt.rs:211-224, 843-850
                        float cont_prob = clamp_f(max3(r.beta), 0.0f, 1.0f) * 0.95f;
                        if (next_1d<PMJ>(p, r.smp) >= cont_prob)
                            cont = false;
                        else
                            r.beta = r.beta * div_s(mk3(1, 1, 1), cont_prob);
                    }
                    if (!cont) {
                        terminated = true;
                    } else {  // pt.rs:851-865
                        r.prev_bsdf_pdf = bs.pdf;
                        r.ro = offset_ray_origin(si.p, face_forward(si.ng, bs.wi));
                        r.rd = bs.wi;
                        r.ray_ex0 = hit.gid;
                    }
                }
            }
            };  // shade_vertex
            const DMaterial& folded = sc.materials[si.material];
            if (TEX) {
                // ONE pass over the shading code for the lanes on textured and on constant materials alike (two call sites
                // would be two copies of it, run one after the other by every wave that holds both kinds of lane)
                DMaterial mat_here = folded;
                material_at(sc.tex, si.material, si.uv, mat_here);
                shade_vertex(mat_here);
            } else {
                shade_vertex(folded);
            }
        }
        if (PARK) {
            r.film_rgb = mk3(u2f(park_get(park, PK_FILM + 0)), u2f(park_get(park, PK_FILM + 1)), u2f(park_get(park, PK_FILM + 2)));
            r.film_w = u2f(park_get(park, PK_FILM_W));
            r.c_samples = park_get(park, PK_CNT + 0); r.c_closest = park_get(park, PK_CNT + 1); r.c_shadow = park_get(park, PK_CNT + 2);
            r.samples_done = park_get(park, PK_SPP + 0); r.pass_idx = park_get(park, PK_SPP + 1); r.cur_spp = park_get(park, PK_SPP + 2);
            if (PARK == 1) { r.d_gid = park_get(park, PK_DEFER + 0); r.d_u = u2f(park_get(park, PK_DEFER + 1)); r.d_v = u2f(park_get(park, PK_DEFER + 2)); }
        }
        if (terminated) {
            // this sample draws no more random numbers: account for it and start the next camera ray now; its
            // radiance is finished (above) after the shadow ray still pending has been resolved
            r.finalize = true;
            r.samples_done++;
            r.c_samples++;
            bool more = true;
            if (r.samples_done == r.cur_spp) {
                // end of a pass: Drop for IndependentSampler (sampler/mod.rs:168-177) = advance(-dim); the next
                // pass re-creates the sampler from that state with dim = 0 (sampler/mod.rs:317-327)
                sampler_end_pass<PMJ>(p, r.smp);
                r.samples_done = 0;
                r.pass_idx++;
                r.cur_spp = (r.pass_idx + 1 == p.n_passes) ? p.last_pass_spp : p.pass_spp;
                more = r.pass_idx < p.n_passes;
            }
            if (more) {
                sampler_start<PMJ>(p, r.smp);
                const uint32_t sx = PARK ? park_get(park, PK_SX) : sx_in, sy = PARK ? park_get(park, PK_SY) : sy_in;
                generate_ray<PMJ>(p, sx, sy, r.smp, r.ro, r.rd);
                r.ray_ex0 = kInvalid;
            } else {
                r.has_ray = false;
                r.lane_done = true;
            }
        }
    }
}

// per-launch counters (akr_pt_stats): one atomic per wave
AKR_D void flush_counters(const PtParams& p, const PathRegs& r, const TraceCounters& tc, bool bvh) {
    if (p.counters == nullptr) return;
    uint64_t* const ctr = p.counters + 8u * (blockIdx.x % kStatStripes);
    uint32_t a = wave_sum_u32(r.c_samples), b = wave_sum_u32(r.c_closest), c = wave_sum_u32(r.c_shadow), e = wave_sum_u32(r.c_shaded);
    uint32_t nn = wave_sum_u32(tc.nodes), nt = wave_sum_u32(tc.tris), ov = wave_sum_u32(tc.overflow);
    if ((threadIdx.x & 63u) == 0) {
        if (a) atomicAdd((unsigned long long*)&ctr[0], (unsigned long long)a);
        if (b) atomicAdd((unsigned long long*)&ctr[1], (unsigned long long)b);
        if (c) atomicAdd((unsigned long long*)&ctr[2], (unsigned long long)c);
        if (e) atomicAdd((unsigned long long*)&ctr[3], (unsigned long long)e);
        if (nn) atomicAdd((unsigned long long*)&ctr[4], (unsigned long long)nn);
        unsigned long long tt = bvh ? (unsigned long long)nt : (unsigned long long)(b + c) * p.sc.n_tris;
        if (tt) atomicAdd((unsigned long long*)&ctr[5], tt);
        if (ov) atomicAdd((unsigned long long*)&ctr[6], (unsigned long long)ov);
    }
}

}  // namespace akr
