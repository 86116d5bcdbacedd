// dbsdf.h -- BSDF evaluation / sampling of the HIP path tracer.
//
// The reference builds, per shading point, a tree of `Rc<dyn Surface>` closures and walks it three times
// (crates/akari_render/src/svm/surface/{mod,principled,diffuse,glass}.rs, microfacet.rs). Here every
// material's constant inputs are folded on the host into one DMaterial record (host/scene_build.cpp) and the
// tree is evaluated as straight-line code; lobes whose weight is exactly zero for the whole material are
// skipped through per-material flags (bit-identical as long as the skipped lobe is finite).
// file:line references are to the reference tree the arithmetic follows.
#pragma once
#include "dgeom.h"

namespace akr {

enum : uint32_t {
    MF_SPEC = 1u << 0,        // specular layer present: f0 != 0                       (principled.rs:55-83)
    MF_COAT = 1u << 1,        // coat_weight != 0                                      (principled.rs:84-100)
    MF_EVAL_BASE = 1u << 2,   // Mix(metallic) evaluates its `a` side: metallic < 1 - 1e-4   (mod.rs:611-615)
    MF_EVAL_METAL = 1u << 3,  // Mix(metallic) evaluates its `b` side: metallic > 1e-4
    MF_EVAL_DIFF = 1u << 4,   // Mix(transmission) evaluates diffuse: transmission < 1 - 1e-4
    MF_EVAL_DIEL = 1u << 5,   // Mix(transmission) evaluates the dielectric: transmission > 1e-4
    MF_NORMAL_MAP = 1u << 6,  // principled `normal` socket is non-zero                 (mod.rs:1380-1417)
    MF_EMISSIVE = 1u << 7,    // emission != 0
    MF_TEXTURED = 1u << 8,    // at least one input is fed by a texture expression: re-folded per hit (dtex.h)
    MF_ALPHA_TEXTURED = 1u << 9,  // ... and the alpha of the base colour is not provably 1 everywhere (alpha test evaluates the graph)
};

enum : uint32_t { MAT_PRINCIPLED = 0, MAT_DIFFUSE = 1, MAT_GLASS = 2, MAT_EMISSION = 3 };

// What a scene's shader graphs can never produce, as a compile-time mask of the kernels (`absent`): the reference traces its kernel
// from the scene's graphs (svm/compiler.rs:16-76, svm/eval.rs:428-467), so a closure no graph builds is not in its kernel either.
// A bit may be set only when the VALUE that switches the lobe on is exactly zero for every material and no texture expression
// feeds it (coat_weight, transmission_weight, metallic: a selection probability of exactly 0 is never taken and remaps u to
// (u - 0) / (1 - 0) = u; the normal socket; no Glass node) -- the same branches are skipped as at run time, but their code is gone.
// AB_SIMPLE: the four bits of the precompiled SIMPLE instantiations (scenes without textures); per-scene kernels
// (host/specialise.cpp) get the scene's own mask.
enum : uint32_t { AB_COAT = 1u, AB_TRANSMISSION = 2u, AB_NORMAL_MAP = 4u, AB_GLASS = 8u, AB_METAL = 16u, AB_SIMPLE = 15u };
AKR_HD constexpr uint32_t absent_flags(uint32_t absent) {
    return ((absent & AB_COAT) ? (uint32_t)MF_COAT : 0u) | ((absent & AB_TRANSMISSION) ? (uint32_t)MF_EVAL_DIEL : 0u) |
           ((absent & AB_NORMAL_MAP) ? (uint32_t)MF_NORMAL_MAP : 0u) | ((absent & AB_METAL) ? (uint32_t)MF_EVAL_METAL : 0u);
}

// One folded material. 64 x 4 B = 256 B, 16-byte aligned rows.
struct alignas(16) DMaterial {
    uint32_t kind, flags;
    float base_alpha, metallic;
    vec3 color;               float transmission;
    vec3 diffuse_refl;        float eta;            // color * FRAC_1_PI            (principled.rs:39-41)
    vec3 transmission_color;  float f0;             // sqrt(color)                  (principled.rs:25)
    vec3 emission;            float eta_s;          // emission_color * strength    (principled.rs:29-30)
    vec3 spec_color;          float roughness;      // specular_tint * f0           (principled.rs:71)
    vec3 spec_tint;           float coat_weight;
    vec3 coat_scale;          float coat_roughness; // lerp(1, coat_tint, coat_weight) (principled.rs:195-198)
    vec3 metal_n;             float coat_eta;
    vec3 metal_k;             float z_spec;         // sqrt(|(eta_s - 1) / (eta_s + 1)|): the table coordinate of the specular layer
    vec2 alpha;               vec2 coat_alpha;      // max(roughness^2, 1e-4)       (microfacet.rs:29-43)
    vec3 nm_normal;           float z_coat;         // normalize((-nx, -ny, nz)); the coat's table coordinate
    uint32_t tex_first_node, tex_n_nodes;           // MF_TEXTURED: pruned node list in DScene.tex.nodes
    uint32_t tex_input[14];                         // node feeding each input (dtex.h IN_*), kNodeNone = constant
};
static_assert(sizeof(DMaterial) == 256, "DMaterial layout");

// The evaluated inputs of a surface node = akr_material_desc (include/akari_hip.h), 26 words.
struct MatInputs {
    uint32_t kind;
    float base_color[3];
    float base_alpha;
    float metallic, roughness, ior, specular_ior_level;
    float specular_tint[3];
    float transmission_weight;
    float coat_weight, coat_roughness, coat_ior;
    float coat_tint[3];
    float emission_color[3];
    float emission_strength;
    float normal[3];
};
static_assert(sizeof(MatInputs) == 104, "MatInputs = akr_material_desc");

struct BsdfEval {
    vec3 f;
    float pdf;
};

// ---- colour pipeline (color.rs:262-275,614-628; svm/texture/mod.rs:9-43) -------------------------------------------
// PtParams.color / TexScene.color: ColorPipeline bits. Shading runs in the space of color_repr; constants enter through
// rgb_to_target_colorspace (their Rgb node's space -> rgb_colorspace) and spectral_uplift (rgb_colorspace -> repr space);
// the film converts every sample back to sRGB primaries. M * v = (c0 x + c1 y) + c2 z, the AKR-F32 matrix product.
enum : uint32_t { COLOR_REPR_ACES = 1u, COLOR_RGB_ACES = 2u };
enum : uint32_t { MAT_KIND_MASK = 0xffu, MAT_CS_BASE_COLOR = 0x100u, MAT_CS_SPECULAR_TINT = 0x200u, MAT_CS_COAT_TINT = 0x400u, MAT_CS_EMISSION_COLOR = 0x800u };
AKR_HD vec3 cs_convert(vec3 v, bool from_aces, bool to_aces) {
    if (from_aces == to_aces) return v;
    if (to_aces)  // srgb_to_aces_with_cat_mat, color.rs:614-620
        return mk3((0.612494199f * v.x + 0.338737252f * v.y) + 0.048855526f * v.z, (0.070594252f * v.x + 0.917671484f * v.y) + 0.011704306f * v.z,
                   (0.020727335f * v.x + 0.106882232f * v.y) + 0.872338062f * v.z);
    // aces_to_srgb_with_cat_mat, color.rs:622-628
    return mk3((1.707062673f * v.x + -0.619959540f * v.y) + -0.087259850f * v.z, (-0.130976829f * v.x + 1.139032275f * v.y) + -0.007956297f * v.z,
               (-0.024510601f * v.x + -0.124810932f * v.y) + 1.149395971f * v.z);
}
// an Rgb constant of colour space `tag_aces` through both conversions
AKR_HD vec3 color_input(vec3 v, bool tag_aces, uint32_t color) {
    v = cs_convert(v, tag_aces, (color & COLOR_RGB_ACES) != 0);
    return cs_convert(v, (color & COLOR_RGB_ACES) != 0, (color & COLOR_REPR_ACES) != 0);
}
// The constant colour inputs of a surface node (those no graph node feeds: fed[k] == false) under the pipeline `color`.
AKR_HD void convert_color_inputs(MatInputs& in, uint32_t cs_flags, uint32_t color, const bool fed[4]) {
    if (color == 0 && (cs_flags & 0xf00u) == 0) return;
    float* slot[4] = {in.base_color, in.specular_tint, in.coat_tint, in.emission_color};
    const uint32_t bit[4] = {MAT_CS_BASE_COLOR, MAT_CS_SPECULAR_TINT, MAT_CS_COAT_TINT, MAT_CS_EMISSION_COLOR};
    for (int k = 0; k < 4; k++) {
        if (fed[k]) continue;
        vec3 v = color_input(mk3(slot[k][0], slot[k][1], slot[k][2]), (cs_flags & bit[k]) != 0, color);
        slot[k][0] = v.x; slot[k][1] = v.y; slot[k][2] = v.z;
    }
}

// ---- Frame trig (geometry.rs:80-151); the reference's cos_phi uses w.y and sin_phi uses w.x ----
AKR_HD float cos_theta(vec3 w) { return w.z; }
AKR_HD float cos2_theta(vec3 w) { return w.z * w.z; }
AKR_HD float abs_cos_theta(vec3 w) { return abs_f(w.z); }
AKR_HD float sin2_theta(vec3 w) { return max_f(1.0f - cos2_theta(w), 0.0f); }
AKR_HD float sin_theta(vec3 w) { return __builtin_sqrtf(max_f(1.0f - cos2_theta(w), 0.0f)); }
AKR_HD float tan2_theta(vec3 w) { return sin2_theta(w) / cos2_theta(w); }
AKR_HD float tan_theta(vec3 w) { return sin_theta(w) / cos_theta(w); }
AKR_HD float sin_phi(vec3 w) {
    float st = sin_theta(w);
    return st == 0.0f ? 0.0f : clamp_f(w.x / st, -1.0f, 1.0f);
}
AKR_HD float cos_phi(vec3 w) {
    float st = sin_theta(w);
    return st == 0.0f ? 1.0f : clamp_f(w.y / st, -1.0f, 1.0f);
}
AKR_HD bool same_hemisphere(vec3 a, vec3 b) { return (a.z * b.z) >= 0.0f; }

// ---- Trowbridge-Reitz (microfacet.rs:45-66, 117-138, 196-206) ----
AKR_HD float tr_d(vec3 wh, vec2 alpha) {
    float tan2 = tan2_theta(wh);
    float cos4 = sqr(cos2_theta(wh));
    float e = tan2 * (sqr(cos_phi(wh) / alpha.x) + sqr(sin_phi(wh) / alpha.y));
    float inv_d = kPi * alpha.x * alpha.y * cos4 * sqr(1.0f + e);
    if (!is_finite(tan2) || !is_finite(inv_d) || inv_d == 0.0f) return 0.0f;
    return 1.0f / inv_d;
}
AKR_HD float tr_lambda(vec3 w, vec2 alpha) {
    float abs_tan = abs_f(tan_theta(w));
    float alpha2 = sqr(cos_phi(w)) * sqr(alpha.x) + sqr(sin_phi(w)) * sqr(alpha.y);
    float a2t2 = alpha2 * sqr(abs_tan);
    float l = (-1.0f + __builtin_sqrtf(1.0f + a2t2)) * 0.5f;
    return !is_finite(abs_tan) ? 0.0f : l;
}
AKR_HD float tr_g1(vec3 w, vec2 alpha) { return 1.0f / (1.0f + tr_lambda(w, alpha)); }
AKR_HD float tr_g(vec3 wo, vec3 wi, vec2 alpha) { return 1.0f / (1.0f + tr_lambda(wo, alpha) + tr_lambda(wi, alpha)); }
AKR_HD vec3 tr_sample_wh(vec3 w, vec2 u, vec2 alpha) {  // visible-normal sampling, Heitz 2018
    vec3 wh = normalize(mk3(alpha.x * w.x, alpha.y * w.y, w.z));
    if (wh.z < 0.0f) wh = -wh;
    vec3 t1 = (wh.z < 0.99999f) ? normalize(cross(mk3(0, 0, 1), wh)) : mk3(1, 0, 0);
    vec3 t2 = normalize(cross(wh, t1));
    vec2 p = uniform_sample_disk(u);
    float h = __builtin_sqrtf(1.0f - sqr(p.x));
    p.y = lerp_f(h, p.y, (1.0f + wh.z) * 0.5f);
    float pz = __builtin_sqrtf(max_f(1.0f - (p.x * p.x + p.y * p.y), 0.0f));
    vec3 nh = (t1 * p.x + t2 * p.y) + wh * pz;
    return normalize(mk3(alpha.x * nh.x, alpha.y * nh.y, max_f(nh.z, 1e-6f)));
}
AKR_HD float tr_pdf(vec3 wo, vec3 wh, vec2 alpha) {
    return tr_d(wh, alpha) * tr_g1(wo, alpha) * abs_f(dot(wo, wh)) / abs_cos_theta(wo);
}

// ---- Fresnel (svm/surface/mod.rs:1009-1098) ----
AKR_HD float fr_dielectric(float cos_i, float eta) {
    cos_i = clamp_f(cos_i, -1.0f, 1.0f);
    eta = cos_i > 0.0f ? eta : 1.0f / eta;
    cos_i = abs_f(cos_i);
    float sin2_i = 1.0f - sqr(cos_i);
    float sin2_t = sin2_i / sqr(eta);
    if (sin2_t >= 1.0f) return 1.0f;
    float cos_t = __builtin_sqrtf(max_f(1.0f - sin2_t, 0.0f));
    float r_parl = (eta * cos_i - cos_t) / (eta * cos_i + cos_t);
    float r_perp = (cos_i - eta * cos_t) / (cos_i + eta * cos_t);
    float fr = (sqr(r_parl) + sqr(r_perp)) * 0.5f;
    return clamp_f(fr, 0.0f, 1.0f);
}
struct Cplx {  // util/mod.rs:517-604
    float re, im;
};
AKR_HD Cplx cx(float re, float im) { return Cplx{re, im}; }
AKR_HD float cx_norm(Cplx a) { return a.re * a.re + a.im * a.im; }
AKR_HD Cplx cx_add(Cplx a, Cplx b) { return cx(a.re + b.re, a.im + b.im); }
AKR_HD Cplx cx_sub(Cplx a, Cplx b) { return cx(a.re - b.re, a.im - b.im); }
AKR_HD Cplx cx_mul(Cplx a, Cplx b) { return cx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
AKR_HD Cplx cx_muls(Cplx a, float s) { return cx(a.re * s, a.im * s); }
AKR_HD Cplx cx_div(Cplx a, Cplx b) {
    float scale = 1.0f / (b.re * b.re + b.im * b.im);
    return cx((a.re * b.re + a.im * b.im) * scale, (a.im * b.re - a.re * b.im) * scale);
}
AKR_HD Cplx cx_sqrt(Cplx a) {
    float n = __builtin_sqrtf(cx_norm(a));
    float t1 = __builtin_sqrtf(0.5f * (n + abs_f(a.re)));
    float t2 = 0.5f * a.im / t1;
    if (n == 0.0f) return cx(0.0f, 0.0f);
    if (a.re >= 0.0f) return cx(t1, t2);
    return cx(abs_f(t2), __builtin_copysignf(t1, a.im));
}
AKR_HD float fr_complex(float cos_i, Cplx eta) {
    cos_i = clamp_f(cos_i, 0.0f, 0.999f);
    float sin2 = 1.0f - sqr(cos_i);
    Cplx sin2_t = cx_div(cx(sin2, 0.0f), cx_mul(eta, eta));
    Cplx cos_t = cx_sqrt(cx_sub(cx(1.0f, 0.0f), sin2_t));
    Cplx r_parl = cx_div(cx_sub(cx_muls(eta, cos_i), cos_t), cx_add(cx_muls(eta, cos_i), cos_t));
    Cplx r_perp = cx_div(cx_sub(cx(cos_i, 0.0f), cx_mul(eta, cos_t)), cx_add(cx(cos_i, 0.0f), cx_mul(eta, cos_t)));
    return (cx_norm(r_parl) + cx_norm(r_perp)) * 0.5f;
}
AKR_HD void artistic_to_conductor(vec3 color, vec3 tint, vec3& n_out, vec3& k_out) {  // Gulbrandsen, mod.rs:1040-1052
    float rr[3] = {clamp_f(color.x, 0.0f, 0.99f), clamp_f(color.y, 0.0f, 0.99f), clamp_f(color.z, 0.0f, 0.99f)};
    float g[3] = {tint.x, tint.y, tint.z};
    float n[3], k[3];
    for (int i = 0; i < 3; i++) {
        float r = rr[i];
        float r_sqrt = __builtin_sqrtf(r);
        float n_min = (1.0f - r) / (1.0f + r);
        float n_max = (1.0f + r_sqrt) / (1.0f - r_sqrt);
        n[i] = lerp_f(n_max, n_min, g[i]);
        float k2 = ((n[i] + 1.0f) * (n[i] + 1.0f) * r - (n[i] - 1.0f) * (n[i] - 1.0f)) / (1.0f - r);
        k2 = max_f(k2, 0.0f);
        k[i] = __builtin_sqrtf(k2);
    }
    n_out = mk3(n[0], n[1], n[2]);
    k_out = mk3(k[0], k[1], k[2]);
}
AKR_HD float ior_from_f0(float f0) {
    float s = __builtin_sqrtf(clamp_f(f0, 0.0f, 0.99f));
    return (1.0f + s) / (1.0f - s);
}
AKR_HD float f0_from_ior(float ior) {
    float f0 = (ior - 1.0f) / (ior + 1.0f);
    return sqr(f0);
}

// ---- ggx_dielectric_s table lookup (svm/surface/mod.rs:1145-1154, 1211-1261), 16 x 16 x 16 ----
AKR_HD float table_read_1d(const float* __restrict__ buf, float x, uint32_t offset, uint32_t size) {
    x = clamp_f(x, 0.0f, 1.0f) * ((float)size - 1.0f);
    uint32_t index = (uint32_t)__builtin_floorf(x);
    uint32_t nindex = index + 1 < size - 1 ? index + 1 : size - 1;
    float t = x - (float)index;
    return (1.0f - t) * buf[offset + index] + t * buf[offset + nindex];
}
AKR_HD float table_read_2d(const float* __restrict__ buf, float x, float y, uint32_t offset, uint32_t xs, uint32_t ys) {
    y = clamp_f(y, 0.0f, 1.0f) * ((float)ys - 1.0f);
    uint32_t index = (uint32_t)__builtin_floorf(y);
    uint32_t nindex = index + 1 < ys - 1 ? index + 1 : ys - 1;
    float t = y - (float)index;
    float d0 = table_read_1d(buf, x, offset + xs * index, xs);
    float d1 = table_read_1d(buf, x, offset + xs * nindex, xs);
    return (1.0f - t) * d0 + t * d1;
}
AKR_HD float table_read_3d(const float* __restrict__ buf, float x, float y, float z) {
    const uint32_t xs = 16, ys = 16, zs = 16;
    z = clamp_f(z, 0.0f, 1.0f) * ((float)zs - 1.0f);
    uint32_t index = (uint32_t)__builtin_floorf(z);
    uint32_t nindex = index + 1 < zs - 1 ? index + 1 : zs - 1;
    float t = z - (float)index;
    float d0 = table_read_2d(buf, x, y, xs * ys * index, xs, ys);
    float d1 = table_read_2d(buf, x, y, xs * ys * nindex, xs, ys);
    return (1.0f - t) * d0 + t * d1;
}
AKR_HD float ggx_table_z(float eta) { return __builtin_sqrtf(abs_f((eta - 1.0f) / (eta + 1.0f))); }
AKR_HD float ggx_dielectric_albedo_z(const float* __restrict__ table, float roughness, float cos_i, float z) {
    cos_i = abs_f(clamp_f(cos_i, -0.999f, 0.999f));
    return table_read_3d(table, roughness, abs_f(cos_i), z);
}
AKR_HD float ggx_dielectric_albedo(const float* __restrict__ table, float roughness, float cos_i, float eta) {
    return ggx_dielectric_albedo_z(table, roughness, cos_i, ggx_table_z(eta));
}

// ---- lobes, in local shading space (z = normal) ----
enum FresnelKind { FR_DIELECTRIC = 0, FR_COMPLEX = 1 };

AKR_HD BsdfEval eval_diffuse(vec3 reflectance, vec3 wo, vec3 wi) {  // diffuse.rs:22-38
    BsdfEval r{mk3(0, 0, 0), 0.0f};
    if (same_hemisphere(wo, wi)) {
        r.pdf = abs_cos_theta(wi) * kInvPi;
        r.f = reflectance * abs_cos_theta(wi);
    }
    return r;
}
// MicrofacetReflection::evaluate_impl, svm/surface/mod.rs:831-858
template <FresnelKind FK>
AKR_HD BsdfEval eval_reflection(vec3 color, float eta, vec3 fn, vec3 fk, vec2 alpha, vec3 wo, vec3 wi) {
    BsdfEval r{mk3(0, 0, 0), 0.0f};
    vec3 wh = wo + wi;
    float cos_o = cos_theta(wo), cos_i = cos_theta(wi);
    if ((dot(wh, wo) * dot(wi, wh)) < 0.0f || (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) || cos_i == 0.0f ||
        cos_o == 0.0f || !same_hemisphere(wo, wi))
        return r;
    wh = normalize(wh);
    float c = dot(wi, face_forward(wh, mk3(0, 0, 1)));
    vec3 fr;
    if (FK == FR_DIELECTRIC) {
        float f = fr_dielectric(c, eta);
        fr = mk3(1.0f * f, 1.0f * f, 1.0f * f);
    } else {
        float ca = abs_f(c);  // FresnelComplex::evaluate passes |cos| (mod.rs:1181-1184)
        fr = mk3(fr_complex(ca, cx(fn.x, fk.x)), fr_complex(ca, cx(fn.y, fk.y)), fr_complex(ca, cx(fn.z, fk.z)));
    }
    float d = tr_d(wh, alpha);
    float g = tr_g(wo, wi, alpha);
    float k = abs_f(0.25f * d * g / (cos_i * cos_o));
    r.f = ((color * fr) * k) * abs_f(cos_i);
    r.pdf = tr_pdf(wo, wh, alpha) / (4.0f * abs_f(dot(wo, wh)));
    return r;
}
// MicrofacetTransmission::evaluate_impl, svm/surface/mod.rs:914-967 (Fresnel is always dielectric here)
AKR_HD BsdfEval eval_transmission(vec3 color, float eta_mat, vec2 alpha, vec3 wo, vec3 wi) {
    BsdfEval r{mk3(0, 0, 0), 0.0f};
    float cos_o = cos_theta(wo), cos_i = cos_theta(wi);
    float eta = cos_o > 0.0f ? eta_mat : 1.0f / eta_mat;
    vec3 wh = normalize(wo + wi * eta);
    wh = face_forward(wh, mk3(0, 0, 1));
    bool backfacing = (dot(wh, wi) * cos_i) < 0.0f || (dot(wh, wo) * cos_o) < 0.0f;
    if ((dot(wh, wo) * dot(wi, wh)) > 0.0f || cos_i == 0.0f || cos_o == 0.0f || backfacing || same_hemisphere(wo, wi)) return r;
    float f = fr_dielectric(dot(wo, wh), eta_mat);
    vec3 fr = mk3(1.0f * f, 1.0f * f, 1.0f * f);
    float denom = sqr(dot(wi, wh) + dot(wo, wh) / eta) * cos_i * cos_o;
    if (denom == 0.0f) {
        r.f = mk3(0, 0, 0);
    } else {
        float k = abs_f(tr_d(wh, alpha) * tr_g(wo, wi, alpha) / sqr(eta) * abs_f(dot(wi, wh)) * abs_f(dot(wo, wh)) / denom);
        vec3 omf = mk3(1.0f - fr.x, 1.0f - fr.y, 1.0f - fr.z);
        r.f = ((omf * color) * k) * abs_f(cos_i);
    }
    float denom2 = sqr(dot(wi, wh) + dot(wo, wh) / eta);
    float dwh_dwi = abs_f(dot(wi, wh)) / denom2;
    r.pdf = denom2 == 0.0f ? 0.0f : tr_pdf(wo, wh, alpha) * dwh_dwi;
    return r;
}
// BsdfMixture{Addictive, frac = fr_dielectric(cos wo, eta), a = transmission(kt), b = reflection(kr)}:
// the `dielectric` of principled.rs:101-131 and the whole of glass.rs:13-45
AKR_HD BsdfEval eval_dielectric(vec3 kr, vec3 kt, float eta, vec2 alpha, vec3 wo, vec3 wi) {
    float frac = fr_dielectric(cos_theta(wo), eta);
    BsdfEval a = eval_transmission(kt, eta, alpha, wo, wi);
    BsdfEval b = eval_reflection<FR_DIELECTRIC>(kr, eta, mk3(0, 0, 0), mk3(0, 0, 0), alpha, wo, wi);
    return BsdfEval{a.f + b.f, lerp_f(a.pdf, b.pdf, frac)};
}

// CoatedBsdf.e_top(w) of the specular layer / of the coat (principled.rs:158-162, 188-193)
// Folds evaluated inputs into the record the closure code reads: the part of SvmPrincipledBsdf::closure
// (principled.rs:23-131,195-205), DiffuseBsdf (diffuse.rs:83-104), SvmGlassBsdf (glass.rs:13-45) and SvmEmission
// (svm/mod.rs:114-123) that does not depend on wo / wi. One definition for the host (constant materials, once) and
// the device (textured materials, per hit). Returns false for an unknown kind. Does not touch the tex_* fields.
AKR_HD bool fold_inputs(const MatInputs& m, DMaterial& d) {
    const vec3 color = mk3(m.base_color[0], m.base_color[1], m.base_color[2]);
    d.kind = m.kind;
    d.flags = 0;
    d.base_alpha = m.base_alpha;
    d.metallic = 0.0f; d.transmission = 0.0f; d.eta = 0.0f; d.f0 = 0.0f; d.eta_s = 0.0f; d.roughness = 0.0f;
    d.coat_weight = 0.0f; d.coat_roughness = 0.0f; d.coat_eta = 0.0f; d.z_spec = 0.0f; d.z_coat = 0.0f;
    d.color = color;
    d.diffuse_refl = mk3(0, 0, 0); d.transmission_color = mk3(0, 0, 0); d.spec_color = mk3(0, 0, 0); d.spec_tint = mk3(0, 0, 0);
    d.coat_scale = mk3(0, 0, 0); d.metal_n = mk3(0, 0, 0); d.metal_k = mk3(0, 0, 0);
    d.alpha = mk2(0, 0); d.coat_alpha = mk2(0, 0);
    d.emission = mk3(m.emission_color[0], m.emission_color[1], m.emission_color[2]) * m.emission_strength;
    d.nm_normal = mk3(0, 0, 1);
    switch (m.kind) {
        case MAT_PRINCIPLED: {
            d.metallic = m.metallic;
            d.transmission = m.transmission_weight;
            d.roughness = m.roughness;
            d.eta = m.ior;
            // (the lobes a material's flags switch off never read their constants: skip the square roots and divisions of the
            // dielectric's transmission colour and of the conductor's (n, k) -- this runs per hit for texture-fed materials)
            if (m.transmission_weight > 1e-4f) d.transmission_color = mk3(__builtin_sqrtf(color.x), __builtin_sqrtf(color.y), __builtin_sqrtf(color.z));
            d.diffuse_refl = color * kInvPi;
            d.spec_tint = mk3(m.specular_tint[0], m.specular_tint[1], m.specular_tint[2]);
            float eta_s = m.ior, f0 = f0_from_ior(eta_s);
            if (m.specular_ior_level != 0.5f) {
                f0 *= 2.0f * m.specular_ior_level;
                eta_s = ior_from_f0(f0);
            }
            d.f0 = f0;
            d.eta_s = eta_s;
            d.spec_color = d.spec_tint * f0;
            d.coat_weight = m.coat_weight;
            d.coat_roughness = m.coat_roughness;
            d.coat_eta = m.coat_ior;
            // the eta coordinate of the albedo table depends on the material only: once here, not at each of the five lookups of a vertex
            // The reference evaluates the specular layer and the coat whatever their weight (principled.rs:55-202). A layer of
            // weight exactly 0 adds exactly 0 and may be skipped -- unless its colour is not finite (a shader graph can feed inf
            // or NaN into specular_tint; 0 * inf is NaN and the reference's sample goes black): then it is evaluated like there.
            // (found by tools/soak.py)
            auto finite = [](float x) { return (x - x) == 0.0f; };
            const bool spec_layer = f0 != 0.0f || !(finite(d.spec_tint.x) && finite(d.spec_tint.y) && finite(d.spec_tint.z));
            if (spec_layer) d.z_spec = ggx_table_z(eta_s);
            if (m.coat_weight != 0.0f) d.z_coat = ggx_table_z(m.coat_ior);
            d.coat_scale = lerp3(mk3(1, 1, 1), mk3(m.coat_tint[0], m.coat_tint[1], m.coat_tint[2]), m.coat_weight);
            d.alpha = mk2(max_f(m.roughness * m.roughness, 1e-4f), max_f(m.roughness * m.roughness, 1e-4f));
            d.coat_alpha = mk2(max_f(m.coat_roughness * m.coat_roughness, 1e-4f), max_f(m.coat_roughness * m.coat_roughness, 1e-4f));
            if (m.metallic > 1e-4f) artistic_to_conductor(color, d.spec_tint, d.metal_n, d.metal_k);
            uint32_t fl = 0;
            if (spec_layer) fl |= MF_SPEC;
            if (m.coat_weight != 0.0f) fl |= MF_COAT;
            if (m.metallic < 1.0f - 1e-4f) fl |= MF_EVAL_BASE;
            if (m.metallic > 1e-4f) fl |= MF_EVAL_METAL;
            if (m.transmission_weight < 1.0f - 1e-4f) fl |= MF_EVAL_DIFF;
            if (m.transmission_weight > 1e-4f) fl |= MF_EVAL_DIEL;
            vec3 normal = mk3(-m.normal[0], -m.normal[1], m.normal[2]);  // principled.rs:203-205
            if (!(normal.x == 0.0f && normal.y == 0.0f && normal.z == 0.0f)) {
                fl |= MF_NORMAL_MAP;
                d.nm_normal = normalize(normal);
            }
            d.flags = fl;
            break;
        }
        case MAT_DIFFUSE:
            d.diffuse_refl = color * kInvPi;
            d.emission = mk3(0, 0, 0);
            break;
        case MAT_GLASS:
            d.eta = m.ior;
            d.roughness = m.roughness;
            d.alpha = mk2(max_f(m.roughness * m.roughness, 1e-4f), max_f(m.roughness * m.roughness, 1e-4f));
            d.emission = mk3(0, 0, 0);
            d.base_alpha = 1.0f;
            break;
        case MAT_EMISSION:
            d.base_alpha = 1.0f;
            break;
        default: return false;
    }
    if (d.emission.x != 0.0f || d.emission.y != 0.0f || d.emission.z != 0.0f) d.flags |= MF_EMISSIVE;
    return true;
}

// fold_inputs for a record that is already folded: `d` holds fold_inputs of the material's CONSTANT inputs, `m` the inputs at a
// shading point (constants + the inputs a texture expression feeds, evaluated), `fed` which inputs those are (bit k = input k of
// dtex.h IN_*), `kind` the material's kind. Recomputes exactly the fields (and flag bits) that read a fed input, with the
// expressions of fold_inputs above -- the result is fold_inputs(m, d) bit for bit (tests/test_specialise.py checks all masks on
// the host), but an input nobody feeds costs nothing. `kind` and `fed` are literals in the per-scene code that calls this
// (host/specialise.cpp), so the conditions fold away at compile time.
AKR_HD void fold_inputs_fed(uint32_t kind, uint32_t fed, const MatInputs& m, DMaterial& d) {
    const bool f_color = fed & (1u << 0), f_metallic = fed & (1u << 1), f_rough = fed & (1u << 2), f_ior = fed & (1u << 3), f_level = fed & (1u << 4),
               f_tint = fed & (1u << 5), f_trans = fed & (1u << 6), f_cw = fed & (1u << 7), f_cr = fed & (1u << 8), f_ci = fed & (1u << 9),
               f_ct = fed & (1u << 10), f_ecol = fed & (1u << 11), f_estr = fed & (1u << 12), f_normal = fed & (1u << 13);
    const vec3 color = mk3(m.base_color[0], m.base_color[1], m.base_color[2]);
    uint32_t fl = d.flags;
    auto set_flag = [&](uint32_t bit, bool on) { fl = on ? (fl | bit) : (fl & ~bit); };
    if (kind == MAT_PRINCIPLED) {
        if (f_color) {
            d.color = color;
            d.base_alpha = m.base_alpha;
            d.diffuse_refl = color * kInvPi;
        }
        if (f_metallic) d.metallic = m.metallic;
        if (f_trans) d.transmission = m.transmission_weight;
        if (f_rough) {
            d.roughness = m.roughness;
            d.alpha = mk2(max_f(m.roughness * m.roughness, 1e-4f), max_f(m.roughness * m.roughness, 1e-4f));
        }
        if (f_ior) d.eta = m.ior;
        if (f_color || f_trans)
            d.transmission_color = m.transmission_weight > 1e-4f ? mk3(__builtin_sqrtf(color.x), __builtin_sqrtf(color.y), __builtin_sqrtf(color.z)) : mk3(0, 0, 0);
        const vec3 tint = mk3(m.specular_tint[0], m.specular_tint[1], m.specular_tint[2]);
        if (f_tint) d.spec_tint = tint;
        if (f_ior || f_level) {
            float eta_s = m.ior, f0 = f0_from_ior(eta_s);
            if (m.specular_ior_level != 0.5f) {
                f0 *= 2.0f * m.specular_ior_level;
                eta_s = ior_from_f0(f0);
            }
            d.f0 = f0;
            d.eta_s = eta_s;
        }
        if (f_ior || f_level || f_tint) {
            d.spec_color = tint * d.f0;
            auto finite = [](float x) { return (x - x) == 0.0f; };
            const bool spec_layer = d.f0 != 0.0f || !(finite(tint.x) && finite(tint.y) && finite(tint.z));
            d.z_spec = spec_layer ? ggx_table_z(d.eta_s) : 0.0f;
            set_flag(MF_SPEC, spec_layer);
        }
        if (f_cw) {
            d.coat_weight = m.coat_weight;
            set_flag(MF_COAT, m.coat_weight != 0.0f);
        }
        if (f_cr) {
            d.coat_roughness = m.coat_roughness;
            d.coat_alpha = mk2(max_f(m.coat_roughness * m.coat_roughness, 1e-4f), max_f(m.coat_roughness * m.coat_roughness, 1e-4f));
        }
        if (f_ci) d.coat_eta = m.coat_ior;
        if (f_cw || f_ci) d.z_coat = m.coat_weight != 0.0f ? ggx_table_z(m.coat_ior) : 0.0f;
        if (f_cw || f_ct) d.coat_scale = lerp3(mk3(1, 1, 1), mk3(m.coat_tint[0], m.coat_tint[1], m.coat_tint[2]), m.coat_weight);
        if (f_color || f_tint || f_metallic) {
            d.metal_n = mk3(0, 0, 0);
            d.metal_k = mk3(0, 0, 0);
            if (m.metallic > 1e-4f) artistic_to_conductor(color, tint, d.metal_n, d.metal_k);
        }
        if (f_metallic) {
            set_flag(MF_EVAL_BASE, m.metallic < 1.0f - 1e-4f);
            set_flag(MF_EVAL_METAL, m.metallic > 1e-4f);
        }
        if (f_trans) {
            set_flag(MF_EVAL_DIFF, m.transmission_weight < 1.0f - 1e-4f);
            set_flag(MF_EVAL_DIEL, m.transmission_weight > 1e-4f);
        }
        if (f_normal) {
            vec3 normal = mk3(-m.normal[0], -m.normal[1], m.normal[2]);
            const bool nm = !(normal.x == 0.0f && normal.y == 0.0f && normal.z == 0.0f);
            d.nm_normal = nm ? normalize(normal) : mk3(0, 0, 1);
            set_flag(MF_NORMAL_MAP, nm);
        }
        if (f_ecol || f_estr) d.emission = mk3(m.emission_color[0], m.emission_color[1], m.emission_color[2]) * m.emission_strength;
    } else if (kind == MAT_DIFFUSE) {
        if (f_color) {
            d.color = color;
            d.base_alpha = m.base_alpha;
            d.diffuse_refl = color * kInvPi;
        }
    } else if (kind == MAT_GLASS) {
        if (f_color) d.color = color;
        if (f_ior) d.eta = m.ior;
        if (f_rough) {
            d.roughness = m.roughness;
            d.alpha = mk2(max_f(m.roughness * m.roughness, 1e-4f), max_f(m.roughness * m.roughness, 1e-4f));
        }
    } else {  // MAT_EMISSION
        if (f_color) d.color = color;
        if (f_ecol || f_estr) d.emission = mk3(m.emission_color[0], m.emission_color[1], m.emission_color[2]) * m.emission_strength;
    }
    if ((f_ecol || f_estr) && (kind == MAT_PRINCIPLED || kind == MAT_EMISSION))
        set_flag(MF_EMISSIVE, d.emission.x != 0.0f || d.emission.y != 0.0f || d.emission.z != 0.0f);
    d.flags = fl;
}

AKR_HD float albedo_spec(const DMaterial& m, const float* __restrict__ table, vec3 w) {
    return ggx_dielectric_albedo_z(table, m.roughness, abs_cos_theta(w), m.z_spec);
}
AKR_HD float albedo_coat(const DMaterial& m, const float* __restrict__ table, vec3 w) {
    return ggx_dielectric_albedo_z(table, m.coat_roughness, abs_cos_theta(w), m.z_coat);
}
AKR_HD vec3 etop_spec_of(const DMaterial& m, float albedo) { return (m.spec_tint * albedo) * m.f0; }
AKR_HD vec3 etop_coat_of(const DMaterial& m, float albedo) { return (mk3(1, 1, 1) * albedo) * m.coat_weight; }
AKR_HD vec3 etop_spec(const DMaterial& m, const float* __restrict__ table, vec3 w) { return etop_spec_of(m, albedo_spec(m, table, w)); }
AKR_HD vec3 etop_coat(const DMaterial& m, const float* __restrict__ table, vec3 w) { return etop_coat_of(m, albedo_coat(m, table, w)); }
// The table values of the OUTGOING direction of a vertex: the NEE evaluation, the lobe selection and the evaluation of the
// sampled direction all ask for them (same material, same wo, same answer) -- looked up once per vertex by the path tracer
// (shade_point_cache_wo); nullptr = look them up here.
struct WoAlbedo {
    float spec, coat;
};
AKR_HD float avg3(vec3 e) { return ((e.x + e.y) + e.z) / 3.0f; }

// The Principled closure tree of principled.rs:133-202 (inside the wrapper), evaluated for (wo, wi).
// absent (a compile-time constant where it matters; AB_* above): lobes the scene cannot have, so their flags are known to be clear
// -- the same branches are skipped as at run time, but their code is not in the kernel.
AKR_HD BsdfEval principled_eval(const DMaterial& m, const float* __restrict__ table, vec3 wo, vec3 wi, const WoAlbedo* wc = nullptr, uint32_t absent = 0) {
    const uint32_t fl = m.flags & ~absent_flags(absent);
    BsdfEval b2{mk3(0, 0, 0), 0.0f};
    if (fl & MF_EVAL_BASE) {
        // Mix(transmission){diffuse, dielectric}
        BsdfEval a{mk3(0, 0, 0), 0.0f}, b{mk3(0, 0, 0), 0.0f};
        if (fl & MF_EVAL_DIFF) a = eval_diffuse(m.diffuse_refl, wo, wi);
        if (fl & MF_EVAL_DIEL) b = eval_dielectric(m.color, m.transmission_color, m.eta, m.alpha, wo, wi);
        BsdfEval b1{lerp3(a.f, b.f, m.transmission), lerp_f(a.pdf, b.pdf, m.transmission)};
        // Coated{top: specular, bottom: b1}
        if (fl & MF_SPEC) {
            BsdfEval top = eval_reflection<FR_DIELECTRIC>(m.spec_color, m.eta_s, mk3(0, 0, 0), mk3(0, 0, 0), m.alpha, wo, wi);
            vec3 eo = wc ? etop_spec_of(m, wc->spec) : etop_spec(m, table, wo), ei = etop_spec(m, table, wi);
            float ps_top = avg3(eo);
            float ps_bottom = 1.0f - ps_top;
            b2.pdf = top.pdf * ps_top + b1.pdf * ps_bottom;
            vec3 mn = mk3(min_f(1.0f - eo.x, 1.0f - ei.x), min_f(1.0f - eo.y, 1.0f - ei.y), min_f(1.0f - eo.z, 1.0f - ei.z));
            b2.f = top.f + b1.f * mn;
        } else {
            b2 = b1;
        }
    }
    // Mix(metallic){b2, metal}
    BsdfEval mt{mk3(0, 0, 0), 0.0f};
    if (fl & MF_EVAL_METAL) mt = eval_reflection<FR_COMPLEX>(mk3(1, 1, 1), 0.0f, m.metal_n, m.metal_k, m.alpha, wo, wi);
    BsdfEval b3{lerp3(b2.f, mt.f, m.metallic), lerp_f(b2.pdf, mt.pdf, m.metallic)};
    // Emissive{b3} passes through; Scaled{lerp(1, coat_tint, coat_weight)}
    BsdfEval sc{b3.f * m.coat_scale, b3.pdf};
    if (!(fl & MF_COAT)) return sc;
    // Coated{top: coat, bottom: scaled}
    BsdfEval top = eval_reflection<FR_DIELECTRIC>(splat3(1.0f) * m.coat_weight, m.coat_eta, mk3(0, 0, 0), mk3(0, 0, 0), m.coat_alpha, wo, wi);
    vec3 eo = wc ? etop_coat_of(m, wc->coat) : etop_coat(m, table, wo), ei = etop_coat(m, table, wi);
    float ps_top = avg3(eo);
    float ps_bottom = 1.0f - ps_top;
    BsdfEval r;
    r.pdf = top.pdf * ps_top + sc.pdf * ps_bottom;
    vec3 mn = mk3(min_f(1.0f - eo.x, 1.0f - ei.x), min_f(1.0f - eo.y, 1.0f - ei.y), min_f(1.0f - eo.z, 1.0f - ei.z));
    r.f = top.f + sc.f * mn;
    return r;
}

// Lobe selection of the same tree (sample_wi_impl of CoatedBsdf mod.rs:504-522 and BsdfMixture mod.rs:627-644),
// followed by one lobe sampler. Every stochastic choice consumes and remaps u_select exactly like the reference
// (including u_select == 1.0, which the PCG stream can produce).
enum LobeKind { LOBE_DIFFUSE = 0, LOBE_REFLECT = 1, LOBE_TRANSMIT = 2 };
AKR_HD bool sample_lobe(LobeKind lobe, vec2 alpha, float eta, vec3 wo, vec2 u, vec3& wi) {
    if (lobe == LOBE_DIFFUSE) {  // diffuse.rs:40-52
        vec3 w = cos_sample_hemisphere(u);
        wi = same_hemisphere(wo, w) ? w : -w;
        return true;
    }
    vec3 wh = tr_sample_wh(wo, u, alpha);
    if (lobe == LOBE_REFLECT) {  // mod.rs:860-873
        wi = reflect(wo, wh);
        return same_hemisphere(wo, wi);
    }
    bool refracted = refract(wo, wh, eta, wi);  // mod.rs:969-982
    return refracted && !same_hemisphere(wo, wi);
}
// which lobe, which alpha, and -- for the roughness AOV -- whether it is the coat
AKR_HD void principled_select_lobe(const DMaterial& m, const float* __restrict__ table, vec3 wo, float u, LobeKind& lobe, vec2& alpha, bool& coat,
                                   const WoAlbedo* wc = nullptr, uint32_t absent = 0) {
    const uint32_t fl = m.flags & ~absent_flags(absent);
    lobe = LOBE_DIFFUSE;
    alpha = m.alpha;
    coat = false;
    float r = u;
    // Coated{coat | Scaled{Emissive{...}}}: top iff u < avg(E_coat(wo))
    // (absent: the probability is 0 -- never taken, and the remapped number is (u - 0) / (1 - 0) = u exactly; r == u holds here and
    // after every "u = r" below, so skipping a choice leaves both as they are)
    float p_coat = (fl & MF_COAT) ? avg3(wc ? etop_coat_of(m, wc->coat) : etop_coat(m, table, wo)) : 0.0f;
    if (!(absent & AB_COAT) && weighted_choice2_and_remap(p_coat, u, r)) {
        lobe = LOBE_REFLECT;
        alpha = m.coat_alpha;
        coat = true;
    } else {
        u = r;
        // Mix(metallic): b (metal) iff u < metallic  (AB_METAL: metallic is exactly 0)
        if (!(absent & AB_METAL) && weighted_choice2_and_remap(m.metallic, u, r)) {
            lobe = LOBE_REFLECT;
        } else {
            u = r;
            // Coated{specular | Mix(transmission)}: top iff u < avg(E_spec(wo))
            float p_spec = (fl & MF_SPEC) ? avg3(wc ? etop_spec_of(m, wc->spec) : etop_spec(m, table, wo)) : 0.0f;
            if (weighted_choice2_and_remap(p_spec, u, r)) {
                lobe = LOBE_REFLECT;
            } else {
                u = r;
                // Mix(transmission): b (dielectric) iff u < transmission  (AB_TRANSMISSION: transmission is exactly 0)
                if (!(absent & AB_TRANSMISSION) && weighted_choice2_and_remap(m.transmission, u, r)) {
                    u = r;
                    // Addictive{transmission, reflection}: b (reflection) iff u < fr_dielectric(cos wo, eta)
                    float frac = fr_dielectric(cos_theta(wo), m.eta);
                    lobe = weighted_choice2_and_remap(frac, u, r) ? LOBE_REFLECT : LOBE_TRANSMIT;
                } else {
                    lobe = LOBE_DIFFUSE;
                }
            }
        }
    }
}
AKR_HD bool principled_sample_wi(const DMaterial& m, const float* __restrict__ table, vec3 wo, float u, vec2 u2, vec3& wi, const WoAlbedo* wc = nullptr,
                                 uint32_t absent = 0) {
    LobeKind lobe;
    vec2 alpha;
    bool coat;
    principled_select_lobe(m, table, wo, u, lobe, alpha, coat, wc, absent);
    if ((absent & AB_TRANSMISSION) && lobe == LOBE_TRANSMIT) lobe = LOBE_DIFFUSE;  // unreachable: tells the compiler to drop the refraction code
    return sample_lobe(lobe, alpha, m.eta, wo, u2, wi);
}

// ---- SurfaceClosure (svm/surface/mod.rs:697-816): light-leak guard + world <-> local ----
AKR_HD bool check_wo_wi_valid(vec3 ns, vec3 ng, vec3 wo, vec3 wi) {
    auto sgn = [](float x) { return x > 0.0f ? 1.0f : -1.0f; };
    float flipped = sgn(dot(ng, ns));
    bool a = sgn(flipped * dot(wo, ns)) * sgn(dot(wo, ng)) > 0.0f;
    bool b = sgn(flipped * dot(wi, ns)) * sgn(dot(wi, ng)) > 0.0f;
    return a & b;
}

// Everything the shading functions need at one path vertex.
// lean: the inner (normal-map) frame and the local geometric normal are not stored but recomputed where they are read -- the same
// operations on the same inputs, the same bits, and 12 fewer registers live from the NEE evaluation to the end of the BSDF sample.
// The path tracer's kernels for scenes without textures run lean (137 -> 51 spilled registers in the full-graph exhaustive
// kernel at 128 VGPRs, +2 % on C3); the TEX kernels, which run at 168 VGPRs, lose 3-5 % to the recomputation and do not.
// `lean` is a compile-time constant wherever a ShadePoint is used, so the branch and the unused members fold away.
struct ShadePoint {
    Frame frame;        // si.frame
    vec3 ng;            // si.ng
    Frame nm_frame_;    // inner (normal-map) frame in local space; identity when !MF_NORMAL_MAP   (not lean)
    vec3 ng_local_;     // frame.to_local(ng)  (normal_map(): ng of the inner SurfaceClosure)       (not lean)
    bool lean;
    uint32_t absent;    // AB_* mask, see principled_eval
    bool force_diffuse;
    bool wo_cached;     // wo_albedo holds the table values of the vertex's outgoing direction (shade_point_cache_wo)
    WoAlbedo wo_albedo;
};
// normal_map(), svm/surface/mod.rs:1380-1417, for a constant `normal` input: the inner frame of a normal-mapped Principled material
AKR_HD Frame nm_frame_compute(const Frame& frame, const DMaterial& m) {
    vec3 n_world = to_world(frame, m.nm_normal);
    Frame nf = frame_from_n_t(n_world, frame.t);
    Frame r;
    r.t = to_local(frame, nf.t);
    r.s = to_local(frame, nf.s);
    r.n = to_local(frame, nf.n);
    return r;
}
AKR_HD Frame sp_nm_frame(const ShadePoint& sp, const DMaterial& m) {
    if (!sp.lean) return sp.nm_frame_;
    if (!(sp.absent & AB_NORMAL_MAP) && !sp.force_diffuse && m.kind == MAT_PRINCIPLED && (m.flags & MF_NORMAL_MAP)) return nm_frame_compute(sp.frame, m);
    return Frame{mk3(0, 0, 1), mk3(1, 0, 0), mk3(0, 1, 0)};
}
AKR_HD vec3 sp_ng_local(const ShadePoint& sp) { return sp.lean ? to_local(sp.frame, sp.ng) : sp.ng_local_; }

AKR_HD void shade_point_init(ShadePoint& sp, const DMaterial& m, Frame frame, vec3 ng, bool force_diffuse, bool lean = false, uint32_t absent = 0) {
    sp.frame = frame;
    sp.ng = ng;
    sp.lean = lean;
    sp.absent = absent;
    sp.force_diffuse = force_diffuse;
    sp.wo_cached = false;
    sp.wo_albedo = WoAlbedo{0.0f, 0.0f};
    sp.nm_frame_ = Frame{mk3(0, 0, 1), mk3(1, 0, 0), mk3(0, 1, 0)};
    sp.ng_local_ = mk3(0, 0, 0);
    if (!lean) {
        sp.ng_local_ = to_local(frame, ng);
        if (!(absent & AB_NORMAL_MAP) && !force_diffuse && m.kind == MAT_PRINCIPLED && (m.flags & MF_NORMAL_MAP)) sp.nm_frame_ = nm_frame_compute(frame, m);
    }
}

// After shade_point_init, for a vertex all of whose evaluate / sample calls use this `wo` (world space): the albedo-table values
// of wo, once. The calls that follow must pass the same wo.
AKR_HD void shade_point_cache_wo(ShadePoint& sp, const DMaterial& m, const float* __restrict__ table, vec3 wo) {
    if (sp.force_diffuse || m.kind != MAT_PRINCIPLED) return;
    vec3 lo = to_local(sp.frame, wo);
    if (!(sp.absent & AB_NORMAL_MAP) && (m.flags & MF_NORMAL_MAP)) lo = to_local(sp_nm_frame(sp, m), lo);
    if (m.flags & MF_SPEC) sp.wo_albedo.spec = albedo_spec(m, table, lo);
    if (!(sp.absent & AB_COAT) && (m.flags & MF_COAT)) sp.wo_albedo.coat = albedo_coat(m, table, lo);
    sp.wo_cached = true;
}

// closure.evaluate(wo, wi) for world-space directions -> (f * |cos|, pdf); pt.rs:268-279 for force_diffuse
AKR_HD BsdfEval shade_evaluate(const ShadePoint& sp, const DMaterial& m, const float* __restrict__ table, vec3 wo, vec3 wi) {
    BsdfEval zero{mk3(0, 0, 0), 0.0f};
    if (!check_wo_wi_valid(sp.frame.n, sp.ng, wo, wi)) return zero;
    vec3 lo = to_local(sp.frame, wo), li = to_local(sp.frame, wi);
    if (sp.force_diffuse) {
        float r = (1.0f * kInvPi) * 0.8f;
        return eval_diffuse(splat3(r), lo, li);
    }
    switch (m.kind) {
        case MAT_PRINCIPLED: {
            if (!(sp.absent & AB_NORMAL_MAP) && (m.flags & MF_NORMAL_MAP)) {
                const Frame nf = sp_nm_frame(sp, m);
                if (!check_wo_wi_valid(nf.n, sp_ng_local(sp), lo, li)) return zero;
                lo = to_local(nf, lo);
                li = to_local(nf, li);
            } else {
                if (!check_wo_wi_valid(mk3(0, 0, 1), sp_ng_local(sp), lo, li)) return zero;
            }
            return principled_eval(m, table, lo, li, sp.wo_cached ? &sp.wo_albedo : nullptr, sp.absent);
        }
        case MAT_DIFFUSE: return eval_diffuse(m.diffuse_refl, lo, li);
        case MAT_GLASS: return (sp.absent & AB_GLASS) ? zero : eval_dielectric(m.color, m.color, m.eta, m.alpha, lo, li);
        default: return zero;  // Emission node: EmissiveSurface{inner: None}
    }
}

struct BsdfSample {  // svm/surface/mod.rs:35-51
    vec3 wi;
    float pdf;
    vec3 color;
    bool valid;
};
// SurfaceClosure::sample, svm/surface/mod.rs:795-815
AKR_HD BsdfSample shade_sample(const ShadePoint& sp, const DMaterial& m, const float* __restrict__ table, vec3 wo, float u_select,
                               vec2 u_sample) {
    BsdfSample s{mk3(0, 0, 0), 0.0f, mk3(0, 0, 0), false};
    vec3 lo = to_local(sp.frame, wo);
    vec3 wl = mk3(0, 0, 0);
    bool valid;
    if (sp.force_diffuse) {
        valid = sample_lobe(LOBE_DIFFUSE, mk2(0, 0), 1.0f, lo, u_sample, wl);
    } else {
        switch (m.kind) {
            case MAT_PRINCIPLED: {
                const bool nm = !(sp.absent & AB_NORMAL_MAP) && (m.flags & MF_NORMAL_MAP) != 0;
                const Frame nf = nm ? sp_nm_frame(sp, m) : Frame{mk3(0, 0, 1), mk3(1, 0, 0), mk3(0, 1, 0)};
                vec3 lo2 = nm ? to_local(nf, lo) : lo;
                vec3 w2;
                valid = principled_sample_wi(m, table, lo2, u_select, u_sample, w2, sp.wo_cached ? &sp.wo_albedo : nullptr, sp.absent);
                wl = nm ? to_world(nf, w2) : w2;
                valid = valid & check_wo_wi_valid(nf.n, sp_ng_local(sp), lo, wl);
                break;
            }
            case MAT_DIFFUSE: valid = sample_lobe(LOBE_DIFFUSE, mk2(0, 0), 1.0f, lo, u_sample, wl); break;
            case MAT_GLASS: {
                if (sp.absent & AB_GLASS) { valid = false; break; }  // no glass material in the scene
                float frac = fr_dielectric(cos_theta(lo), m.eta), r;
                LobeKind lobe = weighted_choice2_and_remap(frac, u_select, r) ? LOBE_REFLECT : LOBE_TRANSMIT;
                valid = sample_lobe(lobe, m.alpha, m.eta, lo, u_sample, wl);
                break;
            }
            default: valid = false; break;
        }
    }
    vec3 wi = to_world(sp.frame, wl);
    valid = valid & check_wo_wi_valid(sp.frame.n, sp.ng, wo, wi);
    if (!valid) return s;
    BsdfEval e = shade_evaluate(sp, m, table, wo, wi);
    s.wi = wi;
    s.color = e.f;
    s.pdf = e.pdf;
    s.valid = valid & (e.pdf > 0.0f);
    return s;
}

// ---- AOVs of the closure (akari_integrator/src/aov.rs:100-160) --------------------------------------------------
// closure.ns(): SurfaceClosure::ns = frame.to_world(inner.ns()) (mod.rs:724-727). Inner: the Principled wrapper sits in the
// normal-map closure (its ns = (0,0,1) seen through nm_frame, principled.rs:224-226, mod.rs:1380-1417); Diffuse, the
// reflection / transmission lobes and an Emission node report (0,0,1); Glass is an additive mixture: normalize(a + b)
// (mod.rs:588-590).
AKR_HD vec3 shade_ns(const ShadePoint& sp, const DMaterial& m) {
    vec3 ns = mk3(0, 0, 1);
    if (m.kind == MAT_PRINCIPLED) ns = to_world(sp_nm_frame(sp, m), mk3(0, 0, 1));
    else if (m.kind == MAT_GLASS) ns = normalize(mk3(0, 0, 1) + mk3(0, 0, 1));
    return to_world(sp.frame, ns);
}
// closure.albedo(wo) + closure.emission(wo): the Principled wrapper answers with its base colour and emission
// (principled.rs:227-274); Diffuse reflectance * PI (diffuse.rs:56-63); Glass kt + kr (mod.rs:659-675,875-882,981-988);
// an Emission node has no inner surface: albedo 0 (mod.rs:372-383).
AKR_HD vec3 shade_albedo_plus_emission(const DMaterial& m) {
    switch (m.kind) {
        case MAT_PRINCIPLED: return m.color + m.emission;
        case MAT_DIFFUSE: return m.diffuse_refl * kPi + mk3(0, 0, 0);
        case MAT_GLASS: return (m.color + m.color) + (mk3(0, 0, 0) + mk3(0, 0, 0));
        default: return mk3(0, 0, 0) + m.emission;
    }
}
// closure.roughness(wo, u): the lobe `u` would select, then that lobe's roughness; diffuse lobes report 1
// (mod.rs:537-556,642-657, diffuse.rs:64-72, microfacet.rs:208-210)
AKR_HD float shade_roughness(const ShadePoint& sp, const DMaterial& m, const float* __restrict__ table, vec3 wo, float u) {
    vec3 lo = to_local(sp.frame, wo);
    switch (m.kind) {
        case MAT_PRINCIPLED: {
            vec3 lo2 = (m.flags & MF_NORMAL_MAP) ? to_local(sp_nm_frame(sp, m), lo) : lo;
            LobeKind lobe;
            vec2 alpha;
            bool coat;
            principled_select_lobe(m, table, lo2, u, lobe, alpha, coat);
            return lobe == LOBE_DIFFUSE ? 1.0f : (coat ? m.coat_roughness : m.roughness);
        }
        case MAT_GLASS: return m.roughness;  // both sides of the mixture carry the same distribution
        default: return 1.0f;
    }
}

}  // namespace akr
