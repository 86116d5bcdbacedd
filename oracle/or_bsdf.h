/* or_bsdf.h -- BSDF closure tree of the reference, restated for the CPU oracle.
 * TEST INFRASTRUCTURE ONLY. Follows crates/akari_render/src/svm/surface/{mod,diffuse,principled,glass}.rs,
 * microfacet.rs and util/mod.rs:509-604 (file:line cited per function). The reference builds a tree of
 * `Rc<dyn Surface>` per shading point; here the same tree is a small array of tagged nodes.
 */
#ifndef OR_BSDF_H
#define OR_BSDF_H
#include "or_geom.h"

typedef enum {
    OR_S_NULL = 0,
    OR_S_DIFFUSE,      /* diffuse.rs:13-82 */
    OR_S_MF_REFL,      /* svm/surface/mod.rs:820-900 */
    OR_S_MF_TRANS,     /* svm/surface/mod.rs:902-1006 */
    OR_S_MIXTURE,      /* svm/surface/mod.rs:568-695 */
    OR_S_COATED,       /* svm/surface/mod.rs:476-567 */
    OR_S_SCALED,       /* svm/surface/mod.rs:412-475 */
    OR_S_EMISSIVE,     /* svm/surface/mod.rs:330-411 */
    OR_S_PRINCIPLED,   /* principled.rs:218-275 PrincipledBsdfWrapper */
    OR_S_CLOSURE       /* svm/surface/mod.rs:697-816 SurfaceClosure */
} or_surface_kind;

enum { OR_FR_DIELECTRIC = 0, OR_FR_COMPLEX = 1, OR_FR_CONST = 2 };
enum { OR_BLEND_ADDICTIVE = 0, OR_BLEND_MIX = 1 };
enum { OR_FRAC_CONST = 0, OR_FRAC_FR_DIELECTRIC = 1 };

typedef struct or_surface {
    int kind;
    v3 color;            /* reflectance / lobe colour / scale weight / wrapper albedo */
    v3 emission;         /* EMISSIVE, PRINCIPLED */
    /* microfacet lobes */
    int fresnel;
    float eta;           /* FresnelDielectric.eta; MicrofacetTransmission.eta */
    v3 fn, fk;           /* FresnelComplex */
    v2 alpha;            /* TrowbridgeReitzDistribution.alpha */
    float roughness;     /* TrowbridgeReitzDistribution.roughness (what from_roughness was given), microfacet.rs:27,208-210 */
    /* combinators */
    struct or_surface *a, *b; /* MIXTURE: bsdf_a/bsdf_b; COATED: a = top, b = bottom; others: a = inner */
    int mode, frac_kind;
    float frac_const, frac_eta;
    /* CoatedBsdf.e_top(w) = etop_tint * A(etop_roughness, |cos w|, etop_eta) * etop_weight */
    v3 etop_tint;
    float etop_weight, etop_roughness, etop_eta;
    const float *table;  /* 16x16x16 ggx_dielectric_s */
    /* CLOSURE */
    or_frame frame;
    v3 ng;
} or_surface;

/* ---------- Frame trig (geometry.rs:80-151); note cos_phi uses w.y and sin_phi uses w.x ---------- */
static inline float fr_cos_theta(v3 w) { return w.z; }
static inline float fr_cos2_theta(v3 w) { return w.z * w.z; }
static inline float fr_abs_cos_theta(v3 w) { return fabsf(w.z); }
static inline float fr_sin2_theta(v3 w) { return or_max(1.0f - fr_cos2_theta(w), 0.0f); }
static inline float fr_sin_theta(v3 w) { return sqrtf(or_max(1.0f - fr_cos2_theta(w), 0.0f)); }
static inline float fr_tan2_theta(v3 w) { return fr_sin2_theta(w) / fr_cos2_theta(w); }
static inline float fr_tan_theta(v3 w) { return fr_sin_theta(w) / fr_cos_theta(w); }
static inline float fr_sin_phi(v3 w) {
    float st = fr_sin_theta(w);
    return st == 0.0f ? 0.0f : or_clamp(w.x / st, -1.0f, 1.0f);
}
static inline float fr_cos_phi(v3 w) {
    float st = fr_sin_theta(w);
    return st == 0.0f ? 1.0f : or_clamp(w.y / st, -1.0f, 1.0f);
}
static inline int fr_same_hemisphere(v3 a, v3 b) { return (a.z * b.z) >= 0.0f; }

/* ---------- GGX (microfacet.rs:29-138, 196-206) ---------- */
static inline v2 tr_alpha_from_roughness(float rx, float ry) {
    return V2(or_max(rx * rx, 1e-4f), or_max(ry * ry, 1e-4f));
}
static inline float tr_d(v3 wh, v2 alpha) { /* microfacet.rs:45-58 */
    float tan2_theta = fr_tan2_theta(wh);
    float cos4_theta = or_sqr(fr_cos2_theta(wh));
    float e = tan2_theta * (or_sqr(fr_cos_phi(wh) / alpha.x) + or_sqr(fr_sin_phi(wh) / alpha.y));
    float inv_d = OR_PI * alpha.x * alpha.y * cos4_theta * or_sqr(1.0f + e);
    if (!or_isfinite(tan2_theta) || !or_isfinite(inv_d) || inv_d == 0.0f) return 0.0f;
    return 1.0f / inv_d;
}
static inline float tr_lambda(v3 w, v2 alpha) { /* microfacet.rs:59-66 */
    float abs_tan_theta = fabsf(fr_tan_theta(w));
    float alpha2 = or_sqr(fr_cos_phi(w)) * or_sqr(alpha.x) + or_sqr(fr_sin_phi(w)) * or_sqr(alpha.y);
    float alpha2_tan2_theta = alpha2 * or_sqr(abs_tan_theta);
    float l = (-1.0f + sqrtf(1.0f + alpha2_tan2_theta)) * 0.5f;
    return !or_isfinite(abs_tan_theta) ? 0.0f : l;
}
static inline float tr_g1(v3 w, v2 alpha) { return 1.0f / (1.0f + tr_lambda(w, alpha)); }
static inline float tr_g(v3 wo, v3 wi, v2 alpha) { return 1.0f / (1.0f + tr_lambda(wo, alpha) + tr_lambda(wi, alpha)); }
/* microfacet.rs:117-138 (sample_visible = true everywhere on this path) */
static inline v3 tr_sample_wh(v3 w, v2 u, v2 alpha) {
    v3 wh = v3normalize(V3(alpha.x * w.x, alpha.y * w.y, w.z));
    if (wh.z < 0.0f) wh = v3neg(wh);
    v3 t1 = (wh.z < 0.99999f) ? v3normalize(v3cross(V3(0, 0, 1), wh)) : V3(1, 0, 0);
    v3 t2 = v3normalize(v3cross(wh, t1));
    v2 p = or_uniform_sample_disk(u);
    float h = sqrtf(1.0f - or_sqr(p.x));
    p.y = or_lerp(h, p.y, (1.0f + wh.z) * 0.5f);
    float pz = sqrtf(or_max(1.0f - (p.x * p.x + p.y * p.y), 0.0f));
    v3 nh = v3add(v3add(v3scale(t1, p.x), v3scale(t2, p.y)), v3scale(wh, pz));
    return v3normalize(V3(alpha.x * nh.x, alpha.y * nh.y, or_max(nh.z, 1e-6f)));
}
static inline float tr_pdf(v3 wo, v3 wh, v2 alpha) { /* microfacet.rs:196-206 */
    return tr_d(wh, alpha) * tr_g1(wo, alpha) * fabsf(v3dot(wo, wh)) / fr_abs_cos_theta(wo);
}

/* ---------- Fresnel (svm/surface/mod.rs:1009-1098) ---------- */
static inline float or_fr_dielectric(float cos_theta_i, float eta) {
    cos_theta_i = or_clamp(cos_theta_i, -1.0f, 1.0f);
    eta = cos_theta_i > 0.0f ? eta : 1.0f / eta;
    cos_theta_i = fabsf(cos_theta_i);
    float sin2_theta_i = 1.0f - or_sqr(cos_theta_i);
    float sin2_theta_t = sin2_theta_i / or_sqr(eta);
    if (sin2_theta_t >= 1.0f) return 1.0f;
    float cos_theta_t = sqrtf(or_max(1.0f - sin2_theta_t, 0.0f));
    float r_parl = (eta * cos_theta_i - cos_theta_t) / (eta * cos_theta_i + cos_theta_t);
    float r_perp = (cos_theta_i - eta * cos_theta_t) / (cos_theta_i + eta * cos_theta_t);
    float fr = (or_sqr(r_parl) + or_sqr(r_perp)) * 0.5f;
    return or_clamp(fr, 0.0f, 1.0f);
}
typedef struct { float re, im; } or_cplx; /* util/mod.rs:517-604 */
static inline or_cplx cx(float re, float im) { or_cplx c = {re, im}; return c; }
static inline float cx_norm(or_cplx a) { return a.re * a.re + a.im * a.im; }
static inline or_cplx cx_add(or_cplx a, or_cplx b) { return cx(a.re + b.re, a.im + b.im); }
static inline or_cplx cx_sub(or_cplx a, or_cplx b) { return cx(a.re - b.re, a.im - b.im); }
static inline or_cplx cx_mul(or_cplx a, or_cplx b) { return cx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline or_cplx cx_muls(or_cplx a, float s) { return cx(a.re * s, a.im * s); }
static inline or_cplx cx_div(or_cplx a, or_cplx b) {
    float scale = 1.0f / (b.re * b.re + b.im * b.im);
    return cx((a.re * b.re + a.im * b.im) * scale, (a.im * b.re - a.re * b.im) * scale);
}
static inline or_cplx cx_sqrt(or_cplx a) {
    float n = sqrtf(cx_norm(a));
    float t1 = sqrtf(0.5f * (n + fabsf(a.re)));
    float t2 = 0.5f * a.im / t1;
    if (n == 0.0f) return cx(0.0f, 0.0f);
    if (a.re >= 0.0f) return cx(t1, t2);
    return cx(fabsf(t2), copysignf(t1, a.im));
}
static inline float or_fr_complex(float cos_theta_i, or_cplx eta) { /* svm/surface/mod.rs:1055-1068 */
    cos_theta_i = or_clamp(cos_theta_i, 0.0f, 0.999f);
    float sin2_theta = 1.0f - or_sqr(cos_theta_i);
    or_cplx sin2_theta_t = cx_div(cx(sin2_theta, 0.0f), cx_mul(eta, eta));
    or_cplx cos_theta_t = cx_sqrt(cx_sub(cx(1.0f, 0.0f), sin2_theta_t));
    or_cplx r_parl = cx_div(cx_sub(cx_muls(eta, cos_theta_i), cos_theta_t), cx_add(cx_muls(eta, cos_theta_i), cos_theta_t));
    or_cplx r_perp = cx_div(cx_sub(cx(cos_theta_i, 0.0f), cx_mul(eta, cos_theta_t)),
                            cx_add(cx(cos_theta_i, 0.0f), cx_mul(eta, cos_theta_t)));
    return (cx_norm(r_parl) + cx_norm(r_perp)) * 0.5f;
}
/* svm/surface/mod.rs:1040-1052 Gulbrandsen */
static inline void or_artistic_to_conductor(v3 color, v3 tint, v3 *n_out, v3 *k_out) {
    float rr[3] = {or_clamp(color.x, 0.0f, 0.99f), or_clamp(color.y, 0.0f, 0.99f), or_clamp(color.z, 0.0f, 0.99f)};
    float g[3] = {tint.x, tint.y, tint.z};
    float n[3], k[3];
    for (int i = 0; i < 3; i++) {
        float r = rr[i];
        float r_sqrt = sqrtf(r);
        float n_min = (1.0f - r) / (1.0f + r);
        float n_max = (1.0f + r_sqrt) / (1.0f - r_sqrt);
        n[i] = or_lerp(n_max, n_min, g[i]);
        float k2 = ((n[i] + 1.0f) * (n[i] + 1.0f) * r - (n[i] - 1.0f) * (n[i] - 1.0f)) / (1.0f - r);
        k2 = or_max(k2, 0.0f);
        k[i] = sqrtf(k2);
    }
    *n_out = V3(n[0], n[1], n[2]);
    *k_out = V3(k[0], k[1], k[2]);
}
static inline float or_ior_from_f0(float f0) { /* :1090-1093 */
    float sqrt_f0 = sqrtf(or_clamp(f0, 0.0f, 0.99f));
    return (1.0f + sqrt_f0) / (1.0f - sqrt_f0);
}
static inline float or_f0_from_ior(float ior) { float f0 = (ior - 1.0f) / (ior + 1.0f); return or_sqr(f0); } /* :1095-1098 */
static inline v3 or_fresnel_eval(const or_surface *s, float cos_theta_i) {
    if (s->fresnel == OR_FR_DIELECTRIC) { float f = or_fr_dielectric(cos_theta_i, s->eta); return V3(1.0f * f, 1.0f * f, 1.0f * f); }
    if (s->fresnel == OR_FR_COMPLEX) { /* FresnelComplex::evaluate passes |cos| (:1181-1184) */
        float c = fabsf(cos_theta_i);
        return V3(or_fr_complex(c, cx(s->fn.x, s->fk.x)), or_fr_complex(c, cx(s->fn.y, s->fk.y)), or_fr_complex(c, cx(s->fn.z, s->fk.z)));
    }
    return V3(1, 1, 1);
}

/* ---------- precomputed albedo table (svm/surface/mod.rs:1145-1154, 1211-1261) ---------- */
static inline float or_table_read_1d(const float *buf, float x, uint32_t offset, uint32_t size) {
    x = or_clamp(x, 0.0f, 1.0f) * ((float)size - 1.0f);
    uint32_t index = (uint32_t)floorf(x);
    uint32_t nindex = index + 1 < size - 1 ? index + 1 : size - 1;
    float t = x - (float)index;
    return (1.0f - t) * buf[offset + index] + t * buf[offset + nindex];
}
static inline float or_table_read_2d(const float *buf, float x, float y, uint32_t offset, uint32_t xs, uint32_t ys) {
    y = or_clamp(y, 0.0f, 1.0f) * ((float)ys - 1.0f);
    uint32_t index = (uint32_t)floorf(y);
    uint32_t nindex = index + 1 < ys - 1 ? index + 1 : ys - 1;
    float t = y - (float)index;
    float d0 = or_table_read_1d(buf, x, offset + xs * index, xs);
    float d1 = or_table_read_1d(buf, x, offset + xs * nindex, xs);
    return (1.0f - t) * d0 + t * d1;
}
static inline float or_table_read_3d(const float *buf, float x, float y, float z) {
    const uint32_t xs = 16, ys = 16, zs = 16;
    z = or_clamp(z, 0.0f, 1.0f) * ((float)zs - 1.0f);
    uint32_t index = (uint32_t)floorf(z);
    uint32_t nindex = index + 1 < zs - 1 ? index + 1 : zs - 1;
    float t = z - (float)index;
    float d0 = or_table_read_2d(buf, x, y, xs * ys * index, xs, ys);
    float d1 = or_table_read_2d(buf, x, y, xs * ys * nindex, xs, ys);
    return (1.0f - t) * d0 + t * d1;
}
static inline float or_ggx_dielectric_albedo(const float *table, float roughness, float cos_theta_i, float eta) {
    float z = sqrtf(fabsf((eta - 1.0f) / (eta + 1.0f)));
    cos_theta_i = fabsf(or_clamp(cos_theta_i, -0.999f, 0.999f));
    return or_table_read_3d(table, roughness, fabsf(cos_theta_i), z);
}

/* ---------- Surface trait: evaluate / sample_wi / emission ---------- */
static void or_surf_evaluate(const or_surface *s, v3 wo, v3 wi, v3 *f, float *pdf);
static int or_surf_sample_wi(const or_surface *s, v3 wo, float u_select, v2 u_sample, v3 *wi);
static v3 or_surf_emission(const or_surface *s, v3 wo);

static inline v3 or_etop(const or_surface *s, v3 w) { /* principled.rs:158-162, 188-193 */
    float albedo = or_ggx_dielectric_albedo(s->table, s->etop_roughness, fr_abs_cos_theta(w), s->etop_eta);
    return v3scale(v3scale(s->etop_tint, albedo), s->etop_weight);
}
static inline float or_frac(const or_surface *s, v3 wo) {
    return s->frac_kind == OR_FRAC_CONST ? s->frac_const : or_fr_dielectric(fr_cos_theta(wo), s->frac_eta);
}
/* SurfaceClosure::check_wo_wi_valid, svm/surface/mod.rs:705-719 */
static inline int or_check_wo_wi_valid(const or_surface *s, v3 wo, v3 wi) {
#define OR_SGN(x) (((x) > 0.0f) ? 1.0f : -1.0f)
    v3 ns = s->frame.n, ng = s->ng;
    float flipped = OR_SGN(v3dot(ng, ns));
    int a = OR_SGN(flipped * v3dot(wo, ns)) * OR_SGN(v3dot(wo, ng)) > 0.0f;
    int b = OR_SGN(flipped * v3dot(wi, ns)) * OR_SGN(v3dot(wi, ng)) > 0.0f;
#undef OR_SGN
    return a & b;
}

static void or_surf_evaluate(const or_surface *s, v3 wo, v3 wi, v3 *f, float *pdf) {
    *f = V3(0, 0, 0);
    *pdf = 0.0f;
    switch (s->kind) {
    case OR_S_NULL: return;
    case OR_S_DIFFUSE: { /* diffuse.rs:22-38 */
        if (fr_same_hemisphere(wo, wi)) {
            *pdf = fr_abs_cos_theta(wi) * OR_INV_PI;
            *f = v3scale(s->color, fr_abs_cos_theta(wi));
        }
        return;
    }
    case OR_S_MF_REFL: { /* svm/surface/mod.rs:831-858 */
        v3 wh = v3add(wo, wi);
        float cos_o = fr_cos_theta(wo), cos_i = fr_cos_theta(wi);
        if ((v3dot(wh, wo) * v3dot(wi, wh)) < 0.0f || (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) ||
            cos_i == 0.0f || cos_o == 0.0f || !fr_same_hemisphere(wo, wi))
            return;
        wh = v3normalize(wh);
        v3 fr = or_fresnel_eval(s, v3dot(wi, or_face_forward(wh, V3(0, 0, 1))));
        float d = tr_d(wh, s->alpha);
        float g = tr_g(wo, wi, s->alpha);
        float k = fabsf(0.25f * d * g / (cos_i * cos_o));
        *f = v3scale(v3scale(v3mul(s->color, fr), k), fabsf(cos_i));
        *pdf = tr_pdf(wo, wh, s->alpha) / (4.0f * fabsf(v3dot(wo, wh)));
        return;
    }
    case OR_S_MF_TRANS: { /* svm/surface/mod.rs:914-967 */
        float cos_o = fr_cos_theta(wo), cos_i = fr_cos_theta(wi);
        float eta = cos_o > 0.0f ? s->eta : 1.0f / s->eta;
        v3 wh = v3normalize(v3add(wo, v3scale(wi, eta)));
        wh = or_face_forward(wh, V3(0, 0, 1));
        int backfacing = (v3dot(wh, wi) * cos_i) < 0.0f || (v3dot(wh, wo) * cos_o) < 0.0f;
        if ((v3dot(wh, wo) * v3dot(wi, wh)) > 0.0f || cos_i == 0.0f || cos_o == 0.0f || backfacing ||
            fr_same_hemisphere(wo, wi))
            return;
        v3 fr = or_fresnel_eval(s, v3dot(wo, wh));
        float denom = or_sqr(v3dot(wi, wh) + v3dot(wo, wh) / eta) * cos_i * cos_o;
        if (denom == 0.0f) {
            *f = V3(0, 0, 0);
        } else {
            float k = fabsf(tr_d(wh, s->alpha) * tr_g(wo, wi, s->alpha) / or_sqr(eta) * fabsf(v3dot(wi, wh)) *
                            fabsf(v3dot(wo, wh)) / denom);
            v3 one_minus_f = V3(1.0f - fr.x, 1.0f - fr.y, 1.0f - fr.z);
            *f = v3scale(v3scale(v3mul(one_minus_f, s->color), k), fabsf(cos_i));
        }
        float denom2 = or_sqr(v3dot(wi, wh) + v3dot(wo, wh) / eta);
        float dwh_dwi = fabsf(v3dot(wi, wh)) / denom2;
        *pdf = denom2 == 0.0f ? 0.0f : tr_pdf(wo, wh, s->alpha) * dwh_dwi;
        return;
    }
    case OR_S_MIXTURE: { /* svm/surface/mod.rs:591-625 */
        float frac = or_frac(s, wo);
        v3 fa = V3(0, 0, 0), fb = V3(0, 0, 0);
        float pa = 0.0f, pb = 0.0f;
        if (s->mode == OR_BLEND_ADDICTIVE) {
            or_surf_evaluate(s->a, wo, wi, &fa, &pa);
            or_surf_evaluate(s->b, wo, wi, &fb, &pb);
            *f = v3add(fa, fb);
            *pdf = or_lerp(pa, pb, frac);
        } else {
            if (frac < 1.0f - 1e-4f) or_surf_evaluate(s->a, wo, wi, &fa, &pa);
            if (frac > 1e-4f) or_surf_evaluate(s->b, wo, wi, &fb, &pb);
            *f = v3lerp(fa, fb, frac);
            *pdf = or_lerp(pa, pb, frac);
        }
        return;
    }
    case OR_S_COATED: { /* svm/surface/mod.rs:486-503 */
        v3 f_top, f_bottom;
        float pdf_top, pdf_bottom;
        or_surf_evaluate(s->a, wo, wi, &f_top, &pdf_top);
        or_surf_evaluate(s->b, wo, wi, &f_bottom, &pdf_bottom);
        v3 eo = or_etop(s, wo), ei = or_etop(s, wi);
        float pdf_select_top = ((eo.x + eo.y) + eo.z) / 3.0f;
        float pdf_select_bottom = 1.0f - pdf_select_top;
        *pdf = pdf_top * pdf_select_top + pdf_bottom * pdf_select_bottom;
        v3 m = V3(or_min(1.0f - eo.x, 1.0f - ei.x), or_min(1.0f - eo.y, 1.0f - ei.y), or_min(1.0f - eo.z, 1.0f - ei.z));
        *f = v3add(f_top, v3mul(f_bottom, m));
        return;
    }
    case OR_S_SCALED: { /* svm/surface/mod.rs:420-430 */
        or_surf_evaluate(s->a, wo, wi, f, pdf);
        *f = v3mul(*f, s->color);
        return;
    }
    case OR_S_EMISSIVE: /* svm/surface/mod.rs:343-355 */
        if (s->a) or_surf_evaluate(s->a, wo, wi, f, pdf);
        return;
    case OR_S_PRINCIPLED: /* principled.rs:238-246 */
        or_surf_evaluate(s->a, wo, wi, f, pdf);
        return;
    case OR_S_CLOSURE: { /* svm/surface/mod.rs:730-748 */
        if (!or_check_wo_wi_valid(s, wo, wi)) return;
        or_surf_evaluate(s->a, or_to_local(&s->frame, wo), or_to_local(&s->frame, wi), f, pdf);
        return;
    }
    }
}

static int or_surf_sample_wi(const or_surface *s, v3 wo, float u_select, v2 u_sample, v3 *wi) {
    *wi = V3(0, 0, 0);
    switch (s->kind) {
    case OR_S_NULL: return 0;
    case OR_S_DIFFUSE: { /* diffuse.rs:40-52 */
        v3 w = or_cos_sample_hemisphere(u_sample);
        *wi = fr_same_hemisphere(wo, w) ? w : v3neg(w);
        return 1;
    }
    case OR_S_MF_REFL: { /* svm/surface/mod.rs:860-873 */
        v3 wh = tr_sample_wh(wo, u_sample, s->alpha);
        *wi = or_reflect(wo, wh);
        return fr_same_hemisphere(wo, *wi);
    }
    case OR_S_MF_TRANS: { /* svm/surface/mod.rs:969-982 */
        v3 wh = tr_sample_wh(wo, u_sample, s->alpha);
        int refracted = or_refract(wo, wh, s->eta, wi);
        return refracted && !fr_same_hemisphere(wo, *wi);
    }
    case OR_S_MIXTURE: { /* svm/surface/mod.rs:627-644: picks b iff u < frac */
        float frac = or_frac(s, wo), remapped;
        int pick_b = or_weighted_choice2_and_remap(frac, u_select, &remapped);
        return or_surf_sample_wi(pick_b ? s->b : s->a, wo, remapped, u_sample, wi);
    }
    case OR_S_COATED: { /* svm/surface/mod.rs:504-522: picks top iff u < avg(E(wo)) */
        v3 eo = or_etop(s, wo);
        float pdf_select_top = ((eo.x + eo.y) + eo.z) / 3.0f, remapped;
        int pick_top = or_weighted_choice2_and_remap(pdf_select_top, u_select, &remapped);
        return or_surf_sample_wi(pick_top ? s->a : s->b, wo, remapped, u_sample, wi);
    }
    case OR_S_SCALED: return or_surf_sample_wi(s->a, wo, u_select, u_sample, wi);
    case OR_S_EMISSIVE: return s->a ? or_surf_sample_wi(s->a, wo, u_select, u_sample, wi) : 0;
    case OR_S_PRINCIPLED: return or_surf_sample_wi(s->a, wo, u_select, u_sample, wi);
    case OR_S_CLOSURE: { /* svm/surface/mod.rs:750-764 */
        v3 wl;
        int valid = or_surf_sample_wi(s->a, or_to_local(&s->frame, wo), u_select, u_sample, &wl);
        *wi = or_to_world(&s->frame, wl);
        return valid & or_check_wo_wi_valid(s, wo, *wi);
    }
    }
    return 0;
}

static v3 or_surf_emission(const or_surface *s, v3 wo) {
    switch (s->kind) {
    case OR_S_MIXTURE: { /* svm/surface/mod.rs:678-694 */
        float frac = or_frac(s, wo);
        v3 ea = or_surf_emission(s->a, wo), eb = or_surf_emission(s->b, wo);
        if (s->mode == OR_BLEND_ADDICTIVE) return v3add(ea, eb);
        return v3add(v3scale(ea, 1.0f - frac), v3scale(eb, frac));
    }
    case OR_S_COATED: { /* svm/surface/mod.rs:553-566 */
        v3 eo = or_etop(s, wo);
        v3 et = or_surf_emission(s->a, wo), eb = or_surf_emission(s->b, wo);
        return v3add(v3mul(et, eo), v3mul(eb, V3(1.0f - eo.x, 1.0f - eo.y, 1.0f - eo.z)));
    }
    case OR_S_SCALED: return v3mul(or_surf_emission(s->a, wo), s->color);
    case OR_S_EMISSIVE: return s->a ? v3add(s->emission, or_surf_emission(s->a, wo)) : s->emission;
    case OR_S_PRINCIPLED: return s->emission; /* principled.rs:267-274 */
    case OR_S_CLOSURE: return or_surf_emission(s->a, or_to_local(&s->frame, wo));
    default: return V3(0, 0, 0);
    }
}

/* ---- AOV queries of the closure tree (trait Surface: ns / albedo / roughness) ---------------------------------- */
static v3 or_surf_ns(const or_surface *s) {
    switch (s->kind) {
    case OR_S_MIXTURE: return v3normalize(v3add(or_surf_ns(s->a), or_surf_ns(s->b))); /* mod.rs:588-590 */
    case OR_S_COATED: return or_surf_ns(s->b);                                          /* bottom, mod.rs:482-484 */
    case OR_S_SCALED: return or_surf_ns(s->a);
    case OR_S_EMISSIVE: return s->a ? or_surf_ns(s->a) : V3(0, 0, 1);                  /* mod.rs:338-342 */
    case OR_S_CLOSURE: return or_to_world(&s->frame, or_surf_ns(s->a));                 /* mod.rs:724-727 */
    default: return V3(0, 0, 1); /* null, diffuse, microfacet lobes, the Principled wrapper (principled.rs:224-226) */
    }
}
static v3 or_surf_albedo(const or_surface *s, v3 wo) {
    switch (s->kind) {
    case OR_S_DIFFUSE: return v3scale(s->color, OR_PI);           /* reflectance * PI, diffuse.rs:56-63 */
    case OR_S_MF_REFL: case OR_S_MF_TRANS: return s->color;      /* mod.rs:875-882, 981-988 */
    case OR_S_MIXTURE: { /* mod.rs:659-675 */
        float frac = or_frac(s, wo);
        v3 aa = or_surf_albedo(s->a, wo), ab = or_surf_albedo(s->b, wo);
        if (s->mode == OR_BLEND_ADDICTIVE) return v3add(aa, ab);
        return v3add(v3scale(aa, 1.0f - frac), v3scale(ab, frac));
    }
    case OR_S_COATED: { /* mod.rs:524-535 */
        v3 eo = or_etop(s, wo);
        v3 at = or_surf_albedo(s->a, wo), ab = or_surf_albedo(s->b, wo);
        return v3add(v3mul(at, eo), v3mul(ab, V3(1.0f - eo.x, 1.0f - eo.y, 1.0f - eo.z)));
    }
    case OR_S_SCALED: return v3mul(or_surf_albedo(s->a, wo), s->color);                 /* mod.rs:446-454 */
    case OR_S_EMISSIVE: return s->a ? or_surf_albedo(s->a, wo) : V3(0, 0, 0);           /* mod.rs:372-383 */
    case OR_S_PRINCIPLED: return s->color;                                              /* wrapper albedo, principled.rs:227-234 */
    case OR_S_CLOSURE: return or_surf_albedo(s->a, or_to_local(&s->frame, wo));         /* mod.rs:766-773 */
    default: return V3(0, 0, 0);
    }
}
static float or_surf_roughness(const or_surface *s, v3 wo, float u_select) {
    switch (s->kind) {
    case OR_S_MF_REFL: case OR_S_MF_TRANS: return s->roughness;
    case OR_S_MIXTURE: { /* mod.rs:642-657: which = 0 (-> bsdf_a) iff NOT (u < frac) */
        float frac = or_frac(s, wo), remapped;
        int pick_b = or_weighted_choice2_and_remap(frac, u_select, &remapped);
        return or_surf_roughness(pick_b ? s->b : s->a, wo, remapped);
    }
    case OR_S_COATED: { /* mod.rs:537-556 */
        v3 eo = or_etop(s, wo);
        float pdf_select_top = ((eo.x + eo.y) + eo.z) / 3.0f, remapped;
        int pick_top = or_weighted_choice2_and_remap(pdf_select_top, u_select, &remapped);
        return or_surf_roughness(pick_top ? s->a : s->b, wo, remapped);
    }
    case OR_S_SCALED: case OR_S_PRINCIPLED: return or_surf_roughness(s->a, wo, u_select);
    case OR_S_EMISSIVE: return s->a ? or_surf_roughness(s->a, wo, u_select) : 1.0f;
    case OR_S_CLOSURE: return or_surf_roughness(s->a, or_to_local(&s->frame, wo), u_select);
    default: return 1.0f; /* null, diffuse (diffuse.rs:64-72) */
    }
}

/* SurfaceClosure::sample, svm/surface/mod.rs:795-815 */
typedef struct { v3 wi; float pdf; v3 color; int valid; } or_bsdf_sample;
static inline or_bsdf_sample or_closure_sample(const or_surface *closure, v3 wo, float u_select, v2 u_sample) {
    or_bsdf_sample r;
    r.valid = or_surf_sample_wi(closure, wo, u_select, u_sample, &r.wi);
    if (!r.valid) { r.wi = V3(0, 0, 0); r.pdf = 0.0f; r.color = V3(0, 0, 0); return r; }
    or_surf_evaluate(closure, wo, r.wi, &r.color, &r.pdf);
    r.valid = r.valid & (r.pdf > 0.0f);
    return r;
}
#endif
