/* or_geom.h -- sampling warps, alias tables, frames, ray-origin offsetting for the CPU oracle.
 * TEST INFRASTRUCTURE ONLY. Reference: crates/akari_render/src/{sampling.rs, util/distribution.rs,
 * geometry.rs}; LuisaCompute primitives as assumed in SURVEY.md Appendix C (parity unpinned there).
 */
#ifndef OR_GEOM_H
#define OR_GEOM_H
#include "or_math.h"
#include <stdlib.h>

/* sampling.rs:5-9 */
static inline v2 or_uniform_sample_disk(v2 u) {
    float r = sqrtf(u.x);
    float phi = u.y * 2.0f * OR_PI;
    float s, c;
    or_sincosf(phi, &s, &c);
    return V2(r * c, r * s);
}
/* sampling.rs:17-21 */
static inline v3 or_cos_sample_hemisphere(v2 u) {
    v2 d = or_uniform_sample_disk(u);
    float z = sqrtf(or_max(1.0f - d.x * d.x - d.y * d.y, 0.0f));
    return V3(d.x, d.y, z);
}
/* sampling.rs:32-44 */
static inline v2 or_uniform_sample_triangle(v2 u) {
    if (u.x < u.y) {
        float b0 = u.x / 2.0f;
        float b1 = u.y - b0;
        return V2(b0, b1);
    } else {
        float b1 = u.y / 2.0f;
        float b0 = u.x - b1;
        return V2(b0, b1);
    }
}
/* sampling.rs:54-59 */
static inline uint32_t or_uniform_discrete_choice_and_remap(uint32_t n, float u, float *remapped) {
    float fi = floorf(u * (float)n);
    int32_t i = (int32_t)fi;
    int32_t hi = (int32_t)n - 1;
    if (i < 0) i = 0;
    if (i > hi) i = hi;
    *remapped = u * (float)n - (float)i;
    return (uint32_t)i;
}
/* sampling.rs:61-71: returns 1 when `a` (first) is chosen */
static inline int or_weighted_choice2_and_remap(float weight_a, float u, float *remapped) {
    int first = u < weight_a;
    *remapped = first ? u / weight_a : (u - weight_a) / (1.0f - weight_a);
    return first;
}

/* util/distribution.rs:12-88 */
typedef struct { uint32_t j; float t; } or_alias_entry;
typedef struct { uint32_t n; or_alias_entry *table; float *pdf; } or_alias_table;

static inline void or_alias_build(or_alias_table *at, const float *weights, uint32_t n) {
    at->n = n;
    at->table = (or_alias_entry *)calloc(n, sizeof(or_alias_entry));
    at->pdf = (float *)malloc(n * sizeof(float));
    float *prob = (float *)malloc(n * sizeof(float));
    float sum = 0.0f;
    for (uint32_t i = 0; i < n; i++) sum += weights[i]; /* iter().sum::<f32>() = sequential */
    for (uint32_t i = 0; i < n; i++) prob[i] = weights[i] / sum * (float)n;
    /* two FIFO work lists (VecDeque push_back / pop_front) */
    uint32_t *small = (uint32_t *)malloc(2 * n * sizeof(uint32_t) + 8), *large = (uint32_t *)malloc(2 * n * sizeof(uint32_t) + 8);
    uint32_t sh = 0, st = 0, lh = 0, lt = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (prob[i] >= 1.0f) large[lt++] = i; else small[st++] = i;
    }
    while (sh < st && lh < lt) {
        uint32_t l = small[sh++], g = large[lh++];
        at->table[l].t = prob[l];
        at->table[l].j = g;
        prob[g] = (prob[g] + prob[l]) - 1.0f;
        if (prob[g] < 1.0f) small[st++] = g; else large[lt++] = g;
    }
    while (lh < lt) { uint32_t g = large[lh++]; at->table[g].t = 1.0f; at->table[g].j = g; }
    while (sh < st) { uint32_t l = small[sh++]; at->table[l].t = 1.0f; at->table[l].j = l; }
    for (uint32_t i = 0; i < n; i++) at->pdf[i] = weights[i] / sum;
    free(prob); free(small); free(large);
}
static inline void or_alias_free(or_alias_table *at) { free(at->table); free(at->pdf); at->table = 0; at->pdf = 0; }
/* util/distribution.rs:81-87 */
static inline uint32_t or_alias_sample_and_remap(const or_alias_table *at, float u, float *pdf, float *remapped) {
    float u1;
    uint32_t idx = or_uniform_discrete_choice_and_remap(at->n, u, &u1);
    or_alias_entry e = at->table[idx];
    float u2;
    int first = or_weighted_choice2_and_remap(e.t, u1, &u2);
    idx = first ? idx : e.j;
    *pdf = at->pdf[idx];
    *remapped = u2;
    return idx;
}

/* geometry.rs:72-78, 156-200 */
typedef struct { v3 n, t, s; } or_frame;
static inline or_frame or_frame_from_n(v3 n) {
    v3 t;
    if (fabsf(n.x) > fabsf(n.y)) t = v3divs(V3(-n.z, 0.0f, n.x), sqrtf(n.x * n.x + n.z * n.z));
    else t = v3divs(V3(0.0f, n.z, -n.y), sqrtf(n.y * n.y + n.z * n.z));
    or_frame f = {n, t, v3cross(n, t)};
    return f;
}
static inline or_frame or_frame_from_n_t(v3 n, v3 tt_in) {
    v3 tt = v3sub(tt_in, v3scale(n, v3dot(n, tt_in)));
    int good = 1;
    or_frame f;
    if (v3len(tt) < 1e-4f) good = 0; else tt = v3normalize(tt);
    if (good) {
        v3 ss = v3cross(n, tt);
        if (v3len(ss) < 1e-4f) good = 0;
        else { ss = v3normalize(ss); f.n = n; f.t = tt; f.s = ss; }
    }
    if (!good) f = or_frame_from_n(n);
    return f;
}
static inline v3 or_to_world(const or_frame *f, v3 v) {
    return v3add(v3add(v3scale(f->t, v.x), v3scale(f->s, v.y)), v3scale(f->n, v.z));
}
static inline v3 or_to_local(const or_frame *f, v3 v) { return V3(v3dot(f->t, v), v3dot(f->s, v), v3dot(f->n, v)); }
/* geometry.rs:264-271 */
static inline v3 or_face_forward(v3 v, v3 n) { return v3dot(v, n) < 0.0f ? v3neg(v) : v; }
/* geometry.rs:275-279 */
static inline v3 or_reflect(v3 w, v3 n) {
    float k = 2.0f * v3dot(w, n);
    return v3add(v3neg(w), v3scale(n, k));
}
/* geometry.rs:283-302 */
static inline int or_refract(v3 w, v3 n, float eta, v3 *wt) {
    float cos_theta_i = v3dot(w, n);
    if (!(cos_theta_i >= 0.0f)) { eta = 1.0f / eta; n = v3neg(n); }
    cos_theta_i = fabsf(cos_theta_i);
    float sin2_theta_i = or_max(1.0f - or_sqr(cos_theta_i), 0.0f);
    float sin2_theta_t = sin2_theta_i / or_sqr(eta);
    if (sin2_theta_t >= 1.0f) { *wt = V3(0, 0, 0); return 0; }
    float cos_theta_t = sqrtf(1.0f - sin2_theta_t);
    *wt = v3add(v3divs(v3neg(w), eta), v3scale(n, cos_theta_i / eta - cos_theta_t));
    return 1;
}
/* luisa::rtx::offset_ray_origin -- Waechter & Binder, Ray Tracing Gems ch.6 */
static inline float or_offset_comp(float p, float n) {
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int32_t of_i = (int32_t)(int_scale * n);
    int32_t pi = (int32_t)f2u(p) + (p < 0.0f ? -of_i : of_i);
    float p_i = u2f((uint32_t)pi);
    return fabsf(p) < origin ? p + float_scale * n : p_i;
}
static inline v3 or_offset_ray_origin(v3 p, v3 n) {
    return V3(or_offset_comp(p.x, n.x), or_offset_comp(p.y, n.y), or_offset_comp(p.z, n.z));
}
#endif
