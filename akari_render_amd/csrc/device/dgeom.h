// dgeom.h -- sampling warps, alias-table lookup, frames and ray-origin offsetting on the device.
// Reference: crates/akari_render/src/sampling.rs, util/distribution.rs:81-87, geometry.rs:156-302.
#pragma once
#include "dmath.h"

namespace akr {

AKR_HD vec2 uniform_sample_disk(vec2 u) {  // sampling.rs:5-9 (polar)
    float r = __builtin_sqrtf(u.x);
    float phi = u.y * 2.0f * kPi;
    float s, c;
    sincos_f(phi, s, c);
    return mk2(r * c, r * s);
}
AKR_HD vec3 cos_sample_hemisphere(vec2 u) {  // sampling.rs:17-21
    vec2 d = uniform_sample_disk(u);
    float z = __builtin_sqrtf(max_f(1.0f - d.x * d.x - d.y * d.y, 0.0f));
    return mk3(d.x, d.y, z);
}
AKR_HD vec2 uniform_sample_triangle(vec2 u) {  // sampling.rs:32-44
    if (u.x < u.y) {
        float b0 = u.x / 2.0f;
        float b1 = u.y - b0;
        return mk2(b0, b1);
    }
    float b1 = u.y / 2.0f;
    float b0 = u.x - b1;
    return mk2(b0, b1);
}
AKR_HD uint32_t uniform_discrete_choice_and_remap(uint32_t n, float u, float& remapped) {  // sampling.rs:54-59
    float fi = __builtin_floorf(u * (float)n);
    int32_t i = (int32_t)fi;
    int32_t hi = (int32_t)n - 1;
    i = i < 0 ? 0 : i;
    i = i > hi ? hi : i;
    remapped = u * (float)n - (float)i;
    return (uint32_t)i;
}
// sampling.rs:61-71; true = the first alternative (weight_a) was taken
AKR_HD bool weighted_choice2_and_remap(float weight_a, float u, float& remapped) {
    bool first = u < weight_a;
    remapped = first ? u / weight_a : (u - weight_a) / (1.0f - weight_a);
    return first;
}

struct AliasEntry {  // util/distribution.rs:12-15
    uint32_t j;
    float t;
};
// util/distribution.rs:81-87 over device arrays (entries[n], pdf[n])
AKR_D uint32_t alias_sample_and_remap(const AliasEntry* __restrict__ entries, const float* __restrict__ pdfs, uint32_t n,
                                      float u, float& pdf, float& remapped) {
    float u1;
    uint32_t idx = uniform_discrete_choice_and_remap(n, u, u1);
    AliasEntry e = entries[idx];
    float u2;
    bool first = weighted_choice2_and_remap(e.t, u1, u2);
    idx = first ? idx : e.j;
    pdf = pdfs[idx];
    remapped = u2;
    return idx;
}

// The same table with both candidate probabilities next to the entry: one 16-byte load per level instead of the entry
// followed by a dependent pdfs[idx] load (the shading phase is a chain of dependent gathers; every link is ~a cache latency).
struct AliasPacked {
    uint32_t j;
    float t, pdf_i, pdf_j;
};
AKR_D uint32_t alias_sample_and_remap(const AliasPacked* __restrict__ entries, uint32_t n, float u, float& pdf, float& remapped) {
    float u1;
    uint32_t idx = uniform_discrete_choice_and_remap(n, u, u1);
    AliasPacked e = entries[idx];
    float u2;
    bool first = weighted_choice2_and_remap(e.t, u1, u2);
    pdf = first ? e.pdf_i : e.pdf_j;
    remapped = u2;
    return first ? idx : e.j;
}

struct Frame {  // geometry.rs:72-78
    vec3 n, t, s;
};
AKR_HD Frame frame_from_n(vec3 n) {  // geometry.rs:159-167
    vec3 t;
    if (abs_f(n.x) > abs_f(n.y))
        t = div_s(mk3(-n.z, 0.0f, n.x), __builtin_sqrtf(n.x * n.x + n.z * n.z));
    else
        t = div_s(mk3(0.0f, n.z, -n.y), __builtin_sqrtf(n.y * n.y + n.z * n.z));
    return Frame{n, t, cross(n, t)};
}
AKR_HD Frame frame_from_n_t(vec3 n, vec3 tt_in) {  // geometry.rs:168-191
    vec3 tt = tt_in - n * dot(n, tt_in);
    bool good = true;
    Frame f{n, n, n};
    if (length(tt) < 1e-4f)
        good = false;
    else
        tt = normalize(tt);
    if (good) {
        vec3 ss = cross(n, tt);
        if (length(ss) < 1e-4f) {
            good = false;
        } else {
            ss = normalize(ss);
            f = Frame{n, tt, ss};
        }
    }
    if (!good) f = frame_from_n(n);
    return f;
}
AKR_HD vec3 to_world(const Frame& f, vec3 v) { return (f.t * v.x + f.s * v.y) + f.n * v.z; }
AKR_HD vec3 to_local(const Frame& f, vec3 v) { return mk3(dot(f.t, v), dot(f.s, v), dot(f.n, v)); }
AKR_HD vec3 face_forward(vec3 v, vec3 n) { return dot(v, n) < 0.0f ? -v : v; }  // geometry.rs:264-271
AKR_HD vec3 reflect(vec3 w, vec3 n) {                                           // geometry.rs:275-279
    float k = 2.0f * dot(w, n);
    return (-w) + n * k;
}
AKR_HD bool refract(vec3 w, vec3 n, float eta, vec3& wt) {  // geometry.rs:283-302
    float cos_theta_i = dot(w, n);
    if (!(cos_theta_i >= 0.0f)) {
        eta = 1.0f / eta;
        n = -n;
    }
    cos_theta_i = abs_f(cos_theta_i);
    float sin2_theta_i = max_f(1.0f - sqr(cos_theta_i), 0.0f);
    float sin2_theta_t = sin2_theta_i / sqr(eta);
    if (sin2_theta_t >= 1.0f) {
        wt = mk3(0, 0, 0);
        return false;
    }
    float cos_theta_t = __builtin_sqrtf(1.0f - sin2_theta_t);
    wt = div_s(-w, eta) + n * (cos_theta_i / eta - cos_theta_t);
    return true;
}
// luisa::rtx::offset_ray_origin == Waechter & Binder, Ray Tracing Gems ch. 6
AKR_HD float offset_comp(float p, float n) {
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int32_t of_i = (int32_t)(int_scale * n);
    int32_t pi = (int32_t)f2u(p) + (p < 0.0f ? -of_i : of_i);
    float p_i = u2f((uint32_t)pi);
    return abs_f(p) < origin ? p + float_scale * n : p_i;
}
AKR_HD vec3 offset_ray_origin(vec3 p, vec3 n) { return mk3(offset_comp(p.x, n.x), offset_comp(p.y, n.y), offset_comp(p.z, n.z)); }

}  // namespace akr
