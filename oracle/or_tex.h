/* or_tex.h -- texture sampling and shader-graph node evaluation of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Follows crates/akari_render/src/svm/eval.rs:97-269 (node semantics), 301-349 (auto-convert rules),
 * svm/texture/mod.rs:44-51 + color.rs:555-558 (sRGB decode), load.rs:680-702 (sampler). The reference samples
 * through LuisaCompute's `tex2d.sample` (third party, source absent): the filter below -- texel centres at
 * +0.5, weights = fractional parts, lerp a + (b - a) t, unorm8 = byte / 255 -- is this project's restatement
 * ("parity unpinned" for the sampling itself, like the BVH hit selection).
 */
#ifndef OR_TEX_H
#define OR_TEX_H
#include "or_api.h"
#include "or_math.h"

typedef struct { float v[4]; } or_val;

/* Repeat / mirror reduce the coordinate to one period first (u - floor(u); triangle wave of period 2): the filter's texel
 * indices then lie in [-1, n] and wrap without integer division. Edge clamps, Zero reports "outside". */
static inline float or_tex_wrap_coord(float u, uint32_t mode) {
    if (mode == OR_TEX_REPEAT) return u - floorf(u);
    if (mode == OR_TEX_MIRROR) { float t = u - 2.0f * floorf(u * 0.5f); return t > 1.0f ? 2.0f - t : t; }
    return u;
}
static inline int or_tex_wrap(int *i, int n, uint32_t mode) {
    int k = *i;
    if (mode == OR_TEX_CLIP) return k >= 0 && k < n; /* Zero */
    if (mode == OR_TEX_REPEAT) k = k < 0 ? k + n : (k >= n ? k - n : k);
    *i = k < 0 ? 0 : (k > n - 1 ? n - 1 : k); /* mirror, edge; repeat: what is still outside (NaN coordinates) */
    return 1;
}
static inline or_val or_tex_fetch(const or_image_desc *im, int i, int j) {
    or_val r = {{0, 0, 0, 0}};
    if (!or_tex_wrap(&i, (int)im->width, im->address) || !or_tex_wrap(&j, (int)im->height, im->address)) return r;
    size_t t = (size_t)j * im->width + (size_t)i;
    if (im->format == OR_IMAGE_RGBA8) {
        const uint8_t *p = (const uint8_t *)im->texels + 4 * t;
        for (int c = 0; c < 4; c++) r.v[c] = (float)p[c] / 255.0f;
    } else {
        const float *p = (const float *)im->texels + 4 * t;
        for (int c = 0; c < 4; c++) r.v[c] = p[c];
    }
    return r;
}
static inline int or_floor_int(float x, float *fl) {
    if (!(x > -1.0e9f)) x = -1.0e9f;
    if (x > 1.0e9f) x = 1.0e9f;
    *fl = floorf(x);
    return (int)*fl;
}
static inline or_val or_tex_sample(const or_image_desc *im, float u, float v) {
    float x = or_tex_wrap_coord(u, im->address) * (float)im->width, y = or_tex_wrap_coord(v, im->address) * (float)im->height, fx, fy;
    if (im->filter == OR_TEX_NEAREST) {
        int i = or_floor_int(x, &fx), j = or_floor_int(y, &fy);
        return or_tex_fetch(im, i, j);
    }
    x = x - 0.5f; y = y - 0.5f;
    int i = or_floor_int(x, &fx), j = or_floor_int(y, &fy);
    float tx = x - fx, ty = y - fy;
    if (!(tx >= 0.0f)) tx = 0.0f;
    if (!(ty >= 0.0f)) ty = 0.0f;
    if (tx > 1.0f) tx = 1.0f;
    if (ty > 1.0f) ty = 1.0f;
    or_val a = or_tex_fetch(im, i, j), b = or_tex_fetch(im, i + 1, j), c = or_tex_fetch(im, i, j + 1), d = or_tex_fetch(im, i + 1, j + 1), r;
    for (int k = 0; k < 4; k++) {
        float r0 = or_lerp(a.v[k], b.v[k], tx), r1 = or_lerp(c.v[k], d.v[k], tx);
        r.v[k] = or_lerp(r0, r1, ty);
    }
    return r;
}
static inline float or_srgb_to_linear(float s) { return s <= 0.04045f ? s / 12.92f : or_powf((s + 0.055f) / 1.055f, 2.4f); }

/* Color::to_rgb / rgb_to_target_colorspace (color.rs:262-275, svm/texture/mod.rs:9-30) with the CAT matrices of
 * color.rs:614-628; M * v = (c0 x + c1 y) + c2 z. Identity when the two spaces agree. */
static inline void or_cs_convert(float *v, int from_aces, int to_aces) {
    if ((from_aces != 0) == (to_aces != 0)) return;
    const float x = v[0], y = v[1], z = v[2];
    if (to_aces) { /* srgb_to_aces_with_cat_mat */
        v[0] = (0.612494199f * x + 0.338737252f * y) + 0.048855526f * z;
        v[1] = (0.070594252f * x + 0.917671484f * y) + 0.011704306f * z;
        v[2] = (0.020727335f * x + 0.106882232f * y) + 0.872338062f * z;
    } else { /* aces_to_srgb_with_cat_mat */
        v[0] = (1.707062673f * x + -0.619959540f * y) + -0.087259850f * z;
        v[1] = (-0.130976829f * x + 1.139032275f * y) + -0.007956297f * z;
        v[2] = (-0.024510601f * x + -0.124810932f * y) + 1.149395971f * z;
    }
}
/* eval_shader (eval.rs:363-380): all nodes in order. Values are 4 floats, narrower types zero-extended. `color` = the render's
 * ColorPipeline bits (OR_COLOR_*). */
static inline void or_eval_graph(const or_material_graph *g, const or_image_desc *images, uint32_t color, float u, float v, or_val *val) {
    for (uint32_t i = 0; i < g->n_nodes; i++) {
        const or_shader_node *n = &g->nodes[i];
        or_val r = {{0, 0, 0, 0}};
        switch (n->op) {
        case OR_NODE_CONST: r.v[0] = n->k[0]; r.v[1] = n->k[1]; r.v[2] = n->k[2]; break;
        case OR_NODE_RGB: /* eval.rs:125-135: node space (arg0: 1 = ACEScg) -> pipeline.rgb_colorspace */
            r.v[0] = n->k[0]; r.v[1] = n->k[1]; r.v[2] = n->k[2]; r.v[3] = 1.0f;
            or_cs_convert(r.v, n->arg[0] == 1u, (color & OR_COLOR_RGB_ACES) != 0);
            break;
        case OR_NODE_TEXCOORDS: r.v[0] = u; r.v[1] = v; break;
        case OR_NODE_IMAGE: {
            float su = u, sv = v;
            if (n->arg[1] != OR_NODE_NONE) { su = val[n->arg[1]].v[0]; sv = val[n->arg[1]].v[1]; }
            r = or_tex_sample(&images[n->arg[0]], su, sv);
            if (n->arg[2]) for (int c = 0; c < 3; c++) r.v[c] = or_srgb_to_linear(r.v[c]);
            break;
        }
        case OR_NODE_MAPPING: {
            const or_val *a = &val[n->arg[0]], *loc = &val[n->arg[1]], *sc = &val[n->arg[2]];
            for (int c = 0; c < 3; c++)
                r.v[c] = n->arg[3] == 0 ? a->v[c] * sc->v[c] + loc->v[c] : (a->v[c] - loc->v[c]) / sc->v[c];
            break;
        }
        case OR_NODE_CHECKERBOARD: {
            float su = u, sv = v, fx, fy;
            if (n->arg[0] != OR_NODE_NONE) { su = val[n->arg[0]].v[0]; sv = val[n->arg[0]].v[1]; }
            float scale = val[n->arg[1]].v[0];
            int px = or_floor_int((su * scale) * 2.0f, &fx), py = or_floor_int((sv * scale) * 2.0f, &fy);
            r = (((px + py) % 2) == 0) ? val[n->arg[2]] : val[n->arg[3]];
            break;
        }
        case OR_NODE_SPECTRAL_UPLIFT: /* eval.rs:155-175 -> spectral_uplift: rgb_colorspace -> the space of color_repr */
            r = val[n->arg[0]];
            or_cs_convert(r.v, (color & OR_COLOR_RGB_ACES) != 0, (color & OR_COLOR_REPR_ACES) != 0);
            break;
        case OR_NODE_SEPARATE_COLOR: r = val[n->arg[0]]; break;
        case OR_NODE_EXTRACT:
            if (n->arg[1] < 3) r.v[0] = val[n->arg[0]].v[n->arg[1]];
            else { r.v[0] = val[n->arg[0]].v[0]; r.v[1] = val[n->arg[0]].v[1]; }
            break;
        case OR_NODE_NORMAL_MAP: {
            float s = val[n->arg[1]].v[0];
            for (int c = 0; c < 3; c++) r.v[c] = 2.0f * val[n->arg[0]].v[c] - 1.0f;
            if (s != 1.0f) { r.v[0] = r.v[0] * s; r.v[1] = r.v[1] * s; r.v[2] = r.v[2] * 1.0f; }
            break;
        }
        default: break;
        }
        val[i] = r;
    }
}

/* The evaluated inputs of a material at uv: the constants of `m`, overridden by the nodes that feed inputs
 * (principled.rs:13-131 read rules: colours xyz (+ alpha for base_color), scalars x). */
static inline void or_material_at(const or_material_desc *m, const or_material_graph *g, const or_image_desc *images, uint32_t color, float u,
                                  float v, or_material_desc *out) {
    *out = *m;
    out->kind = m->kind & OR_MAT_KIND_MASK;
    { /* constant colour inputs (an Rgb node behind a spectral_uplift, folded by the scene reader): both conversions */
        float *slot[4] = {out->base_color, out->specular_tint, out->coat_tint, out->emission_color};
        const uint32_t bit[4] = {OR_MAT_CS_BASE_COLOR, OR_MAT_CS_SPECULAR_TINT, OR_MAT_CS_COAT_TINT, OR_MAT_CS_EMISSION_COLOR};
        const uint32_t key[4] = {OR_IN_BASE_COLOR, OR_IN_SPECULAR_TINT, OR_IN_COAT_TINT, OR_IN_EMISSION_COLOR};
        for (int k = 0; k < 4; k++) {
            if (g && g->n_nodes && g->input[key[k]] != OR_NODE_NONE) continue; /* fed by the graph: converted at its nodes */
            or_cs_convert(slot[k], (m->kind & bit[k]) != 0, (color & OR_COLOR_RGB_ACES) != 0);
            or_cs_convert(slot[k], (color & OR_COLOR_RGB_ACES) != 0, (color & OR_COLOR_REPR_ACES) != 0);
        }
    }
    if (!g || g->n_nodes == 0) return;
    or_val val[256];
    if (g->n_nodes > 256) return;
    or_eval_graph(g, images, color, u, v, val);
    const uint32_t *in = g->input;
#define OR_IN3(K, F) if (in[K] != OR_NODE_NONE) { out->F[0] = val[in[K]].v[0]; out->F[1] = val[in[K]].v[1]; out->F[2] = val[in[K]].v[2]; }
#define OR_IN1(K, F) if (in[K] != OR_NODE_NONE) out->F = val[in[K]].v[0];
    OR_IN3(OR_IN_BASE_COLOR, base_color)
    if (in[OR_IN_BASE_COLOR] != OR_NODE_NONE) out->base_alpha = val[in[OR_IN_BASE_COLOR]].v[3];
    OR_IN1(OR_IN_METALLIC, metallic) OR_IN1(OR_IN_ROUGHNESS, roughness) OR_IN1(OR_IN_IOR, ior)
    OR_IN1(OR_IN_SPECULAR_IOR_LEVEL, specular_ior_level) OR_IN3(OR_IN_SPECULAR_TINT, specular_tint)
    OR_IN1(OR_IN_TRANSMISSION_WEIGHT, transmission_weight) OR_IN1(OR_IN_COAT_WEIGHT, coat_weight)
    OR_IN1(OR_IN_COAT_ROUGHNESS, coat_roughness) OR_IN1(OR_IN_COAT_IOR, coat_ior) OR_IN3(OR_IN_COAT_TINT, coat_tint)
    OR_IN3(OR_IN_EMISSION_COLOR, emission_color) OR_IN1(OR_IN_EMISSION_STRENGTH, emission_strength) OR_IN3(OR_IN_NORMAL, normal)
#undef OR_IN3
#undef OR_IN1
}
/* does any node of the sub-graph feeding `node` vary over the surface (texcoords, image, checkerboard on si.uv)? */
static inline int or_node_varies(const or_material_graph *g, uint32_t node) {
    if (node == OR_NODE_NONE) return 0;
    const or_shader_node *n = &g->nodes[node];
    switch (n->op) {
    case OR_NODE_CONST: case OR_NODE_RGB: return 0;
    case OR_NODE_TEXCOORDS: case OR_NODE_IMAGE: return 1;
    case OR_NODE_MAPPING: return or_node_varies(g, n->arg[0]) || or_node_varies(g, n->arg[1]) || or_node_varies(g, n->arg[2]);
    case OR_NODE_CHECKERBOARD:
        return n->arg[0] == OR_NODE_NONE || or_node_varies(g, n->arg[0]) || or_node_varies(g, n->arg[1]) || or_node_varies(g, n->arg[2]) ||
               or_node_varies(g, n->arg[3]);
    case OR_NODE_NORMAL_MAP: return or_node_varies(g, n->arg[0]) || or_node_varies(g, n->arg[1]);
    default: return or_node_varies(g, n->arg[0]);
    }
}
#endif
