// dradiance.h -- PathTracerBase::run_pt_hybrid_shift_mapping (crates/akari_integrator/src/pt.rs:329-900) as one function of a
// sampler type: the bounce loop WITH the reconnection shift mapping that the gpt integrator drives (gpt_kernels.hip), and,
// with SM = false, the plain loop (PathTracer::radiance) that the mcmc_opt integrator calls with its primary-sample-space
// sampler (mcmc_kernels.hip). The path tracer proper has its own persistent-lane formulation of the same loop (dpath.h).
#pragma once
#include "dpath.h"

namespace akr {

enum : uint32_t { VT_INVALID = 0, VT_LAST_HIT_LIGHT = 1, VT_LAST_NEE = 2, VT_INTERIOR = 3 };  // pt.rs:975-980

struct ReconVertex {  // ReconnectionVertex, pt.rs:981-1000; (inst_id, prim_id) = global triangle id here
    vec3 direct, indirect;
    vec2 bary;
    vec3 direct_wi;
    float direct_light_pdf;
    vec3 wo;
    uint32_t gid;
    vec3 wi;
    float prev_bsdf_pdf, bsdf_pdf, u_bsdf_select, dist;
    uint32_t depth, type;
};
struct ShiftMapping {  // ReconnectionShiftMapping, pt.rs:1008-1017
    float min_dist, min_roughness;
    bool enabled, is_base, success;
    float jacobian;
};

// Film::add_splat (film.rs:167-194): color.remove_nan() * weight, converted to the film's sRGB primaries when the pipeline
// shades in ACEScg (`aces`; color.to_rgb(SRgb), color.rs:262-275), NaN components flushed
AKR_D vec3 splat_value(vec3 c, float weight, bool aces = false) {
    if (is_nan(c.x) || is_nan(c.y) || is_nan(c.z)) c = mk3(0, 0, 0);
    c = c * weight;
    if (aces) c = cs_convert(c, true, false);
    return mk3(is_nan(c.x) ? 0.0f : c.x, is_nan(c.y) ? 0.0f : c.y, is_nan(c.z) ? 0.0f : c.z);
}

// the random numbers of a path: draw_1d(p, sampler) per sampler type; next_2d = two draws, next_3d = (next_1d, next_2d)
AKR_D float draw_1d(const PtParams& p, Sampler& s) { return next_1d<false>(p, s); }
template <class S>
AKR_D vec2 draw_2d(const PtParams& p, S& s) {
    float a = draw_1d(p, s);
    float b = draw_1d(p, s);
    return mk2(a, b);
}
template <class S>
AKR_D vec3 draw_3d(const PtParams& p, S& s) {
    float a = draw_1d(p, s);
    vec2 b = draw_2d(p, s);
    return mk3(a, b.x, b.y);
}

// run_pt_hybrid_shift_mapping with min_reconnect_depth = 1, no denoising features, no cached first hit.
// SM = false compiles the shift mapping out (PathTracer::radiance = run_megakernel); S = the sampler type, read through draw_1d / draw_3d.
// INST: the scene is kept as meshes + instances (two-level traversal + on-the-fly records, dinst_trav.h); BVH is true then.
template <bool BVH, bool TEX, bool SM, bool INST, class S>
AKR_D vec3 radiance_sm(const PtParams& p, TraceCtx& tc, vec3 ro, vec3 rd, S& smp, ShiftMapping& sm, ReconVertex& vx, vec3& base_out,
                       uint32_t& n_rays) {
    const DScene& sc = p.sc;
    auto closest = [&](vec3 o, vec3 d, uint32_t ex0, Hit& h) {
        n_rays++;
        if (INST) return trace_inst<false, TEX>(sc, o, d, 0.0f, 1e20f, ex0, kInvalid, h, tc.stack, tc.cnt);
        return BVH ? trace_bvh<false, TEX>(sc, o, d, 0.0f, 1e20f, ex0, kInvalid, h, tc.stack, tc.cnt)
                   : trace_exhaustive<false, TEX>(sc, o, d, 0.0f, 1e20f, ex0, kInvalid, h);
    };
    auto occluded_ray = [&](vec3 o, vec3 d, float tmax, uint32_t ex0, uint32_t ex1) {
        Hit h;
        n_rays++;
        if (INST) return trace_inst<true, TEX>(sc, o, d, 0.0f, tmax, ex0, ex1, h, tc.stack, tc.cnt);
        return BVH ? trace_bvh<true, TEX>(sc, o, d, 0.0f, tmax, ex0, ex1, h, tc.stack, tc.cnt)
                   : trace_exhaustive<true, TEX>(sc, o, d, 0.0f, tmax, ex0, ex1, h);
    };
    auto material_of = [&](const SurfacePoint& s, DMaterial& m) {
        m = sc.materials[s.material];
        if (TEX) material_at(sc.tex, s.material, s.uv, m);
    };
    vec3 radiance = mk3(0, 0, 0), beta = mk3(1, 1, 1), rrad = mk3(0, 0, 0), rbeta = mk3(1, 1, 1), base = mk3(0, 0, 0);
    uint32_t depth = 0, ray_ex0 = kInvalid;
    float prev_bsdf_pdf = 0.0f, prev_roughness = 0.0f;
    vec3 prev_p = mk3(0, 0, 0);
    bool rejected = false;
    const bool use_sm = SM && sm.enabled;
    if (use_sm && !sm.is_base) {
        sm.success = false;
        sm.jacobian = 0.0f;
    }
    auto add_radiance = [&](vec3 r) {  // pt.rs:134-149
        radiance = radiance + beta * r;
        if (use_sm) rrad = rrad + rbeta * r;
    };
    auto mul_beta = [&](vec3 r) {  // pt.rs:150-155
        beta = beta * r;
        if (use_sm) rbeta = rbeta * r;
    };
    for (;;) {
        Hit hit;
        if (!closest(ro, rd, ray_ex0, hit)) break;
        SurfacePoint si = surface_interaction_any<INST>(sc, hit.gid, mk2(hit.u, hit.v));
        DMaterial mat;
        material_of(si, mat);
        const vec3 wo = -rd;
        {  // handle_surface_light, pt.rs:230-258
            vec3 direct = mk3(0, 0, 0);
            float w = 0.0f;
            if (si.light >= 0 && (!p.indirect_only || depth > 1)) {
                vec3 emission = material_emission(mat);
                direct = dot(si.ng, rd) < 0.0f ? emission : mk3(0, 0, 0);
                if (depth == 0 || !p.use_nee) w = 1.0f;
                else w = mis_weight(prev_bsdf_pdf, pdf_direct(sc, si, hit.gid, ro));
            }
            add_radiance(direct * w);
        }
        if (depth == 0) base = radiance;
        const float dist_prev = length(prev_p - si.p);
        const bool dist_crit = use_sm && dist_prev >= sm.min_dist, prev_rough_crit = use_sm && prev_roughness >= sm.min_roughness;
        if (use_sm) {  // the last segment hit a light: that hit is the reconnection vertex, pt.rs:418-464
            const bool is_last = depth == p.max_depth, can_connect = dist_crit && prev_rough_crit;
            if (depth >= 1 && can_connect) {
                if (vx.type == VT_INVALID && sm.is_base && is_last) {
                    vx.direct = mk3(0, 0, 0); vx.indirect = mk3(0, 0, 0); vx.bary = mk2(hit.u, hit.v); vx.direct_wi = mk3(0, 0, 0);
                    vx.direct_light_pdf = 0.0f; vx.wo = wo; vx.gid = hit.gid; vx.wi = mk3(0, 0, 0); vx.prev_bsdf_pdf = prev_bsdf_pdf;
                    vx.bsdf_pdf = 0.0f; vx.u_bsdf_select = 0.0f; vx.dist = dist_prev; vx.depth = depth; vx.type = VT_LAST_HIT_LIGHT;
                } else if (!sm.is_base && is_last) {
                    rejected = true;
                    break;
                }
            }
        }
        if (depth >= p.max_depth) break;
        depth += 1;
        const vec3 u_direct = draw_3d(p, smp);
        LightSample dl;
        dl.valid = false;
        if (p.use_nee && (!p.indirect_only || depth > 1)) dl = sample_direct<TEX, INST>(sc, si.p, si.ng, u_direct.x, mk2(u_direct.y, u_direct.z));
        if (!dl.valid) {  // DirectLighting::invalid, pt.rs:67-77
            dl.li = mk3(0, 0, 0); dl.wi = mk3(0, 0, 0); dl.pdf = 0.0f;
        }
        bool occluded = true;
        const vec3 u_bsdf = draw_3d(p, smp);
        ShadePoint sp;
        shade_point_init(sp, mat, si.frame, si.ng, false);
        vec3 direct = mk3(0, 0, 0);
        if (dl.valid) {  // sample_surface_and_shade_direct, pt.rs:297-323
            BsdfEval e = shade_evaluate(sp, mat, sc.ggx_table, wo, dl.wi);
            float w = mis_weight(dl.pdf, e.pdf);
            direct = div_s((dl.li * e.f) * w, dl.pdf);
        }
        const BsdfSample bs = shade_sample(sp, mat, sc.ggx_table, wo, u_bsdf.x, mk2(u_bsdf.y, u_bsdf.z));
        const float u_select = u_bsdf.x;
        const float roughness = shade_roughness(sp, mat, sc.ggx_table, wo, u_bsdf.x);
        const bool rough_crit = use_sm && roughness >= sm.min_roughness;
        if (dl.valid) {  // pt.rs:504-513
            occluded = occluded_ray(dl.ro, dl.wi, dl.tmax, hit.gid, dl.ex1);
            if (!occluded) add_radiance(direct);
            if (depth == 1) base = radiance;
        }
        if (use_sm && !sm.is_base && vx.type != VT_INVALID) {  // perform the reconnection, pt.rs:515-774
            if (depth > 1 && dist_crit && prev_rough_crit && rough_crit) {  // a vertex the base path would have picked: not reversible
                rejected = true;
                break;
            }
            if (vx.depth == depth) {
                const SurfacePoint rsi = surface_interaction_any<INST>(sc, vx.gid, vx.bary);
                const vec3 dvec = rsi.p - si.p;
                const float dist = length(dvec);
                const vec3 wi = normalize(dvec);
                if (!(dist >= sm.min_dist && rough_crit)) { rejected = true; break; }
                const vec3 vis_o = offset_ray_origin(si.p, face_forward(si.ng, wi));
                const float cos_y2 = abs_f(dot(rsi.ng, wi)), cos_x2 = abs_f(dot(rsi.ng, vx.wo));
                if (cos_y2 == 0.0f) { rejected = true; break; }
                if (occluded_ray(vis_o, wi, dist * (1.0f - 1e-3f), hit.gid, vx.gid)) { rejected = true; break; }
                const BsdfEval e1 = shade_evaluate(sp, mat, sc.ggx_table, wo, wi);
                const vec3 f1 = e1.f;
                const float pdf_y1 = e1.pdf;
                DMaterial mat_y;
                material_of(rsi, mat_y);
                ShadePoint sp_y;
                shade_point_init(sp_y, mat_y, rsi.frame, rsi.ng, false);
                float roughness_y = 0.0f, pdf_y2 = 0.0f;
                vec3 f2 = mk3(0, 0, 0), direct_f = mk3(0, 0, 0);
                if (vx.type != VT_LAST_HIT_LIGHT) {
                    BsdfEval e2 = shade_evaluate(sp_y, mat_y, sc.ggx_table, -wi, vx.wi);
                    f2 = e2.f;
                    pdf_y2 = e2.pdf;
                    roughness_y = shade_roughness(sp_y, mat_y, sc.ggx_table, -wi, vx.u_bsdf_select);
                }
                if (vx.direct_wi.x != 0.0f || vx.direct_wi.y != 0.0f || vx.direct_wi.z != 0.0f) {
                    BsdfEval ed = shade_evaluate(sp_y, mat_y, sc.ggx_table, -wi, vx.direct_wi);
                    direct_f = ed.f * mis_weight(vx.direct_light_pdf, ed.pdf);
                }
                if (vx.type != VT_LAST_HIT_LIGHT && roughness_y < sm.min_roughness) { rejected = true; break; }  // reversibility
                float pdf_ratio = pdf_y1 / vx.prev_bsdf_pdf;
                if (vx.type != VT_LAST_HIT_LIGHT)
                    pdf_ratio *= vx.bsdf_pdf == 0.0f ? (pdf_y2 == 0.0f ? 1.0f : 0.0f) : pdf_y2 / vx.bsdf_pdf;
                if (pdf_ratio <= 0.0f) { rejected = true; break; }
                vec3 le = mk3(0, 0, 0);
                float light_pdf = 0.0f;
                if (rsi.light >= 0) {
                    le = dot(rsi.ng, wi) < 0.0f ? material_emission(mat_y) : mk3(0, 0, 0);
                    light_pdf = pdf_direct(sc, rsi, vx.gid, si.p);
                }
                const float w = p.use_nee ? mis_weight(pdf_y1, light_pdf) : 1.0f;
                vec3 vertex_le = le * w;
                if (p.indirect_only && depth == 1) vertex_le = mk3(0, 0, 0);
                const vec3 f_pdf = div_s(f1, pdf_y1);
                float cont_prob = 1.0f;  // compute_contibue_prob(vertex.depth, reconnect_beta * f_pdf), pt.rs:211-218
                if (vx.depth > p.rr_depth) cont_prob = clamp_f(max3(rbeta * f_pdf), 0.0f, 1.0f) * 0.95f;
                vec3 sum = vertex_le + direct_f * vx.direct;
                sum = sum + (pdf_y2 > 0.0f ? div_s(f2 * vx.indirect, pdf_y2) : mk3(0, 0, 0));
                add_radiance(div_s(f_pdf * sum, cont_prob));
                float jac = (pdf_ratio * abs_f(cos_y2 / cos_x2)) * sqr(vx.dist / dist);
                if (!is_finite(jac)) jac = 0.0f;
                sm.success = jac > 0.0f;
                sm.jacobian = jac;
                if (!sm.success) rejected = true;
                break;
            }
        }
        mul_beta(div_s(bs.color, bs.pdf));  // pt.rs:783
        if (use_sm && depth > 1) {  // the base path picks its reconnection vertex, pt.rs:784-831
            const bool can_connect = dist_crit && prev_rough_crit && rough_crit;
            if (vx.type == VT_INVALID && sm.is_base && can_connect) {
                vx.direct = (dl.valid && !occluded) ? div_s(dl.li, dl.pdf) : mk3(0, 0, 0);
                vx.indirect = mk3(0, 0, 0); vx.bary = mk2(hit.u, hit.v); vx.direct_wi = dl.wi; vx.direct_light_pdf = dl.pdf; vx.wo = wo;
                vx.gid = hit.gid; vx.wi = bs.wi; vx.prev_bsdf_pdf = prev_bsdf_pdf; vx.bsdf_pdf = bs.pdf; vx.u_bsdf_select = u_select;
                vx.dist = dist_prev; vx.depth = depth - 1; vx.type = VT_LAST_NEE;
                rbeta = mk3(1, 1, 1);
                rrad = mk3(0, 0, 0);
            }
            if (!sm.is_base && can_connect) {
                rejected = true;
                break;
            }
        }
        if (bs.pdf <= 0.0f || !bs.valid || min3(bs.color) < 0.0f) break;  // pt.rs:832-842
        if (depth > p.rr_depth) {                                          // pt.rs:843-850
            float cont_prob = clamp_f(max3(beta), 0.0f, 1.0f) * 0.95f;
            if (draw_1d(p, smp) >= cont_prob) break;
            mul_beta(div_s(mk3(1, 1, 1), cont_prob));
        }
        prev_bsdf_pdf = bs.pdf;
        prev_p = si.p;
        prev_roughness = roughness;
        ro = offset_ray_origin(si.p, face_forward(si.ng, bs.wi));
        rd = bs.wi;
        ray_ex0 = hit.gid;
    }
    {  // pt.rs:871-876
        vec3 ind = radiance - base;
        ind = mk3(clamp_f(ind.x, 0.0f, 1000.0f), clamp_f(ind.y, 0.0f, 1000.0f), clamp_f(ind.z, 0.0f, 1000.0f));
        radiance = base + ind;
    }
    if (use_sm) {  // pt.rs:878-899
        if (vx.type != VT_INVALID && vx.type != VT_LAST_HIT_LIGHT && sm.is_base) vx.indirect = rrad;
        if (!sm.is_base && vx.type == VT_INVALID) {
            sm.success = !rejected;
            sm.jacobian = sm.success ? 1.0f : 0.0f;
        }
    }
    base_out = base;
    return radiance;
}

}  // namespace akr
