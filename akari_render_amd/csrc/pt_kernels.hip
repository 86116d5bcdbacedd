// pt_kernels.hip -- hand-written gfx950 kernels of the `pt` integrator.
//
// k_pt_pass runs `n_passes` consecutive passes of the reference's render loop for its pixels
// (crates/akari_integrator/src/pt.rs:1075-1103,1126-1133): in every pass a pixel takes `pass_spp` samples (the last
// pass of a render may be shorter). Fusing passes keeps the sampler state and the film accumulator in registers
// between them -- the values are those of separate launches, because a pass boundary is only "advance(-dim); dim = 0"
// on the sampler -- and it averages the path-length variance of a wave's 64 lanes over more samples. The reference runs this as one JIT-compiled thread per pixel with two
// nested loops (samples, bounces), so a wave idles on its longest path. Here a lane is a small state machine that
// advances ONE path vertex per iteration and, when its path ends, splats the sample and regenerates the next
// camera ray in the same iteration -- all 64 lanes of a wave stay on the same code (intersect / shade / shadow)
// until the lane's pixel has all its samples. Sample values, RNG consumption order and film arithmetic are the
// reference's; only the schedule differs.
#include "device/disect.h"
#include "device/drng.h"
#include "kernels.h"

namespace akr {

// ----------------------------------------------------------------------------------------------------------
// work distribution: item index -> pixel. Items enumerate the pixels of the tiles this rank owns
// (tile t belongs to rank t % shard_count), tile by tile, and inside a tile in 8x8 blocks so that one wave
// covers an 8x8 pixel square (coherent primary rays, one film cache line per row segment).
AKR_D bool item_to_pixel(const PtParams& p, uint32_t item, uint32_t& px, uint32_t& py) {
    const uint32_t tile_px = p.tile_w * p.tile_h;
    uint32_t j = item / tile_px, within = item - j * tile_px;
    uint32_t tile = p.shard_rank + j * p.shard_count;
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

struct Sampler {  // IndependentSampler, sampler/mod.rs:161-217
    Pcg32 pcg;
    uint32_t dim;
};
AKR_D float next_1d(Sampler& s) {
    s.dim += 1;
    return pcg_next_1d(s.pcg);
}
AKR_D vec2 next_2d(Sampler& s) {
    float a = next_1d(s);
    float b = next_1d(s);
    return mk2(a, b);
}
AKR_D vec3 next_3d(Sampler& s) {
    float a = next_1d(s);
    vec2 b = next_2d(s);
    return mk3(a, b.x, b.y);
}

// camera/mod.rs:70-103
AKR_D void generate_ray(const PtParams& p, uint32_t px, uint32_t py, Sampler& smp, vec3& o, vec3& d) {
    vec2 fpixel = mk2((float)px + 0.5f, (float)py + 0.5f);
    vec2 offset = filter_sample(p, next_2d(smp));
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

AKR_D float mis_weight(float a, float b) {  // pt.rs:962-973 with power = 1
    float pa = 1.0f * a, pb = 1.0f * b;
    return pa / (pa + pb);
}

// emission of the material at a surface point (AreaLightExpr::emission, light/area.rs:19-31): Principled returns
// its emission constant (principled.rs:267-274), an Emission node likewise, everything else is black.
AKR_D vec3 material_emission(const DMaterial& m) {
    return (m.kind == MAT_PRINCIPLED || m.kind == MAT_EMISSION) ? m.emission : mk3(0, 0, 0);
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
    uint32_t light = alias_sample_and_remap(sc.light_entries, sc.light_pdf, sc.n_lights, u_select, light_choice_pdf, u_sel2);
    uint32_t off = sc.light_tri_offset[light];
    uint32_t prim = alias_sample_and_remap(sc.area_entries + off, sc.area_pdf + off, sc.light_n_tris[light], u_sel2, pdf_prim, u_unused);
    uint32_t gid = sc.inst_tri_offset[sc.light_inst[light]] + prim;
    vec2 bary = uniform_sample_triangle(u_sample);
    SurfacePoint y = surface_interaction(sc, gid, bary);
    vec3 wi = y.p - pn_p;
    if (length2(wi) == 0.0f) return s;
    float dist2 = length2(wi);
    wi = div_s(wi, __builtin_sqrtf(dist2));
    vec3 emission = material_emission(sc.materials[y.material]);
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
    uint32_t prim = gid - sc.inst_tri_offset[si.inst];
    float prim_pdf = sc.area_pdf[sc.light_tri_offset[light] + prim];
    vec3 wi = si.p - pn_p;
    float dist2 = length2(wi);
    wi = div_s(wi, __builtin_sqrtf(dist2));
    float pdf = prim_pdf / si.prim_area * dist2 / max_f(abs_f(dot(si.ng, wi)), 1e-6f);
    return light_choice_pdf * pdf;
}

// Per-thread intersection context: the LDS stack slot of this lane and the traversal counters.
struct TraceCtx {
    uint32_t* stack;
    TraceCounters cnt;
};
template <bool BVH, bool ANY_HIT>
AKR_D bool trace(const PtParams& p, TraceCtx& tc, vec3 o, vec3 d, float tmin, float tmax, uint32_t ex0, uint32_t ex1, Hit& hit) {
    if (BVH) return trace_bvh4<ANY_HIT>(p.sc, o, d, tmin, tmax, ex0, ex1, hit, tc.stack, tc.cnt);
    return trace_exhaustive<ANY_HIT>(p.sc, o, d, tmin, tmax, ex0, ex1, hit);
}

AKR_D uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ----------------------------------------------------------------------------------------------------------
#ifndef AKR_PT_MIN_WAVES
#define AKR_PT_MIN_WAVES 4  // waves per SIMD the register allocator must leave room for (occupancy)
#endif
#ifndef AKR_PT_MIN_WAVES_BVH
#define AKR_PT_MIN_WAVES_BVH 4
#endif
template <bool BVH>
__global__ __launch_bounds__(256, BVH ? AKR_PT_MIN_WAVES_BVH : AKR_PT_MIN_WAVES) void k_pt_pass(const PtParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];  // BVH: kBvhStackDepth x 256 words
    TraceCtx tc;
    tc.stack = lds_stack + threadIdx.x;
    tc.cnt = TraceCounters{0, 0, 0};
    const DScene& sc = p.sc;
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    uint32_t px = 0, py = 0;
    bool active = item < p.n_items && item_to_pixel(p, item, px, py);
    const uint32_t pix = px + py * p.width;
    const size_t N = (size_t)p.width * p.height;

    Sampler smp;
    smp.pcg = Pcg32{0, 1};
    smp.dim = 0;
    vec3 film_rgb = mk3(0, 0, 0);
    float film_w = 0.0f;
    if (active) {
        smp.pcg = p.states[pix];  // SamplerCreator::create, sampler/mod.rs:317-327
        film_rgb = mk3(p.film[3 * (size_t)pix + 0], p.film[3 * (size_t)pix + 1], p.film[3 * (size_t)pix + 2]);
        film_w = p.film[6 * N + pix];
    }
    // shifted pixel (pt.rs:1084-1088)
    int32_t sxi = (int32_t)px + p.pixel_offset[0], syi = (int32_t)py + p.pixel_offset[1];
    sxi = sxi < 0 ? 0 : (sxi > (int32_t)p.width - 1 ? (int32_t)p.width - 1 : sxi);
    syi = syi < 0 ? 0 : (syi > (int32_t)p.height - 1 ? (int32_t)p.height - 1 : syi);
    const uint32_t sx = (uint32_t)sxi, sy = (uint32_t)syi;

    // per-path registers (PathTracerBase, pt.rs:28-57)
    vec3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1);
    uint32_t ray_ex0 = kInvalid;
    vec3 radiance = mk3(0, 0, 0), beta = mk3(1, 1, 1), base = mk3(0, 0, 0);
    uint32_t depth = 0;
    float prev_bsdf_pdf = 0.0f;
    // the shadow ray of the vertex shaded last iteration, traced together with the next closest-hit ray
    vec3 s_o = mk3(0, 0, 0), s_d = mk3(0, 0, 1), s_contrib = mk3(0, 0, 0);
    float s_tmax = -1.0f;
    uint32_t s_ex0 = kInvalid, s_ex1 = kInvalid;
    bool has_ray = active, has_shadow = false, s_add = false, s_depth1 = false;
    bool finalize = false, lane_done = false;
    uint32_t samples_done = 0, pass_idx = 0, c_samples = 0;
    uint32_t cur_spp = (p.n_passes == 1) ? p.last_pass_spp : p.pass_spp;
    uint32_t c_closest = 0, c_shadow = 0, c_shaded = 0;

    if (active) {
        pcg_start(smp.pcg, p.start);  // sampler.start(), sampler/mod.rs:199-203
        generate_ray(p, sx, sy, smp, ro, rd);
    }

    while (__builtin_amdgcn_ballot_w64(active) != 0) {
        if (active) {
            // ---- 1. intersection: next closest-hit ray + pending shadow ray ----
            Hit hit;
            bool found = false, occluded = false;
            c_closest += has_ray ? 1u : 0u;
            c_shadow += has_shadow ? 1u : 0u;
            if (BVH) {
                if (has_ray) found = trace_bvh4<false>(sc, ro, rd, 0.0f, 1e20f, ray_ex0, kInvalid, hit, tc.stack, tc.cnt);
                if (has_shadow) {
                    Hit sh;
                    occluded = trace_bvh4<true>(sc, s_o, s_d, 0.0f, s_tmax, s_ex0, s_ex1, sh, tc.stack, tc.cnt);
                }
            } else {
                trace_pair_exhaustive(sc, ro, rd, has_ray ? 1e20f : -1.0f, ray_ex0, s_o, s_d, has_shadow ? s_tmax : -1.0f, s_ex0, s_ex1, hit,
                                      found, occluded);
            }
            // ---- 2. resolve the shadow ray (pt.rs:504-513) ----
            if (has_shadow) {
                if (!occluded && s_add) radiance = radiance + s_contrib;
                if (s_depth1) base = radiance;
                has_shadow = false;
            }
            // ---- 3. finish the sample whose last vertex was shaded in the previous iteration ----
            if (finalize) {
                // pt.rs:871-876 (clamp_indirect = 1000), then film.add_sample with weight 1 (film.rs:196-229)
                vec3 ind = radiance - base;
                ind = mk3(clamp_f(ind.x, 0.0f, 1000.0f), clamp_f(ind.y, 0.0f, 1000.0f), clamp_f(ind.z, 0.0f, 1000.0f));
                vec3 L = base + ind;
                if (is_nan(L.x) || is_nan(L.y) || is_nan(L.z)) L = mk3(0, 0, 0);
                film_rgb = mk3(film_rgb.x + L.x * 1.0f, film_rgb.y + L.y * 1.0f, film_rgb.z + L.z * 1.0f);
                film_w = film_w + 1.0f;
                radiance = mk3(0, 0, 0);
                beta = mk3(1, 1, 1);
                base = mk3(0, 0, 0);
                depth = 0;
                prev_bsdf_pdf = 0.0f;
                finalize = false;
                if (lane_done) {
                    active = false;
                    p.states[pix] = smp.pcg;
                    p.film[3 * (size_t)pix + 0] = film_rgb.x;
                    p.film[3 * (size_t)pix + 1] = film_rgb.y;
                    p.film[3 * (size_t)pix + 2] = film_rgb.z;
                    p.film[6 * N + pix] = film_w;
                }
            }
            // ---- 4. shade the vertex the closest-hit ray found ----
            if (active && has_ray) {
                bool terminated = false;
                if (!found) {
                    terminated = true;  // pt.rs:381-396 (hit_envmap adds zero)
                } else {
                    SurfacePoint si = surface_interaction(sc, hit.gid, mk2(hit.u, hit.v));
                    const DMaterial& mat = sc.materials[si.material];
                    vec3 wo = -rd;
                    {  // handle_surface_light, pt.rs:230-258
                        vec3 direct = mk3(0, 0, 0);
                        float w = 0.0f;
                        if (si.light >= 0 && (!p.indirect_only || depth > 1)) {
                            vec3 emission = material_emission(mat);
                            direct = dot(si.ng, rd) < 0.0f ? emission : mk3(0, 0, 0);
                            if (depth == 0 || !p.use_nee)
                                w = 1.0f;
                            else
                                w = mis_weight(prev_bsdf_pdf, pdf_direct(sc, si, hit.gid, ro));
                        }
                        if (p.debug_depth < 0 || depth == (uint32_t)p.debug_depth) radiance = radiance + beta * (direct * w);
                    }
                    if (depth == 0) base = radiance;
                    if (depth >= p.max_depth) {
                        terminated = true;
                    } else {
                        depth += 1;
                        c_shaded++;
                        vec3 u_direct = next_3d(smp);
                        LightSample dl;
                        dl.valid = false;
                        if (p.use_nee && (!p.indirect_only || depth > 1))
                            dl = sample_direct(sc, si.p, si.ng, u_direct.x, mk2(u_direct.y, u_direct.z));
                        vec3 u_bsdf = next_3d(smp);
                        // sample_surface_and_shade_direct, pt.rs:297-323
                        ShadePoint sp;
                        shade_point_init(sp, mat, si.frame, si.ng, p.force_diffuse != 0);
                        if (dl.valid) {
                            BsdfEval e = shade_evaluate(sp, mat, sc.ggx_table, wo, dl.wi);
                            float w = mis_weight(dl.pdf, e.pdf);
                            vec3 direct = div_s((dl.li * e.f) * w, dl.pdf);
                            // the shadow ray is traced next iteration; what it would add is fixed now (beta of THIS vertex)
                            s_contrib = beta * direct;
                            s_add = p.debug_depth < 0 || depth == (uint32_t)p.debug_depth;
                            s_depth1 = depth == 1;
                            s_o = dl.ro;
                            s_d = dl.wi;
                            s_tmax = dl.tmax;
                            s_ex0 = hit.gid;
                            s_ex1 = dl.ex1;
                            has_shadow = true;
                        }
                        BsdfSample bs = shade_sample(sp, mat, sc.ggx_table, wo, u_bsdf.x, mk2(u_bsdf.y, u_bsdf.z));
                        beta = beta * div_s(bs.color, bs.pdf);  // pt.rs:783
                        if (bs.pdf <= 0.0f || !bs.valid || min3(bs.color) < 0.0f) {
                            terminated = true;  // pt.rs:832-842
                        } else {
                            bool cont = true;
                            if (depth > p.rr_depth) {  // pt.rs:211-224, 843-850
                                float cont_prob = clamp_f(max3(beta), 0.0f, 1.0f) * 0.95f;
                                if (next_1d(smp) >= cont_prob)
                                    cont = false;
                                else
                                    beta = beta * div_s(mk3(1, 1, 1), cont_prob);
                            }
                            if (!cont) {
                                terminated = true;
                            } else {  // pt.rs:851-865
                                prev_bsdf_pdf = bs.pdf;
                                ro = offset_ray_origin(si.p, face_forward(si.ng, bs.wi));
                                rd = bs.wi;
                                ray_ex0 = hit.gid;
                            }
                        }
                    }
                }
                if (terminated) {
                    // this sample draws no more random numbers: account for it and start the next camera ray now; its
                    // radiance is finished (step 3) after the shadow ray still pending has been resolved
                    finalize = true;
                    samples_done++;
                    c_samples++;
                    bool more = true;
                    if (samples_done == cur_spp) {
                        // end of a pass: Drop for IndependentSampler (sampler/mod.rs:168-177) = advance(-dim); the next
                        // pass re-creates the sampler from that state with dim = 0 (sampler/mod.rs:317-327)
                        pcg_advance(smp.pcg, -(int64_t)smp.dim);
                        smp.dim = 0;
                        samples_done = 0;
                        pass_idx++;
                        cur_spp = (pass_idx + 1 == p.n_passes) ? p.last_pass_spp : p.pass_spp;
                        more = pass_idx < p.n_passes;
                    }
                    if (more) {
                        pcg_start(smp.pcg, p.start);
                        generate_ray(p, sx, sy, smp, ro, rd);
                        ray_ex0 = kInvalid;
                    } else {
                        has_ray = false;
                        lane_done = true;
                    }
                }
            }
        }
    }
    if (p.counters != nullptr) {
        uint32_t a = wave_sum_u32(c_samples), b = wave_sum_u32(c_closest), c = wave_sum_u32(c_shadow), e = wave_sum_u32(c_shaded);
        uint32_t nn = wave_sum_u32(tc.cnt.nodes), nt = wave_sum_u32(tc.cnt.tris), ov = wave_sum_u32(tc.cnt.overflow);
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd((unsigned long long*)&p.counters[0], (unsigned long long)a);
            atomicAdd((unsigned long long*)&p.counters[1], (unsigned long long)b);
            atomicAdd((unsigned long long*)&p.counters[2], (unsigned long long)c);
            atomicAdd((unsigned long long*)&p.counters[3], (unsigned long long)e);
            atomicAdd((unsigned long long*)&p.counters[4], (unsigned long long)nn);
            atomicAdd((unsigned long long*)&p.counters[5], BVH ? (unsigned long long)nt : (unsigned long long)(b + c) * p.sc.n_tris);
            if (ov) atomicAdd((unsigned long long*)&p.counters[6], (unsigned long long)ov);
        }
    }
}

// init_pcg32_buffer_with_seed's device half (sampler/mod.rs:152-158)
__global__ void k_init_pcg32(const uint64_t* __restrict__ seeds, Pcg32* __restrict__ states, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) states[i] = pcg_new_seq_offset(i, seeds[i]);
}

// Film resolve: copy_to_rgba_image with hdr = true, splat_scale = 1 (film.rs:120-148)
__global__ void k_film_resolve(const float* __restrict__ film, uint64_t n, float* __restrict__ rgb) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float w = film[6 * n + i];
    float inv = w == 0.0f ? 1.0f : w;
#pragma unroll
    for (int c = 0; c < 3; c++) rgb[3 * i + c] = film[3 * i + c] / inv + film[3 * n + 3 * i + c] * 1.0f;
}

// PreComputedTables::init "ggx_dielectric_s" (svm/surface/precompute.rs:56-94,133-145; mod.rs:1336-1356):
// one thread per table entry, 2^20 sequential samples each (the f32 running sum makes the order part of the value).
__global__ void k_ggx_dielectric_table(const uint64_t* __restrict__ seeds, float* __restrict__ table, uint32_t samples) {
    const uint32_t dim = 16;
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= dim * dim * dim) return;
    uint32_t tx = gid % dim, ty = (gid / dim) % dim, tz = gid / (dim * dim);
    Pcg32 rng = pcg_new_seq_offset(gid, seeds[gid]);
    float fx = clamp_f((float)tx / ((float)dim - 1.0f), 1e-4f, 0.9999f);
    float fy = clamp_f((float)ty / ((float)dim - 1.0f), 1e-4f, 0.9999f);
    float fz = clamp_f((float)tz / ((float)dim - 1.0f), 1e-4f, 0.9999f);
    float ior = ior_from_f0(sqr(sqr(fz)));
    vec2 alpha = mk2(max_f(fx * fx, 1e-4f), max_f(fx * fx, 1e-4f));
    Frame frame = frame_from_n(mk3(0, 0, 1));
    vec3 ng = mk3(0, 0, 1);
    vec3 wo = mk3(__builtin_sqrtf(1.0f - sqr(fy)), 0.0f, fy);
    float sum = 0.0f;
    for (uint32_t s = 0; s < samples; s++) {
        float u0 = pcg_next_1d(rng), u1 = pcg_next_1d(rng), u2 = pcg_next_1d(rng);
        (void)u0;
        // SurfaceClosure::sample on MicrofacetReflection{color 1, FresnelDielectric(ior), GGX(roughness)}
        vec3 lo = to_local(frame, wo), wl;
        bool valid = sample_lobe(LOBE_REFLECT, alpha, ior, lo, mk2(u1, u2), wl);
        vec3 wi = to_world(frame, wl);
        valid = valid & check_wo_wi_valid(frame.n, ng, wo, wi);
        float val = 0.0f;
        if (valid) {
            BsdfEval e{mk3(0, 0, 0), 0.0f};
            if (check_wo_wi_valid(frame.n, ng, wo, wi))
                e = eval_reflection<FR_DIELECTRIC>(mk3(1, 1, 1), ior, mk3(0, 0, 0), mk3(0, 0, 0), alpha, to_local(frame, wo), to_local(frame, wi));
            if (e.pdf > 0.0f) val = e.f.x / e.pdf;
        }
        sum += val;
    }
    table[gid] = sum / (float)samples;
}

// ---------------------------------------------------------------------------------------------------- probes
__global__ void k_probe_math(uint32_t n, const float* __restrict__ x, float* s, float* c, float* l) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sv, cv;
    sincos_f(x[i], sv, cv);
    s[i] = sv;
    c[i] = cv;
    l[i] = log_f(x[i]);
}
__global__ void k_probe_bsdf(const DMaterial* __restrict__ m, const float* __restrict__ table, int mode, vec3 wo, uint32_t n,
                             const float* __restrict__ in, float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ShadePoint sp;
    shade_point_init(sp, *m, frame_from_n(mk3(0, 0, 1)), mk3(0, 0, 1), false);
    if (mode == 0) {
        BsdfEval e = shade_evaluate(sp, *m, table, wo, mk3(in[3 * i], in[3 * i + 1], in[3 * i + 2]));
        out[4 * i + 0] = e.f.x;
        out[4 * i + 1] = e.f.y;
        out[4 * i + 2] = e.f.z;
        out[4 * i + 3] = e.pdf;
    } else {
        BsdfSample s = shade_sample(sp, *m, table, wo, in[3 * i], mk2(in[3 * i + 1], in[3 * i + 2]));
        float* o = out + 8 * (size_t)i;
        o[0] = s.wi.x; o[1] = s.wi.y; o[2] = s.wi.z;
        o[3] = s.color.x; o[4] = s.color.y; o[5] = s.color.z;
        o[6] = s.pdf;
        o[7] = s.valid ? 1.0f : 0.0f;
    }
}
template <bool BVH>
__global__ __launch_bounds__(256) void k_probe_intersect(PtParams p, uint32_t n, const float* __restrict__ rays, uint32_t* __restrict__ out, float* __restrict__ bary) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    TraceCtx tc;
    tc.stack = lds_stack + threadIdx.x;
    tc.cnt = TraceCounters{0, 0, 0};
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool act = i < n;
    const float* r = rays + 8 * (size_t)(act ? i : 0);
    Hit h;
    bool found = trace<BVH, false>(p, tc, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), r[6], r[7], kInvalid, kInvalid, h);
    if (!act) return;
    uint32_t inst = 0, prim = 0;
    if (found) {
        inst = f2u(p.sc.shade[(size_t)h.gid * SHADE_ROWS + 6].z);
        prim = h.gid - p.sc.inst_tri_offset[inst];
    }
    out[3 * i + 0] = found ? 1u : 0u;
    out[3 * i + 1] = inst;
    out[3 * i + 2] = prim;
    bary[2 * i + 0] = found ? h.u : 0.0f;
    bary[2 * i + 1] = found ? h.v : 0.0f;
}
__global__ void k_probe_si(PtParams p, uint32_t n, const uint32_t* __restrict__ inst_prim, const float* __restrict__ bary, float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t gid = p.sc.inst_tri_offset[inst_prim[2 * i]] + inst_prim[2 * i + 1];
    SurfacePoint s = surface_interaction(p.sc, gid, mk2(bary[2 * i], bary[2 * i + 1]));
    float* o = out + 19 * (size_t)i;
    o[0] = s.p.x; o[1] = s.p.y; o[2] = s.p.z;
    o[3] = s.ng.x; o[4] = s.ng.y; o[5] = s.ng.z;
    o[6] = s.frame.n.x; o[7] = s.frame.n.y; o[8] = s.frame.n.z;
    o[9] = s.frame.t.x; o[10] = s.frame.t.y; o[11] = s.frame.t.z;
    o[12] = s.frame.s.x; o[13] = s.frame.s.y; o[14] = s.frame.s.z;
    const float4* r = p.sc.shade + (size_t)gid * SHADE_ROWS;
    vec2 b = mk2(bary[2 * i], bary[2 * i + 1]);
    float w = 1.0f - b.x - b.y;
    o[15] = (r[0].w * w + r[2].w * b.x) + r[4].w * b.y;  // uv = interp(uv0, uv1, uv2)
    o[16] = (r[1].w * w + r[3].w * b.x) + r[5].w * b.y;
    o[17] = s.prim_area;
    o[18] = (float)s.material;
}

// ---------------------------------------------------------------------------------------------------- launchers
hipError_t launch_pt_pass(const PtParams& p, hipStream_t stream) {
    uint32_t blocks = (p.n_items + 255u) / 256u;
    if (blocks == 0) return hipSuccess;
    if (p.sc.bvh_nodes != nullptr)
        hipLaunchKernelGGL(k_pt_pass<true>, dim3(blocks), dim3(256), kBvhStackDepth * 256 * 4, stream, p);
    else
        hipLaunchKernelGGL(k_pt_pass<false>, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_init_pcg32(const uint64_t* seeds, void* states, uint64_t n, hipStream_t stream) {
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_init_pcg32, dim3(blocks), dim3(256), 0, stream, seeds, (Pcg32*)states, n);
    return hipGetLastError();
}
hipError_t launch_film_resolve(const float* film, uint64_t n, float* rgb, hipStream_t stream) {
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_film_resolve, dim3(blocks), dim3(256), 0, stream, film, n, rgb);
    return hipGetLastError();
}
hipError_t launch_ggx_table(const uint64_t* seeds, float* table, uint32_t samples, hipStream_t stream) {
    hipLaunchKernelGGL(k_ggx_dielectric_table, dim3(64), dim3(64), 0, stream, seeds, table, samples);
    return hipGetLastError();
}
hipError_t launch_probe_math(uint32_t n, const float* x, float* s, float* c, float* l, hipStream_t stream) {
    hipLaunchKernelGGL(k_probe_math, dim3((n + 255) / 256), dim3(256), 0, stream, n, x, s, c, l);
    return hipGetLastError();
}
hipError_t launch_probe_bsdf(const DMaterial* m, const float* table, int mode, const float* wo, uint32_t n, const float* in, float* out,
                             hipStream_t stream) {
    hipLaunchKernelGGL(k_probe_bsdf, dim3((n + 255) / 256), dim3(256), 0, stream, m, table, mode, mk3(wo[0], wo[1], wo[2]), n, in, out);
    return hipGetLastError();
}
hipError_t launch_probe_intersect(const PtParams& p, uint32_t n, const float* rays, uint32_t* out, float* bary, hipStream_t stream) {
    if (p.sc.bvh_nodes != nullptr)
        hipLaunchKernelGGL(k_probe_intersect<true>, dim3((n + 255) / 256), dim3(256), kBvhStackDepth * 256 * 4, stream, p, n, rays, out, bary);
    else
        hipLaunchKernelGGL(k_probe_intersect<false>, dim3((n + 255) / 256), dim3(256), 0, stream, p, n, rays, out, bary);
    return hipGetLastError();
}
hipError_t launch_probe_si(const PtParams& p, uint32_t n, const uint32_t* inst_prim, const float* bary, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_probe_si, dim3((n + 255) / 256), dim3(256), 0, stream, p, n, inst_prim, bary, out);
    return hipGetLastError();
}

}  // namespace akr
