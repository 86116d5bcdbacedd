// aov_kernels.hip -- the `aov` integrator (crates/akari_integrator/src/aov.rs:57-173) on gfx950: per pixel `spp` camera rays,
// the value of one closure / geometry attribute at the first hit, accumulated into the film like a radiance sample. Shares
// the camera, sampler, intersectors, hit reconstruction and material records of the path tracer (device/*.h).
#include "device/dpath.h"

namespace akr {

enum : uint32_t { AOV_NS = 0, AOV_NG = 1, AOV_TANGENT = 2, AOV_BITANGENT = 3, AOV_ALBEDO = 4, AOV_ROUGHNESS = 5 };

template <bool BVH, bool TEX, bool PMJ, bool INST = false>
__global__ __launch_bounds__(256) void k_aov(const PtParams p_in, uint32_t spp, uint32_t aov, uint32_t remap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];  // BVH: traversal stacks; else: the staged scene tables
    PtParams staged = p_in;
    if (!BVH) stage_scene_tables<false, TEX>(p_in, lds_stack, staged);
    const PtParams& p = BVH ? p_in : staged;
    TraceCtx tc;
    tc.stack = lds_stack + threadIdx.x;
    tc.cnt = TraceCounters{0, 0, 0};
    const DScene& sc = p.sc;
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    uint32_t px = 0, py = 0;
    const bool in_frame = item < p.n_items && item_to_pixel(p, item, px, py);
    uint32_t n_closest = 0;
    if (in_frame) {
        const uint32_t pix = px + py * p.width;
        const size_t N = (size_t)p.width * p.height;
        Sampler smp;
        smp.pcg = p.states[pix];
        smp.dim = 0;
        vec3 acc = mk3(p.film[3 * (size_t)pix + 0], p.film[3 * (size_t)pix + 1], p.film[3 * (size_t)pix + 2]);
        float wsum = p.film[6 * N + pix];
        for (uint32_t s = 0; s < spp; s++) {
            sampler_start<PMJ>(p, smp);  // sampler.start(), aov.rs:86
            vec3 o, d;
            generate_ray<PMJ>(p, px, py, smp, o, d);
            Hit hit;
            n_closest++;
            bool found = INST ? trace_inst<false, TEX>(sc, o, d, 0.0f, 1e20f, kInvalid, kInvalid, hit, tc.stack, tc.cnt)
                              : (BVH ? trace_bvh<false, TEX>(sc, o, d, 0.0f, 1e20f, kInvalid, kInvalid, hit, tc.stack, tc.cnt)
                                     : trace_exhaustive<false, TEX>(sc, o, d, 0.0f, 1e20f, kInvalid, kInvalid, hit));
            vec3 c = mk3(0, 0, 0);
            if (found) {
                SurfacePoint si = surface_interaction_any<INST>(sc, hit.gid, mk2(hit.u, hit.v));
                auto remapped = [&](vec3 v) { return remap ? v * 0.5f + mk3(0.5f, 0.5f, 0.5f) : v; };
                if (aov == AOV_NG) {
                    c = remapped(si.ng);
                } else if (aov == AOV_TANGENT) {
                    c = remapped(si.frame.t);
                } else if (aov == AOV_BITANGENT) {
                    c = remapped(si.frame.s);
                } else {
                    DMaterial mat = sc.materials[si.material];
                    if (TEX) material_at(sc.tex, si.material, si.uv, mat);
                    ShadePoint sp;
                    shade_point_init(sp, mat, si.frame, si.ng, false);
                    if (aov == AOV_NS) c = remapped(shade_ns(sp, mat));
                    else if (aov == AOV_ALBEDO) c = shade_albedo_plus_emission(mat);
                    else c = mk3(1, 1, 1) * shade_roughness(sp, mat, sc.ggx_table, -d, next_1d<PMJ>(p, smp));
                }
            }
            // film.add_sample(p, color, swl, ray_w = 1), film.rs:196-229
            if (is_nan(c.x) || is_nan(c.y) || is_nan(c.z)) c = mk3(0, 0, 0);
            c = c * 1.0f;
            if (p.color & COLOR_REPR_ACES) c = cs_convert(c, true, false);  // Color::Rgb(v, the space of color_repr) -> the sRGB film (aov.rs:98-124, film.rs:218)
            acc = mk3(acc.x + c.x, acc.y + c.y, acc.z + c.z);
            wsum = wsum + 1.0f;
        }
        sampler_end_pass<PMJ>(p, smp);  // Drop of the sampler
        p.states[pix] = smp.pcg;
        p.film[3 * (size_t)pix + 0] = acc.x;
        p.film[3 * (size_t)pix + 1] = acc.y;
        p.film[3 * (size_t)pix + 2] = acc.z;
        p.film[6 * N + pix] = wsum;
    }
    if (p.counters != nullptr) {
        uint32_t a = wave_sum_u32(n_closest), nn = wave_sum_u32(tc.cnt.nodes), nt = wave_sum_u32(tc.cnt.tris), ov = wave_sum_u32(tc.cnt.overflow);
        if ((threadIdx.x & 63u) == 0) {
            if (a) {
                atomicAdd((unsigned long long*)&p.counters[0], (unsigned long long)a);
                atomicAdd((unsigned long long*)&p.counters[1], (unsigned long long)a);
            }
            if (nn) atomicAdd((unsigned long long*)&p.counters[4], (unsigned long long)nn);
            unsigned long long tt = BVH ? (unsigned long long)nt : (unsigned long long)a * p.sc.n_tris;
            if (tt) atomicAdd((unsigned long long*)&p.counters[5], tt);
            if (ov) atomicAdd((unsigned long long*)&p.counters[6], (unsigned long long)ov);
        }
    }
}

hipError_t launch_aov(const PtParams& p, uint32_t spp, uint32_t aov, uint32_t remap, hipStream_t stream) {
    uint32_t blocks = (p.n_items + 255u) / 256u;
    if (blocks == 0) return hipSuccess;
    const bool bvh = p.sc.bvh_nodes != nullptr, tex = p.sc.tex.nodes != nullptr;
    size_t lds;
    const PtParams q = with_tex_slots(p, bvh ? p.sc.bvh_stack_depth * 256 * 4 : p.stage_total, lds);
#define AKR_AOV(B, T)                                                                                                  \
    {                                                                                                                \
        if (p.sampler) hipLaunchKernelGGL((k_aov<B, T, true>), dim3(blocks), dim3(256), lds, stream, q, spp, aov, remap); \
        else hipLaunchKernelGGL((k_aov<B, T, false>), dim3(blocks), dim3(256), lds, stream, q, spp, aov, remap);      \
    }
#define AKR_AOV_INST(T)                                                                                                          \
    {                                                                                                                            \
        if (p.sampler) hipLaunchKernelGGL((k_aov<true, T, true, true>), dim3(blocks), dim3(256), lds, stream, q, spp, aov, remap); \
        else hipLaunchKernelGGL((k_aov<true, T, false, true>), dim3(blocks), dim3(256), lds, stream, q, spp, aov, remap);      \
    }
    if (p.sc.in2.on) { if (tex) AKR_AOV_INST(true) else AKR_AOV_INST(false) }  // meshes + instances (aov.rs:57-173 over the reference's two-level accel)
    else if (bvh) { if (tex) AKR_AOV(true, true) else AKR_AOV(true, false) }
    else { if (tex) AKR_AOV(false, true) else AKR_AOV(false, false) }
#undef AKR_AOV
#undef AKR_AOV_INST
    return hipGetLastError();
}

}  // namespace akr
