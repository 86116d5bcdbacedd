// pt_kernels.hip -- hand-written gfx950 kernels of the `pt` integrator. k_pt_pass: device/pt_pass.h.
#include <algorithm>
#include "device/pt_pass.h"
#include "pt_launch.h"

namespace akr {

// init_pcg32_buffer_with_seed's device half (sampler/mod.rs:152-158)
__global__ void k_init_pcg32(const uint64_t* __restrict__ seeds, Pcg32* __restrict__ states, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) states[i] = pcg_new_seq_offset(i, seeds[i]);
}

// Film resolve: copy_to_rgba_image with hdr = true (film.rs:120-148)
__global__ void k_film_resolve(const float* __restrict__ film, uint64_t n, float splat_scale, float* __restrict__ rgb) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float w = film[6 * n + i];
    float inv = w == 0.0f ? 1.0f : w;
#pragma unroll
    for (int c = 0; c < 3; c++) rgb[3 * i + c] = film[3 * i + c] / inv + film[3 * n + 3 * i + c] * splat_scale;
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
template <bool BVH, bool INST = false>
__global__ __launch_bounds__(256) void k_probe_intersect(PtParams p, uint32_t n, const float* __restrict__ rays, uint32_t* __restrict__ out, float* __restrict__ bary) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    TraceCtx tc;
    tc.stack = lds_stack + threadIdx.x;
    tc.cnt = TraceCounters{0, 0, 0};
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool act = i < n;
    const float* r = rays + 8 * (size_t)(act ? i : 0);
    Hit h;
    bool found = INST ? trace_inst<false, false>(p.sc, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), r[6], r[7], kInvalid, kInvalid, h, tc.stack, tc.cnt)
                      : trace<BVH, false>(p, tc, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), r[6], r[7], kInvalid, kInvalid, h);
    if (!act) return;
    uint32_t inst = 0, prim = 0;
    if (found) {
        inst = INST ? inst_of_gid(p.sc, h.gid) : f2u(p.sc.shade[(size_t)h.gid * SHADE_ROWS + 6].z);
        prim = h.gid - p.sc.inst_tri_offset[inst];
    }
    out[3 * i + 0] = found ? 1u : 0u;
    out[3 * i + 1] = inst;
    out[3 * i + 2] = prim;
    bary[2 * i + 0] = found ? h.u : 0.0f;
    bary[2 * i + 1] = found ? h.v : 0.0f;
}
template <bool INST>
__global__ void k_probe_si(PtParams p, uint32_t n, const uint32_t* __restrict__ inst_prim, const float* __restrict__ bary, float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t gid = p.sc.inst_tri_offset[inst_prim[2 * i]] + inst_prim[2 * i + 1];
    SurfacePoint s = surface_interaction_any<INST>(p.sc, gid, mk2(bary[2 * i], bary[2 * i + 1]));
    float* o = out + 19 * (size_t)i;
    o[0] = s.p.x; o[1] = s.p.y; o[2] = s.p.z;
    o[3] = s.ng.x; o[4] = s.ng.y; o[5] = s.ng.z;
    o[6] = s.frame.n.x; o[7] = s.frame.n.y; o[8] = s.frame.n.z;
    o[9] = s.frame.t.x; o[10] = s.frame.t.y; o[11] = s.frame.t.z;
    o[12] = s.frame.s.x; o[13] = s.frame.s.y; o[14] = s.frame.s.z;
    o[15] = s.uv.x;  // uv = interp(uv0, uv1, uv2)
    o[16] = s.uv.y;
    o[17] = s.prim_area;
    o[18] = (float)s.material;
}

// evaluated inputs (26 words = akr_material_desc) of `material` at n uv points
__global__ void k_probe_material(PtParams p, uint32_t material, uint32_t n, const float* __restrict__ uv, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DMaterial& m = p.sc.materials[material];
    MatInputs in = p.sc.tex.mat_inputs[material];
    if (m.flags & MF_TEXTURED) {
        eval_material_graph(p.sc.tex, m.tex_first_node, m.tex_n_nodes & kTexCountMask, mk2(uv[2 * i], uv[2 * i + 1]), in);
    }
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&in);
    for (uint32_t k = 0; k < 26; k++) out[26 * (size_t)i + k] = w[k];
}

// ---------------------------------------------------------------------------------------------------- launchers
hipError_t launch_probe_material(const PtParams& p, uint32_t material, uint32_t n, const float* uv, uint32_t* out, hipStream_t stream) {
    size_t lds;
    const PtParams q = with_tex_slots(p, 0, lds);
    hipLaunchKernelGGL(k_probe_material, dim3((n + 127) / 128), dim3(128), lds, stream, q, material, n, uv, out);
    return hipGetLastError();
}
hipError_t launch_init_pcg32(const uint64_t* seeds, void* states, uint64_t n, hipStream_t stream) {
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_init_pcg32, dim3(blocks), dim3(256), 0, stream, seeds, (Pcg32*)states, n);
    return hipGetLastError();
}
hipError_t launch_film_resolve(const float* film, uint64_t n, float splat_scale, float* rgb, hipStream_t stream) {
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_film_resolve, dim3(blocks), dim3(256), 0, stream, film, n, splat_scale, rgb);
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
    if (p.sc.in2.on)
        hipLaunchKernelGGL((k_probe_intersect<true, true>), dim3((n + 255) / 256), dim3(256), p.sc.bvh_stack_depth * 256 * 4, stream, p, n, rays, out, bary);
    else if (p.sc.bvh_nodes != nullptr)
        hipLaunchKernelGGL(k_probe_intersect<true>, dim3((n + 255) / 256), dim3(256), p.sc.bvh_stack_depth * 256 * 4, stream, p, n, rays, out, bary);
    else
        hipLaunchKernelGGL(k_probe_intersect<false>, dim3((n + 255) / 256), dim3(256), 0, stream, p, n, rays, out, bary);
    return hipGetLastError();
}
hipError_t launch_probe_si(const PtParams& p, uint32_t n, const uint32_t* inst_prim, const float* bary, float* out, hipStream_t stream) {
    if (p.sc.in2.on) hipLaunchKernelGGL(k_probe_si<true>, dim3((n + 255) / 256), dim3(256), 0, stream, p, n, inst_prim, bary, out);
    else hipLaunchKernelGGL(k_probe_si<false>, dim3((n + 255) / 256), dim3(256), 0, stream, p, n, inst_prim, bary, out);
    return hipGetLastError();
}

}  // namespace akr
