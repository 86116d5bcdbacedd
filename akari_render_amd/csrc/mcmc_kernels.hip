// mcmc_kernels.hip -- the `mcmc_opt` integrator (crates/akari_integrator/src/mcmc_opt.rs) on gfx950: primary-sample-space
// Metropolis light transport (Kelemen mutations, lazily mutated sample vectors). One lane per Markov chain; a chain's path is
// the plain bounce loop of the path tracer (device/dradiance.h with SM = false) reading its "random numbers" from the chain's
// primary-sample vector through McmcSampler. Primary samples are stored dimension-major ([dim][chain], 16-byte records), so
// the lanes of a wave -- consecutive chains on the same dimension -- touch consecutive records.
//
// Film splats are float atomics, as in the reference (film.rs:167-194): every chain's sequence of splat VALUES is
// deterministic (tests compare the chain states with the oracle bit for bit), the order in which different chains' splats
// reach a pixel is not, here or there.
#include "device/dradiance.h"

namespace akr {

struct McmcSampler {  // LazyMcmcSampler + Mutator, mcmc_opt.rs:61-129
    Pcg32 rng;        // the IndependentSampler underneath (from_pcg32: no start / drop bookkeeping)
    PssSample* samples;  // this chain's dimension 0; dimension i is samples[i * stride]
    uint32_t stride;
    uint32_t cur_dim, mcmc_dim;
    bool mutate, is_large_step, is_image_mutation;
    uint32_t last_large_iter, cur_iter;
    const McmcParams* mp;
};

AKR_D float erf_inv(float x) {  // util/mod.rs:149-186
    float cx = clamp_f(x, -0.99999f, 0.99999f);
    float w = -log_f((1.0f - cx) * (1.0f + cx));
    float q;
    if (w < 0.5f) {
        w -= 2.5f;
        q = 2.81022636e-08f; q = 3.43273939e-07f + q * w; q = -3.5233877e-06f + q * w; q = -4.39150654e-06f + q * w;
        q = 0.00021858087f + q * w; q = -0.00125372503f + q * w; q = -0.00417768164f + q * w; q = 0.246640727f + q * w; q = 1.50140941f + q * w;
    } else {
        w = __builtin_sqrtf(w) - 3.0f;
        q = -0.000200214257f; q = 0.000100950558f + q * w; q = 0.00134934322f + q * w; q = -0.00367342844f + q * w;
        q = 0.00573950773f + q * w; q = -0.0076224613f + q * w; q = 0.00943887047f + q * w; q = 1.00167406f + q * w; q = 2.83297682f + q * w;
    }
    return q * cx;
}
AKR_D float kelemen_mutate(float cur, float u) {  // KELEMEN_MUTATE, sampler/mcmc.rs:111-134; mutation sizes 1/1024 .. 1/64
    const float size_high = 1.0f / 64.0f, log_ratio = -2.7725887f;  // -(size_high / size_low).ln() = -ln 16
    const bool add = u < 0.5f;
    u = add ? u * 2.0f : (u - 0.5f) * 2.0f;
    const float dv = size_high * exp_f(log_ratio * u);
    if (add) {
        float n = cur + dv;
        return n > 1.0f ? n - 1.0f : n;
    }
    float n = cur - dv;
    return n < 0.0f ? n + 1.0f : n;
}
AKR_D PssSample mutate_one(McmcSampler& s, uint32_t i) {  // Mutator::mutate_one, mcmc_opt.rs:131-226
    const McmcParams& c = *s.mp;
    PssSample sp = s.samples[(size_t)i * s.stride];
    float u = pcg_next_1d(s.rng);
    if (sp.last_modified < s.last_large_iter) {
        sp.cur = pcg_next_1d(s.rng);
        sp.last_modified = s.last_large_iter;
    }
    sp.backup = sp.cur;
    sp.modified_backup = sp.last_modified;
    if (s.is_large_step) {
        sp.cur = u;
    } else {
        const bool has_img = c.image_mutation_size > 0.0f;
        const bool under_image = has_img && s.is_image_mutation;
        const bool should_mutate = !under_image || i < 2;
        const uint32_t target_iter = should_mutate ? s.cur_iter : s.cur_iter - 1;
        const uint32_t n_small = target_iter - sp.last_modified;
        if (c.exponential_mutation) {
            float x = sp.cur;
            for (uint32_t k = 0; k < n_small; k++) {
                float v = pcg_next_1d(s.rng);
                if (v < 1.0f - c.image_mutation_prob) x = kelemen_mutate(x, v / (1.0f - c.image_mutation_prob));
            }
            sp.cur = x;
        } else if (n_small > 0) {
            float dv = __builtin_sqrtf(2.0f) * erf_inv(2.0f * u - 1.0f);  // sample_gaussian(u), sampling.rs:46-48
            float n = sp.cur + (dv * c.small_sigma) * __builtin_sqrtf((1.0f - c.image_mutation_prob) * (float)n_small);
            n = n - __builtin_floorf(n);
            sp.cur = is_finite(n) ? n : 0.0f;
        }
        if (has_img && s.is_image_mutation && i < 2) {  // mutate_image_space_single, sampler/mcmc.rs:180-200
            float v = pcg_next_1d(s.rng);
            const bool add = v < 0.5f;
            v = add ? v * 2.0f : (v - 0.5f) * 2.0f;
            float off = v * c.image_mutation_size;
            off = add ? off : -off;
            float n = sp.cur + off / (i == 0 ? (float)c.width : (float)c.height);
            sp.cur = n - __builtin_floorf(n);
        }
    }
    sp.last_modified = s.cur_iter;
    s.samples[(size_t)i * s.stride] = sp;
    return sp;
}
AKR_D float draw_1d(const PtParams&, McmcSampler& s) {  // LazyMcmcSampler::next_1d, mcmc_opt.rs:88-103
    if (s.cur_dim < s.mcmc_dim) {
        float r = s.mutate ? mutate_one(s, s.cur_dim).cur : s.samples[(size_t)s.cur_dim * s.stride].cur;
        s.cur_dim += 1;
        return r;
    }
    s.cur_dim += 1;
    return pcg_next_1d(s.rng);
}

struct McmcEval {
    uint32_t px, py;
    vec3 l;
    float f;
};
// McmcOpt::evaluate, mcmc_opt.rs:253-305
template <bool BVH, bool TEX, bool INST>
AKR_D McmcEval mcmc_evaluate(const PtParams& p, TraceCtx& tc, McmcSampler& s, uint32_t& n_rays) {
    s.cur_dim = 0;  // sampler.start()
    const vec2 u = draw_2d(p, s);
    int32_t ix = (int32_t)(u.x * (float)p.width), iy = (int32_t)(u.y * (float)p.height);
    ix = ix < 0 ? 0 : (ix > (int32_t)p.width - 1 ? (int32_t)p.width - 1 : ix);
    iy = iy < 0 ? 0 : (iy > (int32_t)p.height - 1 ? (int32_t)p.height - 1 : iy);
    vec3 o, d;
    generate_ray_from(p, (uint32_t)ix, (uint32_t)iy, draw_2d(p, s), o, d);
    ShiftMapping sm{0.0f, 0.0f, false, true, false, 0.0f};
    ReconVertex vx;
    vx.type = VT_INVALID;
    vec3 base;
    vec3 l = radiance_sm<BVH, TEX, false, INST>(p, tc, o, d, s, sm, vx, base, n_rays);
    l = l * 1.0f;  // * ray_w
    McmcEval e;
    e.px = (uint32_t)ix;
    e.py = (uint32_t)iy;
    e.l = l;
    e.f = clamp_f(max3(l), 0.0f, 1e5f);  // scalar_contribution, mcmc_opt.rs:306-309
    return e;
}

// bootstrap (mcmc_opt.rs:331-349): the contribution of n_bootstrap independent paths
template <bool BVH, bool TEX, bool INST = false>
__global__ __launch_bounds__(256, 2) void k_mcmc_bootstrap(const PtParams p_in, const McmcParams m) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];  // BVH: traversal stacks; else: the staged scene tables
    PtParams staged = p_in;
    if (!BVH) stage_scene_tables<false, TEX>(p_in, lds_stack, staged);
    const PtParams& p = BVH ? p_in : staged;
    TraceCtx tc;
    tc.stack = lds_stack + threadIdx.x;
    tc.cnt = TraceCounters{0, 0, 0};
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m.n_bootstrap) return;
    McmcSampler s;
    s.rng = m.seeds[i];
    s.samples = nullptr; s.stride = 0; s.mcmc_dim = 0; s.mutate = false; s.is_large_step = false; s.is_image_mutation = false;
    s.last_large_iter = 0; s.cur_iter = 0; s.mp = &m;
    uint32_t n_rays = 0;
    m.fs[i] = mcmc_evaluate<BVH, TEX, INST>(p, tc, s, n_rays).f;
}
// the chains' initial states (mcmc_opt.rs:356-386): chain i starts from bootstrap path resampled[i]
template <bool BVH, bool TEX, bool INST = false>
__global__ __launch_bounds__(256, 2) void k_mcmc_init(const PtParams p_in, const McmcParams m) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];  // BVH: traversal stacks; else: the staged scene tables
    PtParams staged = p_in;
    if (!BVH) stage_scene_tables<false, TEX>(p_in, lds_stack, staged);
    const PtParams& p = BVH ? p_in : staged;
    TraceCtx tc;
    tc.stack = lds_stack + threadIdx.x;
    tc.cnt = TraceCounters{0, 0, 0};
    const uint32_t local = blockIdx.x * 256u + threadIdx.x;
    if (local >= m.chain_count) return;
    const uint32_t i = m.chain_begin + local;  // arrays are indexed by the chain's global id: a shard touches its own range only
    McmcSampler s;
    s.rng = m.seeds[m.resampled[i]];
    s.samples = m.pss + i; s.stride = m.n_chains; s.mcmc_dim = m.dim; s.mutate = false; s.is_large_step = false; s.is_image_mutation = false;
    s.last_large_iter = 0; s.cur_iter = 0; s.mp = &m;
    for (uint32_t j = 0; j < m.dim; j++) s.samples[(size_t)j * s.stride] = PssSample{pcg_next_1d(s.rng), 0.0f, 0u, 0u};
    uint32_t n_rays = 0;
    McmcEval e = mcmc_evaluate<BVH, TEX, INST>(p, tc, s, n_rays);
    m.cur_colors[i] = make_float4(e.l.x, e.l.y, e.l.z, 0.0f);
    m.states[i] = MarkovState{{e.px, e.py}, i, e.f, 0.0f, 0u, 0u, 0u, 0u, 0u};
}
// advance_chain + mutate_chain (mcmc_opt.rs:409-552)
template <bool BVH, bool TEX, bool INST = false>
__global__ __launch_bounds__(256, 2) void k_mcmc_advance(const PtParams p_in, const McmcParams m, uint32_t mutations_per_chain, float contribution) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];  // BVH: traversal stacks; else: the staged scene tables
    PtParams staged = p_in;
    if (!BVH) stage_scene_tables<false, TEX>(p_in, lds_stack, staged);
    const PtParams& p = BVH ? p_in : staged;
    TraceCtx tc;
    tc.stack = lds_stack + threadIdx.x;
    tc.cnt = TraceCounters{0, 0, 0};
    const uint32_t local = blockIdx.x * 256u + threadIdx.x, i = m.chain_begin + local;
    uint32_t n_rays = 0, n_paths = 0;
    if (local < m.chain_count) {
        const size_t N = (size_t)p.width * p.height;
        float* splat = m.film + 3 * N;
        McmcSampler s;
        s.rng = m.rngs[i];
        s.samples = m.pss + i; s.stride = m.n_chains; s.mcmc_dim = m.dim; s.mutate = true; s.mp = &m;
        MarkovState st = m.states[i];
        float4 cc = m.cur_colors[i];
        vec3 cur_color = mk3(cc.x, cc.y, cc.z);
        for (uint32_t it = 0; it < mutations_per_chain; it++) {
            if (st.cur_iter == 0xffffffffu - 1u) {  // the iteration counter is about to overflow: rebase it, mcmc_opt.rs:518-531
                for (uint32_t j = 0; j < m.dim; j++) {
                    PssSample sp = s.samples[(size_t)j * s.stride];
                    if (sp.last_modified < st.last_large_iter) sp.cur = pcg_next_1d(s.rng);
                    sp.last_modified = 0u;
                    s.samples[(size_t)j * s.stride] = sp;
                }
                st.cur_iter -= st.last_large_iter;
                st.last_large_iter = 0u;
            }
            st.cur_iter += 1;
            const float u = pcg_next_1d(s.rng);
            s.is_large_step = u < m.large_step_prob;
            s.is_image_mutation = pcg_next_1d(s.rng) < m.image_mutation_prob;
            s.last_large_iter = st.last_large_iter;
            s.cur_iter = st.cur_iter;
            const McmcEval e = mcmc_evaluate<BVH, TEX, INST>(p, tc, s, n_rays);
            n_paths++;
            const float proposal_f = e.f;
            if (s.is_large_step && st.b_cnt < 1024u * 1024u) {
                st.b += proposal_f;
                st.b_cnt += 1;
            }
            const float cur_f = st.cur_f;
            float accept = 0.0f;
            if (is_finite(proposal_f)) accept = (cur_f == 0.0f || !is_finite(cur_f)) ? 1.0f : clamp_f(proposal_f / cur_f, 0.0f, 1.0f);
            {  // the two expected-value splats, mcmc_opt.rs:463-474
                const vec3 a = splat_value(div_s(e.l, proposal_f), accept * contribution, (p.color & COLOR_REPR_ACES) != 0);
                float* d = splat + 3 * ((size_t)e.px + (size_t)e.py * p.width);
                unsafeAtomicAdd(d + 0, a.x); unsafeAtomicAdd(d + 1, a.y); unsafeAtomicAdd(d + 2, a.z);
                const vec3 b = splat_value(div_s(cur_color, cur_f), (1.0f - accept) * contribution, (p.color & COLOR_REPR_ACES) != 0);
                d = splat + 3 * ((size_t)st.cur_pixel[0] + (size_t)st.cur_pixel[1] * p.width);
                unsafeAtomicAdd(d + 0, b.x); unsafeAtomicAdd(d + 1, b.y); unsafeAtomicAdd(d + 2, b.z);
            }
            if (pcg_next_1d(s.rng) < accept) {
                st.cur_f = proposal_f;
                cur_color = e.l;
                st.cur_pixel[0] = e.px;
                st.cur_pixel[1] = e.py;
                if (!s.is_large_step) st.n_accepted += 1;
                else st.last_large_iter = st.cur_iter;
            } else {  // reject: restore the samples the proposal touched
                st.cur_iter -= 1;
                const uint32_t nd = s.cur_dim < m.dim ? s.cur_dim : m.dim;
                for (uint32_t j = 0; j < nd; j++) {
                    PssSample sp = s.samples[(size_t)j * s.stride];
                    sp.cur = sp.backup;
                    sp.last_modified = sp.modified_backup;
                    s.samples[(size_t)j * s.stride] = sp;
                }
            }
            if (!s.is_large_step) st.n_mutations += 1;
        }
        m.cur_colors[i] = make_float4(cur_color.x, cur_color.y, cur_color.z, 0.0f);
        m.rngs[i] = s.rng;
        m.states[i] = st;
    }
    if (p.counters != nullptr) {
        uint32_t a = wave_sum_u32(n_paths), r = wave_sum_u32(n_rays), nn = wave_sum_u32(tc.cnt.nodes), nt = wave_sum_u32(tc.cnt.tris),
                 ov = wave_sum_u32(tc.cnt.overflow);
        if ((threadIdx.x & 63u) == 0) {
            if (a) atomicAdd((unsigned long long*)&p.counters[0], (unsigned long long)a);
            if (r) atomicAdd((unsigned long long*)&p.counters[1], (unsigned long long)r);
            if (nn) atomicAdd((unsigned long long*)&p.counters[4], (unsigned long long)nn);
            unsigned long long tt = BVH ? (unsigned long long)nt : (unsigned long long)r * p.sc.n_tris;
            if (tt) atomicAdd((unsigned long long*)&p.counters[5], tt);
            if (ov) atomicAdd((unsigned long long*)&p.counters[6], (unsigned long long)ov);
        }
    }
}

#define AKR_MCMC_LAUNCH(KERNEL, COUNT, ...)                                                                         \
    {                                                                                                             \
        uint32_t blocks = ((COUNT) + 255u) / 256u;                                                                \
        if (blocks == 0) return hipSuccess;                                                                       \
        const bool bvh = p_in.sc.bvh_nodes != nullptr, tex = p_in.sc.tex.nodes != nullptr;                        \
        size_t lds;                                                                                               \
        const PtParams p = with_tex_slots(p_in, bvh ? p_in.sc.bvh_stack_depth * 256 * 4 : p_in.stage_total, lds);           \
        if (p_in.sc.in2.on) { /* meshes + instances (mcmc_opt.rs:686-746 over the reference's two-level accel) */   \
            if (tex) hipLaunchKernelGGL((KERNEL<true, true, true>), dim3(blocks), dim3(256), lds, stream, __VA_ARGS__);  \
            else hipLaunchKernelGGL((KERNEL<true, false, true>), dim3(blocks), dim3(256), lds, stream, __VA_ARGS__);     \
        } else if (bvh) {                                                                                         \
            if (tex) hipLaunchKernelGGL((KERNEL<true, true>), dim3(blocks), dim3(256), lds, stream, __VA_ARGS__);   \
            else hipLaunchKernelGGL((KERNEL<true, false>), dim3(blocks), dim3(256), lds, stream, __VA_ARGS__);      \
        } else {                                                                                                  \
            if (tex) hipLaunchKernelGGL((KERNEL<false, true>), dim3(blocks), dim3(256), lds, stream, __VA_ARGS__);  \
            else hipLaunchKernelGGL((KERNEL<false, false>), dim3(blocks), dim3(256), lds, stream, __VA_ARGS__);     \
        }                                                                                                         \
        return hipGetLastError();                                                                                 \
    }
hipError_t launch_mcmc_bootstrap(const PtParams& p_in, const McmcParams& m, hipStream_t stream) AKR_MCMC_LAUNCH(k_mcmc_bootstrap, m.n_bootstrap, p, m)
hipError_t launch_mcmc_init(const PtParams& p_in, const McmcParams& m, hipStream_t stream) AKR_MCMC_LAUNCH(k_mcmc_init, m.chain_count, p, m)
hipError_t launch_mcmc_advance(const PtParams& p_in, const McmcParams& m, uint32_t mutations_per_chain, float contribution, hipStream_t stream)
    AKR_MCMC_LAUNCH(k_mcmc_advance, m.chain_count, p, m, mutations_per_chain, contribution)
#undef AKR_MCMC_LAUNCH

}  // namespace akr
