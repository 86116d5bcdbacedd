// gpt_kernels.hip -- the `gpt` integrator (crates/akari_integrator/src/gpt.rs) on gfx950: gradient-domain path tracing with
// the reconnection shift mapping of PathTracerBase::run_pt_hybrid_shift_mapping (crates/akari_integrator/src/pt.rs:329-900,
// the `Some(sm)` branches the plain path tracer never takes). One lane per pixel per sample: the base path, then the four
// offset paths through the neighbouring pixels on the same random numbers (gpt.rs:144-351). Shares camera, sampler,
// intersectors, hit reconstruction, light sampling and material records with the path tracer (device/*.h).
//
// The reference splats the five contributions with float atomics into a scratch film and lets a second kernel fold the
// scratch film into the accumulators (gpt.rs:424-461); the order in which a pixel's own and its neighbours' splats arrive is
// whatever the hardware does. Here k_gpt_sample writes what a pixel splats onto itself (`own`) and what it splats for each
// neighbour (`shifted[i]`) to per-pixel slots and k_gpt_update gathers them in a fixed order -- no atomics, reproducible.
#include "device/dradiance.h"

namespace akr {

enum : uint32_t { RECON_NONE = 0, RECON_UNIFORM = 1, RECON_WEIGHTED = 2 };
AKR_D uint32_t gpt_reflect(int32_t x, uint32_t r) {  // gpt.rs:131-139
    return x < 0 ? (uint32_t)(-x) : ((uint32_t)x >= r ? r - ((uint32_t)x - r) - 1u : (uint32_t)x);
}
AKR_D void gpt_shifted(const GptParams& g, uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t i, uint32_t& sx, uint32_t& sy) {
    const int32_t ox = i == 0 ? 1 : (i == 2 ? -1 : 0), oy = i == 1 ? 1 : (i == 3 ? -1 : 0);
    sx = gpt_reflect((int32_t)x + ox * (int32_t)g.stride, W);
    sy = gpt_reflect((int32_t)y + oy * (int32_t)g.stride, H);
}
// render_one_spp, gpt.rs:144-351
// Register allocation aims at 2 waves per SIMD (230 VGPRs, no spills). Measured on the 1080p cbox, Mpaths/s: 1 wave 425,
// 2 waves 810, 4 waves (128 VGPRs, 330 spilled) 630. Running a lane's five paths through one flattened loop (a finished
// path starts the next one while the neighbours still bounce) was slower than the nested loops: 716 at 2 waves.
template <bool BVH, bool TEX, bool INST = false>
__global__ __launch_bounds__(256, 2) void k_gpt_sample(const PtParams p_in, const GptParams g) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];  // BVH: traversal stacks; else: the staged scene tables
    PtParams staged = p_in;
    if (!BVH) stage_scene_tables<false, TEX>(p_in, lds_stack, staged);
    const PtParams& p = BVH ? p_in : staged;
    TraceCtx tc;
    tc.stack = lds_stack + threadIdx.x;
    tc.cnt = TraceCounters{0, 0, 0};
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    uint32_t px = 0, py = 0;
    bool in_frame;
    if (g.item_pixels != nullptr) {  // sharded: the host's list of the pixels this rank has to sample (own tiles + halo)
        in_frame = item < p.n_items;
        if (in_frame) {
            const uint32_t q = g.item_pixels[item];
            py = q / p.width;
            px = q - py * p.width;
        }
    } else {
        in_frame = item < p.n_items && item_to_pixel(p, item, px, py);
    }
    uint32_t n_rays = 0;
    if (in_frame) {
        const uint32_t pix = px + py * p.width;
        const Pcg32 backup = p.states[pix];  // sampler_backup = sampler_creator.create(px)
        ReconVertex vx;
        vx.type = VT_INVALID;
        vx.depth = 0;
        vx.indirect = mk3(0, 0, 0);
        ShiftMapping sm{0.03f, 0.2f, g.reconnect != 0, true, false, 0.0f};
        vec3 l0 = mk3(0, 0, 0), rec0 = mk3(0, 0, 0), own = mk3(0, 0, 0);
        for (uint32_t k = 0; k < 5; k++) {  // trace(is_primary, pixel, shift_mapping), gpt.rs:153-203
            uint32_t qx = px, qy = py;
            if (k > 0) gpt_shifted(g, p.width, p.height, px, py, k - 1, qx, qy);
            Sampler smp;
            smp.pcg = backup;  // sampler_backup.clone_box()
            smp.dim = 0;
            sampler_start<false>(p, smp);
            vec3 ro, rd;
            generate_ray<false>(p, qx, qy, smp, ro, rd);
            sm.is_base = k == 0;
            if (k > 0) {
                sm.success = false;
                sm.jacobian = 0.0f;
            }
            vec3 base;
            const vec3 rad = radiance_sm<BVH, TEX, true, INST>(p, tc, ro, rd, smp, sm, vx, base, n_rays);
            vec3 l = rad, rec = mk3(0, 0, 0);
            float jac = 1.0f;
            bool ok = false;
            if (sm.enabled) {
                l = g.separate_weights ? base : rad;
                jac = sm.jacobian;
                ok = sm.success;
                rec = rad - l;
            }
            if (k == 0) {
                l0 = l;
                rec0 = rec;
                if (g.reconstruction != RECON_NONE) own = splat_value(l0 + rec0, 1.0f, (p.color & COLOR_REPR_ACES) != 0);
                continue;
            }
            vec3 out;
            if (g.reconstruction == RECON_NONE) {  // gpt.rs:275-307
                const float wp = ok ? 1.0f / (1.0f + jac) : 1.0f, ws = ok ? 1.0f / (1.0f + jac) : 0.0f;
                vec3 a, b;
                if (g.separate_weights) {
                    a = l0 * 0.5f + rec0 * wp;
                    b = l * 0.5f + (rec * ws) * jac;
                } else {
                    a = l0 * wp;
                    b = (l * ws) * jac;
                }
                own = own + splat_value(a, 1.0f, (p.color & COLOR_REPR_ACES) != 0);
                out = splat_value(b, 1.0f, (p.color & COLOR_REPR_ACES) != 0);
            } else {  // gpt.rs:308-346
                vec3 grad;
                if (sm.enabled) {
                    if (g.separate_weights) {
                        vec3 gr = ok ? div_s(rec * jac - rec0, 1.0f + jac) : mk3(0, 0, 0) - rec0;
                        grad = (l - l0) * 0.5f + gr;
                    } else {
                        grad = ok ? div_s(l * jac - l0, 1.0f + jac) : mk3(0, 0, 0) - l0;
                    }
                } else {
                    grad = (l - l0) * 0.5f;
                }
                out = splat_value(grad, k <= 2 ? 1.0f : -1.0f, (p.color & COLOR_REPR_ACES) != 0);
            }
            float* dst = g.shifted[k - 1] + 3 * (size_t)pix;
            dst[0] = out.x; dst[1] = out.y; dst[2] = out.z;
        }
        g.own[3 * (size_t)pix + 0] = own.x;
        g.own[3 * (size_t)pix + 1] = own.y;
        g.own[3 * (size_t)pix + 2] = own.z;
        Sampler b;  // sampler_backup.start(); its Drop stores the state with dim = 0
        b.pcg = backup;
        b.dim = 0;
        sampler_start<false>(p, b);
        p.states[pix] = b.pcg;
    }
    if (p.counters != nullptr) {
        uint32_t a = wave_sum_u32(in_frame ? 5u : 0u), r = wave_sum_u32(n_rays), nn = wave_sum_u32(tc.cnt.nodes), nt = wave_sum_u32(tc.cnt.tris),
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

// The pixels whose neighbour along one axis (offset o, size r) is coordinate cp: the inverse of gpt_reflect(c + o), in
// the fixed order [unreflected, mirrored at 0, mirrored at r].
AKR_D int gpt_sources(int32_t cp, int32_t o, uint32_t r, uint32_t out[3]) {
    int n = 0;
    const int64_t cand[3] = {(int64_t)cp - o, -(int64_t)cp - o, 2 * (int64_t)r - 1 - cp - o};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int64_t c = cand[k];
        if (c < 0 || c >= (int64_t)r) continue;
        const int64_t q = c + o;
        const int cls = q < 0 ? 1 : (q >= (int64_t)r ? 2 : 0);
        if (cls == k) out[n++] = (uint32_t)c;
    }
    return n;
}

// update_kernel, gpt.rs:424-461: fold one sample's splats into the film (reconstruction none: film.splat += v / 4) or into
// the primal / gradient sums and sums of squares.
AKR_D bool gpt_owned(const GptParams& g, uint32_t x, uint32_t y) {
    if (g.shard_count <= 1) return true;
    return tile_owner(x / g.tile_w, y / g.tile_h, g.shard_count) == g.shard_rank;
}
__global__ __launch_bounds__(256) void k_gpt_update(const GptParams g, uint32_t W, uint32_t H, float* __restrict__ film) {
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    if (!gpt_owned(g, x, y)) return;  // another rank's pixel: its film / accumulator entries stay zero here
    const size_t N = (size_t)W * H, q = x + (size_t)y * W, gq = x + (size_t)y * (W + 1);
    for (int c = 0; c < 3; c++) {
        if (g.reconstruction == RECON_NONE) {
            float v = 0.0f;
            v += g.own[3 * q + c];
            for (int i = 0; i < 4; i++) {
                const int32_t ox = i == 0 ? 1 : (i == 2 ? -1 : 0), oy = i == 1 ? 1 : (i == 3 ? -1 : 0);
                uint32_t src[3];
                if (ox) {
                    int n = gpt_sources((int32_t)x, ox * (int32_t)g.stride, W, src);
                    for (int k = 0; k < n; k++) v += g.shifted[i][3 * (src[k] + (size_t)y * W) + c];
                } else {
                    int n = gpt_sources((int32_t)y, oy * (int32_t)g.stride, H, src);
                    for (int k = 0; k < n; k++) v += g.shifted[i][3 * (x + (size_t)src[k] * W) + c];
                }
            }
            film[3 * N + 3 * q + c] += v * 0.25f;
        } else {
            float v = 0.0f, gx = 0.0f, gy = 0.0f;
            v += g.own[3 * q + c];
            if (x >= 1) gx += g.shifted[0][3 * (q - 1) + c];
            gx += g.shifted[2][3 * q + c];
            if (y >= 1) gy += g.shifted[1][3 * (q - W) + c];
            gy += g.shifted[3][3 * q + c];
            g.acc_p[3 * q + c] += v;
            g.acc_gx[3 * gq + c] += gx;
            g.acc_gy[3 * gq + c] += gy;
            g.sqr_p[3 * q + c] += v * v;
            g.sqr_gx[3 * gq + c] += gx * gx;
            g.sqr_gy[3 * gq + c] += gy * gy;
        }
    }
}

// recon_old = acc.primal / spp (gpt.rs:498-511)
__global__ void k_gpt_recon_init(const float* __restrict__ acc_p, float* __restrict__ old, uint64_t n3, float spp) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) old[i] = acc_p[i] / spp;
}
// one Jacobi sweep of the reconstruction (gpt.rs:525-602): old -> cur (the film's splat channels)
__global__ __launch_bounds__(256) void k_gpt_recon(const GptParams g, uint32_t W, uint32_t H, const float* __restrict__ old, float* __restrict__ cur,
                                                   float scaling, float spp) {
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t q = x + (size_t)y * W;
    for (int c = 0; c < 3; c++) {
        const float primal = old[3 * q + c];
        const float primal2 = g.sqr_p[3 * q + c] / spp;
        const float primal_var = max_f(primal2 - sqr(g.acc_p[3 * q + c] / spp), 1e-6f) / spp;
        const float pw = g.reconstruction == RECON_UNIFORM ? 1.0f : 1.0f / (primal_var * scaling);
        float v = 0.0f, sum_w = 0.0f;
        v += primal * pw;
        sum_w += pw;
        for (uint32_t i = 0; i < 4; i++) {
            const bool is_x = (i & 1u) == 0;
            const float sign = i < 2 ? 1.0f : -1.0f;
            const uint32_t gx_ = x + (i == 0 ? 1u : 0u), gy_ = y + (i == 1 ? 1u : 0u);
            uint32_t sx, sy;
            gpt_shifted(g, W, H, x, y, i, sx, sy);
            const size_t gi = 3 * (gx_ + (size_t)gy_ * (W + 1)) + c;
            const float grad = (is_x ? g.acc_gx[gi] : g.acc_gy[gi]) / spp;
            const float grad2 = (is_x ? g.sqr_gx[gi] : g.sqr_gy[gi]) / spp;
            const float grad_var = max_f((grad2 - sqr(grad)) / spp, 1e-6f);
            const float nb = old[3 * (sx + (size_t)sy * W) + c];
            const float var = primal_var + grad_var;
            const float w = g.reconstruction == RECON_UNIFORM ? 1.0f : 1.0f / var;
            v += (nb - sign * grad) * w;
            sum_w += w;
        }
        cur[3 * q + c] = v / sum_w;
    }
}

hipError_t launch_gpt_sample(const PtParams& p, const GptParams& g, hipStream_t stream) {
    uint32_t blocks = (p.n_items + 255u) / 256u;
    if (blocks == 0) return hipSuccess;
    const bool bvh = p.sc.bvh_nodes != nullptr, tex = p.sc.tex.nodes != nullptr;
    size_t lds;
    const PtParams q = with_tex_slots(p, bvh ? p.sc.bvh_stack_depth * 256 * 4 : p.stage_total, lds);
    if (p.sc.in2.on) {  // meshes + instances (gpt.rs:381-640 traces through the same two-level accel the path tracer does)
        if (tex) hipLaunchKernelGGL((k_gpt_sample<true, true, true>), dim3(blocks), dim3(256), lds, stream, q, g);
        else hipLaunchKernelGGL((k_gpt_sample<true, false, true>), dim3(blocks), dim3(256), lds, stream, q, g);
    } else if (bvh) {
        if (tex) hipLaunchKernelGGL((k_gpt_sample<true, true>), dim3(blocks), dim3(256), lds, stream, q, g);
        else hipLaunchKernelGGL((k_gpt_sample<true, false>), dim3(blocks), dim3(256), lds, stream, q, g);
    } else {
        if (tex) hipLaunchKernelGGL((k_gpt_sample<false, true>), dim3(blocks), dim3(256), lds, stream, q, g);
        else hipLaunchKernelGGL((k_gpt_sample<false, false>), dim3(blocks), dim3(256), lds, stream, q, g);
    }
    return hipGetLastError();
}
hipError_t launch_gpt_update(const GptParams& g, uint32_t W, uint32_t H, float* film, hipStream_t stream) {
    hipLaunchKernelGGL(k_gpt_update, dim3((W + 63u) / 64u, (H + 3u) / 4u), dim3(256), 0, stream, g, W, H, film);
    return hipGetLastError();
}
hipError_t launch_gpt_recon_init(const GptParams& g, uint32_t W, uint32_t H, float* old, float spp, hipStream_t stream) {
    const uint64_t n3 = 3ull * W * H;
    hipLaunchKernelGGL(k_gpt_recon_init, dim3((uint32_t)((n3 + 255) / 256)), dim3(256), 0, stream, g.acc_p, old, n3, spp);
    return hipGetLastError();
}
hipError_t launch_gpt_recon(const GptParams& g, uint32_t W, uint32_t H, const float* old, float* cur, float scaling, float spp, hipStream_t stream) {
    hipLaunchKernelGGL(k_gpt_recon, dim3((W + 63u) / 64u, (H + 3u) / 4u), dim3(256), 0, stream, g, W, H, old, cur, scaling, spp);
    return hipGetLastError();
}

}  // namespace akr
