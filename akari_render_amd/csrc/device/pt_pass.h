// pt_pass.h -- the body of k_pt_pass, the dominant kernel of the `pt` integrator, as a device function template: the library's
// precompiled instantiations (pt_kernels.hip) and the per-scene kernels compiled at run time (host/specialise.cpp) wrap the same code.
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
#pragma once
#include "dpath.h"

namespace akr {

// ----------------------------------------------------------------------------------------------------------
#ifndef AKR_PT_MIN_WAVES
#define AKR_PT_MIN_WAVES 4  // waves per SIMD the register allocator must leave room for (occupancy)
#endif
#ifndef AKR_PT_MIN_WAVES_BVH
#define AKR_PT_MIN_WAVES_BVH 4
#endif
#ifndef AKR_PT_MIN_WAVES_FD
#define AKR_PT_MIN_WAVES_FD 4  // force_diffuse specialisation of the exhaustive kernel
#endif
#ifndef AKR_PT_MIN_WAVES_TEX
#define AKR_PT_MIN_WAVES_TEX 3      // exhaustive full-graph kernel of a scene with texture-fed materials: 168 VGPRs hold the
                                    // re-folded material of the hit (264 -> 96 bytes of scratch), 642 -> 750 Msamples/s
#endif
#ifndef AKR_PT_MIN_WAVES_BVH_TEX
#define AKR_PT_MIN_WAVES_BVH_TEX 3  // BVH kernels of such a scene (399 -> 517)
#endif
// (AKR_WALK_*, AKR_PT_PARK_*, AKR_BVH_TILE, AKR_PT_STRAGGLERS*: kernels.h -- the host sizes the launch's LDS from them too)
#ifndef AKR_PT_MERGED_RAYS
#define AKR_PT_MERGED_RAYS 1  // BVH path: a lane starts its shadow ray the moment its closest-hit ray is done (one loop)
#endif
// ABSENT: lobes the scene cannot have (dbsdf.h AB_*). The precompiled kernels know 0 and AB_SIMPLE (PtParams.simple_scene: scenes without
// textures); a per-scene kernel gets the mask of its scene.
// INST: the scene is kept as meshes + instances (two-level traversal, dinst_trav.h); BVH kernels without STAGE / DEFER only.
template <bool BVH, bool FD, bool TEX, bool PMJ, bool STAGE, bool DEFER, uint32_t ABSENT = 0, bool INST = false>
AKR_D void pt_pass_body(const PtParams& p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];  // BVH: kBvhStackDepth x 256 words; else: staged tables
    TraceCtx tc;
    tc.stack = lds_stack + threadIdx.x;
    tc.cnt = TraceCounters{0, 0, 0};
    // STAGE: the kernel works on a parameter block whose table pointers aim at LDS copies (stage_scene_tables, dpath.h)
    PtParams staged = p;
    if (STAGE) stage_scene_tables<BVH, TEX, !BVH && !FD && TEX>(p, lds_stack, staged);
    const PtParams& q = STAGE ? staged : p;
    const DScene& sc = q.sc;
    constexpr bool TILE = BVH && !TEX && !INST && AKR_BVH_TILE != 0;
    constexpr uint32_t STRAG = BVH ? (INST ? AKR_PT_STRAGGLERS_INST : (TEX ? AKR_PT_STRAGGLERS_TEX : AKR_PT_STRAGGLERS)) : 0;
    const uint4* tile = (const uint4*)(lds_stack + p.tile_offset);
    if (TILE) {  // nodes 0 .. bvh_tile_nodes - 1
        uint32_t* l = lds_stack + p.tile_offset;
        const uint32_t* g = (const uint32_t*)p.sc.bvh_nodes;
        for (uint32_t i = threadIdx.x; i < p.sc.bvh_tile_nodes * kBvhNodeWords; i += 256u) l[i] = g[i];
        __syncthreads();
    }
    constexpr int WALK = FD ? AKR_WALK_FD : (TEX ? AKR_WALK_TEX : AKR_WALK_FULL);
    const float4* lds_recs = nullptr;
    if (!BVH && (WALK == 1 || WALK >= 3)) {  // the triangle records behind the staged tables (launch_pt_pass sizes the block)
        uint32_t* l = lds_stack + (p.stage_total >> 2);
        const uint32_t* g = (const uint32_t*)p.sc.woop;
        for (uint32_t i = threadIdx.x; i < (p.sc.n_tris + 2u) * 12u; i += 256u) l[i] = g[i];
        __syncthreads();
        lds_recs = (const float4*)l;
    }
    // Which 256 work items a workgroup takes. Workgroups are dealt to the chip's 8 XCDs round-robin (workgroup b runs on XCD b % 8)
    // and every XCD has its own 4 MiB L2: with the identity mapping each XCD sees every eighth 32x8-pixel strip of the whole frame,
    // so all eight L2s hold the same mix of the scene. BANDS gives XCD x the x-th contiguous eighth of the item space (items
    // enumerate the rank's tiles in row-major order: a horizontal band of the image), so that an L2 only has to hold the part of
    // the tree its band's rays walk. Only the assignment of pixels to workgroups changes: films are the same bit for bit.
    uint32_t vblock = blockIdx.x;
    if (BVH && AKR_PT_XCD_BANDS) {
        const uint32_t nb = gridDim.x, xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
        vblock = xcd * (nb >> 3) + (xcd < (nb & 7u) ? xcd : (nb & 7u)) + local;
    }
    const uint32_t item = vblock * 256u + threadIdx.x;
    uint32_t px = 0, py = 0;
    const bool in_frame = item < p.n_items && item_to_pixel(p, item, px, py);
    const uint32_t pix = px + py * p.width;
    uint32_t sx, sy;
    shifted_pixel(p, px, py, sx, sy);
    if (PMJ && p.bn_offset != 0) pmj_bluenoise_stage(p, px, py);  // before the first draw (path_regs_init generates the first camera ray)
    PathRegs r;
    path_regs_init<PMJ>(r, q, in_frame, pix, sx, sy);
    constexpr bool PARK = !FD && (TEX ? AKR_PT_PARK_TEX != 0 : (BVH ? AKR_PT_PARK_BVH != 0 : AKR_PT_PARK_FULL != 0));
    uint32_t* park = lds_stack + p.park_offset + threadIdx.x;
    if (PARK) {
        park_put(park, PK_PIX, pix);
        park_put(park, PK_SX, sx);
        park_put(park, PK_SY, sy);
    }

    uint32_t iteration = 0;
    while (__builtin_amdgcn_ballot_w64(r.active) != 0) {
        iteration++;
        if (r.active) {
            // intersection phase: next closest-hit ray + pending shadow ray
            Hit hit;
            bool found = false, occluded = false;
            if (!(STRAG > 0 && r.carry)) {
                r.c_closest += r.has_ray ? 1u : 0u;
                r.c_shadow += r.has_shadow ? 1u : 0u;
            }
            if (BVH && INST) {
                // meshes + instances: the two-level traversal (dinst_trav.h), both rays in one loop, stragglers carried over
                static_assert(kCarrySlotsInstanced == kCarrySlotsInst, "LDS plan and traversal disagree");
                trace_pair_inst<TEX, STRAG>(sc, r.has_ray, r.ro, r.rd, r.ray_ex0, r.has_shadow, r.s_o, r.s_d, r.s_tmax, r.s_ex0, r.s_ex1, r.carry, hit, found, occluded, tc.stack,
                                            lds_stack + p.carry_offset + threadIdx.x, tc.cnt);
            } else if (BVH && AKR_PT_MERGED_RAYS && STRAG > 0) {
                // The merged loop below ends when the wave's LONGEST pair of rays is done: on the 10 M-triangle hall 40 % of its
                // lane-steps do work, the rest is lanes waiting for the tail of the ray-length distribution. Here the phase ends
                // when at most 1/n of the lanes that entered it are still tracing. Those lanes keep their traversal -- position
                // in the tree and best hit so far in a column of LDS, the stack where it is -- skip this iteration's shading and
                // continue in the next phase, while the others shade and start their next rays. Per lane only the iteration in
                // which a vertex is shaded changes (as with DEFER): films and sampler states are the same bit for bit.
                uint32_t* cy = lds_stack + p.carry_offset + threadIdx.x;
                Trav s;
                uint32_t phase;
                hit.t = 1e20f; hit.u = 0.0f; hit.v = 0.0f; hit.gid = kInvalid;
                if (!r.carry) {
                    phase = r.has_ray ? 0u : (r.has_shadow ? 1u : 2u);
                    if (phase == 0) trav_begin(s, r.ro, r.rd, 0.0f, 1e20f, r.ray_ex0, kInvalid);
                    else trav_begin(s, r.s_o, r.s_d, 0.0f, phase == 1 ? r.s_tmax : -1.0f, r.s_ex0, r.s_ex1);
                } else {
                    phase = cy[8 * 256];
                    if (phase == 0) trav_begin(s, r.ro, r.rd, 0.0f, 1e20f, r.ray_ex0, kInvalid);
                    else {
                        trav_begin(s, r.s_o, r.s_d, 0.0f, r.s_tmax, r.s_ex0, r.s_ex1);
                        hit.t = u2f(cy[9 * 256]); hit.u = u2f(cy[10 * 256]); hit.v = u2f(cy[11 * 256]); hit.gid = cy[12 * 256];
                        found = hit.gid != kInvalid;
                    }
                    s.best_t = u2f(cy[0]); s.best_u = u2f(cy[1 * 256]); s.best_v = u2f(cy[2 * 256]); s.best = cy[3 * 256];
                    s.G = cy[4 * 256]; s.T = cy[5 * 256]; s.tbase = cy[6 * 256]; s.sp = cy[7 * 256];
                    s.active = true;
                }
                const uint32_t n_in = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(phase != 2u));
                const uint32_t n_leave = n_in / (STRAG > 0 ? STRAG : 1u);
                while (true) {
                    const uint32_t n_now = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(phase != 2u));
                    if (n_now <= n_leave) break;  // n_leave < n_in: at least one lane of the phase finishes
                    if (phase != 2u) {
                        if (s.active) trav_step<2, TEX, TILE>(sc, s, tc.stack, tc.cnt, phase == 1u, tile);
                        if (!s.active) {
                            if (phase == 0u) {
                                found = s.best != kInvalid;
                                hit.t = s.best_t; hit.u = s.best_u; hit.v = s.best_v; hit.gid = s.best;
                                phase = r.has_shadow ? 1u : 2u;
                                if (r.has_shadow) trav_begin(s, r.s_o, r.s_d, 0.0f, r.s_tmax, r.s_ex0, r.s_ex1);
                            } else {
                                occluded = s.best != kInvalid;
                                phase = 2u;
                            }
                        }
                    }
                }
                r.carry = phase != 2u;
                if (r.carry) {
                    cy[0] = f2u(s.best_t); cy[1 * 256] = f2u(s.best_u); cy[2 * 256] = f2u(s.best_v); cy[3 * 256] = s.best;
                    cy[4 * 256] = s.G; cy[5 * 256] = s.T; cy[6 * 256] = s.tbase; cy[7 * 256] = s.sp;
                    cy[8 * 256] = phase;
                    if (phase == 1u) { cy[9 * 256] = f2u(hit.t); cy[10 * 256] = f2u(hit.u); cy[11 * 256] = f2u(hit.v); cy[12 * 256] = hit.gid; }
                }
            } else if (BVH && AKR_PT_MERGED_RAYS) {
                // Both rays of the iteration through ONE traversal loop: a lane whose closest-hit ray is done goes straight on
                // to its shadow ray, so the wave pays for its longest PAIR of rays instead of its longest closest-hit ray plus
                // its longest shadow ray (rays of a wave differ in length by an order of magnitude; the loop is the same code for
                // both kinds: the visiting order comes from the node layout, not from sorting).
                Trav s;
                uint32_t phase = r.has_ray ? 0u : (r.has_shadow ? 1u : 2u);  // 0: closest-hit ray in flight, 1: shadow ray, 2: done
                if (phase == 0) trav_begin(s, r.ro, r.rd, 0.0f, 1e20f, r.ray_ex0, kInvalid);
                else trav_begin(s, r.s_o, r.s_d, 0.0f, phase == 1 ? r.s_tmax : -1.0f, r.s_ex0, r.s_ex1);
                hit.t = 1e20f; hit.u = 0.0f; hit.v = 0.0f; hit.gid = kInvalid;
                while (phase != 2u) {
                    if (s.active) trav_step<2, TEX, TILE>(sc, s, tc.stack, tc.cnt, phase == 1u, tile);
                    if (!s.active) {
                        if (phase == 0u) {
                            found = s.best != kInvalid;
                            hit.t = s.best_t; hit.u = s.best_u; hit.v = s.best_v; hit.gid = s.best;
                            phase = r.has_shadow ? 1u : 2u;
                            if (r.has_shadow) trav_begin(s, r.s_o, r.s_d, 0.0f, r.s_tmax, r.s_ex0, r.s_ex1);
                        } else {
                            occluded = s.best != kInvalid;
                            phase = 2u;
                        }
                    }
                }
            } else if (BVH) {
                if (r.has_ray) found = trace_bvh<false, TEX>(sc, r.ro, r.rd, 0.0f, 1e20f, r.ray_ex0, kInvalid, hit, tc.stack, tc.cnt);
                if (r.has_shadow) {
                    Hit sh;
                    occluded = trace_bvh<true, TEX>(sc, r.s_o, r.s_d, 0.0f, r.s_tmax, r.s_ex0, r.s_ex1, sh, tc.stack, tc.cnt);
                }
            } else {
                trace_pair_exhaustive<TEX, FD || (!TEX && AKR_WALK_FULL_UNROLL != 0), WALK>(sc, r.ro, r.rd, r.has_ray ? 1e20f : -1.0f, r.ray_ex0, r.s_o, r.s_d, r.has_shadow ? r.s_tmax : -1.0f,
                                            r.s_ex0, r.s_ex1, hit, found, occluded, lds_recs);
            }
            if (DEFER) {
                // A scene with one metal among diffuse surfaces: every wave carries a few lanes on the metal at every
                // iteration, so every iteration pays for the conductor lobe (GGX + complex Fresnel, the dearest code of the
                // material system) with a handful of lanes. Hits on a material with that lobe are therefore shaded on EVEN
                // iterations only: a lane that finds one on an odd iteration keeps the hit and sits the next intersection
                // phase out (the walk is wave-uniform: an idle lane costs nothing), and on odd iterations no lane enters
                // that code at all. Per lane nothing changes but the iteration a vertex is shaded in.
                // The BVH kernels of scenes with textures do the same (round 4), for the conductor lobe and for materials whose
                // shader graph has to be evaluated at the hit: a lane that waits has no ray in the next traversal phase either.
                if (r.deferred) {
                    hit.gid = r.d_gid; hit.u = r.d_u; hit.v = r.d_v; hit.t = 0.0f;
                    found = true;
                    r.has_ray = true;
                    r.deferred = false;
                } else if (r.has_ray && found && (iteration & q.defer_metal)) {
                    const uint32_t mat = f2u(sc.shade[(size_t)hit.gid * SHADE_ROWS + 6].y);
                    if (sc.materials[mat].flags & q.defer_flags) {  // MF_EVAL_METAL and / or MF_TEXTURED, the host's choice (api.cpp fill_params)
                        r.d_gid = hit.gid; r.d_u = hit.u; r.d_v = hit.v;
                        r.deferred = true;
                        r.has_ray = false;  // path_step resolves the shadow ray and finishes the previous sample, no more
                    }
                }
            }
            if (STRAG > 0 && r.carry) {
                // still tracing: nothing to resolve or shade yet
            } else if (PARK) path_step<FD ? 1 : 0, TEX, PMJ, DEFER ? 1 : 2, ABSENT, INST>(q, r, hit, found, occluded, 0, 0, 0, park);
            else path_step<FD ? 1 : 0, TEX, PMJ, 0, ABSENT, INST>(q, r, hit, found, occluded, pix, sx, sy);
        }
    }
    flush_counters(p, r, tc.cnt, BVH);
}

// waves per SIMD the register allocator leaves room for, by instantiation (the second argument of __launch_bounds__)
constexpr int pt_pass_min_waves(bool bvh, bool fd, bool tex) {
    return bvh ? (tex ? AKR_PT_MIN_WAVES_BVH_TEX : AKR_PT_MIN_WAVES_BVH) : (fd ? AKR_PT_MIN_WAVES_FD : (tex ? AKR_PT_MIN_WAVES_TEX : AKR_PT_MIN_WAVES));
}

}  // namespace akr
