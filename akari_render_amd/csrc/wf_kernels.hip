// wf_kernels.hip -- wavefront schedule of the same path tracer, for scenes whose intersection cost is dominated by
// BVH traversal (BASELINE configs[3], 10 M triangles).
//
// The megakernel (pt_kernels.hip) keeps a whole path in one lane's registers; with a BVH that means ~120 live
// registers around a divergent traversal loop (4 waves/SIMD, 26 % VALU lane utilisation, 69 % of wave cycles waiting
// on memory on the 10 M-triangle hall). Here the two halves are separate kernels and the path state streams through
// HBM between them (structure-of-arrays, 16-byte records, one slot per pixel):
//   k_wf_trace  persistent waves pull ray ids from a queue: when a lane's ray ends it takes the next id immediately
//               (wave64 __ballot + popcount prefix + one atomicAdd per wave) instead of idling until the slowest
//               lane of its wave is done; only ray + traversal registers are live, so 7-8 waves/SIMD hide latency;
//   k_wf_shade  one thread per path slot: path_step() of device/dpath.h (identical arithmetic to the megakernel),
//               then stream compaction of the rays it produced into the next queue (ballot/prefix-sum again).
// The reference's author sketched the same decomposition in crates/akari_integrator/src/wfpt.rs:59-225,315-494
// (PathState SoA, KernelWorkQueue, raygen / intersect / shade / test_shadow); that file never runs there.
#include <algorithm>
#include "device/dpath.h"

namespace akr {

enum : uint32_t {
    WF_ACTIVE = 1u, WF_HAS_RAY = 2u, WF_HAS_SHADOW = 4u, WF_S_ADD = 8u, WF_S_DEPTH1 = 16u, WF_FINALIZE = 32u, WF_LANE_DONE = 64u
};

AKR_D void wf_store(const WfBuffers& wf, uint32_t slot, const PathRegs& r) {
    wf.ray_o[slot] = make_float4(r.ro.x, r.ro.y, r.ro.z, u2f(r.ray_ex0));
    wf.ray_d[slot] = make_float4(r.rd.x, r.rd.y, r.rd.z, 0.0f);
    wf.sh_o[slot] = make_float4(r.s_o.x, r.s_o.y, r.s_o.z, u2f(r.s_ex0));
    wf.sh_d[slot] = make_float4(r.s_d.x, r.s_d.y, r.s_d.z, r.s_tmax);
    wf.sh_c[slot] = make_float4(r.s_contrib.x, r.s_contrib.y, r.s_contrib.z, u2f(r.s_ex1));
    wf.beta[slot] = make_float4(r.beta.x, r.beta.y, r.beta.z, r.prev_bsdf_pdf);
    wf.rad[slot] = make_float4(r.radiance.x, r.radiance.y, r.radiance.z, u2f(r.depth));
    uint32_t fl = (r.active ? WF_ACTIVE : 0u) | (r.has_ray ? WF_HAS_RAY : 0u) | (r.has_shadow ? WF_HAS_SHADOW : 0u) |
                  (r.s_add ? WF_S_ADD : 0u) | (r.s_depth1 ? WF_S_DEPTH1 : 0u) | (r.finalize ? WF_FINALIZE : 0u) |
                  (r.lane_done ? WF_LANE_DONE : 0u);
    wf.base[slot] = make_float4(r.base.x, r.base.y, r.base.z, u2f(fl));
    wf.film[slot] = make_float4(r.film_rgb.x, r.film_rgb.y, r.film_rgb.z, r.film_w);
    wf.rng[slot] = make_uint4((uint32_t)r.smp.pcg.state, (uint32_t)(r.smp.pcg.state >> 32), r.smp.dim, r.samples_done);
    wf.misc[slot] = make_uint4(r.pass_idx, r.cur_spp, (uint32_t)r.smp.pcg.inc, (uint32_t)(r.smp.pcg.inc >> 32));
}
AKR_D void wf_load(const WfBuffers& wf, uint32_t slot, PathRegs& r) {
    float4 a = wf.ray_o[slot], b = wf.ray_d[slot], c = wf.sh_o[slot], d = wf.sh_d[slot], e = wf.sh_c[slot];
    float4 f = wf.beta[slot], g = wf.rad[slot], h = wf.base[slot], fm = wf.film[slot];
    uint4 rg = wf.rng[slot], ms = wf.misc[slot];
    r.ro = xyz(a); r.ray_ex0 = f2u(a.w);
    r.rd = xyz(b);
    r.s_o = xyz(c); r.s_ex0 = f2u(c.w);
    r.s_d = xyz(d); r.s_tmax = d.w;
    r.s_contrib = xyz(e); r.s_ex1 = f2u(e.w);
    r.beta = xyz(f); r.prev_bsdf_pdf = f.w;
    r.radiance = xyz(g); r.depth = f2u(g.w);
    r.base = xyz(h);
    uint32_t fl = f2u(h.w);
    r.active = fl & WF_ACTIVE; r.has_ray = fl & WF_HAS_RAY; r.has_shadow = fl & WF_HAS_SHADOW; r.s_add = fl & WF_S_ADD;
    r.s_depth1 = fl & WF_S_DEPTH1; r.finalize = fl & WF_FINALIZE; r.lane_done = fl & WF_LANE_DONE;
    r.film_rgb = xyz(fm); r.film_w = fm.w;
    r.smp.pcg.state = (uint64_t)rg.x | ((uint64_t)rg.y << 32);
    r.smp.dim = rg.z;
    r.samples_done = rg.w;
    r.pass_idx = ms.x; r.cur_spp = ms.y;
    r.smp.pcg.inc = (uint64_t)ms.z | ((uint64_t)ms.w << 32);
    r.c_samples = r.c_closest = r.c_shadow = r.c_shaded = 0;
}

// Workgroup-wide stream compaction: every lane with a ray gets a distinct index into the queue of its kind; the four waves'
// counts meet in LDS and ONE lane per counter adds the workgroup's total (three atomics per workgroup instead of three per wave:
// queue heads and the active counter are single addresses, and their atomics serialise at the L2). Queue order = slot order
// within the workgroup. Must be called by all 256 threads.
// Sort key of a ray (option wf_sort): 21-bit Morton code of the origin's cell in the scene's box (128 cells per axis), then the
// three sign bits of the direction -- rays that start close together and head the same way end up in the same trace wave.
AKR_D uint32_t wf_spread7(uint32_t x) {  // bit i of the low 7 bits -> bit 3 i
    x &= 0x7fu;
    x = (x | (x << 8)) & 0x0000700fu;
    x = (x | (x << 4)) & 0x000430c3u;
    x = (x | (x << 2)) & 0x00049249u;
    return x;
}
AKR_D uint32_t wf_ray_key(const PtParams& p, vec3 o, vec3 d) {
    auto cell = [](float t) { return (uint32_t)(int)min_f(max_f(t, 0.0f), 127.0f); };  // (NaN -> 0)
    const uint32_t cx = cell((o.x - p.sort_lo[0]) * p.sort_scale[0]), cy = cell((o.y - p.sort_lo[1]) * p.sort_scale[1]), cz = cell((o.z - p.sort_lo[2]) * p.sort_scale[2]);
    const uint32_t oct = (d.x >= 0.0f ? 1u : 0u) | (d.y >= 0.0f ? 2u : 0u) | (d.z >= 0.0f ? 4u : 0u);
    return ((wf_spread7(cx) | (wf_spread7(cy) << 1) | (wf_spread7(cz) << 2)) << 3) | oct;
}
// `resume` != 0: the slot's rays of the last trace launch are not all finished (WfBuffers::pend: bit 0 closest-hit ray, bit 1 shadow ray); the
// unfinished ones go back into the queues marked kWfResume -- the trace kernel continues them from their carry records -- and are not counted again.
constexpr uint32_t kWfResume = 0x80000000u;
AKR_D void wf_enqueue(const PtParams& p, const WfBuffers& wf, uint32_t q, uint32_t slot, PathRegs& r, uint32_t resume = 0u) {
    // closest-hit rays and shadow rays go to separate queues so that waves of the trace kernel are homogeneous
    __shared__ uint32_t sh_cnt[4][3], sh_base[3];
    const bool want_c = r.active && (resume ? (resume & 1u) != 0u : r.has_ray), want_s = r.active && (resume ? (resume & 2u) != 0u : r.has_shadow);
    const uint32_t entry = slot | (resume ? kWfResume : 0u);
    const uint64_t mc = __builtin_amdgcn_ballot_w64(want_c), ms = __builtin_amdgcn_ballot_w64(want_s), ma = __builtin_amdgcn_ballot_w64(r.active);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sh_cnt[wave][0] = (uint32_t)__builtin_popcountll(mc);
        sh_cnt[wave][1] = (uint32_t)__builtin_popcountll(ms);
        sh_cnt[wave][2] = (uint32_t)__builtin_popcountll(ma);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const uint32_t tot = sh_cnt[0][threadIdx.x] + sh_cnt[1][threadIdx.x] + sh_cnt[2][threadIdx.x] + sh_cnt[3][threadIdx.x];
        uint32_t* counter = threadIdx.x == 2 ? wf.n_active : &wf.qcount[2 * q + threadIdx.x];
        sh_base[threadIdx.x] = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    uint32_t bc = sh_base[0], bs = sh_base[1];
    for (uint32_t k = 0; k < wave; k++) { bc += sh_cnt[k][0]; bs += sh_cnt[k][1]; }
    const uint64_t below = (1ull << lane) - 1ull;
    if (want_c) {
        const uint32_t at = bc + (uint32_t)__builtin_popcountll(mc & below);
        wf.queue_closest[q][at] = entry;
        if (p.wf_sort) wf.key_closest[q][at] = wf_ray_key(p, r.ro, r.rd);
        if (!resume) r.c_closest++;
    }
    if (want_s) {
        const uint32_t at = bs + (uint32_t)__builtin_popcountll(ms & below);
        wf.queue_shadow[q][at] = entry;
        if (p.wf_sort) wf.key_shadow[q][at] = wf_ray_key(p, r.s_o, r.s_d);
        if (!resume) r.c_shadow++;
    }
}

template <bool PMJ>
__global__ __launch_bounds__(256) void k_wf_init(const PtParams p, const WfBuffers wf) {
    const uint32_t slot = wf.slot_base + blockIdx.x * 256u + threadIdx.x;
    uint32_t px = 0, py = 0;
    const bool in_frame = slot < wf.slot_end && item_to_pixel(p, slot, px, py);
    const uint32_t pix = px + py * p.width;
    uint32_t sx, sy;
    shifted_pixel(p, px, py, sx, sy);
    PathRegs r;
    path_regs_init<PMJ>(r, p, in_frame, pix, sx, sy);
    if (slot < wf.slot_end) {
        wf_store(wf, slot, r);
        if (wf.pend) wf.pend[slot] = 0u;
    }
    wf_enqueue(p, wf, 0, slot, r);
    flush_counters(p, r, TraceCounters{0, 0, 0}, true);
}

#ifndef AKR_WF_SHADE_WAVES
#define AKR_WF_SHADE_WAVES 1  // waves per SIMD the shade kernel's register allocation must leave room for (1 = whatever it needs)
#endif
template <bool TEX, bool PMJ, bool INST = false>
__global__ __launch_bounds__(256, TEX ? 1 : AKR_WF_SHADE_WAVES) void k_wf_shade(const PtParams p, const WfBuffers wf, uint32_t q_out) {
    const uint32_t slot = wf.slot_base + blockIdx.x * 256u + threadIdx.x;
    PathRegs r;
    r.active = false; r.has_ray = false; r.has_shadow = false;
    r.c_samples = r.c_closest = r.c_shadow = r.c_shaded = 0;
    bool live = false;
    if (slot < wf.slot_end) live = (f2u(wf.base[slot].w) & WF_ACTIVE) != 0;
    // a slot one of whose rays the trace launch carried over is not shaded this time: its state stays as it is and the unfinished rays are queued again
    uint32_t resume = 0u;
    if (live && wf.pend) resume = wf.pend[slot];
    if (resume) r.active = true;
    if (live && !resume) {
        uint32_t px = 0, py = 0;
        item_to_pixel(p, slot, px, py);
        const uint32_t pix = px + py * p.width;
        uint32_t sx, sy;
        shifted_pixel(p, px, py, sx, sy);
        wf_load(wf, slot, r);
        float4 hv = wf.hit[slot];
        Hit hit;
        hit.gid = f2u(hv.x); hit.u = hv.y; hit.v = hv.z; hit.t = 0.0f;
        bool found = hit.gid != kInvalid, occluded = f2u(hv.w) != 0;
        path_step<-1, TEX, PMJ, 0, 0u, INST>(p, r, hit, found, occluded, pix, sx, sy);
        wf_store(wf, slot, r);
    }
    // the queue the trace launch before this one emptied is the next shade launch's to fill: its counts and the queue head back to zero
    // (that launch is complete -- stream order -- and nothing in this one reads them; two hipMemsetAsync per iteration did this until round 6)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        wf.qcount[2u * (1u - q_out)] = 0u;
        wf.qcount[2u * (1u - q_out) + 1u] = 0u;
        *wf.qhead = 0u;
    }
    wf_enqueue(p, wf, q_out, slot, r, resume);
    flush_counters(p, r, TraceCounters{0, 0, 0}, true);
}

template <bool INST> struct TravOf { typedef Trav type; };
template <> struct TravOf<true> { typedef TravI type; };
template <bool INST, class T>
AKR_D void wf_trav_begin(T& s, vec3 o, vec3 d, float tmin, float tmax, uint32_t ex0, uint32_t ex1) {
    if constexpr (INST) trav_begin_inst(s, o, d, tmin, tmax, ex0, ex1);
    else trav_begin(s, o, d, tmin, tmax, ex0, ex1);
}
// Carried rays. A trace launch used to end with its slowest rays -- a hundred dependent fetches deep, in waves with a handful of lanes left -- while
// the rest of the chip idled: 19 of 73 ms per 8 spp on the kept 1080p forest, 69 of 144 ms with 100 k-triangle meshes (HISTORY R6.6). Now a wave that
// has found the queue empty, is down to WfBuffers::carry_lanes (16) lanes and has given each of them carry_steps (48) steps since it started or resumed (progress
// is guaranteed) writes those lanes' traversals -- best hit, position in the tree, the stack -- into the rays' carry records, marks the slots
// (WfBuffers::pend) and ends. k_wf_shade leaves a marked slot alone and queues its unfinished rays again (kWfResume); the next trace launch
// picks them up with everything else. Launches with fewer than WfBuffers::carry_queue rays (65 536) trace to the end as before (the last iterations of a launch
// group must drain, and a launch of its own per fifty steps would cost more than the tail). Nothing a path computes changes -- the iteration in
// which a vertex is shaded does, as with the megakernel's stragglers (pt_pass.h). Measured with 8 / 16 / 32 / 48 lanes and 16 / 48 / 128 steps: all
// within 2 % of each other (1080p forest x 100 k: 172 -- 177 Msamples/s against 124 without).
template <bool INST, class T>
AKR_D void wf_carry_save(const WfBuffers& wf, const T& s, const uint32_t* __restrict__ stack, uint32_t slot, bool any) {
    uint32_t* c = wf.carry + ((size_t)(any ? wf.n_slots : 0u) + slot) * wf.carry_words;  // (carry_words is a multiple of 4: 16-byte aligned records)
    ((uint4*)c)[0] = make_uint4(f2u(s.best_t), f2u(s.best_u), f2u(s.best_v), s.best);
    ((uint4*)c)[1] = make_uint4(s.G, s.T, s.tbase, s.sp);
    if constexpr (INST) ((uint4*)c)[2] = make_uint4(s.leaf, s.pend_rec, s.pend_inst, 0u);
    for (uint32_t k = 0; k < s.sp; k++) c[12u + k] = stack[k * 256u];
}
template <bool INST, class T>
AKR_D void wf_carry_restore(const DScene& sc, const WfBuffers& wf, T& s, uint32_t* __restrict__ stack, uint32_t slot, bool any) {  // (after wf_trav_begin on the slot's ray)
    const uint32_t* c = wf.carry + ((size_t)(any ? wf.n_slots : 0u) + slot) * wf.carry_words;
    const uint4 a = ((const uint4*)c)[0], b = ((const uint4*)c)[1];
    s.best_t = u2f(a.x); s.best_u = u2f(a.y); s.best_v = u2f(a.z); s.best = a.w;
    s.G = b.x; s.T = b.y; s.tbase = b.z; s.sp = b.w;
    for (uint32_t k = 0; k < s.sp; k++) stack[k * 256u] = c[12u + k];
    if constexpr (INST) {
        const uint4 d = ((const uint4*)c)[2];
        s.leaf = d.x; s.pend_rec = d.y; s.pend_inst = d.z;
        if (s.leaf != kInvalid) {
            const uint4* lf = sc.in2.tlas_leaves + (size_t)s.leaf * 4;
            trav_into_instance(sc, s, lf[0], lf[1], lf[2], lf[3]);
        }
    }
    s.active = (s.T != 0) | ((s.G >> 24) != 0) | (s.sp != 0);
}

// Persistent traversal kernel. Ray id = slot; ids [0, n_closest) come from the closest-hit queue, the rest from the
// shadow queue. A lane that finishes its ray writes the result and becomes idle; when enough lanes of the wave are
// idle (or all), the wave refills them from the queue head.
#ifndef AKR_WF_REFILL_IDLE
#define AKR_WF_REFILL_IDLE 20  // refill when at least this many of the 64 lanes are idle
#endif
#ifndef AKR_WF_REFILL_IDLE_INST
#define AKR_WF_REFILL_IDLE_INST 8  // ... on a scene kept as meshes + instances (an iteration of its loop costs more: 1 / 4 / 8 / 20 / 32 idle lanes: 232 / 243 / 245 / 241 / 230 Msamples/s, 1080p forest)
#endif
// INST (round 6): the scene is kept as meshes + instances -- the two-level traversal of dinst_trav.h: a lane's candidates wait in its
// pending slot and the wave takes the exact test in batches, as trace_inst does.
#ifndef AKR_WF_TRACE_INST_WAVES
#define AKR_WF_TRACE_INST_WAVES 1  // waves per SIMD the kept-scene trace kernel's register allocation must leave room for (1 = whatever it needs: 128 VGPRs, 4 waves)
#endif
template <bool TEX, bool INST = false>
__global__ __launch_bounds__(256, INST ? AKR_WF_TRACE_INST_WAVES : 1) void k_wf_trace(const PtParams p, const WfBuffers wf, uint32_t q_in) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    uint32_t* stack = lds_stack + threadIdx.x;
    const DScene& sc = p.sc;
    const uint32_t n_closest = wf.qcount[2 * q_in + 0], n_total = n_closest + wf.qcount[2 * q_in + 1];
    const uint32_t lane = threadIdx.x & 63u;
    if (blockIdx.x == 0 && threadIdx.x == 0) *wf.n_active = 0u;  // (the shade launch after this one counts the slots still active; the host reads it after that)
    TraceCounters cnt{0, 0, 0};
    bool has = false, exhausted = false, any = false, resumed = false;
    uint32_t slot = 0, steps = 0, n_carried = 0;
    const bool carry = wf.carry != nullptr && n_total >= wf.carry_queue;
    // 128 ray ids per claim. (ADVICE round 3 suggested 64 -- one wave-fill -- so that near the end of a queue no wave sits on ids that
    // idle waves could have traced; measured in round 4: twice the atomics on the queue head cost more than the shorter tail gains,
    // cbox with a forced BVH 398 against 576 Msamples/s, 10 M-triangle hall 211 against 219.)
    constexpr uint32_t kWfChunk = 128;
    uint32_t c_next = 0, c_end = 0;  // the wave's claimed range of ray ids
    typename TravOf<INST>::type s;
    wf_trav_begin<INST>(s, mk3(0, 0, 0), mk3(0, 0, 1), 0.0f, -1.0f, kInvalid, kInvalid);  // idle: tmax < tmin
    uint32_t waited = 0;
    bool blocked = false;
    for (;;) {
        // Refill idle lanes from the queue. A wave claims kWfChunk consecutive ray ids with ONE atomic and hands them to its idle
        // lanes by ballot + prefix count until the chunk is used up (one atomic per refill -- 150 k of them on one address per
        // launch -- kept the L2's atomic unit busier than the traversal kept the CUs).
        for (int attempt = 0; attempt < 2 && !exhausted; attempt++) {
            const uint64_t idle = __builtin_amdgcn_ballot_w64(!has);
            if (idle == 0) break;
            if (c_next >= c_end) {  // (wave-uniform)
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(wf.qhead, kWfChunk);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                c_next = base;
                c_end = base + kWfChunk < n_total ? base + kWfChunk : n_total;
                if (base >= n_total) { exhausted = true; break; }
            }
            const uint32_t my = c_next + (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
            if (!has && my < c_end) {
                any = my >= n_closest;
                const uint32_t entry = any ? wf.queue_shadow[q_in][my - n_closest] : wf.queue_closest[q_in][my];
                slot = entry & ~kWfResume;
                resumed = (entry & kWfResume) != 0u;
                float4 a = any ? wf.sh_o[slot] : wf.ray_o[slot];
                float4 b = any ? wf.sh_d[slot] : wf.ray_d[slot];
                wf_trav_begin<INST>(s, xyz(a), xyz(b), 0.0f, any ? b.w : 1e20f, f2u(a.w), any ? f2u(wf.sh_c[slot].w) : kInvalid);
                if (resumed) wf_carry_restore<INST>(sc, wf, s, stack, slot, any);
                has = true;
                blocked = false;
                steps = 0;
            }
            const uint32_t n = (uint32_t)__builtin_popcountll(idle);
            c_next = c_next + n < c_end ? c_next + n : c_end;
        }
        if (__builtin_amdgcn_ballot_w64(has) == 0) break;
        for (;;) {
            bool finished;
            if constexpr (INST) {
                bool pending = false, wait = false;
                if (has) {
                    if (s.pend_rec == kInvalid) blocked = false;
                    steps++;
                    if (s.active & !blocked) blocked = trav_step_inst<TEX>(sc, s, stack, cnt);
                    pending = s.pend_rec != kInvalid;
                    wait = pending & (blocked | !s.active);  // cannot go on without the verdict
                }
                if (__builtin_amdgcn_ballot_w64(wait) != 0) {
                    waited++;
                    if (waited >= AKR_INST_PATIENCE || __builtin_amdgcn_ballot_w64(has & s.active & !wait) == 0 ||
                        __builtin_popcountll(__builtin_amdgcn_ballot_w64(pending)) >= AKR_INST_QUORUM) {
                        waited = 0;
                        if (pending) resolve_pending<TEX>(sc, s, any);
                    }
                }
                finished = has && !s.active && s.pend_rec == kInvalid;
            } else {
                if (has && s.active) { steps++; trav_step<2, TEX>(sc, s, stack, cnt, any); }
                finished = has && !s.active;
            }
            if (finished) {  // ray finished: publish the result for k_wf_shade
                float* hp = (float*)&wf.hit[slot];
                if (any) {
                    hp[3] = u2f(s.best != kInvalid ? 1u : 0u);
                } else {
                    hp[0] = u2f(s.best); hp[1] = s.best_u; hp[2] = s.best_v;
                }
                if (resumed) atomicAnd(&wf.pend[slot], any ? ~2u : ~1u);
                has = false;
            }
            const uint32_t n_idle = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(!has));
            if (n_idle == 64u) break;
            if (carry && exhausted && n_idle >= 64u - wf.carry_lanes && __builtin_amdgcn_ballot_w64(has && steps < wf.carry_steps) == 0) {
                if (has) {  // the wave's last rays go on in the next launch
                    wf_carry_save<INST>(wf, s, stack, slot, any);
                    atomicOr(&wf.pend[slot], any ? 2u : 1u);
                    has = false;
                    n_carried++;
                }
                break;
            }
            if (!exhausted && n_idle >= (INST ? AKR_WF_REFILL_IDLE_INST : AKR_WF_REFILL_IDLE)) break;
        }
    }
    // traversal counters
    if (p.counters != nullptr) {
        uint64_t* const ctr = p.counters + 8u * (blockIdx.x % kStatStripes);
        uint32_t nn = wave_sum_u32(cnt.nodes), nt = wave_sum_u32(cnt.tris), ov = wave_sum_u32(cnt.overflow);
        if (lane == 0) {
            if (nn) atomicAdd((unsigned long long*)&ctr[4], (unsigned long long)nn);
            if (nt) atomicAdd((unsigned long long*)&ctr[5], (unsigned long long)nt);
            if (ov) atomicAdd((unsigned long long*)&ctr[6], (unsigned long long)ov);
        }
        if (carry) {
            const uint32_t nc = wave_sum_u32(n_carried);
            if (lane == 0 && nc) atomicAdd((unsigned long long*)&ctr[7], (unsigned long long)nc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------- launchers
hipError_t launch_wf_init(const PtParams& p, const WfBuffers& wf, hipStream_t stream) {
    uint32_t blocks = (wf.slot_end - wf.slot_base + 255u) / 256u;
    if (blocks == 0) return hipSuccess;
    if (p.sampler) hipLaunchKernelGGL(k_wf_init<true>, dim3(blocks), dim3(256), 0, stream, p, wf);
    else hipLaunchKernelGGL(k_wf_init<false>, dim3(blocks), dim3(256), 0, stream, p, wf);
    return hipGetLastError();
}
hipError_t launch_wf_shade(const PtParams& p, const WfBuffers& wf, uint32_t q_out, hipStream_t stream) {
    uint32_t blocks = (wf.slot_end - wf.slot_base + 255u) / 256u;
    if (blocks == 0) return hipSuccess;
    const bool tex = p.sc.tex.nodes != nullptr, pmj = p.sampler != 0;
    const bool inst = p.sc.in2.on != 0;
#define AKR_WF_SHADE(T, S, Q, L)                                                                                              \
    {                                                                                                                       \
        if (inst) hipLaunchKernelGGL((k_wf_shade<T, S, true>), dim3(blocks), dim3(256), L, stream, Q, wf, q_out);              \
        else hipLaunchKernelGGL((k_wf_shade<T, S, false>), dim3(blocks), dim3(256), L, stream, Q, wf, q_out);                  \
    }
    if (tex) {
        size_t lds;
        const PtParams q = with_tex_slots(p, 0, lds);
        if (pmj) AKR_WF_SHADE(true, true, q, lds) else AKR_WF_SHADE(true, false, q, lds)
    } else {
        if (pmj) AKR_WF_SHADE(false, true, p, 0) else AKR_WF_SHADE(false, false, p, 0)
    }
#undef AKR_WF_SHADE
    return hipGetLastError();
}
// Workgroups of the persistent trace kernel one CU holds at once (registers and the LDS stacks of this scene's tree decide).
uint32_t wf_trace_blocks_per_cu(const PtParams& p) {
    int n = 0;
    const bool tex = p.sc.tex.nodes != nullptr;
    size_t lds = (size_t)p.sc.bvh_stack_depth * 256 * 4;
    if (tex) (void)with_tex_slots(p, lds, lds);
    const bool inst = p.sc.in2.on != 0;
    hipError_t e = inst ? (tex ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wf_trace<true, true>, 256, lds)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wf_trace<false, true>, 256, lds))
                        : (tex ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wf_trace<true, false>, 256, lds)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wf_trace<false, false>, 256, lds));
    if (e != hipSuccess || n < 1) n = 4;
    return (uint32_t)std::min(n, 8);
}
hipError_t launch_wf_trace(const PtParams& p, const WfBuffers& wf, uint32_t q_in, uint32_t n_blocks, hipStream_t stream) {
    const bool inst = p.sc.in2.on != 0;
    if (p.sc.tex.nodes != nullptr) {
        size_t lds;
        const PtParams q = with_tex_slots(p, p.sc.bvh_stack_depth * 256 * 4, lds);
        if (inst) hipLaunchKernelGGL((k_wf_trace<true, true>), dim3(n_blocks), dim3(256), lds, stream, q, wf, q_in);
        else hipLaunchKernelGGL((k_wf_trace<true, false>), dim3(n_blocks), dim3(256), lds, stream, q, wf, q_in);
    } else if (inst) hipLaunchKernelGGL((k_wf_trace<false, true>), dim3(n_blocks), dim3(256), p.sc.bvh_stack_depth * 256 * 4, stream, p, wf, q_in);
    else hipLaunchKernelGGL((k_wf_trace<false, false>), dim3(n_blocks), dim3(256), p.sc.bvh_stack_depth * 256 * 4, stream, p, wf, q_in);
    return hipGetLastError();
}

}  // namespace akr
