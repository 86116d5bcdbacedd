// dinst_trav.h -- two-level traversal and hit reconstruction for a scene kept as meshes + instances (host/scene_inst.cpp).
//
// The reference's accel is two-level (crates/akari_render/src/mesh.rs:259-348: one BLAS per mesh, instances pushed with their
// transform) and its hit reconstruction works from object-space buffers + the instance transform (mesh.rs:487-654). Here the two
// box levels only cull; a triangle is accepted, and a hit shaded, by the FLATTENED arithmetic computed on the fly (dinst.h): the
// candidate's object-space vertices through the instance transform in f32, Woop rows in f64, the coplanar-neighbour rule, then
// the same tri_test on the WORLD ray. Films are bit-identical to the flattened scene's (and the oracle's).
//
// State machine = disect.h's trav_step with an instance level: a lane is in the TLAS (inst == kInvalid: world-space ray, "triangles"
// of a leaf are instances) or inside one instance's BLAS (the ray taken through the instance's inverse, unnormalised so that t is the
// world parameter). Entering an instance pushes the TLAS position (pending group, triangle base, pending leaf bits as a sentinel
// whose top byte is 0 -- a BLAS never pushes such a word); popping the sentinel leaves the instance.
#pragma once
#include "dinst.h"
#include "disect.h"

namespace akr {

struct TravI : Trav {
    vec3 wo, wd;                 // the world-space ray (s.o / s.d are the current level's)
    uint32_t inst, node_off, tri_off, gid_base;  // current instance (kInvalid: TLAS), its BLAS's node offset, mesh triangle base, first gid
    uint32_t leaf;                               // ... and its TLAS leaf record (what a carried traversal re-enters the instance from)
#if defined(AKR_INST_PRETEST_CHECK)
    float check_t;                               // >= 0: tri_may_hit rejected the pending candidate against this limit
#endif
    uint32_t pend_rec, pend_inst;                // a candidate that passed tri_may_hit and waits for the exact test: its record in mesh_tris
                                                 // (kInvalid: none) and its instance
};
AKR_D void trav_set_ray(Trav& s, vec3 o, vec3 d) {
    s.o = o; s.d = d;
    s.inv = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    s.noi = mk3(-o.x * s.inv.x, -o.y * s.inv.y, -o.z * s.inv.z);
    const uint32_t oi = (s.inv.x >= 0.0f ? 1u : 0u) | (s.inv.y >= 0.0f ? 2u : 0u) | (s.inv.z >= 0.0f ? 4u : 0u);
    s.octinv4 = oi * 0x01010101u;
}
AKR_D void trav_begin_inst(TravI& s, vec3 o, vec3 d, float tmin, float tmax, uint32_t ex0, uint32_t ex1) {
    trav_begin(s, o, d, tmin, tmax, ex0, ex1);
    s.wo = o; s.wd = d;
    s.inst = kInvalid; s.node_off = 0; s.tri_off = 0; s.gid_base = 0; s.leaf = kInvalid;
    s.pend_rec = kInvalid; s.pend_inst = 0;
#if defined(AKR_INST_PRETEST_CHECK)
    s.check_t = -1.0f;
#endif
}

// the ray into the object space of the instance of a TLAS leaf record (rows of the inverse transform | ids)
AKR_D void trav_into_instance(const DScene& sc, TravI& s, uint4 w0, uint4 w1, uint4 w2, uint4 w3) {
    s.inst = w3.z; s.node_off = w3.x; s.tri_off = w3.y;
    s.gid_base = f2u(sc.inst[(size_t)w3.z * INST_ROWS + 5].w);  // (w3.w = the node of the mesh's tree this leaf record starts at)
    const vec3 r0 = mk3(u2f(w0.x), u2f(w0.y), u2f(w0.z)), r1 = mk3(u2f(w1.x), u2f(w1.y), u2f(w1.z)), r2 = mk3(u2f(w2.x), u2f(w2.y), u2f(w2.z));
    const vec3 oo = mk3(dot(r0, s.wo) + u2f(w0.w), dot(r1, s.wo) + u2f(w1.w), dot(r2, s.wo) + u2f(w2.w));
    const vec3 od = mk3(dot(r0, s.wd), dot(r1, s.wd), dot(r2, s.wd));
    trav_set_ray(s, oo, od);
}

// material of (instance record m, mesh triangle): mats[slots[prim]] (mesh.rs:508-521)
AKR_D uint32_t inst_material(const DScene& sc, const float4* m, uint32_t meta) { return sc.in2.inst_mats[f2u(m[2].w) + (meta & 0x3fffffffu)]; }

// scene.rs:49-86 for a candidate of an instanced scene: alpha of the base colour (folded, or the graph at the candidate's uv)
template <bool TEX>
AKR_D bool alpha_test_inst(const DScene& sc, const float4* m, uint32_t inst, uint32_t prim, uint32_t mesh_base, float4 q0, float4 q1, float4 q2, float4 q3, float u, float v) {
    const uint32_t material = inst_material(sc, m, sc.in2.mesh_meta[mesh_base + prim]);
    const DMaterial& mt = sc.materials[material];
    float alpha = (mt.kind == MAT_PRINCIPLED || mt.kind == MAT_DIFFUSE) ? mt.base_alpha : 1.0f;
    if (TEX) {
        if (mt.flags & MF_ALPHA_TEXTURED) {
            const float w = 1.0f - u - v;
            // uv0 = (q0.w, q1.w), uv1 = (q2.w, q3.x), uv2 = (q3.y, q3.z): the interpolation of textured_alpha (disect.h)
            const vec2 uv = mk2((q0.w * w + q2.w * u) + q3.y * v, (q1.w * w + q3.x * u) + q3.z * v);
            alpha = material_alpha_at(sc.tex, mt, material, uv);
        }
    }
    if (alpha >= 1.0f) return true;
    float h = (float)xxhash32_4(inst, prim, f2u(u), f2u(v)) * 2.3283064365386963e-10f;
    return alpha > h;
}

// The exact test of the pending candidate: its flattened record, computed here (dinst.h), then the flattened test on the world ray.
template <bool TEX>
AKR_D void resolve_pending(const DScene& sc, TravI& s, bool any_hit) {
    const uint32_t inst = s.pend_inst;
    const float4* m = sc.inst + (size_t)inst * INST_ROWS;
    const float4* rec = sc.in2.mesh_tris + (size_t)s.pend_rec * 4;
    s.pend_rec = kInvalid;
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
    const uint32_t tri_off = f2u(m[1].w), prim = f2u(q3.w) & kMeshPrimMask, gid = f2u(m[5].w) + prim;
    const vec3 c0 = xyz(m[0]), c1 = xyz(m[1]), c2 = xyz(m[2]), tr = xyz(m[3]);
    const vec3 A = xf_point(c0, c1, c2, tr, xyz(q0)), B = xf_point(c0, c1, c2, tr, xyz(q1)), C = xf_point(c0, c1, c2, tr, xyz(q2));
    float wr[12];
#if defined(AKR_INST_FAKE_EXACT)  // timing only: f32 rows (films differ)
    {
        const vec3 e1 = B - A, e2 = C - A, n = cross(e1, e2);
        const float det = dot(n, n), inv = 1.0f / det;
        const vec3 r0 = cross(e2, n) * inv, r1 = cross(n, e1) * inv, r2 = n * inv;
        wr[0] = r0.x; wr[1] = r0.y; wr[2] = r0.z; wr[3] = -dot(r0, A);
        wr[4] = r1.x; wr[5] = r1.y; wr[6] = r1.z; wr[7] = -dot(r1, A);
        wr[8] = r2.x; wr[9] = r2.y; wr[10] = r2.z; wr[11] = -dot(r2, A);
    }
#else
    // The coplanar-neighbour rule (dinst.h share_plane_row: an odd triangle lying in its even neighbour's plane carries that neighbour's plane
    // row) was decided once per scene for every instance-triangle (k_inst_share_bits: inst_pair_shares below). Here the third row is computed
    // from the neighbour's vertices instead of the triangle's own -- one code path, no test at the candidate; a triangle that does not share
    // never reads its neighbour (two dependent gathers less), and one that shares in no instance of its mesh (kMeshTriShares clear in its
    // record: every triangle of a mesh that is not made of quads) not even its bit. 1080p forest x 10 k / x 100 k: wavefront schedule 229 -> 239 / 137 -> 146 Msamples/s, megakernel 162 -> 177 / 112 -> 123.
    vec3 P0 = A, P1 = B, P2 = C;
    if ((f2u(q3.w) & kMeshTriShares) && ((sc.in2.share_bits[gid >> 5] >> (gid & 31u)) & 1u)) {
        const float4* nb = sc.in2.mesh_tris + (size_t)(tri_off + sc.in2.mesh_pos[tri_off + prim - 1u]) * 4;
        P0 = xf_point(c0, c1, c2, tr, xyz(nb[0])); P1 = xf_point(c0, c1, c2, tr, xyz(nb[1])); P2 = xf_point(c0, c1, c2, tr, xyz(nb[2]));
    }
    woop_edge_rows(A, B, C, wr);
    woop_plane_row(P0, P1, P2, wr + 8);
#endif
    float t, u, v;
    bool h = tri_test(s.wo, s.wd, make_float4(wr[0], wr[1], wr[2], wr[3]), make_float4(wr[4], wr[5], wr[6], wr[7]), make_float4(wr[8], wr[9], wr[10], wr[11]), s.tmin, s.tmax,
                      t, u, v);
#if defined(AKR_INST_PRETEST_CHECK)
    if (h && s.check_t >= 0.0f && t <= s.check_t) s.check_t = -2.0f;  // (trace_inst turns this into the overflow flag: the render fails)
#endif
    if (h && sc.has_alpha) h = alpha_test_inst<TEX>(sc, m, inst, prim, tri_off, q0, q1, q2, q3, u, v);
    if (h) {
        if (any_hit) {
            s.best = gid;
            s.T = 0; s.G = 0; s.sp = 0;  // any hit: done
            s.active = false;
        } else {
            const bool better = (s.best == kInvalid) | (t < s.best_t) | ((t == s.best_t) & (gid < s.best));
            if (better) { s.best_t = t; s.best_u = u; s.best_v = v; s.best = gid; }
        }
    }
}

// share_plane_row for the odd triangle `prim` of instance `inst`, outside a traversal (k_inst_share_bits): the flattening compiler's decision
// (host/scene_build.cpp) from the same functions on the same values.
AKR_D bool inst_pair_shares(const DScene& sc, uint32_t inst, uint32_t prim, uint32_t& rec_pos) {
    const float4* m = sc.inst + (size_t)inst * INST_ROWS;
    const uint32_t tri_off = f2u(m[1].w);
    rec_pos = tri_off + sc.in2.mesh_pos[tri_off + prim];
    const float4* rec = sc.in2.mesh_tris + (size_t)rec_pos * 4;
    const float4* nb = sc.in2.mesh_tris + (size_t)(tri_off + sc.in2.mesh_pos[tri_off + prim - 1u]) * 4;
    const vec3 c0 = xyz(m[0]), c1 = xyz(m[1]), c2 = xyz(m[2]), tr = xyz(m[3]);
    const vec3 A = xf_point(c0, c1, c2, tr, xyz(rec[0])), B = xf_point(c0, c1, c2, tr, xyz(rec[1])), C = xf_point(c0, c1, c2, tr, xyz(rec[2]));
    const vec3 na = xf_point(c0, c1, c2, tr, xyz(nb[0])), nbv = xf_point(c0, c1, c2, tr, xyz(nb[1])), nc = xf_point(c0, c1, c2, tr, xyz(nb[2]));
    float wr[12], ra[4];
    woop_precompute(A, B, C, wr);
    woop_plane_row(na, nbv, nc, ra);
    const vec3 vb[3] = {A, B, C};
    return plane_row_is_shared(ra, wr + 8, vb);
}

// One step of one lane, in up to four stages -- whatever the lane's state allows, in this order: pop a stack entry (possibly the
// sentinel that ends an instance), enter the instance of a pending top-level leaf, visit a node, test a candidate. (A wave pays for
// every stage some lane is in; a lane that is through with one goes on to the next in the same iteration instead of waiting for the
// wave's next one: a third fewer iterations per ray than one stage per step.) The order of a ray's node visits and candidates is the
// one-stage-per-step order. Returns true when the lane found a second candidate for the exact test while one is pending: the
// candidate is put back (its leaf bit set again) and the lane waits for trace_pair_inst / trace_inst to resolve the pending one.
template <bool TEX>
AKR_D bool trav_step_inst(const DScene& sc, TravI& s, uint32_t* __restrict__ stack, TraceCounters& cnt) {
    bool blocked = false;
    // ---- stage 0: nothing pending at this level -> the next stack entry (the caller guarantees sp > 0 then)
    if ((s.T == 0) & ((s.G >> 24) == 0)) {
        s.sp--;
        const uint32_t e = stack[s.sp * 256u];
        if ((s.inst != kInvalid) & ((e >> 24) == 0)) {
            // the sentinel: this instance is done. Back to the TLAS where it stood: pending leaf bits, their base, the pending group.
            s.T = e;
            s.sp--;
            s.tbase = stack[s.sp * 256u];
            s.sp--;
            s.G = stack[s.sp * 256u];
            s.inst = kInvalid; s.node_off = 0; s.tri_off = 0; s.leaf = kInvalid;
            trav_set_ray(s, s.wo, s.wd);
        } else {
            s.G = e;
        }
    }
    // ---- stage 1: a pending TLAS leaf entry = an instance: remember where the TLAS traversal stands, take the ray into object space,
    // start at the BLAS root
    if ((s.T != 0) & (s.inst == kInvalid)) {
        const uint32_t b = (uint32_t)__builtin_ctz(s.T);
        s.T &= s.T - 1u;
        const uint4* lf = sc.in2.tlas_leaves + (size_t)(s.tbase + b) * 4;
        const uint4 w0 = lf[0], w1 = lf[1], w2 = lf[2], w3 = lf[3];
#if !defined(AKR_INST_COUNT)
        cnt.nodes++;
#endif
        if (s.sp + 3 <= sc.bvh_stack_depth) {
            stack[s.sp * 256u] = s.G; s.sp++;
            stack[s.sp * 256u] = s.tbase; s.sp++;
            stack[s.sp * 256u] = s.T & 0x00ffffffu; s.sp++;
        } else {
            cnt.overflow = 1;
        }
        s.leaf = s.tbase + b;
        trav_into_instance(sc, s, w0, w1, w2, w3);
        s.G = (1u << (24u + (s.octinv4 & 7u))) | w3.w;  // the group {entry node}: base = the node (relative; 0 = the root), slot 0
        s.T = 0; s.tbase = 0;
    }
    // ---- stage 2: a node of either level: disect.h trav_step's box test on the current level's ray
    if ((s.T == 0) & ((s.G >> 24) != 0)) {
        const uint32_t j = 31u - (uint32_t)__builtin_clz(s.G);  // nearest pending sibling
        s.G &= ~(1u << j);
        if ((s.G >> 24) != 0) {  // the others wait as one entry
            if (s.sp < sc.bvh_stack_depth) {
                stack[s.sp * 256u] = s.G;
                s.sp++;
            } else {
                cnt.overflow = 1;
            }
        }
        const uint32_t slot = (j - 24u) ^ (s.octinv4 & 7u);
        const uint32_t idx = s.node_off + (s.G & 0xffffffu) + slot;
        const uint4* p = sc.bvh_nodes + (size_t)idx * (kBvhNodeWords / 4);
        const uint4 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
#if !defined(AKR_INST_COUNT)
        cnt.nodes++;
#endif
        const float limit = s.best_t;
        const float bx = u2f((w0.w & 0xffu) << 23) * s.inv.x, by = u2f(((w0.w >> 8) & 0xffu) << 23) * s.inv.y, bz = u2f(((w0.w >> 16) & 0xffu) << 23) * s.inv.z;
        const float ax = __builtin_fmaf(u2f(w0.x), s.inv.x, s.noi.x), ay = __builtin_fmaf(u2f(w0.y), s.inv.y, s.noi.y), az = __builtin_fmaf(u2f(w0.z), s.inv.z, s.noi.z);
        const bool nx = s.inv.x < 0.0f, ny = s.inv.y < 0.0f, nz = s.inv.z < 0.0f;
        const uint32_t xb = nx ? ((w3.y >> 16) | (w3.y << 16)) : w3.y, yb = ny ? ((w3.z >> 16) | (w3.z << 16)) : w3.z, zb = nz ? ((w3.w >> 16) | (w3.w << 16)) : w3.w;
        const uint32_t qnx[2] = {nx ? w2.z : w1.w, xb}, qfx[2] = {nx ? w1.w : w2.z, xb >> 16};
        const uint32_t qny[2] = {ny ? w2.w : w2.x, yb}, qfy[2] = {ny ? w2.x : w2.w, yb >> 16};
        const uint32_t qnz[2] = {nz ? w3.x : w2.y, zb}, qfz[2] = {nz ? w2.y : w3.x, zb >> 16};
        uint32_t hitmask = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t meta4 = h ? (w1.x >> 16) : w1.y;
            const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
            const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
            const uint32_t bit_index4 = (meta4 ^ (s.octinv4 & inner_mask4)) & 0x1f1f1f1fu;
            const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
#pragma unroll
            for (int i = 0; i < (h ? 2 : 4); i++) {
                const float tnx = __builtin_fmaf((float)byte_of(qnx[h], i), bx, ax), tfx = __builtin_fmaf((float)byte_of(qfx[h], i), bx, ax);
                const float tny = __builtin_fmaf((float)byte_of(qny[h], i), by, ay), tfy = __builtin_fmaf((float)byte_of(qfy[h], i), by, ay);
                const float tnz = __builtin_fmaf((float)byte_of(qnz[h], i), bz, az), tfz = __builtin_fmaf((float)byte_of(qfz[h], i), bz, az);
                const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, s.tmin));
                const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, limit));
                if (tn <= tf) hitmask |= byte_of(child_bits4, i) << byte_of(bit_index4, i);
            }
        }
        s.G = ((w0.w >> 24) | ((w1.x & 0xffffu) << 8)) | (hitmask & 0xff000000u);
        s.T = hitmask & 0x00ffffffu;
        s.tbase = w1.z;
    }
    // ---- stage 3: a candidate triangle of the current instance: the conservative reject (dinst.h tri_may_hit) here; what it cannot
    // decide waits in the lane's pending slot for the exact test, which the caller runs for many lanes at once
    // (two or three candidates, or two or three nodes, per step were measured too: 135 / 125 and 156 / 148 against 159 Msamples/s)
    if ((s.T != 0) & (s.inst != kInvalid)) {
        const uint32_t leaf_bit = (uint32_t)__builtin_ctz(s.T);
        s.T &= s.T - 1u;
        const uint4* p = (const uint4*)sc.in2.mesh_tris + (size_t)(s.tri_off + s.tbase + leaf_bit) * 4;
        const uint4 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
        cnt.tris++;
        const uint32_t prim = w3.w & kMeshPrimMask, gid = s.gid_base + prim;
        if ((gid != s.ex0) & (gid != s.ex1)) {
            const float4* m = sc.inst + (size_t)s.inst * INST_ROWS;
            const vec3 c0 = xyz(m[0]), c1 = xyz(m[1]), c2 = xyz(m[2]), tr = xyz(m[3]);
            const vec3 A = xf_point(c0, c1, c2, tr, mk3(u2f(w0.x), u2f(w0.y), u2f(w0.z))), B = xf_point(c0, c1, c2, tr, mk3(u2f(w1.x), u2f(w1.y), u2f(w1.z)));
            const vec3 C = xf_point(c0, c1, c2, tr, mk3(u2f(w2.x), u2f(w2.y), u2f(w2.z)));
            // (an odd triangle may carry its even neighbour's plane row: that plane is within 1e-6 sqrt(|n_even|) of its vertices;
            // m[7].x bounds sqrt(|n|) over the instance's triangles -- scene_inst.cpp)
            const float shift = (w3.w & kMeshTriShares) ? 2e-6f * m[7].x : 0.0f;
#if defined(AKR_INST_PRETEST_CHECK)  // measurement / test builds: every candidate takes the exact test, which reports a wrong reject
            const bool may = tri_may_hit(s.wo, s.wd, A, B, C, s.tmin, s.best_t, shift);
            if (s.pend_rec == kInvalid && s.check_t != -2.0f) s.check_t = may ? -1.0f : s.best_t;
            if (true) {
#else
            if (tri_may_hit(s.wo, s.wd, A, B, C, s.tmin, s.best_t, shift)) {
#endif
                if (s.pend_rec == kInvalid) {
                    s.pend_rec = s.tri_off + s.tbase + leaf_bit;
                    s.pend_inst = s.inst;
                } else {
                    s.T |= 1u << leaf_bit;
                    cnt.tris--;
                    blocked = true;
                }
            }
        }
    }
    s.active = (s.T != 0) | ((s.G >> 24) != 0) | (s.sp != 0);
    return blocked;
}

// The exact test is ~20 times a node visit (nine f64 divisions) and few candidates need it: the lanes of a wave collect theirs and
// take the test together -- when AKR_INST_QUORUM lanes have one, when a lane that cannot go on without its verdict (a second
// candidate, or the end of its traversal) has waited AKR_INST_PATIENCE steps, or when no lane can go on. The order in which a ray's
// candidates are tested changes neither the closest hit (smallest t, then smallest id) nor whether there is any.
#ifndef AKR_INST_QUORUM
#define AKR_INST_QUORUM 16
#endif
#ifndef AKR_INST_PATIENCE
#define AKR_INST_PATIENCE 2  // (1 / 2 / 4 / 8 steps: within 2 % of each other on the 1080p forest, 2 ahead on most legs)
#endif
template <bool ANY_HIT, bool TEX = false>
AKR_D bool trace_inst(const DScene& sc, vec3 o, vec3 d, float tmin, float tmax, uint32_t ex0, uint32_t ex1, Hit& hit, uint32_t* __restrict__ stack, TraceCounters& cnt) {
    TravI s;
    trav_begin_inst(s, o, d, tmin, tmax, ex0, ex1);
    uint32_t waited = 0;
    bool blocked = false;
    while (s.active | (s.pend_rec != kInvalid)) {
#if defined(AKR_INST_COUNT) && AKR_INST_COUNT == 3
        if ((uint32_t)__builtin_ctzll(__ballot(true)) == (threadIdx.x & 63u)) cnt.nodes++;
#endif
        if (s.pend_rec == kInvalid) blocked = false;
        if (s.active & !blocked) blocked = trav_step_inst<TEX>(sc, s, stack, cnt);
        const bool pending = s.pend_rec != kInvalid;
        const bool wait = pending & (blocked | !s.active);  // cannot go on without the verdict
        const uint64_t waiting = __ballot(wait);
        if (waiting == 0) continue;
        waited++;
        if (waited >= AKR_INST_PATIENCE || __ballot(s.active & !wait) == 0 || __popcll(__ballot(pending)) >= AKR_INST_QUORUM) {
            waited = 0;
#if defined(AKR_INST_COUNT) && AKR_INST_COUNT == 1   // measurement builds: n_node_visits counts exact tests (lanes) instead
            if (pending) cnt.nodes++;
#elif defined(AKR_INST_COUNT) && AKR_INST_COUNT == 2  // ... or the times a wave ran the exact test
            if (pending && (uint32_t)__builtin_ctzll(__ballot(pending)) == (threadIdx.x & 63u)) cnt.nodes++;
#elif defined(AKR_INST_COUNT) && AKR_INST_COUNT == 3  // ... or the loop iterations of the waves
            if (false) cnt.nodes++;
#endif
            if (pending) resolve_pending<TEX>(sc, s, ANY_HIT);
#if defined(AKR_INST_PRETEST_CHECK)
            if (s.check_t == -2.0f) cnt.overflow = 1;
#endif
        }
    }
    hit.t = s.best_t; hit.u = s.best_u; hit.v = s.best_v; hit.gid = s.best;
    return s.best != kInvalid;
}

// Both rays of a path vertex through ONE loop (a lane whose closest-hit ray is done goes straight on to its shadow ray), which ends
// when at most 1/STRAG of the lanes that entered it are still tracing: those keep their traversal -- in 16 words of LDS per lane, the
// stack where it is -- and go on in the next intersection phase while the others shade (pt_pass.h: the flattened scenes' kernels do
// the same, AKR_PT_STRAGGLERS). cy = the lane's LDS column (slot k at cy[k * 256]).
constexpr uint32_t kCarrySlotsInst = 16;
template <bool TEX, uint32_t STRAG>
AKR_D void trace_pair_inst(const DScene& sc, bool has_ray, vec3 ro, vec3 rd, uint32_t ray_ex0, bool has_shadow, vec3 s_o, vec3 s_d, float s_tmax, uint32_t s_ex0,
                           uint32_t s_ex1, bool& carry, Hit& hit, bool& found, bool& occluded, uint32_t* __restrict__ stack, uint32_t* __restrict__ cy, TraceCounters& cnt) {
    TravI s;
    uint32_t phase;  // 0: closest-hit ray in flight, 1: shadow ray, 2: done
    hit.t = 1e20f; hit.u = 0.0f; hit.v = 0.0f; hit.gid = kInvalid;
    if (!(STRAG > 0 && carry)) {
        phase = has_ray ? 0u : (has_shadow ? 1u : 2u);
        if (phase == 0) trav_begin_inst(s, ro, rd, 0.0f, 1e20f, ray_ex0, kInvalid);
        else trav_begin_inst(s, s_o, s_d, 0.0f, phase == 1 ? s_tmax : -1.0f, s_ex0, s_ex1);
    } else {
        phase = cy[8 * 256];
        if (phase == 0) trav_begin_inst(s, ro, rd, 0.0f, 1e20f, ray_ex0, kInvalid);
        else {
            trav_begin_inst(s, s_o, s_d, 0.0f, s_tmax, s_ex0, s_ex1);
            hit.t = u2f(cy[9 * 256]); hit.u = u2f(cy[10 * 256]); hit.v = u2f(cy[11 * 256]); hit.gid = cy[12 * 256];
            found = hit.gid != kInvalid;
        }
        s.best_t = u2f(cy[0]); s.best_u = u2f(cy[1 * 256]); s.best_v = u2f(cy[2 * 256]); s.best = cy[3 * 256];
        s.G = cy[4 * 256]; s.T = cy[5 * 256]; s.tbase = cy[6 * 256]; s.sp = cy[7 * 256];
        s.leaf = cy[13 * 256]; s.pend_rec = cy[14 * 256]; s.pend_inst = cy[15 * 256];
        if (s.leaf != kInvalid) {
            const uint4* lf = sc.in2.tlas_leaves + (size_t)s.leaf * 4;
            trav_into_instance(sc, s, lf[0], lf[1], lf[2], lf[3]);
        }
        s.active = (s.T != 0) | ((s.G >> 24) != 0) | (s.sp != 0);
    }
    const uint32_t n_in = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(phase != 2u));
    const uint32_t n_leave = STRAG > 0 ? n_in / (STRAG > 0 ? STRAG : 1u) : 0u;
    uint32_t waited = 0;
    bool blocked = false;
    while (true) {
        const uint32_t n_now = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(phase != 2u));
        if (n_now <= n_leave) break;  // (n_leave < n_in: at least one lane of the phase finishes)
        if (phase != 2u) {
            if (s.pend_rec == kInvalid) blocked = false;
            if (s.active & !blocked) blocked = trav_step_inst<TEX>(sc, s, stack, cnt);
            const bool pending = s.pend_rec != kInvalid;
            const bool wait = pending & (blocked | !s.active);  // cannot go on without the verdict
            if (__ballot(wait) != 0) {
                waited++;
                if (waited >= AKR_INST_PATIENCE || __ballot(s.active & !wait) == 0 || __popcll(__ballot(pending)) >= AKR_INST_QUORUM) {
                    waited = 0;
                    if (pending) resolve_pending<TEX>(sc, s, phase == 1u);
                }
            }
            if (!s.active & (s.pend_rec == kInvalid)) {
                blocked = false;
                if (phase == 0u) {
                    found = s.best != kInvalid;
                    hit.t = s.best_t; hit.u = s.best_u; hit.v = s.best_v; hit.gid = s.best;
                    phase = has_shadow ? 1u : 2u;
                    if (has_shadow) trav_begin_inst(s, s_o, s_d, 0.0f, s_tmax, s_ex0, s_ex1);
                } else {
                    occluded = s.best != kInvalid;
                    phase = 2u;
                }
            }
        }
    }
    carry = STRAG > 0 && phase != 2u;
    if (STRAG > 0 && carry) {
        cy[0] = f2u(s.best_t); cy[1 * 256] = f2u(s.best_u); cy[2 * 256] = f2u(s.best_v); cy[3 * 256] = s.best;
        cy[4 * 256] = s.G; cy[5 * 256] = s.T; cy[6 * 256] = s.tbase; cy[7 * 256] = s.sp;
        cy[8 * 256] = phase;
        if (phase == 1u) { cy[9 * 256] = f2u(hit.t); cy[10 * 256] = f2u(hit.u); cy[11 * 256] = f2u(hit.v); cy[12 * 256] = hit.gid; }
        cy[13 * 256] = s.leaf; cy[14 * 256] = s.pend_rec; cy[15 * 256] = s.pend_inst;
    }
}

// the instance a global triangle id belongs to: the last i with inst_tri_offset[i] <= gid
AKR_D uint32_t inst_of_gid(const DScene& sc, uint32_t gid) {
    uint32_t lo = 0, hi = sc.in2.n_instances;  // inst_tri_offset has n + 1 entries
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sc.inst_tri_offset[mid] <= gid) lo = mid; else hi = mid;
    }
    return lo;
}

// MeshAggregate::surface_interaction (mesh.rs:487-654) for a scene kept as meshes + instances: the shade record the flattening
// compiler would have stored for this instance-triangle, rebuilt from the mesh triangle and the instance transform (dinst.h).
AKR_D SurfacePoint surface_interaction_inst(const DScene& sc, uint32_t gid, vec2 bary) {
    const uint32_t inst = inst_of_gid(sc, gid);
    const float4* m = sc.inst + (size_t)inst * INST_ROWS;
    const uint32_t mesh_base = f2u(m[1].w), prim = gid - f2u(m[5].w);
    const float4* r = sc.in2.mesh_tris + (size_t)(mesh_base + sc.in2.mesh_pos[mesh_base + prim]) * 4;
    const float4 a = r[0], b = r[1], c = r[2], d = r[3];
    const uint32_t meta = sc.in2.mesh_meta[mesh_base + prim];
    const vec2 uv0 = mk2(a.w, b.w), uv1 = mk2(c.w, d.x), uv2 = mk2(d.y, d.z);
    const TriWorld tw = tri_world(inst_xf_from_rows(m), xyz(a), xyz(b), xyz(c), uv0, uv1, uv2);
    // the eight rows of the flattened shade record (dscene.h)
    const float4 q0 = make_float4(a.x, a.y, a.z, uv0.x), q1 = make_float4(b.x, b.y, b.z, uv0.y), q2 = make_float4(c.x, c.y, c.z, uv1.x);
    const float4 q3 = make_float4(tw.ng.x, tw.ng.y, tw.ng.z, uv1.y), q4 = make_float4(tw.frame.t.x, tw.frame.t.y, tw.frame.t.z, uv2.x);
    const float4 q5 = make_float4(tw.frame.s.x, tw.frame.s.y, tw.frame.s.z, uv2.y);
    const float4 q6 = make_float4(tw.area, u2f(inst_material(sc, m, meta)), u2f(inst), m[3].w);
    const float4 q7 = make_float4(tw.tt.x, tw.tt.y, tw.tt.z, u2f(meta >> 30));
    const bool has_normals = sc.in2.mesh_normals != nullptr;
    return surface_interaction_rows(m, q0, q1, q2, q3, q4, q5, q6, q7, has_normals, has_normals ? sc.in2.mesh_normals + (size_t)(mesh_base + prim) * 6 : nullptr, bary);
}
template <bool INST>
AKR_D SurfacePoint surface_interaction_any(const DScene& sc, uint32_t gid, vec2 bary) {
    if (INST) return surface_interaction_inst(sc, gid, bary);
    return surface_interaction(sc, gid, bary);
}

}  // namespace akr
