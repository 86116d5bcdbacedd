// disect.h -- ray/triangle test and the two scene intersectors (exhaustive for tiny scenes, BVH4 otherwise).
//
// Replaces LuisaCompute's rtx::Accel ray queries (crates/akari_render/src/scene.rs:88-185). Semantics kept from
// the reference: a candidate is rejected when its (inst, prim) equals one of the ray's two exclusion slots
// (scene.rs:98-100) or fails the stochastic alpha test (scene.rs:49-86); closest hit = smallest t.
// Added so that results do not depend on traversal order: ties in t go to the lowest global triangle id.
#pragma once
#include "drng.h"
#include "dscene.h"

namespace akr {

struct Hit {
    float t, u, v;
    uint32_t gid;
};

// Triangle test in Woop's precomputed-transform form. (r0|c0), (r1|c1), (r2|c2) map world space to the space in
// which the triangle is (0,0,0),(1,0,0),(0,1,0); 27 flops + one division, no cross products at run time.
struct PlaneHit {  // the plane solve: t and the hit point from the third row
    float t, px, py, pz;
};
AKR_HD PlaneHit tri_plane(vec3 o, vec3 d, float4 r2) {
    float dz = __builtin_fmaf(r2.x, d.x, __builtin_fmaf(r2.y, d.y, r2.z * d.z));
    float oz = __builtin_fmaf(r2.x, o.x, __builtin_fmaf(r2.y, o.y, __builtin_fmaf(r2.z, o.z, r2.w)));
    PlaneHit h;
    h.t = -oz / dz;
    h.px = __builtin_fmaf(h.t, d.x, o.x);
    h.py = __builtin_fmaf(h.t, d.y, o.y);
    h.pz = __builtin_fmaf(h.t, d.z, o.z);
    return h;
}
// the inside test: the two affine coordinates of the hit point in the triangle's frame, from the first two rows
AKR_HD void tri_uv(const PlaneHit& h, float4 r0, float4 r1, float& u, float& v) {
    u = __builtin_fmaf(r0.x, h.px, __builtin_fmaf(r0.y, h.py, __builtin_fmaf(r0.z, h.pz, r0.w)));
    v = __builtin_fmaf(r1.x, h.px, __builtin_fmaf(r1.y, h.py, __builtin_fmaf(r1.z, h.pz, r1.w)));
}
AKR_HD bool tri_test(vec3 o, vec3 d, float4 r0, float4 r1, float4 r2, float tmin, float tmax, float& t_out, float& u_out,
                     float& v_out) {
    PlaneHit h = tri_plane(o, d, r2);
    float u, v;
    tri_uv(h, r0, r1, u, v);
    t_out = h.t;
    u_out = u;
    v_out = v;
    return (h.t >= tmin) & (h.t <= tmax) & (u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f);
}

// SvmEvalMode::Alpha for a texture-fed base colour (principled.rs:15-21): the graph at the candidate's uv
// (surface_interaction_for_alpha_test, mesh.rs:426-485), alpha = w of the node feeding base_color. Inlined on purpose: a
// call inside the triangle loop makes the compiler give up the scalar (SGPR) path of the wave-uniform record loads.
AKR_D float textured_alpha(const DScene& sc, const float4* r, uint32_t material, float u, float v) {
    float w = 1.0f - u - v;
    vec2 uv = mk2((r[0].w * w + r[2].w * u) + r[4].w * v, (r[1].w * w + r[3].w * u) + r[5].w * v);
    const DMaterial& m = sc.materials[material];
    TexVal val[kMaxGraphNodes];
    eval_graph(sc.tex, m.tex_first_node, m.tex_n_nodes, uv, val);
    return val[m.tex_input[IN_BASE_COLOR]].w;
}
// scene.rs:49-86 for folded materials: alpha = alpha channel of the base-colour node
template <bool TEX>
AKR_D bool alpha_test(const DScene& sc, uint32_t gid, float u, float v) {
    const float4* r = sc.shade + (size_t)gid * SHADE_ROWS;
    float4 q6 = r[6];
    const DMaterial& m = sc.materials[f2u(q6.y)];
    float alpha = (m.kind == MAT_PRINCIPLED || m.kind == MAT_DIFFUSE) ? m.base_alpha : 1.0f;
    if (TEX) {  // only the TEX kernels carry the graph evaluation (and its scratch array)
        if (m.flags & MF_ALPHA_TEXTURED)
            alpha = textured_alpha(sc, r, f2u(q6.y), u, v);
    }
    if (alpha >= 1.0f) return true;
    uint32_t inst = f2u(q6.z);
    uint32_t prim = gid - sc.inst_tri_offset[inst];
    float h = (float)xxhash32_4(inst, prim, f2u(u), f2u(v)) * 2.3283064365386963e-10f;
    return alpha > h;
}

// The five conditions of the test as ONE number: inside the triangle and inside (0, tmax) <=> margin >= 0. For finite values
// this is the comparison chain of tri_test exactly (a - b >= 0 <=> a >= b in IEEE arithmetic, and the minimum of numbers
// is >= 0 iff all of them are); a NaN t makes u, v and every operand NaN, and NaN >= 0 is false, as in the chain. Why: each
// comparison of the chain leaves a lane mask in SGPRs and the masks are combined on the scalar unit, which the four SIMDs of a
// CU share -- at 60 scalar instructions per record the loops below were bound by that unit, not by the VALU.
AKR_D float next_up(float x) {  // the next float above a finite x
    uint32_t b = f2u(x);
    return x > 0.0f ? u2f(b + 1u) : (x < 0.0f ? u2f(b - 1u) : u2f(1u));
}
AKR_D float hit_margin(float t, float u, float v, float tmax) {
    return __builtin_fminf(__builtin_fminf(__builtin_fminf(u, v), 1.0f - (u + v)), __builtin_fminf(t, tmax - t));
}

// Exhaustive intersector: every lane of the wave walks the same triangle list, so the 48-byte records are
// wave-uniform and come in through the scalar cache (s_load_dwordx4 x3), leaving the VALU for the test itself.
template <bool ANY_HIT, bool TEX = false>
AKR_D bool trace_exhaustive(const DScene& sc, vec3 o, vec3 d, float tmin, float tmax, uint32_t ex0, uint32_t ex1, Hit& hit) {
    float best_t = next_up(tmax);  // see trace_pair_exhaustive
    uint32_t best = kInvalid;
    float best_u = 0.0f, best_v = 0.0f;
    const uint32_t n = sc.n_tris;
    // constant address space (4) + wave-uniform index => s_load_dwordx4 into SGPRs; the records are read-only for
    // the whole launch, which is what makes the scalar (non-coherent) cache legal here
    typedef const float __attribute__((address_space(4))) * ConstF;
    ConstF recs = (ConstF)(uintptr_t)sc.woop;
    auto load_rec = [&](uint32_t k, float4& a, float4& b, float4& c) {
        ConstF r = recs + 12 * (size_t)k;
        a = make_float4(r[0], r[1], r[2], r[3]);
        b = make_float4(r[4], r[5], r[6], r[7]);
        c = make_float4(r[8], r[9], r[10], r[11]);
    };
    // software prefetch: the record of triangle k+1 is requested before triangle k is tested, so the scalar-cache
    // latency overlaps the ~40 VALU instructions of the test (the buffer is padded by one record, scene_build.cpp)
    float4 n0, n1, n2;
    load_rec(0, n0, n1, n2);
    PlaneHit ph{0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t k = 0; k < n; k++) {
        const float4 r0 = n0, r1 = n1, r2 = n2;
        load_rec(k + 1, n0, n1, n2);
        if (!((sc.plane_share_mask >> k) & 1ull)) ph = tri_plane(o, d, r2);  // wave-uniform: one solve per coplanar pair of records
        float u, v;
        const float t = ph.t;
        tri_uv(ph, r0, r1, u, v);
        float m = __builtin_fminf(hit_margin(t, u, v, tmax), t - tmin);
        m = (k == ex0) ? -1.0f : m;
        m = (k == ex1) ? -1.0f : m;
        if (sc.has_alpha) {
            if (m >= 0.0f && !alpha_test<TEX>(sc, k, u, v)) m = -1.0f;
        }
        if (ANY_HIT) {
            if (m >= 0.0f) best = k;
            if (__builtin_amdgcn_ballot_w64(best == kInvalid) == 0) break;  // every lane of the wave is occluded
        } else {
            // ascending k: a strict '<' keeps the lowest id among equal t
            const float tc = (m >= 0.0f) ? t : __builtin_inff();
            const bool better = tc < best_t;
            best_t = better ? tc : best_t;
            best_u = better ? u : best_u;
            best_v = better ? v : best_v;
            best = better ? k : best;
        }
    }
    hit.t = best != kInvalid ? best_t : tmax;
    hit.u = best_u;
    hit.v = best_v;
    hit.gid = best;
    return best != kInvalid;
}

// Exhaustive intersector for a PAIR of rays per lane: the closest-hit ray of the next path vertex and the shadow ray
// of the current one are both known once a vertex has been shaded, so one walk over the (wave-uniform, scalar-cache
// resident) records serves both: half the scalar loads and loop overhead of two separate walks, and two independent
// dependency chains per record for the VALU to overlap. A ray that does not exist for a lane is passed with
// tmax < tmin and can never hit.
template <bool TEX = false, bool UNROLL = false>
AKR_D void trace_pair_exhaustive(const DScene& sc, vec3 o, vec3 d, float tmax, uint32_t ex0, vec3 so, vec3 sd, float stmax,
                                 uint32_t sex0, uint32_t sex1, Hit& hit, bool& found, bool& occluded) {
    // best_t starts one ulp above tmax: "t < best_t" then admits a first hit at t == tmax and keeps, among equal t, the
    // lowest id afterwards (ascending k, strict '<') without a separate "no hit yet" test
    float best_t = next_up(tmax);
    uint32_t best = kInvalid;
    float best_u = 0.0f, best_v = 0.0f;
    float occ_margin = -1.0f;  // max over the records of the shadow ray's margin: >= 0 <=> something occludes
    const uint32_t n = sc.n_tris;
    typedef const float __attribute__((address_space(4))) * ConstF;
    ConstF recs = (ConstF)(uintptr_t)sc.woop;
    auto load_rec = [&](uint32_t k, float4& a, float4& b, float4& c) {
        ConstF r = recs + 12 * (size_t)k;
        a = make_float4(r[0], r[1], r[2], r[3]);
        b = make_float4(r[4], r[5], r[6], r[7]);
        c = make_float4(r[8], r[9], r[10], r[11]);
    };
    PlaneHit ph{0.0f, 0.0f, 0.0f, 0.0f}, sph{0.0f, 0.0f, 0.0f, 0.0f};
    auto record = [&](uint32_t k, const float4& r0, const float4& r1, const float4& r2) {
        if (!((sc.plane_share_mask >> k) & 1ull)) {  // wave-uniform: one plane solve per ray per coplanar pair of records
            ph = tri_plane(o, d, r2);
            sph = tri_plane(so, sd, r2);
        }
        float u, v, su, sv;
        const float t = ph.t, st = sph.t;
        tri_uv(ph, r0, r1, u, v);
        tri_uv(sph, r0, r1, su, sv);
        float m = hit_margin(t, u, v, tmax), sm = hit_margin(st, su, sv, stmax);
        m = (k == ex0) ? -1.0f : m;
        sm = (k == sex0) ? -1.0f : sm;
        sm = (k == sex1) ? -1.0f : sm;
        if (sc.has_alpha) {
            // k through readfirstlane: opaque to loop strength reduction, which otherwise keeps this block's record pointer and
            // hash constant as induction variables updated on the scalar unit every trip, alpha or not
            const uint32_t ka = __builtin_amdgcn_readfirstlane(k);
            if (m >= 0.0f && !alpha_test<TEX>(sc, ka, u, v)) m = -1.0f;
            if (sm >= 0.0f && !alpha_test<TEX>(sc, ka, su, sv)) sm = -1.0f;
        }
        const float tc = (m >= 0.0f) ? t : __builtin_inff();
        const bool better = tc < best_t;
        best_t = better ? tc : best_t;
        best_u = better ? u : best_u;
        best_v = better ? v : best_v;
        best = better ? k : best;
        occ_margin = sm > occ_margin ? sm : occ_margin;  // (not fmaxf: that costs two canonicalising v_max per record)
    };
    if (UNROLL) {
        // two records per trip on alternating register sets: the one-record software prefetch without the scalar moves that
        // rotating a single pair of sets costs per record (buffer padded by two records). +5 % in the small force_diffuse
        // kernel, -3 % in the full-graph kernel, whose 111 KB of code already overflow the instruction cache.
        float4 a0, a1, a2, b0, b1, b2;
        load_rec(0, a0, a1, a2);
        uint32_t k = 0;
        for (; k + 1 < n; k += 2) {
            load_rec(k + 1, b0, b1, b2);
            record(k, a0, a1, a2);
            load_rec(k + 2, a0, a1, a2);
            record(k + 1, b0, b1, b2);
        }
        if (k < n) record(k, a0, a1, a2);
    } else {
        float4 n0, n1, n2;
        load_rec(0, n0, n1, n2);
        for (uint32_t k = 0; k < n; k++) {
            const float4 r0 = n0, r1 = n1, r2 = n2;
            load_rec(k + 1, n0, n1, n2);  // prefetch
            record(k, r0, r1, r2);
        }
    }
    found = best != kInvalid;
    hit.t = found ? best_t : tmax;
    hit.u = best_u;
    hit.v = best_v;
    hit.gid = best;
    occluded = occ_margin >= 0.0f;
}

// ------------------------------------------------------------------------------------------------------------
// BVH4 traversal (one ray per lane, while-while). Node layout: host/bvh.cpp. The per-lane stack lives in LDS,
// strided by the workgroup size so that lane i of every wave touches bank i (no conflicts): 4 B x depth x 256.
constexpr uint32_t kBvhStackDepth = 32;
constexpr uint32_t kBvhLeafBit = 0x80000000u;
constexpr uint32_t kBvhDone = 0xfffffffeu;

struct TraceCounters {
    uint32_t nodes, tris, overflow;
};

// 1/d for the slab test, |d| floored at 1e-20 so that 0 * inf never appears. The box test only culls: it does not
// have to follow the AKR-F32 contract (the oracle has no BVH), it only has to be conservative, which the padding of
// the boxes (host/bvh.cpp) guarantees with a margin of ~10^4 ulp; so it may use v_rcp_f32, fma and v_min3/v_max3.
AKR_D float safe_inv(float d) {
    float a = abs_f(d) < 1e-20f ? __builtin_copysignf(1e-20f, d) : d;
    return __builtin_amdgcn_rcpf(a);
}
// One BVH4 node: 64 bytes = 4 x 16-byte loads (host/bvh.cpp). Child boxes are 8-bit offsets from the node's own
// (padded) lower corner in units of a per-axis power of two, rounded outwards, so the decoded boxes contain the exact
// ones: lo = origin + q_lo * 2^e, hi = origin + q_hi * 2^e. With t = plane * inv + noi the slab distances become
// t = q * (2^e * inv) + (origin * inv + noi): one v_cvt_f32_ubyte + one fma per plane.
// Writes the entry distance of every child the ray enters within [tmin, tlimit] (inf otherwise) and the child refs.
AKR_D void bvh4_node_test(const DScene& sc, uint32_t node, vec3 inv, vec3 noi, float tmin, float tlimit, float tn[4], uint32_t ch[4]) {
    const uint4* n = (const uint4*)sc.bvh_nodes + (size_t)node * 4;
    const uint4 r0 = n[0], r1 = n[1], r2 = n[2], r3 = n[3];
    const float sx = u2f((r0.w & 0xffu) << 23), sy = u2f(((r0.w >> 8) & 0xffu) << 23), sz = u2f(((r0.w >> 16) & 0xffu) << 23);
    const float ax = __builtin_fmaf(u2f(r0.x), inv.x, noi.x), ay = __builtin_fmaf(u2f(r0.y), inv.y, noi.y), az = __builtin_fmaf(u2f(r0.z), inv.z, noi.z);
    const float bx = sx * inv.x, by = sy * inv.y, bz = sz * inv.z;
    ch[0] = r2.z; ch[1] = r2.w; ch[2] = r3.x; ch[3] = r3.y;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float lx = (float)((r1.x >> (8 * i)) & 0xffu), ly = (float)((r1.y >> (8 * i)) & 0xffu), lz = (float)((r1.z >> (8 * i)) & 0xffu);
        const float hx = (float)((r1.w >> (8 * i)) & 0xffu), hy = (float)((r2.x >> (8 * i)) & 0xffu), hz = (float)((r2.y >> (8 * i)) & 0xffu);
        float t0x = __builtin_fmaf(lx, bx, ax), t1x = __builtin_fmaf(hx, bx, ax);
        float t0y = __builtin_fmaf(ly, by, ay), t1y = __builtin_fmaf(hy, by, ay);
        float t0z = __builtin_fmaf(lz, bz, az), t1z = __builtin_fmaf(hz, bz, az);
        float near = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                     __builtin_fmaxf(__builtin_fminf(t0z, t1z), tmin));
        float far = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                                    __builtin_fminf(__builtin_fmaxf(t0z, t1z), tlimit));
        tn[i] = ((near <= far) & (ch[i] != kInvalid)) ? near : __builtin_inff();  // empty slots carry ref 0xffffffff
    }
}

template <bool ANY_HIT, bool TEX = false>
AKR_D bool trace_bvh4(const DScene& sc, vec3 o, vec3 d, float tmin, float tmax, uint32_t ex0, uint32_t ex1, Hit& hit,
                      uint32_t* __restrict__ stack, TraceCounters& cnt) {
    const vec3 inv = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    const vec3 noi = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);  // t = plane * inv + noi
    float best_t = tmax;
    uint32_t best = kInvalid;
    float best_u = 0.0f, best_v = 0.0f;
    uint32_t sp = 0;
    uint32_t cur = 0;  // root is always an inner node
#define AKR_PUSH(ref)                                  \
    {                                                  \
        if (sp < kBvhStackDepth) {                     \
            stack[sp * 256u] = (ref);                  \
            sp++;                                      \
        } else {                                       \
            cnt.overflow = 1;                          \
        }                                              \
    }
    for (;;) {
        while (!(cur & kBvhLeafBit)) {
            cnt.nodes++;
            float tn[4];
            uint32_t ch[4];
            bvh4_node_test(sc, cur, inv, noi, tmin, best_t, tn, ch);
            if (!ANY_HIT) {
                // sort the four (tn, ch) pairs ascending with a 5-comparator network
#define AKR_CSWAP(a, b)                                          \
    {                                                            \
        bool sw = tn[b] < tn[a];                                 \
        float tf = sw ? tn[b] : tn[a], tg = sw ? tn[a] : tn[b];  \
        uint32_t cf = sw ? ch[b] : ch[a], cg = sw ? ch[a] : ch[b]; \
        tn[a] = tf; tn[b] = tg; ch[a] = cf; ch[b] = cg;          \
    }
                AKR_CSWAP(0, 1) AKR_CSWAP(2, 3) AKR_CSWAP(0, 2) AKR_CSWAP(1, 3) AKR_CSWAP(1, 2)
#undef AKR_CSWAP
                // push far-to-near so that the nearest is popped first
                if (tn[3] < __builtin_inff()) AKR_PUSH(ch[3])
                if (tn[2] < __builtin_inff()) AKR_PUSH(ch[2])
                if (tn[1] < __builtin_inff()) AKR_PUSH(ch[1])
                if (tn[0] < __builtin_inff()) {
                    cur = ch[0];
                } else if (sp > 0) {
                    sp--; cur = stack[sp * 256u];
                } else {
                    cur = kBvhDone;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (tn[i] < __builtin_inff()) AKR_PUSH(ch[i])
                if (sp > 0) { sp--; cur = stack[sp * 256u]; } else { cur = kBvhDone; }
            }
        }
        if (cur == kBvhDone) break;
        {   // leaf: up to 4 triangles, 48 B each
            const uint32_t first = cur & 0x0fffffffu, count = (cur >> 28) & 7u;
            for (uint32_t i = 0; i < count; i++) {
                const uint32_t k = first + i;
                float4 r0 = sc.woop[3 * (size_t)k + 0], r1 = sc.woop[3 * (size_t)k + 1], r2 = sc.woop[3 * (size_t)k + 2];
                cnt.tris++;
                float t, u, v;
                bool h = tri_test(o, d, r0, r1, r2, tmin, tmax, t, u, v);
                if (h) {
                    uint32_t gid = sc.tri_gid[k];
                    h = (gid != ex0) & (gid != ex1);
                    if (h && sc.has_alpha) h = alpha_test<TEX>(sc, gid, u, v);
                    if (h) {
                        if (ANY_HIT) {
                            best = gid;
                        } else {
                            bool better = (best == kInvalid) | (t < best_t) | ((t == best_t) & (gid < best));
                            if (better) { best_t = t; best_u = u; best_v = v; best = gid; }
                        }
                    }
                }
            }
            if (ANY_HIT && best != kInvalid) break;
            if (sp > 0) { sp--; cur = stack[sp * 256u]; } else break;
        }
    }
#undef AKR_PUSH
    hit.t = best_t; hit.u = best_u; hit.v = best_v; hit.gid = best;
    return best != kInvalid;
}

}  // namespace akr
