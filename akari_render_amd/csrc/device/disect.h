// disect.h -- ray/triangle test and the two scene intersectors (exhaustive for tiny scenes, 6-wide compressed BVH of 64-byte nodes otherwise).
//
// Replaces LuisaCompute's rtx::Accel ray queries (crates/akari_render/src/scene.rs:88-185). Semantics kept from
// the reference: a candidate is rejected when its (inst, prim) equals one of the ray's two exclusion slots
// (scene.rs:98-100) or fails the stochastic alpha test (scene.rs:49-86); closest hit = smallest t.
// Added so that results do not depend on traversal order: ties in t go to the lowest global triangle id.
#pragma once
#include "drng.h"
#include "dscene.h"

namespace akr {

// compile-time flags passed to generic lambdas (the headers keep clear of <type_traits>: they are also compiled by hiprtc, which has no host headers)
template <bool V>
struct BoolTag {
    static constexpr bool value = V;
};

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
    h.t = div_f(-oz, dz);
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
    return material_alpha_at(sc.tex, sc.materials[material], material, uv);  // = w of the node feeding base_color
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
    // (v_min3_f32 instead of two v_min_f32 was measured in round 4: a min3 costs two issue slots, -1.3 %; profiles/r4_ab_walk.txt)
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
//
// WALK selects where a record's coefficients sit when the VALU reads them (measured variants):
//   0  SGPRs: s_load through the scalar cache, every fma of the affine rows names one SGPR operand. A SIMD accepts one
//      scalar-operand VALU instruction per ~4.3 cycles against ~2.15 for a VGPR-only one (tools/micro/valu_rate.hip), and the
//      rows come 16 such instructions in a row.
//   1  VGPRs: the records are staged in LDS with the shading tables and every lane reads the same address (a broadcast
//      ds_read_b128, 4 LDS cycles per wave and row); all of the test's arithmetic is then VGPR-only.
//   2  SGPRs, but the two rays of the pair go through the rows as ONE packed instruction (v_pk_fma_f32 with the coefficient
//      broadcast): half as many scalar-operand instructions; a packed f32 op holds the VALU for two slots either way.
//   3  records in LDS, every fma of plane solve and inside test packed over the two rays, the coefficient broadcast from one half
//      of a register pair by op_sel (no copies, no register-bank conflicts: 417 instead of 535 modelled issue cycles per two records,
//      tools/valu_cost_report.py) -- and 3 % SLOWER than 1 on C2 and C3: half as many independent instructions per wave.
//   4  records in LDS, scalar fmas as in 1, with the cheaper bookkeeping of 3: the alpha test unswitched out of the loop (inside
//      it the wave-uniform flag cost every record a v_cndmask + v_cmp), min / max without the compiler's canonicalising v_max x, x,
//      the shadow ray's margin folded with one v_max, and only (t, id) of the best hit selected per record -- its (u, v) are
//      recomputed once after the walk. +4.2 % on C2, +3.3 % on C3 over 1 (same box, profiles/r4_ab_walk.txt). The default.
// The arithmetic (operation order, fma placement) is identical in all of them: films do not change.
typedef float v2f __attribute__((ext_vector_type(2)));
AKR_D v2f pk_fma(float a, v2f b, v2f c) { return __builtin_elementwise_fma((v2f){a, a}, b, c); }
AKR_D v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// WALK 3: v_pk_fma_f32 / v_pk_mul_f32 with ONE half of a 64-bit register pair feeding both lanes (op_sel / op_sel_hi): the
// coefficient of a record row, read from LDS as part of a 128-bit quad, multiplies the closest-hit ray's value in the low lane and
// the shadow ray's in the high lane without ever being copied. Each lane is the IEEE fma / mul the scalar instruction computes.
//   SA / SC = which half (0 = low, 1 = high) of `a` / `c` is broadcast.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int SA>
AKR_D v2f pk_fma_b(v2f a, v2f b, v2f c) {  // {a[SA], a[SA]} * b + c
    v2f r;
    if (SA == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int SA, int SC>
AKR_D v2f pk_fma_bb(v2f a, v2f b, v2f c) {  // {a[SA], a[SA]} * b + {c[SC], c[SC]}
    v2f r;
    if (SA == 0 && SC == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if (SA == 0 && SC == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if (SA == 1 && SC == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int SA>
AKR_D v2f pk_mul_b(v2f a, v2f b) {  // {a[SA], a[SA]} * b
    v2f r;
    if (SA == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// IEEE minNum / maxNum as the instruction computes them, without the v_max x, x the compiler puts in front of fminf / fmaxf to quiet
// a possible signalling NaN (4.4 cycles of the SIMD each, tools/micro/vgpr_bank.hip; arithmetic never produces one)
AKR_D float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
AKR_D float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool TEX = false, bool UNROLL = false, int WALK = 0>
AKR_D void trace_pair_exhaustive(const DScene& sc, vec3 o, vec3 d, float tmax, uint32_t ex0, vec3 so, vec3 sd, float stmax,
                                 uint32_t sex0, uint32_t sex1, Hit& hit, bool& found, bool& occluded, const float4* lds_recs = nullptr) {
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
        if (WALK == 1) {  // the same LDS address in every lane: broadcast reads
            const float4* r = lds_recs + 3 * (size_t)k;
            a = r[0]; b = r[1]; c = r[2];
            return;
        }
        ConstF r = recs + 12 * (size_t)k;
        a = make_float4(r[0], r[1], r[2], r[3]);
        b = make_float4(r[4], r[5], r[6], r[7]);
        c = make_float4(r[8], r[9], r[10], r[11]);
    };
    PlaneHit ph{0.0f, 0.0f, 0.0f, 0.0f}, sph{0.0f, 0.0f, 0.0f, 0.0f};
    const v2f ox2 = {o.x, so.x}, oy2 = {o.y, so.y}, oz2 = {o.z, so.z}, dx2 = {d.x, sd.x}, dy2 = {d.y, sd.y}, dz2 = {d.z, sd.z};
    v2f hx2 = {0.0f, 0.0f}, hy2 = {0.0f, 0.0f}, hz2 = {0.0f, 0.0f};  // WALK 2: the two rays' hit points on the current plane
    auto record = [&](uint32_t k, const float4& r0, const float4& r1, const float4& r2) {
        float u, v, su, sv;
        if (WALK == 2) {
            // tri_plane / tri_uv for both rays at once, component 0 = closest-hit ray, 1 = shadow ray; the same fma chains
            if (!((sc.plane_share_mask >> k) & 1ull)) {
                const v2f den = pk_fma(r2.x, dx2, pk_fma(r2.y, dy2, (v2f){r2.z, r2.z} * dz2));
                const v2f num = pk_fma(r2.x, ox2, pk_fma(r2.y, oy2, pk_fma(r2.z, oz2, (v2f){r2.w, r2.w})));
                ph.t = div_f(-num.x, den.x);
                sph.t = div_f(-num.y, den.y);
                const v2f t2 = {ph.t, sph.t};
                hx2 = pk_fma(t2, dx2, ox2); hy2 = pk_fma(t2, dy2, oy2); hz2 = pk_fma(t2, dz2, oz2);
            }
            const v2f u2 = pk_fma(r0.x, hx2, pk_fma(r0.y, hy2, pk_fma(r0.z, hz2, (v2f){r0.w, r0.w})));
            const v2f v2 = pk_fma(r1.x, hx2, pk_fma(r1.y, hy2, pk_fma(r1.z, hz2, (v2f){r1.w, r1.w})));
            u = u2.x; su = u2.y; v = v2.x; sv = v2.y;
        } else {
            if (!((sc.plane_share_mask >> k) & 1ull)) {  // wave-uniform: one plane solve per ray per coplanar pair of records
                ph = tri_plane(o, d, r2);
                sph = tri_plane(so, sd, r2);
            }
            tri_uv(ph, r0, r1, u, v);
            tri_uv(sph, r0, r1, su, sv);
        }
        const float t = ph.t, st = sph.t;
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
    if (WALK == 3 || WALK == 4 || WALK == 5) {  // 4: the same loop with scalar fmas instead of packed ones; 5: 4 + lockstep pairs
        // Both rays of the pair through every fma of the plane solve and of the inside test as ONE packed instruction. A scalar
        // v_fma_f32 costs its SIMD 2.2 cycles unless its three source registers all have the same parity (the register file's two
        // banks): then 4.4 -- and which registers a value lands in is the allocator's choice, 13 of the 76 three-source instructions
        // of the two-record trip were hit (tools/valu_cost_report.py). A v_pk_fma_f32 costs 4.4 for its two results whatever its
        // registers are. The records' rows are read from LDS as 128-bit quads; a row's coefficient is one half of a 64-bit register
        // pair and is broadcast to both lanes by op_sel, the rays' components are pairs {closest-hit ray, shadow ray} throughout.
        // The comparisons of a record are cut down too: a v_cmp costs 4.4 cycles and a v_cndmask on its result 3.7.
        const v4f* lrec = (const v4f*)lds_recs;
        const bool has_alpha = __builtin_amdgcn_readfirstlane((int)sc.has_alpha) != 0;  // (a scalar branch per record, nothing per lane)
        const v2f tmax2 = {tmax, stmax};
        v2f T2 = {0.0f, 0.0f};
        auto record3 = [&](auto alpha_tag, auto shared_tag, uint32_t k, const v4f& q0, const v4f& q1, const v4f& q2) {
            constexpr bool ALPHA = decltype(alpha_tag)::value;
            constexpr bool SHARED = decltype(shared_tag)::value;  // the caller knows that this record shares the plane solved last
            const v2f r0xy = __builtin_shufflevector(q0, q0, 0, 1), r0zw = __builtin_shufflevector(q0, q0, 2, 3);
            const v2f r1xy = __builtin_shufflevector(q1, q1, 0, 1), r1zw = __builtin_shufflevector(q1, q1, 2, 3);
            v2f u2, v2;
            if (WALK == 4 || WALK == 5) {
                if (!SHARED && !((sc.plane_share_mask >> k) & 1ull)) {
                    const PlaneHit a = tri_plane(o, d, make_float4(q2.x, q2.y, q2.z, q2.w)), b = tri_plane(so, sd, make_float4(q2.x, q2.y, q2.z, q2.w));
                    T2 = (v2f){a.t, b.t};
                    hx2 = (v2f){a.px, b.px}; hy2 = (v2f){a.py, b.py}; hz2 = (v2f){a.pz, b.pz};
                }
                float u, v, su, sv;
                tri_uv(PlaneHit{T2.x, hx2.x, hy2.x, hz2.x}, make_float4(q0.x, q0.y, q0.z, q0.w), make_float4(q1.x, q1.y, q1.z, q1.w), u, v);
                tri_uv(PlaneHit{T2.y, hx2.y, hy2.y, hz2.y}, make_float4(q0.x, q0.y, q0.z, q0.w), make_float4(q1.x, q1.y, q1.z, q1.w), su, sv);
                u2 = (v2f){u, su}; v2 = (v2f){v, sv};
            } else {
            if (!SHARED && !((sc.plane_share_mask >> k) & 1ull)) {
                const v2f r2xy = __builtin_shufflevector(q2, q2, 0, 1), r2zw = __builtin_shufflevector(q2, q2, 2, 3);
                // tri_plane for both rays: den = fma(x, d.x, fma(y, d.y, z * d.z)), num = fma(x, o.x, fma(y, o.y, fma(z, o.z, w)))
                const v2f den = pk_fma_b<0>(r2xy, dx2, pk_fma_b<1>(r2xy, dy2, pk_mul_b<0>(r2zw, dz2)));
                const v2f num = pk_fma_b<0>(r2xy, ox2, pk_fma_b<1>(r2xy, oy2, pk_fma_bb<0, 1>(r2zw, oz2, r2zw)));
                T2 = (v2f){div_f(-num.x, den.x), div_f(-num.y, den.y)};
                hx2 = pk_fma(T2, dx2, ox2); hy2 = pk_fma(T2, dy2, oy2); hz2 = pk_fma(T2, dz2, oz2);
            }
            // tri_uv: u = fma(r0.x, p.x, fma(r0.y, p.y, fma(r0.z, p.z, r0.w))), v likewise from r1
            u2 = pk_fma_b<0>(r0xy, hx2, pk_fma_b<1>(r0xy, hy2, pk_fma_bb<0, 1>(r0zw, hz2, r0zw)));
            v2 = pk_fma_b<0>(r1xy, hx2, pk_fma_b<1>(r1xy, hy2, pk_fma_bb<0, 1>(r1zw, hz2, r1zw)));
            }
            // hit_margin: min(min(min(u, v), 1 - (u + v)), min(t, tmax - t)) per ray
            v2f s2, w2;
            if (WALK == 4 || WALK == 5) {
                s2 = (v2f){1.0f - (u2.x + v2.x), 1.0f - (u2.y + v2.y)};
                w2 = (v2f){tmax - T2.x, stmax - T2.y};
            } else {
                s2 = (v2f){1.0f, 1.0f} - (u2 + v2);
                w2 = tmax2 - T2;
            }
            float m = min_raw(min_raw(min_raw(u2.x, v2.x), s2.x), min_raw(T2.x, w2.x));
            float sm = min_raw(min_raw(min_raw(u2.y, v2.y), s2.y), min_raw(T2.y, w2.y));
            m = (k == ex0) ? -1.0f : m;
            sm = (k == sex0) ? -1.0f : sm;
            sm = (k == sex1) ? -1.0f : sm;
            if (ALPHA) {
                const uint32_t ka = __builtin_amdgcn_readfirstlane(k);
                if (m >= 0.0f && !alpha_test<TEX>(sc, ka, u2.x, v2.x)) m = -1.0f;
                if (sm >= 0.0f && !alpha_test<TEX>(sc, ka, u2.y, v2.y)) sm = -1.0f;
            }
            // only the distance and the id of the best hit are tracked here; its (u, v) are recomputed after the walk (below)
            const float tc = (m >= 0.0f) ? T2.x : __builtin_inff();
            const bool better = tc < best_t;
            best_t = better ? tc : best_t;
            best = better ? k : best;
            // max over the records of the shadow ray's margin (a NaN margin leaves it as it is, like the comparison it replaces; only
            // the sign of the result is read)
            occ_margin = max_raw(occ_margin, sm);
        };
        // The loop exists twice, with and without the alpha test of a candidate: left as a run-time branch inside ONE loop, the
        // (wave-uniform) flag is turned into a lane mask and back by every record of every scene -- a v_cndmask and a v_cmp, 8.8
        // cycles of 240 -- to merge the two values of the margins behind the branch.
        auto walk3 = [&](auto alpha_tag) {
            v4f a0 = lrec[0], a1 = lrec[1], a2 = lrec[2], b0, b1, b2;
            uint32_t k = 0;
            for (; k + 1 < n; k += 2) {
                b0 = lrec[3 * (k + 1)]; b1 = lrec[3 * (k + 1) + 1]; b2 = lrec[3 * (k + 1) + 2];
                if (WALK == 5 && ((sc.plane_share_mask >> (k + 1)) & 1ull)) {
                    // the two triangles of a quad: one plane solve, then both inside tests in ONE basic block (no branch between
                    // them), so that the scheduler can interleave the two records' chains
                    record3(alpha_tag, BoolTag<false>{}, k, a0, a1, a2);
                    a0 = lrec[3 * (k + 2)]; a1 = lrec[3 * (k + 2) + 1]; a2 = lrec[3 * (k + 2) + 2];
                    record3(alpha_tag, BoolTag<true>{}, k + 1, b0, b1, b2);
                } else {
                    record3(alpha_tag, BoolTag<false>{}, k, a0, a1, a2);
                    a0 = lrec[3 * (k + 2)]; a1 = lrec[3 * (k + 2) + 1]; a2 = lrec[3 * (k + 2) + 2];
                    record3(alpha_tag, BoolTag<false>{}, k + 1, b0, b1, b2);
                }
            }
            if (k < n) record3(alpha_tag, BoolTag<false>{}, k, a0, a1, a2);
        };
        if (has_alpha) walk3(BoolTag<true>{});
        else walk3(BoolTag<false>{});
        // (u, v) of the closest hit, once per walk instead of two selects per record: the record's rows from LDS again (a per-lane
        // address now) and the hit point o + t d with the t the walk kept -- the very operations on the very operands of the walk
        // (hx2.x = fma(T2.x, d.x, o.x), u2.x = the fma chain over row 0), so the bits are the walk's.
        if (best != kInvalid) {
            const v4f q0 = lrec[3 * best], q1 = lrec[3 * best + 1];
            const float px = __builtin_fmaf(best_t, d.x, o.x), py = __builtin_fmaf(best_t, d.y, o.y), pz = __builtin_fmaf(best_t, d.z, o.z);
            best_u = __builtin_fmaf(q0.x, px, __builtin_fmaf(q0.y, py, __builtin_fmaf(q0.z, pz, q0.w)));
            best_v = __builtin_fmaf(q1.x, px, __builtin_fmaf(q1.y, py, __builtin_fmaf(q1.z, pz, q1.w)));
        }
    } else if (UNROLL) {
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
// Compressed wide-BVH traversal: six children in eight octant positions, 64-byte nodes (layout and builder: host/bvh.cpp; after
// Ylitie, Karras & Laine, HPG 2017).
//
// One ray per lane. A lane's state is a NODE GROUP G = child_base (24 bits) | hit bits of the (at most 6) sibling nodes in their 8 octant
// positions, in visiting order (bits 24..31), a TRIANGLE GROUP (tbase, T) = the pending triangles of the node visited last (at most 18:
// six leaves of three; 24 bits), and a
// stack of node groups in LDS (strided by the workgroup size: lane i of every wave touches bank i). A node's children are
// tested together; those the ray enters become the new G, ordered by octant: slot s sits at bit 24 + (s ^ octinv), the
// highest bit is the nearest child, so "pop the nearest" is one count-leading-zeros and nothing is sorted. The siblings
// left over go to the stack as ONE entry: a traversal holds at most one entry per tree level and the stack
// (kBvhStackDepth levels, checked against the tree's depth when the scene is built) cannot overflow.
//
// Every step a lane does ONE thing -- test its next pending triangle, or fetch and test its next node -- and both kinds
// of lane fetch through the SAME four 16-byte loads from a per-lane address (64-byte triangle record: Woop rows + global
// id; 64-byte node: one memory sector each). The wave waits once per step whatever mix of nodes and triangles its lanes are at: on the
// 10 M-triangle hall round 1's while-while loop over a 4-wide BVH ran at 26 % lane utilisation, all of it waiting on dependent
// fetches.
// The box test only culls and does not have to follow the AKR-F32 contract (the oracle has no BVH): it must be conservative,
// which the padding of the boxes (host/bvh.cpp) guarantees; so it may use v_rcp_f32, fma and min3 / max3.
constexpr uint32_t kBvhDone = 0xfffffffeu;

struct TraceCounters {
    uint32_t nodes, tris, overflow;
};

// 1/d for the slab test, |d| floored at 1e-20 so that 0 * inf never appears.
AKR_D float safe_inv(float d) {
    float a = abs_f(d) < 1e-20f ? __builtin_copysignf(1e-20f, d) : d;
    return __builtin_amdgcn_rcpf(a);
}
AKR_D uint32_t byte_of(uint32_t w, int i) { return (w >> (8 * i)) & 0xffu; }

struct Trav {  // one ray in flight
    vec3 o, d, inv, noi;   // t = plane * inv + noi
    float tmin, tmax, best_t, best_u, best_v;
    uint32_t ex0, ex1, best;
    uint32_t G, T, tbase, sp, octinv4;
    bool active;
};
AKR_D void trav_begin(Trav& s, vec3 o, vec3 d, float tmin, float tmax, uint32_t ex0, uint32_t ex1) {
    s.o = o; s.d = d;
    s.inv = mk3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    s.noi = mk3(-o.x * s.inv.x, -o.y * s.inv.y, -o.z * s.inv.z);
    s.tmin = tmin; s.tmax = tmax;
    s.best_t = tmax; s.best_u = 0.0f; s.best_v = 0.0f; s.best = kInvalid;
    s.ex0 = ex0; s.ex1 = ex1;
    // octinv: bit a set = the ray travels towards +a, i.e. meets the children on the low side of axis a first
    const uint32_t oi = (s.inv.x >= 0.0f ? 1u : 0u) | (s.inv.y >= 0.0f ? 2u : 0u) | (s.inv.z >= 0.0f ? 4u : 0u);
    s.octinv4 = oi * 0x01010101u;
    s.G = 1u << (24u + oi);  // the group {root}: base 0, slot 0 at bit 24 + (0 ^ oi)
    s.T = 0; s.tbase = 0; s.sp = 0;
    s.active = tmax >= tmin;
}

// One step of one lane: a triangle test if one is pending, else the next node. MODE 0: closest hit, 1: any hit, 2: `any_rt`
// decides per lane (the wavefront schedule traces both kinds of ray in one loop).
// TILE: the launch keeps the first sc.bvh_tile_nodes nodes (the top levels, breadth-first order: host/bvh.cpp) in LDS at `tile`;
// a lane whose next node is one of them reads it with four ds_read_b128 instead of going through the texture addresser and L1 --
// on the 10 M-triangle hall the traversal keeps that path 58 % busy, and every ray starts with three to five such nodes.
// The same step in two stages: a lane that leaves the node test with leaf triangles tests the first one in the SAME step (two
// dependent fetches per step, a fifth fewer steps per ray). Same visits in the same order. Measured (round 5): BVH kernel of the
// textured room 602 -> 630 Msamples/s, 1 M-triangle hall unchanged, 10 M-triangle hall 317 -> 299: used by the kernels of scenes with
// textures only (AKR_BVH_STAGED: 0 never, 1 those, 2 all).
#ifndef AKR_BVH_STAGED
#define AKR_BVH_STAGED 1
#endif
template <int MODE, bool TEX, bool TILE = false>
AKR_D void trav_step_staged(const DScene& sc, Trav& s, uint32_t* __restrict__ stack, TraceCounters& cnt, bool any_rt = false, const uint4* tile = nullptr) {
    const bool any_hit = MODE == 2 ? any_rt : (MODE == 1);
    if (s.T == 0) {
        if ((s.G >> 24) == 0) {  // the caller guarantees sp > 0 here
            s.sp--;
            s.G = stack[s.sp * 256u];
        }
        const uint32_t j = 31u - (uint32_t)__builtin_clz(s.G);
        s.G &= ~(1u << j);
        if ((s.G >> 24) != 0) {
            if (s.sp < sc.bvh_stack_depth) {
                stack[s.sp * 256u] = s.G;
                s.sp++;
            } else {
                cnt.overflow = 1;
            }
        }
        const uint32_t slot = (j - 24u) ^ (s.octinv4 & 7u);
        const uint32_t idx = (s.G & 0xffffffu) + slot;
        const uint4* p = sc.bvh_nodes + (size_t)idx * (kBvhNodeWords / 4);
        uint4 w0, w1, w2, w3;
        if (TILE && idx < sc.bvh_tile_nodes) {
            typedef const volatile uint32_t __attribute__((address_space(3))) * LdsW;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef const volatile u32x4 __attribute__((address_space(3))) * LdsU4;
            LdsU4 pt = (LdsU4)((LdsW)(const uint32_t*)tile + idx * kBvhNodeWords);
            const u32x4 t0 = pt[0], t1 = pt[1], t2 = pt[2], t3 = pt[3];
            w0 = make_uint4(t0.x, t0.y, t0.z, t0.w); w1 = make_uint4(t1.x, t1.y, t1.z, t1.w); w2 = make_uint4(t2.x, t2.y, t2.z, t2.w);
            w3 = make_uint4(t3.x, t3.y, t3.z, t3.w);
        } else {
            w0 = p[0]; w1 = p[1]; w2 = p[2]; w3 = p[3];
        }
        cnt.nodes++;
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
    if (s.T != 0) {
        const uint32_t b = (uint32_t)__builtin_ctz(s.T);
        s.T &= s.T - 1u;
        const uint4* p = (const uint4*)sc.woop + (size_t)(s.tbase + b) * (kBvhTriWords / 4);
        const uint4 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
        cnt.tris++;
        float t, u, v;
        bool h = tri_test(s.o, s.d, make_float4(u2f(w0.x), u2f(w0.y), u2f(w0.z), u2f(w0.w)), make_float4(u2f(w1.x), u2f(w1.y), u2f(w1.z), u2f(w1.w)),
                          make_float4(u2f(w2.x), u2f(w2.y), u2f(w2.z), u2f(w2.w)), s.tmin, s.tmax, t, u, v);
        if (h) {
            const uint32_t gid = w3.x;
            h = (gid != s.ex0) & (gid != s.ex1);
            if (h && sc.has_alpha) h = alpha_test<TEX>(sc, gid, u, v);
            if (h) {
                if (any_hit) {
                    s.best = gid;
                    s.T = 0; s.G = 0; s.sp = 0;
                } else {
                    const bool better = (s.best == kInvalid) | (t < s.best_t) | ((t == s.best_t) & (gid < s.best));
                    if (better) { s.best_t = t; s.best_u = u; s.best_v = v; s.best = gid; }
                }
            }
        }
    }
    s.active = (s.T != 0) | ((s.G >> 24) != 0) | (s.sp != 0);
}
template <int MODE, bool TEX, bool TILE = false>
AKR_D void trav_step(const DScene& sc, Trav& s, uint32_t* __restrict__ stack, TraceCounters& cnt, bool any_rt = false, const uint4* tile = nullptr) {
    if constexpr (AKR_BVH_STAGED == 2 || (AKR_BVH_STAGED == 1 && TEX)) {
        trav_step_staged<MODE, TEX, TILE>(sc, s, stack, cnt, any_rt, tile);
        return;
    }
    const bool any_hit = MODE == 2 ? any_rt : (MODE == 1);
    const bool do_tri = s.T != 0;
    const uint4* p;
    bool in_tile = false;
    uint32_t tile_idx = 0;
    if (do_tri) {
        const uint32_t b = (uint32_t)__builtin_ctz(s.T);
        s.T &= s.T - 1u;
        p = (const uint4*)sc.woop + (size_t)(s.tbase + b) * (kBvhTriWords / 4);
    } else {
        if ((s.G >> 24) == 0) {  // the caller guarantees sp > 0 here
            s.sp--;
            s.G = stack[s.sp * 256u];
        }
        const uint32_t j = 31u - (uint32_t)__builtin_clz(s.G);  // nearest pending sibling
        s.G &= ~(1u << j);
        if ((s.G >> 24) != 0) {  // the others wait as one entry
            if (s.sp < sc.bvh_stack_depth) {
                stack[s.sp * 256u] = s.G;
                s.sp++;
            } else {
                cnt.overflow = 1;  // unreachable for a tree scene_build.cpp accepted; kept as a tripwire (akr_pt_stats)
            }
        }
        const uint32_t slot = (j - 24u) ^ (s.octinv4 & 7u);
        const uint32_t idx = (s.G & 0xffffffu) + slot;
        p = sc.bvh_nodes + (size_t)idx * (kBvhNodeWords / 4);
        if (TILE && idx < sc.bvh_tile_nodes) {
            in_tile = true;
            tile_idx = idx;
        }
    }
    // the one fetch of the step: four 16-byte loads, a 64-byte triangle record or a 64-byte node -- one sector either way
    uint4 w0, w1, w2, w3;
    if (TILE && in_tile) {
        // explicitly an LDS address: left generic, the compiler folds the two branches into ONE flat load of a selected pointer
        // (and volatile: two plain loads in the arms of an if / else are sunk into one load of a selected -- generic -- pointer)
        typedef const volatile uint32_t __attribute__((address_space(3))) * LdsW;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef const volatile u32x4 __attribute__((address_space(3))) * LdsU4;
        LdsU4 pt = (LdsU4)((LdsW)(const uint32_t*)tile + tile_idx * kBvhNodeWords);
        const u32x4 t0 = pt[0], t1 = pt[1], t2 = pt[2], t3 = pt[3];
        w0 = make_uint4(t0.x, t0.y, t0.z, t0.w); w1 = make_uint4(t1.x, t1.y, t1.z, t1.w); w2 = make_uint4(t2.x, t2.y, t2.z, t2.w);
        w3 = make_uint4(t3.x, t3.y, t3.z, t3.w);
    } else {
        w0 = p[0]; w1 = p[1]; w2 = p[2]; w3 = p[3];
    }
    // All four loads are in flight before anything waits. Without this fence the compiler sinks the words only the node test
    // reads into the node branch -- a second dependent round trip to memory for every node step.
    asm volatile("" : "+v"(w0.x), "+v"(w0.y), "+v"(w0.z), "+v"(w0.w), "+v"(w1.x), "+v"(w1.y), "+v"(w1.z), "+v"(w1.w), "+v"(w2.x), "+v"(w2.y),
                      "+v"(w2.z), "+v"(w2.w), "+v"(w3.x), "+v"(w3.y), "+v"(w3.z), "+v"(w3.w));
    if (do_tri) {
        cnt.tris++;
        float t, u, v;
        bool h = tri_test(s.o, s.d, make_float4(u2f(w0.x), u2f(w0.y), u2f(w0.z), u2f(w0.w)), make_float4(u2f(w1.x), u2f(w1.y), u2f(w1.z), u2f(w1.w)),
                          make_float4(u2f(w2.x), u2f(w2.y), u2f(w2.z), u2f(w2.w)), s.tmin, s.tmax, t, u, v);
        if (h) {
            const uint32_t gid = w3.x;
            h = (gid != s.ex0) & (gid != s.ex1);
            if (h && sc.has_alpha) h = alpha_test<TEX>(sc, gid, u, v);
            if (h) {
                if (any_hit) {
                    s.best = gid;
                    s.T = 0; s.G = 0; s.sp = 0;  // any hit: done
                } else {
                    const bool better = (s.best == kInvalid) | (t < s.best_t) | ((t == s.best_t) & (gid < s.best));
                    if (better) { s.best_t = t; s.best_u = u; s.best_v = v; s.best = gid; }
                }
            }
        }
    } else {
        cnt.nodes++;
        const float limit = s.best_t;  // closest hit: culls with the best distance so far (non-strict: equal-t lower ids stay reachable)
        const float bx = u2f((w0.w & 0xffu) << 23) * s.inv.x, by = u2f(((w0.w >> 8) & 0xffu) << 23) * s.inv.y, bz = u2f(((w0.w >> 16) & 0xffu) << 23) * s.inv.z;
        const float ax = __builtin_fmaf(u2f(w0.x), s.inv.x, s.noi.x), ay = __builtin_fmaf(u2f(w0.y), s.inv.y, s.noi.y), az = __builtin_fmaf(u2f(w0.z), s.inv.z, s.noi.z);
        // Node words (host/bvh.cpp): w0 = origin | exponents + child_base[7:0]; w1 = child_base[23:8] + meta[4..5] | meta[0..3] | tri_base |
        // lo.x of entries 0..3; w2 = lo.y, lo.z, hi.x, hi.y of entries 0..3; w3 = hi.z of entries 0..3 | x, y, z of entries 4, 5 (lo lo hi hi).
        // per axis: the byte planes the ray enters through (near) and leaves through (far)
        const bool nx = s.inv.x < 0.0f, ny = s.inv.y < 0.0f, nz = s.inv.z < 0.0f;
        const uint32_t xb = nx ? ((w3.y >> 16) | (w3.y << 16)) : w3.y, yb = ny ? ((w3.z >> 16) | (w3.z << 16)) : w3.z, zb = nz ? ((w3.w >> 16) | (w3.w << 16)) : w3.w;
        // [0]: entries 0..3 (four bytes), [1]: entries 4, 5 (near in bytes 0, 1; far in bytes 2, 3 after the swap above)
        const uint32_t qnx[2] = {nx ? w2.z : w1.w, xb}, qfx[2] = {nx ? w1.w : w2.z, xb >> 16};
        const uint32_t qny[2] = {ny ? w2.w : w2.x, yb}, qfy[2] = {ny ? w2.x : w2.w, yb >> 16};
        const uint32_t qnz[2] = {nz ? w3.x : w2.y, zb}, qfz[2] = {nz ? w2.y : w3.x, zb >> 16};
        uint32_t hitmask = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t meta4 = h ? (w1.x >> 16) : w1.y;  // (h = 1: entries 4, 5; the two upper bytes are zero = empty)
            // inner children (index bits 3 and 4 set: 24..31) get their bit position xor-ed with the ray's octant
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
                if (tn <= tf) hitmask |= byte_of(child_bits4, i) << byte_of(bit_index4, i);  // empty entries have no child bits
            }
        }
        s.G = ((w0.w >> 24) | ((w1.x & 0xffffu) << 8)) | (hitmask & 0xff000000u);
        s.T = hitmask & 0x00ffffffu;
        s.tbase = w1.z;
    }
    s.active = (s.T != 0) | ((s.G >> 24) != 0) | (s.sp != 0);
}

template <bool ANY_HIT, bool TEX = false>
AKR_D bool trace_bvh(const DScene& sc, vec3 o, vec3 d, float tmin, float tmax, uint32_t ex0, uint32_t ex1, Hit& hit,
                     uint32_t* __restrict__ stack, TraceCounters& cnt) {
    Trav s;
    trav_begin(s, o, d, tmin, tmax, ex0, ex1);
    while (s.active) trav_step<ANY_HIT ? 1 : 0, TEX>(sc, s, stack, cnt);
    hit.t = s.best_t; hit.u = s.best_u; hit.v = s.best_v; hit.gid = s.best;
    return s.best != kInvalid;
}

}  // namespace akr
