// dinst.h -- what one triangle of one mesh instance looks like in world space: the records host/scene_build.cpp folds per
// instance-triangle when it FLATTENS a scene (shade record rows 3-7, the Woop rows of the triangle test, the plane row two coplanar
// neighbours share), as functions of (instance transform, object-space triangle). One definition for two users:
//   * the host's flattening scene compiler (every scene up to round 4, and the default);
//   * the device, at a candidate hit / at a shaded hit, when a scene is kept as meshes + instances (two-level acceleration
//     structure, disect.h trav_step_inst): nothing per instance-triangle is stored then, and what is computed on the fly has to be
//     the flattened record bit for bit -- the oracle flattens (mesh.rs:487-654 evaluates these per hit from object-space buffers).
// f32 throughout (AKR-F32: same helpers, same order) except the Woop rows and the sharing test, which are f64 on both sides.
#pragma once
#include "dscene.h"

namespace akr {

struct InstXf {  // AffineTransform of an instance (geometry.rs:203-209) with what the per-triangle code needs of it
    vec3 c0, c1, c2, t;   // matrix columns, translation
    vec3 k0, k1, k2;      // cofactor columns: (M^T)^-1 = k / det
    float det, inv_det;   // MeshInstance.transform_det, mesh.rs:309-310
};
// from the 8 x float4 instance record (dscene.h `inst`: c0 | det, c1, c2, t, k0 | 1 / det, k1, k2)
AKR_HD InstXf inst_xf_from_rows(const float4* m) {
    InstXf x;
    const float4 a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6];
    x.c0 = xyz(a); x.c1 = xyz(b); x.c2 = xyz(c); x.t = xyz(d);
    x.k0 = xyz(e); x.k1 = xyz(f); x.k2 = xyz(g);
    x.det = a.w; x.inv_det = e.w;
    return x;
}
AKR_HD vec3 inst_xf_normal(const InstXf& x, vec3 n) {  // (M^T)^-1 n
    vec3 r = (x.k0 * n.x + x.k1 * n.y) + x.k2 * n.z;
    return r * x.inv_det;
}

// What does not depend on the barycentrics of a hit (mesh.rs:527-546, 572-589, 608-635): geometric normal, flat frame, area, dpdu.
struct TriWorld {
    vec3 ng, tt;     // world geometric normal; world dpdu (zero = none)
    Frame frame;     // flat-shaded frame
    float area;      // world area of the triangle
    vec3 ng_local;   // object-space geometric normal (corner normals default to it)
};
AKR_HD TriWorld tri_world(const InstXf& x, vec3 v0, vec3 v1, vec3 v2, vec2 uv0, vec2 uv1, vec2 uv2) {
    TriWorld w;
    // mesh.rs:527-535
    vec3 ngc = cross(v1 - v0, v2 - v0);
    float len = length(ngc);
    float area_local = len * 0.5f;
    vec3 ng_local = div_s(ngc, len);
    // default tangent = dpdu (mesh.rs:572-589)
    vec3 tt_local = mk3(0, 0, 0);
    {
        vec2 duv02 = mk2(uv0.x - uv2.x, uv0.y - uv2.y), duv12 = mk2(uv1.x - uv2.x, uv1.y - uv2.y);
        vec3 dp02 = v0 - v2, dp12 = v1 - v2;
        float determinant = difference_of_products(duv02.x, duv12.y, duv02.y, duv12.x);
        bool degenerate_uv = abs_f(determinant) < 1e-8f;
        if (!degenerate_uv) {
            float inv_det = 1.0f / determinant;
            tt_local.x = difference_of_products(duv12.y, dp02.x, duv02.y, dp12.x) * inv_det;
            tt_local.y = difference_of_products(duv12.y, dp02.y, duv02.y, dp12.y) * inv_det;
            tt_local.z = difference_of_products(duv12.y, dp02.z, duv02.y, dp12.z) * inv_det;
        }
        if (degenerate_uv || length2(tt_local) == 0.0f) tt_local = frame_from_n(ng_local).t;
    }
    // world space (mesh.rs:608-635)
    vec3 tt = xf_vector(x.c0, x.c1, x.c2, tt_local);
    vec3 c = xf_vector(x.c0, x.c1, x.c2, ng_local);
    vec3 ng = normalize(inst_xf_normal(x, ng_local));
    w.area = (area_local == 0.0f || x.det == 0.0f) ? 0.0f : abs_f(area_local * x.det / dot(ng, c));
    w.frame = (tt.x != 0.0f || tt.y != 0.0f || tt.z != 0.0f) ? frame_from_n_t(ng, tt) : frame_from_n(ng);
    w.ng = ng;
    w.tt = tt;
    w.ng_local = ng_local;
    return w;
}
// the default uvs of a mesh without uvs (mesh.rs:541-546)
AKR_HD void tri_default_uvs(vec2& uv0, vec2& uv1, vec2& uv2) { uv0 = mk2(0.0f, 0.0f); uv1 = mk2(1.0f, 0.0f); uv2 = mk2(1.0f, 0.1f); }

// Woop's precomputed transform, in double, from the f32 world-space vertices: 12 floats = rows (r0 | c0), (r1 | c1), (r2 | c2).
AKR_HD void woop_precompute(vec3 A, vec3 B, vec3 C, float* w) {
    double ax = A.x, ay = A.y, az = A.z;
    double e1x = (double)B.x - ax, e1y = (double)B.y - ay, e1z = (double)B.z - az;
    double e2x = (double)C.x - ax, e2y = (double)C.y - ay, e2z = (double)C.z - az;
    double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    double det = nx * nx + ny * ny + nz * nz;
    if (!(det > 0.0)) {
        for (int i = 0; i < 12; i++) w[i] = 0.0f;
        return;
    }
    double r0x = (e2y * nz - e2z * ny) / det, r0y = (e2z * nx - e2x * nz) / det, r0z = (e2x * ny - e2y * nx) / det;
    double r1x = (ny * e1z - nz * e1y) / det, r1y = (nz * e1x - nx * e1z) / det, r1z = (nx * e1y - ny * e1x) / det;
    double r2x = nx / det, r2y = ny / det, r2z = nz / det;
    w[0] = (float)r0x; w[1] = (float)r0y; w[2] = (float)r0z; w[3] = (float)(-(r0x * ax + r0y * ay + r0z * az));
    w[4] = (float)r1x; w[5] = (float)r1y; w[6] = (float)r1z; w[7] = (float)(-(r1x * ax + r1y * ay + r1z * az));
    w[8] = (float)r2x; w[9] = (float)r2y; w[10] = (float)r2z; w[11] = (float)(-(r2x * ax + r2y * ay + r2z * az));
}
// the first two rows alone, bit for bit woop_precompute's (the device's exact test of a kept scene takes the third from woop_plane_row of
// either this triangle or the even neighbour whose plane it shares: dinst_trav.h resolve_pending)
AKR_HD void woop_edge_rows(vec3 A, vec3 B, vec3 C, float* w) {
    double ax = A.x, ay = A.y, az = A.z;
    double e1x = (double)B.x - ax, e1y = (double)B.y - ay, e1z = (double)B.z - az;
    double e2x = (double)C.x - ax, e2y = (double)C.y - ay, e2z = (double)C.z - az;
    double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    double det = nx * nx + ny * ny + nz * nz;
    if (!(det > 0.0)) {
        for (int i = 0; i < 8; i++) w[i] = 0.0f;
        return;
    }
    double r0x = (e2y * nz - e2z * ny) / det, r0y = (e2z * nx - e2x * nz) / det, r0z = (e2x * ny - e2y * nx) / det;
    double r1x = (ny * e1z - nz * e1y) / det, r1y = (nz * e1x - nx * e1z) / det, r1z = (nx * e1y - ny * e1x) / det;
    w[0] = (float)r0x; w[1] = (float)r0y; w[2] = (float)r0z; w[3] = (float)(-(r0x * ax + r0y * ay + r0z * az));
    w[4] = (float)r1x; w[5] = (float)r1y; w[6] = (float)r1z; w[7] = (float)(-(r1x * ax + r1y * ay + r1z * az));
}
// the third row alone (what share_plane_row needs of the even neighbour)
AKR_HD void woop_plane_row(vec3 A, vec3 B, vec3 C, float* r2) {
    double ax = A.x, ay = A.y, az = A.z;
    double e1x = (double)B.x - ax, e1y = (double)B.y - ay, e1z = (double)B.z - az;
    double e2x = (double)C.x - ax, e2y = (double)C.y - ay, e2z = (double)C.z - az;
    double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    double det = nx * nx + ny * ny + nz * nz;
    if (!(det > 0.0)) {
        r2[0] = r2[1] = r2[2] = r2[3] = 0.0f;
        return;
    }
    double r2x = nx / det, r2y = ny / det, r2z = nz / det;
    r2[0] = (float)r2x; r2[1] = (float)r2y; r2[2] = (float)r2z; r2[3] = (float)(-(r2x * ax + r2y * ay + r2z * az));
}

// Coplanar neighbours share a plane row: triangles 2j and 2j+1 of an instance -- the two halves of a quad in every mesh an
// exporter triangulated -- get the SAME third row (plane equation) when the second one's vertices lie in the first one's
// plane to within 1e-6 of the triangle's size: the second record's row is overwritten with the first's. Every intersector
// then computes bit-identical t and hit point for the two (same ray, same row), which the exhaustive pair walk uses to
// solve the plane once per quad (disect.h). A data-level definition: the oracle applies the same rule when it builds its
// scene (oracle/akr_oracle.c: or_share_plane_row); nothing in either tracer depends on it.
// ra = the even triangle's third row (4 floats), rb = the odd one's (overwritten if shared), vb = the odd one's world vertices.
AKR_HD bool plane_row_is_shared(const float* ra, const float* rb, const vec3 vb[3]) {
    const double rx = ra[0], ry = ra[1], rz = ra[2], c = ra[3];
    const double len = __builtin_sqrt(rx * rx + ry * ry + rz * rz);  // = 1 / |n| = 1 / (2 area)
    if (!(len > 0.0) || (rb[0] == 0.0f && rb[1] == 0.0f && rb[2] == 0.0f)) return false;  // a degenerate triangle on either side
    const double tol = 1e-6 * __builtin_sqrt(len);  // height / sqrt(|n|) <= 1e-6
    for (int i = 0; i < 3; i++) {
        const double s = ((rx * (double)vb[i].x + ry * (double)vb[i].y) + rz * (double)vb[i].z) + c;
        if (!(__builtin_fabs(s) <= tol)) return false;
    }
    return true;
}
AKR_HD void share_plane_row(const float* ra, float* rb, const vec3 vb[3]) {
    if (plane_row_is_shared(ra, rb, vb)) { rb[0] = ra[0]; rb[1] = ra[1]; rb[2] = ra[2]; rb[3] = ra[3]; }
}


// A cheap, CONSERVATIVE reject ahead of the exact test of a candidate (dinst_trav.h): Moeller-Trumbore in f32 on the world-space
// vertices the exact test uses, with a bound on how far its t, u, v can be from the ones tri_test computes from the f64 Woop rows.
// Returns false only when tri_test cannot accept; true = "undecided", the exact test decides. With eps = 2^-24 and 2-norms bounded
// by 1-norms (M = |o| + |A| + |t||d| the magnitudes that cancel, L = the longer of the edges e1, e2, D = |d . n| = |det|):
//   * tri_test: the rows are exact to eps per component (the f64 arithmetic is far below that), each 4-term fma chain loses
//     <= 4 eps of its absolute sum; t = -(r2.o + c2) / (r2.d) is off by <= 5 eps M |n| / D, the hit point by 6 eps M |n||d| / D, and
//     u = r0.P + c0 with |r0| = |e2| / |n| by <= 14 eps |e2| M |d| / D;
//   * this function: edges and o - A are exact to eps per component, a cross product and a dot product on top lose <= 11 eps of
//     |a||b||c|: u is off by <= 11 eps |e2||d| (|o - A| + |u||e1|) / D, the determinant by the fraction rho = 11 eps |e1||e2||d| / D;
//   * a shared plane row (share_plane_row: odd triangles) moves the plane by <= `plane_shift` along its normal.
// The sum, doubled: E below. Slivers and grazing rays make rho large: undecided. Every comparison is false on a NaN: undecided.
AKR_HD bool tri_may_hit(vec3 o, vec3 d, vec3 A, vec3 B, vec3 C, float tmin, float tlimit, float plane_shift) {
    const float eps = 5.9604645e-8f;
    const vec3 e1 = B - A, e2 = C - A, tv = o - A;
    const vec3 pv = cross(d, e2), qv = cross(tv, e1);
    const float det = dot(e1, pv);
    const float inv = 1.0f / det;
    const float u = dot(tv, pv) * inv, v = dot(d, qv) * inv, t = dot(e2, qv) * inv;
    const float l1 = (abs_f(e1.x) + abs_f(e1.y)) + abs_f(e1.z), l2 = (abs_f(e2.x) + abs_f(e2.y)) + abs_f(e2.z);
    const float L = max_f(l1, l2), dn = (abs_f(d.x) + abs_f(d.y)) + abs_f(d.z);
    const float M = ((abs_f(o.x) + abs_f(o.y)) + abs_f(o.z)) + ((abs_f(A.x) + abs_f(A.y)) + abs_f(A.z)) + abs_f(t) * dn;
    const float k = 1.0f / abs_f(det);
    const float rho = 11.0f * eps * L * L * dn * k;
    const float E = 2.0f * L * dn * k * (eps * (25.0f * M + 11.0f * ((abs_f(u) + abs_f(v)) + 1.0f) * L) + plane_shift);
    const float Et = 2.0f * L * L * k * (16.0f * eps * M + plane_shift) + 4.0f * (eps + rho) * abs_f(t);
    const bool reject = (u < -E) | (v < -E) | (u + v > 1.0f + 2.0f * E) | (t + Et < tmin) | (t - Et > tlimit);
    return !(reject & (rho < 0.25f));
}

}  // namespace akr
