/* or_accel.h -- checker-side helpers around the oracle's intersector. TEST INFRASTRUCTURE ONLY.
 *
 * Included by akr_oracle.c after or_tri_test / or_alpha_test / or_trace. Three things, none of which the product knows:
 *
 * 1. An OPTIONAL bounding-volume hierarchy for the ORACLE (or_scene_build_bvh). The definition of a hit stays the
 *    exhaustive loop of or_trace (every triangle through or_tri_test, min t, ties -> lowest global id; any-hit = some
 *    triangle passes). The hierarchy only skips triangles that cannot pass: median-split binary tree, boxes grown by
 *    1e-5 x scene diagonal, slab test in f64 with no early-out that depends on visiting order other than "node entry
 *    distance > best t so far (+ slack)". The result is therefore independent of the tree, and the tests check it against
 *    the exhaustive loop ray by ray (tests/test_oracle_accel.py) before it is used for the 1 M / 10 M-triangle
 *    configurations where the exhaustive loop would take hours.
 *    It shares nothing with akari_render_amd/csrc/host/bvh.cpp (binned SAH, 4-wide, quantised boxes).
 *
 * 2. A switch for the coplanar-neighbour plane-row rule (or_set_share_plane_rows): lets a test render the same scene with
 *    the unmodified per-triangle records and bound the effect of the rule (tests/test_oracle_frozen.py).
 *
 * 3. An independent f64 Moeller-Trumbore intersector straight from the f32 world-space vertices (or_mt_f64_*): no Woop
 *    records, no shared code with or_tri_test. The reference's intersector (Embree through LuisaCompute, scene.rs:88-110)
 *    is absent; this is the cross-check that the build's own triangle test decides hit / miss and (t, u, v) like a textbook
 *    one up to rounding.
 */
#ifndef OR_ACCEL_H
#define OR_ACCEL_H

/* ---------------------------------------------------------------- world-space vertices (f32, as the scene stores them) */
static float *or_world_vertices(const or_scene *sc) { /* 9 floats per global triangle; caller frees */
    float *wv = (float *)malloc(36ull * (sc->n_tris ? sc->n_tris : 1));
    for (uint32_t i = 0; i < sc->n_instances; i++) {
        const or_instance *in = &sc->instances[i];
        const or_mesh_desc *g = &sc->meshes[in->mesh].d;
        for (uint32_t p = 0; p < g->n_triangles; p++) {
            float *o = wv + 9ull * (in->tri_offset + p);
            for (int c = 0; c < 3; c++) {
                v3 a = xf_point(&in->xf, ld3(g->vertices, g->indices[3 * p + c]));
                o[3 * c] = a.x; o[3 * c + 1] = a.y; o[3 * c + 2] = a.z;
            }
        }
    }
    return wv;
}

/* ---------------------------------------------------------------- 1. oracle-only BVH */
typedef struct { float lo[3], hi[3]; uint32_t first, count; /* count == 0: inner node, children first, first + 1 */ } or_bnode;
struct or_bvh { or_bnode *nodes; uint32_t n_nodes; uint32_t *order; double pad; };

static void or_bvh_bounds(const float *wv, const uint32_t *order, uint32_t first, uint32_t count, float *lo, float *hi, float *clo, float *chi) {
    for (int a = 0; a < 3; a++) { lo[a] = clo[a] = INFINITY; hi[a] = chi[a] = -INFINITY; }
    for (uint32_t i = first; i < first + count; i++) {
        const float *t = wv + 9ull * order[i];
        for (int a = 0; a < 3; a++) {
            float mn = fminf(t[a], fminf(t[3 + a], t[6 + a])), mx = fmaxf(t[a], fmaxf(t[3 + a], t[6 + a]));
            float c = 0.5f * (mn + mx);
            lo[a] = fminf(lo[a], mn); hi[a] = fmaxf(hi[a], mx);
            clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c);
        }
    }
}
static float or_bvh_centroid(const float *wv, uint32_t tri, int axis) {
    const float *t = wv + 9ull * tri;
    float mn = fminf(t[axis], fminf(t[3 + axis], t[6 + axis])), mx = fmaxf(t[axis], fmaxf(t[3 + axis], t[6 + axis]));
    return 0.5f * (mn + mx);
}
/* quickselect: order[first + k] gets the element of rank k along `axis`, smaller-or-equal ones before it */
static void or_bvh_select(const float *wv, uint32_t *order, uint32_t first, uint32_t count, uint32_t k, int axis) {
    int64_t lo = first, hi = (int64_t)first + count - 1, target = (int64_t)first + k;
    uint64_t rng = 0x9e3779b97f4a7c15ull ^ ((uint64_t)first << 20) ^ count;
    while (lo < hi) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        int64_t pi = lo + (int64_t)((rng >> 33) % (uint64_t)(hi - lo + 1));
        float pv = or_bvh_centroid(wv, order[pi], axis);
        int64_t i = lo, j = hi;
        while (i <= j) { /* Hoare partition */
            while (or_bvh_centroid(wv, order[i], axis) < pv) i++;
            while (or_bvh_centroid(wv, order[j], axis) > pv) j--;
            if (i <= j) { uint32_t t = order[i]; order[i] = order[j]; order[j] = t; i++; j--; }
        }
        if (target <= j) hi = j; else if (target >= i) lo = i; else break;
    }
}
OR_EXPORT void or_scene_free_bvh(or_scene *sc) {
    if (!sc->bvh) return;
    free(sc->bvh->nodes); free(sc->bvh->order); free(sc->bvh);
    sc->bvh = 0;
}
OR_EXPORT uint32_t or_scene_build_bvh(or_scene *sc) { /* returns the number of nodes */
    or_scene_free_bvh(sc);
    const uint32_t n = sc->n_tris;
    if (n == 0) return 0;
    float *wv = or_world_vertices(sc);
    struct or_bvh *b = (struct or_bvh *)calloc(1, sizeof *b);
    b->order = (uint32_t *)malloc(4ull * n);
    for (uint32_t i = 0; i < n; i++) b->order[i] = i;
    b->nodes = (or_bnode *)malloc(sizeof(or_bnode) * (2ull * n + 2));
    typedef struct { uint32_t node, first, count; } item;
    item *stack = (item *)malloc(sizeof(item) * 128);
    int sp = 0;
    b->n_nodes = 1;
    stack[sp++] = (item){0, 0, n};
    float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    while (sp > 0) {
        item it = stack[--sp];
        or_bnode *nd = &b->nodes[it.node];
        float clo[3], chi[3];
        or_bvh_bounds(wv, b->order, it.first, it.count, nd->lo, nd->hi, clo, chi);
        if (it.node == 0) for (int a = 0; a < 3; a++) { slo[a] = nd->lo[a]; shi[a] = nd->hi[a]; }
        if (it.count <= 4) { nd->first = it.first; nd->count = it.count; continue; }
        int axis = 0;
        float ext = chi[0] - clo[0];
        for (int a = 1; a < 3; a++) if (chi[a] - clo[a] > ext) { ext = chi[a] - clo[a]; axis = a; }
        uint32_t half = it.count / 2;
        if (ext > 0.0f) or_bvh_select(wv, b->order, it.first, it.count, half, axis);
        uint32_t l = b->n_nodes;
        b->n_nodes += 2;
        nd->first = l; nd->count = 0;
        /* median split: depth <= log2(n) + 1, the larger half is pushed first so the stack stays shallow */
        stack[sp++] = (item){l, it.first, half};
        stack[sp++] = (item){l + 1, it.first + half, it.count - half};
        if (sp > 120) { fprintf(stderr, "or_scene_build_bvh: stack\n"); abort(); }
    }
    double dx = (double)shi[0] - slo[0], dy = (double)shi[1] - slo[1], dz = (double)shi[2] - slo[2];
    b->pad = 1e-5 * sqrt(dx * dx + dy * dy + dz * dz) + 1e-30;
    free(stack); free(wv);
    sc->bvh = b;
    return b->n_nodes;
}
/* entry distance of the ray into the padded box within [tmin, tmax], or -1 when it misses (f64; NaN-safe: a NaN makes
 * the comparisons false and the node is VISITED, never skipped) */
static inline int or_bvh_box_miss(const or_bnode *nd, double pad, const double *o, const double *d, double tmin, double tmax) {
    double t0 = tmin, t1 = tmax;
    for (int a = 0; a < 3; a++) {
        double lo = (double)nd->lo[a] - pad, hi = (double)nd->hi[a] + pad;
        if (d[a] == 0.0) { if (o[a] < lo || o[a] > hi) return 1; continue; }
        double inv = 1.0 / d[a];
        double ta = (lo - o[a]) * inv, tb = (hi - o[a]) * inv;
        if (ta > tb) { double s = ta; ta = tb; tb = s; }
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
    }
    return t0 > t1; /* false for NaN */
}
static int or_trace_bvh(const or_scene *sc, const or_ray *r, int any_hit, uint32_t *o_inst, uint32_t *o_prim, v2 *o_bary, or_stats *st) {
    const struct or_bvh *b = sc->bvh;
    const double o[3] = {r->o.x, r->o.y, r->o.z}, d[3] = {r->d.x, r->d.y, r->d.z};
    float best_t = 0.0f; uint32_t best = OR_INVALID; v2 best_b = V2(0, 0);
    uint32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    uint64_t tests = 0;
    while (sp > 0) {
        const or_bnode *nd = &b->nodes[stack[--sp]];
        /* the far limit: t_max, or the best t so far with slack (candidates at EQUAL t and a lower id must stay reachable) */
        double lim = (double)r->t_max;
        if (best != OR_INVALID) { double bt = (double)best_t; bt += 1e-6 * fabs(bt) + 1e-30; if (bt < lim) lim = bt; }
        double lo_lim = (double)r->t_min; lo_lim -= 1e-6 * fabs(lo_lim) + 1e-30;
        if (or_bvh_box_miss(nd, b->pad, o, d, lo_lim, lim)) continue;
        if (nd->count == 0) { stack[sp++] = nd->first; stack[sp++] = nd->first + 1; continue; }
        for (uint32_t i = nd->first; i < nd->first + nd->count; i++) {
            uint32_t k = b->order[i];
            float t, u, v;
            tests++;
            if (!or_tri_test(r->o, r->d, sc->woop + 12ull * k, r->t_min, r->t_max, &t, &u, &v)) continue;
            uint32_t inst = sc->tri_inst[k], prim = sc->tri_prim[k];
            if (!((inst != r->ex0_inst || prim != r->ex0_prim) && (inst != r->ex1_inst || prim != r->ex1_prim))) continue;
            if (!or_alpha_test(sc, inst, prim, u, v)) continue;
            if (any_hit) { if (st) st->n_tri_tests += tests; return 1; }
            if (best == OR_INVALID || t < best_t || (t == best_t && k < best)) { best = k; best_t = t; best_b = V2(u, v); }
        }
    }
    if (st) st->n_tri_tests += tests;
    if (any_hit || best == OR_INVALID) return 0;
    *o_inst = sc->tri_inst[best]; *o_prim = sc->tri_prim[best]; *o_bary = best_b;
    return 1;
}

/* ---------------------------------------------------------------- batch intersection (tests) */
typedef struct { const or_scene *sc; uint32_t n; const float *rays; int any_hit; uint32_t *out; float *tuv; volatile uint32_t *next; } or_isect_job;
static float or_hit_t(const or_scene *sc, const or_ray *r, uint32_t inst, uint32_t prim) { /* t of a known hit */
    uint32_t k = sc->instances[inst].tri_offset + prim;
    float t = 0, u, v;
    or_tri_test(r->o, r->d, sc->woop + 12ull * k, r->t_min, r->t_max, &t, &u, &v);
    return t;
}
static void *or_isect_worker(void *arg) {
    or_isect_job *j = (or_isect_job *)arg;
    for (;;) {
        uint32_t c = __sync_fetch_and_add(j->next, 1024u);
        if (c >= j->n) break;
        uint32_t e = c + 1024u < j->n ? c + 1024u : j->n;
        for (uint32_t i = c; i < e; i++) {
            const float *q = j->rays + 8ull * i;
            or_ray r = {V3(q[0], q[1], q[2]), V3(q[3], q[4], q[5]), q[6], q[7], OR_INVALID, OR_INVALID, OR_INVALID, OR_INVALID};
            uint32_t inst = 0, prim = 0; v2 b = V2(0, 0);
            int hit = or_trace(j->sc, &r, j->any_hit, &inst, &prim, &b, 0);
            j->out[3ull * i] = (uint32_t)hit; j->out[3ull * i + 1] = hit ? inst : 0; j->out[3ull * i + 2] = hit ? prim : 0;
            j->tuv[3ull * i] = (hit && !j->any_hit) ? or_hit_t(j->sc, &r, inst, prim) : 0.0f;
            j->tuv[3ull * i + 1] = hit ? b.x : 0.0f; j->tuv[3ull * i + 2] = hit ? b.y : 0.0f;
        }
    }
    return 0;
}
/* rays: 8 floats each (o, d, tmin, tmax). out: (hit, inst, prim) per ray; tuv: (t, u, v) per ray. Through or_trace, i.e. the
 * exhaustive loop, or the BVH if or_scene_build_bvh was called. */
OR_EXPORT void or_scene_intersect_many(const or_scene *sc, uint32_t n, const float *rays, int any_hit, uint32_t *out, float *tuv, uint32_t n_threads) {
    volatile uint32_t next = 0;
    or_isect_job job = {sc, n, rays, any_hit, out, tuv, &next};
    pthread_t th[256];
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    for (uint32_t t = 1; t < n_threads; t++) pthread_create(&th[t], 0, or_isect_worker, &job);
    or_isect_worker(&job);
    for (uint32_t t = 1; t < n_threads; t++) pthread_join(th[t], 0);
}

/* ---------------------------------------------------------------- 3. independent f64 Moeller-Trumbore */
/* Moeller & Trumbore 1997, "Fast, minimum storage ray/triangle intersection", two-sided, in double from the f32 vertices.
 * u, v here are the weights of vertex 1 and vertex 2 -- the same barycentric convention as the product's (u, v). Returns 0
 * for a ray parallel to the plane. No range test: the caller decides. */
static inline int or_mt_f64(const float *tri, const double *o, const double *d, double *t, double *u, double *v) {
    double e1[3], e2[3], p[3], s[3], q[3];
    for (int a = 0; a < 3; a++) { e1[a] = (double)tri[3 + a] - tri[a]; e2[a] = (double)tri[6 + a] - tri[a]; s[a] = o[a] - (double)tri[a]; }
    p[0] = d[1] * e2[2] - d[2] * e2[1]; p[1] = d[2] * e2[0] - d[0] * e2[2]; p[2] = d[0] * e2[1] - d[1] * e2[0];
    double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (det == 0.0) return 0;
    double inv = 1.0 / det;
    *u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inv;
    q[0] = s[1] * e1[2] - s[2] * e1[1]; q[1] = s[2] * e1[0] - s[0] * e1[2]; q[2] = s[0] * e1[1] - s[1] * e1[0];
    *v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
    *t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    return 1;
}
typedef struct { const or_scene *sc; const float *wv; uint32_t n; const float *rays; const uint32_t *gids; uint32_t *out_gid; double *out; volatile uint32_t *next; } or_mt_job;
static void *or_mt_worker(void *arg) {
    or_mt_job *j = (or_mt_job *)arg;
    for (;;) {
        uint32_t c = __sync_fetch_and_add(j->next, 256u);
        if (c >= j->n) break;
        uint32_t e = c + 256u < j->n ? c + 256u : j->n;
        for (uint32_t i = c; i < e; i++) {
            const float *r = j->rays + 8ull * i;
            const double o[3] = {r[0], r[1], r[2]}, d[3] = {r[3], r[4], r[5]};
            double *out = j->out + 4ull * i;
            if (j->gids) { /* one given triangle: (t, u, v, margin) without any range test */
                double t = NAN, u = NAN, v = NAN;
                if (j->gids[i] != OR_INVALID) or_mt_f64(j->wv + 9ull * j->gids[i], o, d, &t, &u, &v);
                out[0] = t; out[1] = u; out[2] = v; out[3] = fmin(fmin(u, v), 1.0 - (u + v));
                continue;
            }
            double bt = INFINITY, bu = 0, bv = 0; uint32_t best = OR_INVALID;
            for (uint32_t k = 0; k < j->sc->n_tris; k++) {
                double t, u, v;
                if (!or_mt_f64(j->wv + 9ull * k, o, d, &t, &u, &v)) continue;
                if (!(t >= (double)r[6] && t <= (double)r[7] && u >= 0.0 && v >= 0.0 && u + v <= 1.0)) continue;
                if (t < bt) { bt = t; bu = u; bv = v; best = k; }
            }
            j->out_gid[i] = best;
            out[0] = best != OR_INVALID ? bt : NAN; out[1] = bu; out[2] = bv; out[3] = fmin(fmin(bu, bv), 1.0 - (bu + bv));
        }
    }
    return 0;
}
/* gids == NULL: closest hit per ray over ALL triangles -> out_gid[n] (global triangle id or 0xffffffff), out[4n] = (t, u, v,
 * inside margin). gids != NULL: the solve for that one triangle per ray, no range test (NaN when parallel / gid invalid). */
OR_EXPORT void or_mt_f64_many(const or_scene *sc, uint32_t n, const float *rays, const uint32_t *gids, uint32_t *out_gid, double *out, uint32_t n_threads) {
    float *wv = or_world_vertices(sc);
    volatile uint32_t next = 0;
    or_mt_job job = {sc, wv, n, rays, gids, out_gid, out, &next};
    pthread_t th[256];
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    for (uint32_t t = 1; t < n_threads; t++) pthread_create(&th[t], 0, or_mt_worker, &job);
    or_mt_worker(&job);
    for (uint32_t t = 1; t < n_threads; t++) pthread_join(th[t], 0);
    free(wv);
}
/* the f32 world-space vertices the scene was built from (9 floats per global triangle), for test-side ray construction */
OR_EXPORT void or_scene_world_vertices(const or_scene *sc, float *out9) {
    float *wv = or_world_vertices(sc);
    memcpy(out9, wv, 36ull * sc->n_tris);
    free(wv);
}
#endif
