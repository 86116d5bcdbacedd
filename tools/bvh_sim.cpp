// bvh_sim.cpp -- host-side model of the wide-BVH traversal (device/disect.h: trav_step) over the tree host/bvh.cpp builds, for
// judging builder changes without a GPU: builds the BVH of a triangle dump, traces path-like rays (camera rays, cosine-weighted
// bounces, shadow rays towards a ceiling light) with the device's visiting order, culling and any-hit rules, and reports node
// visits / triangle tests per ray, tree depth, node count and -- a proxy for the wave's divergence -- the ratio between the
// mean of a ray pair's steps and the mean of the maximum over 64 pairs.
//   python -c "..."  dumps the hall: see tools/bvh_sim.sh
//   g++ -O2 -std=c++17 -I akari_render_amd/csrc tools/bvh_sim.cpp akari_render_amd/csrc/host/bvh.cpp -o /tmp/bvh_sim
//   /tmp/bvh_sim tris.f32 [n_paths=20000] [max_depth=6]
// Measurement only; nothing in the library or the tests depends on it.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

namespace akr {
void build_bvh8(const std::vector<float>& tri_bounds, uint32_t n_tris, float pad, uint32_t stride, bool balanced, std::vector<uint32_t>& order_out,
                std::vector<uint32_t>& out_nodes, uint32_t& depth_out);
}
static const uint32_t kStride = 16;
// 64-byte node, 6 entries (host/bvh.cpp): entry e -> meta byte and quantised box
struct Kid { uint32_t meta; double lo[3], hi[3]; };
static float u2f_(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static void decode(const uint32_t* n, Kid k[6], uint32_t& child_base, uint32_t& tri_base) {
    child_base = (n[3] >> 24) | ((n[4] & 0xffffu) << 8);
    tri_base = n[6];
    for (int e = 0; e < 6; e++) {
        k[e].meta = e < 4 ? (n[5] >> (8 * e)) & 0xffu : (n[4] >> (16 + 8 * (e - 4))) & 0xffu;
        for (int a = 0; a < 3; a++) {
            const double scale = u2f_(((n[3] >> (8 * a)) & 0xffu) << 23);
            uint32_t qlo, qhi;
            if (e < 4) { qlo = (n[7 + a] >> (8 * e)) & 0xffu; qhi = (n[10 + a] >> (8 * e)) & 0xffu; }
            else { qlo = (n[13 + a] >> (8 * (e - 4))) & 0xffu; qhi = (n[13 + a] >> (16 + 8 * (e - 4))) & 0xffu; }
            k[e].lo[a] = u2f_(n[a]) + qlo * scale;
            k[e].hi[a] = u2f_(n[a]) + qhi * scale;
        }
    }
}
struct V3 { double x, y, z; };
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 norm(V3 a) { double l = std::sqrt(dot(a, a)); return a * (1.0 / l); }

struct Scene {
    std::vector<float> tris;  // 9 floats per triangle, traversal order
    std::vector<uint32_t> nodes;
    uint32_t depth = 0;
};
struct Counters { uint64_t nodes = 0, tris = 0, rays = 0; };

static float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static bool g_groupcull = false;  // SIM_GROUPCULL=1: a pending group carries the smallest entry distance of its nodes and is dropped at pop if that is beyond the best hit
static bool g_sorted = false;  // SIM_SORT=1: children visited in order of their entry distance (what a distance sort would buy)
// the same traversal with an exact near-to-far order: a stack of single nodes, pushed far to near
static int trace_sorted(const Scene& sc, V3 o, V3 d, double tmax, bool any_hit, double& t_out, Counters& c, uint32_t& steps) {
    const double inv[3] = {1.0 / (std::fabs(d.x) < 1e-20 ? std::copysign(1e-20, d.x) : d.x), 1.0 / (std::fabs(d.y) < 1e-20 ? std::copysign(1e-20, d.y) : d.y),
                           1.0 / (std::fabs(d.z) < 1e-20 ? std::copysign(1e-20, d.z) : d.z)};
    const double org[3] = {o.x, o.y, o.z};
    struct E { uint32_t node; double tn; };
    std::vector<E> stack{{0u, 0.0}};
    double best_t = tmax;
    int best = -1;
    c.rays++;
    steps = 0;
    while (!stack.empty()) {
        const E e = stack.back();
        stack.pop_back();
        if (e.tn > best_t) continue;
        const uint32_t* n = &sc.nodes[(size_t)kStride * e.node];
        c.nodes++; steps++;
        E kids[8];
        int nk = 0;
        uint32_t T = 0;
        Kid kd[6];
        uint32_t cbase, tbase_;
        decode(n, kd, cbase, tbase_);
        for (int s = 0; s < 6; s++) {
            const uint32_t meta = kd[s].meta;
            if (meta == 0) continue;
            double tn = 0.0, tf = best_t;
            for (int a = 0; a < 3; a++) {
                const double t0 = (kd[s].lo[a] - org[a]) * inv[a], t1 = (kd[s].hi[a] - org[a]) * inv[a];
                tn = std::max(tn, std::min(t0, t1));
                tf = std::min(tf, std::max(t0, t1));
            }
            if (tn > tf) continue;
            if ((meta & 0x18u) == 0x18u && (meta >> 5) == 1u) kids[nk++] = E{cbase + ((meta & 0x1fu) - 24u), tn};
            else T |= (meta >> 5) << (meta & 0x1fu);
        }
        while (T) {
            const uint32_t b = (uint32_t)__builtin_ctz(T);
            T &= T - 1;
            const float* v = &sc.tris[9ull * (tbase_ + b)];
            c.tris++; steps++;
            const V3 A{v[0], v[1], v[2]}, B{v[3], v[4], v[5]}, C{v[6], v[7], v[8]};
            const V3 e1 = B - A, e2 = C - A, p = cross(d, e2);
            const double det = dot(e1, p);
            if (std::fabs(det) <= 1e-30) continue;
            const double id = 1.0 / det;
            const V3 sv = o - A;
            const double u = dot(sv, p) * id;
            const V3 q = cross(sv, e1);
            const double vv = dot(d, q) * id, t = dot(e2, q) * id;
            if (u >= 0 && vv >= 0 && u + vv <= 1 && t > 1e-9 && t <= best_t) {
                if (any_hit) { t_out = t; return (int)(tbase_ + b); }
                if (t < best_t) { best_t = t; best = (int)(tbase_ + b); }
            }
        }
        std::sort(kids, kids + nk, [](const E& a, const E& b) { return a.tn > b.tn; });
        for (int i = 0; i < nk; i++) stack.push_back(kids[i]);
    }
    t_out = best_t;
    return best;
}
// returns the hit triangle (traversal order) or -1; t_out; steps = node + triangle steps of this ray
static int trace(const Scene& sc, V3 o, V3 d, double tmax, bool any_hit, double& t_out, Counters& c, uint32_t& steps) {
    if (g_sorted) return trace_sorted(sc, o, d, tmax, any_hit, t_out, c, steps);
    const double inv[3] = {1.0 / (std::fabs(d.x) < 1e-20 ? std::copysign(1e-20, d.x) : d.x), 1.0 / (std::fabs(d.y) < 1e-20 ? std::copysign(1e-20, d.y) : d.y),
                           1.0 / (std::fabs(d.z) < 1e-20 ? std::copysign(1e-20, d.z) : d.z)};
    const double org[3] = {o.x, o.y, o.z};
    const uint32_t oi = (inv[0] >= 0 ? 1u : 0u) | (inv[1] >= 0 ? 2u : 0u) | (inv[2] >= 0 ? 4u : 0u);
    uint32_t G = 1u << (24 + oi), T = 0, tbase = 0;
    std::vector<uint32_t> stack;
    std::vector<double> stack_tn;
    double tn_child[8];
    double g_tn = 0.0;
    double best_t = tmax;
    int best = -1;
    c.rays++;
    steps = 0;
    while (true) {
        if (T != 0) {
            const uint32_t b = (uint32_t)__builtin_ctz(T);
            T &= T - 1;
            const float* v = &sc.tris[9ull * (tbase + b)];
            c.tris++; steps++;
            const V3 A{v[0], v[1], v[2]}, B{v[3], v[4], v[5]}, C{v[6], v[7], v[8]};
            const V3 e1 = B - A, e2 = C - A, p = cross(d, e2);
            const double det = dot(e1, p);
            if (std::fabs(det) > 1e-30) {
                const double id = 1.0 / det;
                const V3 s = o - A;
                const double u = dot(s, p) * id;
                const V3 q = cross(s, e1);
                const double vv = dot(d, q) * id, t = dot(e2, q) * id;
                if (u >= 0 && vv >= 0 && u + vv <= 1 && t > 1e-9 && t <= best_t) {
                    if (any_hit) { t_out = t; return (int)(tbase + b); }
                    if (t < best_t) { best_t = t; best = (int)(tbase + b); }
                }
            }
        } else {
            if ((G >> 24) == 0) {
                if (stack.empty()) break;
                G = stack.back();
                stack.pop_back();
                const double gt = stack_tn.back();
                stack_tn.pop_back();
                if (g_groupcull && gt > best_t) { G = 0; continue; }
                g_tn = gt;
            }
            const uint32_t j = 31u - (uint32_t)__builtin_clz(G);
            G &= ~(1u << j);
            if ((G >> 24) != 0) { stack.push_back(G); stack_tn.push_back(g_tn); }
            const uint32_t slot = (j - 24u) ^ oi;
            const uint32_t* n = &sc.nodes[(size_t)kStride * ((G & 0xffffffu) + slot)];
            c.nodes++; steps++;
            uint32_t hitmask = 0;
            Kid kd[6];
            uint32_t cbase, tb;
            decode(n, kd, cbase, tb);
            for (int s = 0; s < 6; s++) {
                const uint32_t meta = kd[s].meta;
                if (meta == 0) continue;
                double tn = 0.0, tf = best_t;
                for (int a = 0; a < 3; a++) {
                    const double t0 = (kd[s].lo[a] - org[a]) * inv[a], t1 = (kd[s].hi[a] - org[a]) * inv[a];
                    tn = std::max(tn, std::min(t0, t1));
                    tf = std::min(tf, std::max(t0, t1));
                }
                if (tn <= tf) {
                    const uint32_t is_inner = (meta & 0x18u) == 0x18u && (meta >> 5) == 1u;
                    if (is_inner) hitmask |= 1u << (24 + (((meta & 0x1fu) - 24u) ^ oi));
                    else hitmask |= (meta >> 5) << (meta & 0x1fu);
                }
            }
            g_tn = 0.0;
            (void)tn_child;
            G = cbase | (hitmask & 0xff000000u);
            T = hitmask & 0x00ffffffu;
            tbase = tb;
        }
        if (T == 0 && (G >> 24) == 0 && stack.empty()) break;
    }
    t_out = best_t;
    return best;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: bvh_sim tris.f32 [n_paths] [max_depth]\n"); return 1; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    std::fseek(f, 0, SEEK_END);
    const size_t bytes = (size_t)std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const uint32_t n = (uint32_t)(bytes / 36);
    std::vector<float> src(9ull * n);
    if (std::fread(src.data(), 4, src.size(), f) != src.size()) return 1;
    std::fclose(f);
    g_sorted = std::getenv("SIM_SORT") != nullptr;
    g_groupcull = std::getenv("SIM_GROUPCULL") != nullptr;
    const uint32_t n_paths = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 20000u, max_depth = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 6u;
    std::vector<float> bounds(6ull * n);
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            const float x0 = src[9ull * i + a], x1 = src[9ull * i + 3 + a], x2 = src[9ull * i + 6 + a];
            bounds[6ull * i + a] = std::min(x0, std::min(x1, x2));
            bounds[6ull * i + 3 + a] = std::max(x0, std::max(x1, x2));
            lo[a] = std::min(lo[a], bounds[6ull * i + a]);
            hi[a] = std::max(hi[a], bounds[6ull * i + 3 + a]);
        }
    // SIM_SPLIT=<f> (round 6): SPATIAL SPLITS as early split clipping (Ernst & Greiner 2007): the reference whose box has
    // the largest surface area is cut at the middle of its box's longest axis -- the triangle's polygon clipped against the plane, a tight box
    // per half -- until there are f x n references. A triangle then sits in several leaves (every copy is the same full test: the min-t /
    // lowest-id rule gives the same hit; the copies cost fetches). The tree is built over the references' boxes.
    std::vector<uint32_t> ref_tri(n);
    for (uint32_t i = 0; i < n; i++) ref_tri[i] = i;
    const double split_f = std::getenv("SIM_SPLIT") ? std::atof(std::getenv("SIM_SPLIT")) : 1.0;
    if (split_f > 1.0) {
        struct Ref { uint32_t tri; int nv; float v[9][3]; float lo[3], hi[3]; float area; };
        auto finish = [](Ref& r) {
            for (int a = 0; a < 3; a++) { r.lo[a] = 1e30f; r.hi[a] = -1e30f; }
            for (int k = 0; k < r.nv; k++)
                for (int a = 0; a < 3; a++) { r.lo[a] = std::min(r.lo[a], r.v[k][a]); r.hi[a] = std::max(r.hi[a], r.v[k][a]); }
            const float dx = r.hi[0] - r.lo[0], dy = r.hi[1] - r.lo[1], dz = r.hi[2] - r.lo[2];
            r.area = dx * dy + dy * dz + dz * dx;
        };
        std::vector<Ref> refs(n);
        for (uint32_t i = 0; i < n; i++) {
            refs[i].tri = i; refs[i].nv = 3;
            for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) refs[i].v[k][a] = src[9ull * i + 3 * k + a];
            finish(refs[i]);
        }
        auto cmp = [&](uint32_t x, uint32_t y) { return refs[x].area < refs[y].area; };
        std::vector<uint32_t> heap(n);
        for (uint32_t i = 0; i < n; i++) heap[i] = i;
        std::make_heap(heap.begin(), heap.end(), cmp);
        const size_t target = (size_t)(split_f * n);
        refs.reserve(target + 8);
        while (refs.size() < target && !heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            const uint32_t ri = heap.back(); heap.pop_back();
            Ref r = refs[ri];
            int ax = 0;
            for (int a = 1; a < 3; a++) if (r.hi[a] - r.lo[a] > r.hi[ax] - r.lo[ax]) ax = a;
            const float mid = 0.5f * (r.lo[ax] + r.hi[ax]);
            if (!(mid > r.lo[ax] && mid < r.hi[ax])) continue;
            Ref L = r, R = r; L.nv = R.nv = 0;
            for (int k = 0; k < r.nv; k++) {  // Sutherland-Hodgman against the plane, both sides at once
                const float* p = r.v[k]; const float* q = r.v[(k + 1) % r.nv];
                const bool pin = p[ax] <= mid, qin = q[ax] <= mid;
                if (pin) { std::memcpy(L.v[L.nv++], p, 12); }
                if (!pin || p[ax] == mid) { std::memcpy(R.v[R.nv++], p, 12); }
                if (pin != qin) {
                    const float t = (mid - p[ax]) / (q[ax] - p[ax]);
                    float x[3];
                    for (int a = 0; a < 3; a++) x[a] = p[a] + t * (q[a] - p[a]);
                    x[ax] = mid;
                    std::memcpy(L.v[L.nv++], x, 12); std::memcpy(R.v[R.nv++], x, 12);
                }
            }
            if (L.nv < 3 || R.nv < 3 || L.nv > 8 || R.nv > 8) continue;
            finish(L); finish(R);
            refs[ri] = L;
            refs.push_back(R);
            heap.push_back(ri); std::push_heap(heap.begin(), heap.end(), cmp);
            heap.push_back((uint32_t)refs.size() - 1); std::push_heap(heap.begin(), heap.end(), cmp);
        }
        bounds.resize(6ull * refs.size());
        ref_tri.resize(refs.size());
        for (size_t i = 0; i < refs.size(); i++) {
            ref_tri[i] = refs[i].tri;
            for (int a = 0; a < 3; a++) { bounds[6 * i + a] = refs[i].lo[a]; bounds[6 * i + 3 + a] = refs[i].hi[a]; }
        }
        std::fprintf(stderr, "spatial splits: %zu references for %u triangles (x%.3f)\n", refs.size(), n, (double)refs.size() / n);
    }
    const uint32_t n_refs = (uint32_t)ref_tri.size();
    const float diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    Scene sc;
    std::vector<uint32_t> order;
    auto t0 = std::chrono::steady_clock::now();
    akr::build_bvh8(bounds, n_refs, 4e-6f * diag, kStride, false, order, sc.nodes, sc.depth);
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    sc.tris.resize(9ull * n_refs);
    for (uint32_t k = 0; k < n_refs; k++) std::memcpy(&sc.tris[9ull * k], &src[9ull * ref_tri[order[k]]], 36);
    // camera and light of akari_render_amd/procedural.py (hall 30 x 12 x 15)
    const double L = 30, H = 12;
    const V3 eye{-L / 2 + 1.0, 1.7, 0.3}, fwd = norm(V3{1.0, 0.08, 0.05}), right = norm(cross(fwd, V3{0, 1, 0})), up = cross(right, fwd);
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    Counters c, cc, cs;
    std::vector<uint32_t> pair_steps;
    const double tanh_ = std::tan(70.0 * M_PI / 360.0), aspect = 1920.0 / 1080.0;
    for (uint32_t p = 0; p < n_paths; p++) {
        const double sx = (2 * U(rng) - 1) * tanh_ * aspect, sy = (2 * U(rng) - 1) * tanh_;
        V3 o = eye, d = norm(fwd + right * sx + up * sy);
        uint32_t pending_shadow = 0;
        for (uint32_t depth = 0; depth <= max_depth; depth++) {
            double t;
            uint32_t st;
            Counters before = c;
            const int hit = trace(sc, o, d, 1e20, false, t, c, st);
            cc.nodes += c.nodes - before.nodes; cc.tris += c.tris - before.tris; cc.rays++;
            pair_steps.push_back(st + pending_shadow);
            pending_shadow = 0;
            if (hit < 0) break;
            const float* v = &sc.tris[9ull * hit];
            V3 ng = norm(cross(V3{v[3] - v[0], v[4] - v[1], v[5] - v[2]}, V3{v[6] - v[0], v[7] - v[1], v[8] - v[2]}));
            if (dot(ng, d) > 0) ng = ng * -1.0;
            const V3 x = o + d * t + ng * 1e-4;
            {  // shadow ray to a point of the 2 x 2 light under the ceiling
                const V3 y{2 * U(rng) - 1, H - 0.35, 2 * U(rng) - 1};
                V3 w = y - x;
                const double dist = std::sqrt(dot(w, w));
                double ts;
                Counters b2 = c;
                trace(sc, x, w * (1.0 / dist), dist * (1 - 1e-3), true, ts, c, st);
                cs.nodes += c.nodes - b2.nodes; cs.tris += c.tris - b2.tris; cs.rays++;
                pending_shadow = st;
            }
            // cosine-weighted bounce
            const double r1 = U(rng), r2 = U(rng), r = std::sqrt(r1), ph = 2 * M_PI * r2;
            const V3 tt = norm(std::fabs(ng.x) > 0.5 ? cross(ng, V3{0, 1, 0}) : cross(ng, V3{1, 0, 0})), bb = cross(ng, tt);
            d = norm(tt * (r * std::cos(ph)) + bb * (r * std::sin(ph)) + ng * std::sqrt(std::max(0.0, 1 - r1)));
            o = x;
            if (depth >= 3 && U(rng) > 0.7) break;  // roulette-like thinning of long paths
        }
    }
    // divergence proxy: lanes of a wave hold path vertices of random depth; utilisation = mean(pair) / mean(max over 64 pairs)
    std::shuffle(pair_steps.begin(), pair_steps.end(), rng);
    double sum = 0, summax = 0;
    size_t groups = pair_steps.size() / 64;
    for (size_t g = 0; g < groups; g++) {
        uint32_t mx = 0;
        for (int i = 0; i < 64; i++) { sum += pair_steps[g * 64 + i]; mx = std::max(mx, pair_steps[g * 64 + i]); }
        summax += mx;
    }
    std::printf("{\"references\": %u, \"n_tris\": %u, \"node_slots\": %zu, \"depth\": %u, \"build_s\": %.2f, \"rays\": %llu, \"nodes_per_ray\": %.3f, \"tris_per_ray\": %.3f, "
                "\"closest\": {\"nodes\": %.3f, \"tris\": %.3f}, \"shadow\": {\"nodes\": %.3f, \"tris\": %.3f}, \"lane_utilisation_proxy\": %.3f}\n",
                n_refs, n, sc.nodes.size() / kStride, sc.depth, build_s, (unsigned long long)c.rays, (double)c.nodes / c.rays, (double)c.tris / c.rays,
                (double)cc.nodes / cc.rays, (double)cc.tris / cc.rays, (double)cs.nodes / cs.rays, (double)cs.tris / cs.rays, sum / (64.0 * summax));
    return 0;
}
