// ray_sort_sim.cpp -- would sorting the wavefront schedule's ray queues make the BVH traversal of a big scene miss less in the
// caches? A host-side model, to price the idea before building it.
//
// The 10 M-triangle hall is bound by the number of L2 misses per ray (profiles/r4_fetch_size_calibration.json, r4_pmc_c4.json:
// 8.2 misses per ray at 23 record fetches per ray). This program builds the same tree (host/bvh.cpp), traces the wavefront
// schedule's iterations breadth-first (all camera rays, then all first-bounce rays + their shadow rays, ...) with the device's
// traversal rules (group stack, octant order; tools/bvh_sim.cpp's traversal as a steppable state machine), and pushes every
// 64-byte record fetch through a model of the memory side:
//   * waves of 64 consecutive queue entries; a wave's lanes step together and identical addresses of one step are ONE request
//   * 4096 waves resident (256 CUs x 4 SIMDs x 4), 512 per XCD, served round-robin one step at a time; a finished wave claims the
//     next 64 entries of the queue (the persistent trace kernel's queue head)
//   * per XCD a 4 MiB, 16-way, 128-byte-line LRU L2; behind the eight of them one 256 MiB, 16-way LRU Infinity Cache
// for four orders of each iteration's queue:
//   slot     queue order as today (slot order = pixel order: 8x8 pixel blocks, tiles of 32x32)
//   morton   sorted by (21-bit Morton code of the origin, 3 octant bits of the direction), closest-hit and shadow queues apart
//   merged   the same key, both kinds of ray in one queue
//   octant   sorted by (octant, Morton) -- direction first
// and reports, per order: record fetches per ray, requests per ray after the per-wave merge, L2 misses and Infinity Cache misses
// per ray. Build + run: tools/ray_sort_sim.sh. Measurement only.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

namespace akr {
void build_bvh8(const std::vector<float>& tri_bounds, uint32_t n_tris, float pad, uint32_t stride, bool balanced, std::vector<uint32_t>& order_out,
                std::vector<uint32_t>& out_nodes, uint32_t& depth_out);
}
static const uint32_t kStride = 16;
struct Kid { uint32_t meta; float lo[3], hi[3]; };
static float u2f_(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static void decode(const uint32_t* n, Kid k[6], uint32_t& child_base, uint32_t& tri_base) {
    child_base = (n[3] >> 24) | ((n[4] & 0xffffu) << 8);
    tri_base = n[6];
    for (int e = 0; e < 6; e++) {
        k[e].meta = e < 4 ? (n[5] >> (8 * e)) & 0xffu : (n[4] >> (16 + 8 * (e - 4))) & 0xffu;
        for (int a = 0; a < 3; a++) {
            const float scale = u2f_(((n[3] >> (8 * a)) & 0xffu) << 23);
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

// One ray's traversal, one record fetch per step() (disect.h trav_step: a node of the pending group, or a triangle of the leaf mask)
struct Ray {
    V3 o, d;
    double tmax;
    bool any_hit;
    uint32_t slot;  // the path slot this ray belongs to
    uint32_t kind;  // 0 closest-hit, 1 shadow
    // state
    double inv[3], org[3], best_t;
    int best;
    uint32_t oi, G, T, tbase;
    uint32_t stack[28], sp;
    bool done;
    void begin() {
        inv[0] = 1.0 / (std::fabs(d.x) < 1e-20 ? std::copysign(1e-20, d.x) : d.x);
        inv[1] = 1.0 / (std::fabs(d.y) < 1e-20 ? std::copysign(1e-20, d.y) : d.y);
        inv[2] = 1.0 / (std::fabs(d.z) < 1e-20 ? std::copysign(1e-20, d.z) : d.z);
        org[0] = o.x; org[1] = o.y; org[2] = o.z;
        oi = (inv[0] >= 0 ? 1u : 0u) | (inv[1] >= 0 ? 2u : 0u) | (inv[2] >= 0 ? 4u : 0u);
        G = 1u << (24 + oi); T = 0; tbase = 0;
        sp = 0;
        best_t = tmax; best = -1; done = false;
    }
    // performs one record fetch; returns its address (node: index * 64; triangle: (1 << 40) + index * 64), or ~0 when the ray is done
    uint64_t step(const Scene& sc) {
        if (done) return ~0ull;
        if (T != 0) {
            const uint32_t b = (uint32_t)__builtin_ctz(T);
            T &= T - 1;
            const uint32_t tri = tbase + b;
            const float* v = &sc.tris[9ull * tri];
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
                    if (any_hit) { best_t = t; best = (int)tri; done = true; }
                    else if (t < best_t) { best_t = t; best = (int)tri; }
                }
            }
            if (!done && T == 0 && (G >> 24) == 0 && sp == 0) done = true;
            return (1ull << 40) + 64ull * tri;
        }
        if ((G >> 24) == 0) {
            if (sp == 0) { done = true; return ~0ull; }
            G = stack[--sp];
        }
        const uint32_t j = 31u - (uint32_t)__builtin_clz(G);
        G &= ~(1u << j);
        if ((G >> 24) != 0) stack[sp++] = G;
        const uint32_t slot_ = (j - 24u) ^ oi;
        const uint32_t node = (G & 0xffffffu) + slot_;
        const uint32_t* n = &sc.nodes[(size_t)kStride * node];
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
        G = cbase | (hitmask & 0xff000000u);
        T = hitmask & 0x00ffffffu;
        tbase = tb;
        if (T == 0 && (G >> 24) == 0 && sp == 0) done = true;
        return 64ull * node;
    }
};

// set-associative LRU cache of 128-byte lines
struct Cache {
    uint32_t sets, ways;
    std::vector<uint64_t> tag;   // sets * ways, ~0 = empty
    std::vector<uint32_t> stamp;
    uint32_t clock = 0;
    uint64_t hits = 0, misses = 0;
    Cache(size_t bytes, uint32_t ways_) : ways(ways_) {
        sets = (uint32_t)(bytes / 128 / ways_);
        tag.assign((size_t)sets * ways, ~0ull);
        stamp.assign((size_t)sets * ways, 0);
    }
    bool access(uint64_t addr) {  // true = hit
        const uint64_t line = addr >> 7;
        const uint32_t set = (uint32_t)((line * 0x9e3779b97f4a7c15ull) >> 40) % sets;
        uint64_t* t = &tag[(size_t)set * ways];
        uint32_t* s = &stamp[(size_t)set * ways];
        clock++;
        uint32_t victim = 0;
        for (uint32_t w = 0; w < ways; w++) {
            if (t[w] == line) { s[w] = clock; hits++; return true; }
            if (s[w] < s[victim]) victim = w;
        }
        t[victim] = line;
        s[victim] = clock;
        misses++;
        return false;
    }
};

struct Stats { uint64_t rays = 0, fetches = 0, requests = 0, l2_miss = 0, mall_miss = 0, wave_steps = 0, lane_steps = 0; };

// Traces `rays` (already in queue order) through the machine model; fills hit results back into the rays.
static void trace_queue(const Scene& sc, std::vector<Ray>& rays, std::vector<Cache>& l2, Cache& mall, Stats& st, uint32_t n_resident) {
    const size_t n = rays.size();
    size_t head = 0;
    struct Wave { size_t first, count; };
    std::vector<Wave> resident(n_resident, Wave{0, 0});
    auto claim = [&](Wave& w) {
        w.first = head;
        w.count = std::min<size_t>(64, n - head);
        head += w.count;
        for (size_t i = 0; i < w.count; i++) rays[w.first + i].begin();
    };
    for (auto& w : resident) claim(w);
    std::vector<uint64_t> addrs;
    bool any = true;
    while (any) {
        any = false;
        for (uint32_t wi = 0; wi < n_resident; wi++) {
            Wave& w = resident[wi];
            if (w.count == 0) continue;
            addrs.clear();
            for (size_t i = 0; i < w.count; i++) {
                const uint64_t a = rays[w.first + i].step(sc);
                if (a != ~0ull) addrs.push_back(a);
            }
            if (addrs.empty()) {  // every lane done: the wave takes the next entries of the queue
                claim(w);
                if (w.count) any = true;
                continue;
            }
            any = true;
            st.wave_steps++;
            st.lane_steps += addrs.size();
            st.fetches += addrs.size();
            std::sort(addrs.begin(), addrs.end());
            addrs.erase(std::unique(addrs.begin(), addrs.end()), addrs.end());
            st.requests += addrs.size();
            Cache& c = l2[wi % l2.size()];  // wave -> XCD: workgroups are dealt round-robin
            for (uint64_t a : addrs)
                if (!c.access(a)) {
                    st.l2_miss++;
                    if (!mall.access(a)) st.mall_miss++;
                }
        }
    }
    st.rays += n;
}

static uint32_t part1by2(uint32_t x) {  // spread the low 7 bits
    x &= 0x7fu;
    x = (x | (x << 8)) & 0x0000700fu;
    x = (x | (x << 4)) & 0x000430c3u;
    x = (x | (x << 2)) & 0x00049249u;
    return x;
}
struct Bounds { double lo[3], hi[3]; };
static uint32_t morton21(const Bounds& b, V3 p) {
    const double q[3] = {p.x, p.y, p.z};
    uint32_t c[3];
    for (int a = 0; a < 3; a++) {
        double t = (q[a] - b.lo[a]) / (b.hi[a] - b.lo[a]);
        t = std::min(std::max(t, 0.0), 0.999999);
        c[a] = (uint32_t)(t * 128.0);
    }
    return part1by2(c[0]) | (part1by2(c[1]) << 1) | (part1by2(c[2]) << 2);
}
static uint32_t octant(V3 d) { return (d.x >= 0 ? 1u : 0u) | (d.y >= 0 ? 2u : 0u) | (d.z >= 0 ? 4u : 0u); }

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: ray_sort_sim tris.f32 [width=512] [height=288] [max_depth=6] [out.json]\n"); return 1; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    std::fseek(f, 0, SEEK_END);
    const size_t bytes = (size_t)std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const uint32_t n = (uint32_t)(bytes / 36);
    std::vector<float> src(9ull * n);
    if (std::fread(src.data(), 4, src.size(), f) != src.size()) return 1;
    std::fclose(f);
    const uint32_t W = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 512u, H = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 288u;
    const uint32_t max_depth = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 6u;
    std::vector<float> bounds(6ull * n);
    Bounds sb{{1e30, 1e30, 1e30}, {-1e30, -1e30, -1e30}};
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            const float x0 = src[9ull * i + a], x1 = src[9ull * i + 3 + a], x2 = src[9ull * i + 6 + a];
            bounds[6ull * i + a] = std::min(x0, std::min(x1, x2));
            bounds[6ull * i + 3 + a] = std::max(x0, std::max(x1, x2));
            sb.lo[a] = std::min<double>(sb.lo[a], bounds[6ull * i + a]);
            sb.hi[a] = std::max<double>(sb.hi[a], bounds[6ull * i + 3 + a]);
        }
    const float diag = (float)std::sqrt((sb.hi[0] - sb.lo[0]) * (sb.hi[0] - sb.lo[0]) + (sb.hi[1] - sb.lo[1]) * (sb.hi[1] - sb.lo[1]) + (sb.hi[2] - sb.lo[2]) * (sb.hi[2] - sb.lo[2]));
    Scene sc;
    std::vector<uint32_t> order;
    auto t0 = std::chrono::steady_clock::now();
    akr::build_bvh8(bounds, n, 4e-6f * diag, kStride, false, order, sc.nodes, sc.depth);
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    sc.tris.resize(9ull * n);
    for (uint32_t k = 0; k < n; k++) std::memcpy(&sc.tris[9ull * k], &src[9ull * order[k]], 36);
    src.clear(); src.shrink_to_fit();
    std::fprintf(stderr, "%u triangles, %zu node slots, depth %u, built in %.1f s\n", n, sc.nodes.size() / kStride, sc.depth, build_s);

    // camera and light of akari_render_amd/procedural.py (hall 30 x 12 x 15)
    const double L = 30, Hh = 12;
    const V3 eye{-L / 2 + 1.0, 1.7, 0.3}, fwd = norm(V3{1.0, 0.08, 0.05}), right = norm(cross(fwd, V3{0, 1, 0})), up = cross(right, fwd);
    const double tanh_ = std::tan(70.0 * M_PI / 360.0), aspect = (double)W / H;
    // slot order of the wavefront schedule = item order of the megakernel: 32x32 tiles, 8x8 blocks inside (dpath.h item_to_pixel)
    std::vector<uint32_t> slot_px, slot_py;
    for (uint32_t ty = 0; ty < (H + 31) / 32; ty++)
        for (uint32_t tx = 0; tx < (W + 31) / 32; tx++)
            for (uint32_t by = 0; by < 4; by++)
                for (uint32_t bx = 0; bx < 4; bx++)
                    for (uint32_t l = 0; l < 64; l++) {
                        const uint32_t px = tx * 32 + bx * 8 + (l & 7), py = ty * 32 + by * 8 + (l >> 3);
                        if (px < W && py < H) { slot_px.push_back(px); slot_py.push_back(py); }
                    }
    const uint32_t n_slots = (uint32_t)slot_px.size();

    const char* names[4] = {"slot", "morton", "merged", "octant"};
    std::string json = "{";
    json += "\"n_tris\": " + std::to_string(n) + ", \"node_slots\": " + std::to_string(sc.nodes.size() / kStride) + ", \"frame\": [" + std::to_string(W) + ", " + std::to_string(H) +
            "], \"max_depth\": " + std::to_string(max_depth) + ", \"resident_waves\": 4096, \"l2\": \"8 x 4 MiB, 16-way, 128 B lines, LRU\", \"infinity_cache\": \"256 MiB, 16-way, LRU\", \"orders\": {";
    for (int mode = 0; mode < 4; mode++) {
        std::vector<Cache> l2;
        for (int x = 0; x < 8; x++) l2.emplace_back((size_t)4 << 20, 16);
        Cache mall((size_t)256 << 20, 16);
        Stats total, per_iter[16];
        std::mt19937_64 rng(12345);  // the same paths for every order: random numbers are drawn per slot in slot order below
        std::uniform_real_distribution<double> U(0.0, 1.0);
        struct Path { V3 o, d; bool alive, has_shadow; V3 so, sd; double stmax; uint32_t depth; };
        std::vector<Path> paths(n_slots);
        for (uint32_t s = 0; s < n_slots; s++) {
            const double sx = (2 * ((slot_px[s] + U(rng)) / W) - 1) * tanh_ * aspect, sy = (1 - 2 * ((slot_py[s] + U(rng)) / H)) * tanh_;
            paths[s] = Path{eye, norm(fwd + right * sx + up * sy), true, false, {}, {}, 0, 0};
        }
        for (uint32_t iter = 0; iter <= max_depth; iter++) {
            std::vector<Ray> qc, qs;
            for (uint32_t s = 0; s < n_slots; s++) {
                if (paths[s].alive) { Ray r; r.o = paths[s].o; r.d = paths[s].d; r.tmax = 1e20; r.any_hit = false; r.slot = s; r.kind = 0; qc.push_back(r); }
                if (paths[s].has_shadow) { Ray r; r.o = paths[s].so; r.d = paths[s].sd; r.tmax = paths[s].stmax; r.any_hit = true; r.slot = s; r.kind = 1; qs.push_back(r); }
            }
            if (qc.empty() && qs.empty()) break;
            auto key = [&](const Ray& r) -> uint32_t {
                const uint32_t m = morton21(sb, r.o), oc = octant(r.d);
                return mode == 3 ? (oc << 21) | m : (m << 3) | oc;
            };
            auto by_key = [&](const Ray& a, const Ray& b) { return key(a) < key(b); };
            Stats it;
            if (mode == 2) {
                std::vector<Ray> q(qc);
                q.insert(q.end(), qs.begin(), qs.end());
                std::stable_sort(q.begin(), q.end(), by_key);
                trace_queue(sc, q, l2, mall, it, 4096);
                qc.clear(); qs.clear();
                for (Ray& r : q) (r.kind == 0 ? qc : qs).push_back(r);
            } else {
                if (mode != 0) { std::stable_sort(qc.begin(), qc.end(), by_key); std::stable_sort(qs.begin(), qs.end(), by_key); }
                // (the schedule traces both queues in one persistent launch: closest-hit entries first, then the shadow entries)
                std::vector<Ray> q(qc);
                q.insert(q.end(), qs.begin(), qs.end());
                trace_queue(sc, q, l2, mall, it, 4096);
                std::copy(q.begin(), q.begin() + qc.size(), qc.begin());
                std::copy(q.begin() + qc.size(), q.end(), qs.begin());
            }
            per_iter[iter] = it;
            total.rays += it.rays; total.fetches += it.fetches; total.requests += it.requests; total.l2_miss += it.l2_miss; total.mall_miss += it.mall_miss;
            total.wave_steps += it.wave_steps; total.lane_steps += it.lane_steps;
            // shade: in slot order, so that every order draws the same random numbers for the same slot
            std::vector<int> hit_of(n_slots, -2);
            std::vector<double> t_of(n_slots, 0.0);
            for (const Ray& r : qc) { hit_of[r.slot] = r.best; t_of[r.slot] = r.best_t; }
            for (uint32_t s = 0; s < n_slots; s++) {
                Path& p = paths[s];
                p.has_shadow = false;
                if (!p.alive) continue;
                const int hit = hit_of[s];
                if (hit < 0) { p.alive = false; continue; }
                const float* v = &sc.tris[9ull * hit];
                V3 ng = norm(cross(V3{v[3] - v[0], v[4] - v[1], v[5] - v[2]}, V3{v[6] - v[0], v[7] - v[1], v[8] - v[2]}));
                if (dot(ng, p.d) > 0) ng = ng * -1.0;
                const V3 x = p.o + p.d * t_of[s] + ng * 1e-4;
                const V3 y{2 * U(rng) - 1, Hh - 0.35, 2 * U(rng) - 1};
                V3 w = y - x;
                const double dist = std::sqrt(dot(w, w));
                p.has_shadow = true; p.so = x; p.sd = w * (1.0 / dist); p.stmax = dist * (1 - 1e-3);
                const double r1 = U(rng), r2 = U(rng), r = std::sqrt(r1), ph = 2 * M_PI * r2;
                const V3 tt = norm(std::fabs(ng.x) > 0.5 ? cross(ng, V3{0, 1, 0}) : cross(ng, V3{1, 0, 0})), bb = cross(ng, tt);
                p.d = norm(tt * (r * std::cos(ph)) + bb * (r * std::sin(ph)) + ng * std::sqrt(std::max(0.0, 1 - r1)));
                p.o = x;
                p.depth++;
                const double rr = U(rng);
                if (p.depth >= 3 && rr > 0.7) p.alive = false;
                if (p.depth > max_depth) p.alive = false;
            }
            std::fprintf(stderr, "  %-7s iteration %u: %llu rays, %.2f fetches/ray, %.2f requests/ray, %.2f L2 misses/ray, %.2f MALL misses/ray, lane utilisation %.2f\n", names[mode], iter,
                         (unsigned long long)it.rays, (double)it.fetches / it.rays, (double)it.requests / it.rays, (double)it.l2_miss / it.rays, (double)it.mall_miss / it.rays,
                         (double)it.lane_steps / (64.0 * it.wave_steps));
        }
        char buf[1024];
        std::snprintf(buf, sizeof buf,
                      "%s\"%s\": {\"rays\": %llu, \"fetches_per_ray\": %.3f, \"requests_per_ray\": %.3f, \"l2_misses_per_ray\": %.3f, \"infinity_cache_misses_per_ray\": %.3f, "
                      "\"l2_hit\": %.3f, \"lane_utilisation\": %.3f, \"wave_steps_per_ray\": %.4f, \"per_iteration_l2_misses_per_ray\": [",
                      mode ? ", " : "", names[mode], (unsigned long long)total.rays, (double)total.fetches / total.rays, (double)total.requests / total.rays,
                      (double)total.l2_miss / total.rays, (double)total.mall_miss / total.rays, 1.0 - (double)total.l2_miss / total.requests,
                      (double)total.lane_steps / (64.0 * total.wave_steps), (double)total.wave_steps / total.rays);
        json += buf;
        for (uint32_t i = 0; i <= max_depth && per_iter[i].rays; i++) {
            std::snprintf(buf, sizeof buf, "%s%.3f", i ? ", " : "", (double)per_iter[i].l2_miss / per_iter[i].rays);
            json += buf;
        }
        json += "]}";
    }
    json += "}}";
    std::printf("%s\n", json.c_str());
    if (argc > 5) {
        FILE* o = std::fopen(argv[5], "w");
        if (o) { std::fprintf(o, "%s\n", json.c_str()); std::fclose(o); }
    }
    return 0;
}
