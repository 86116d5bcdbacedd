// bvh.cpp -- host BVH4 builder (replaces the BLAS/TLAS build the reference delegates to LuisaCompute,
// crates/akari_render/src/mesh.rs:288-294,331-333).
//
// Binned-SAH binary build over world-space triangle boxes (instances are flattened: every triangle is stored
// once per instance, which is exact for the rigid/affine instance transforms of the scene graph), collapsed to
// a 4-wide tree. Node = 64 B = 4 x 16-byte words (half of an f32 SoA node, so twice as many nodes per cache line
// and four instead of seven loads per visit):
//   word 0: origin.xyz (f32, the node's own padded lower corner) | exponent bytes ex, ey, ez (scale_a = 2^(e_a - 127))
//   word 1: q_lo.x[4] | q_lo.y[4] | q_lo.z[4] | q_hi.x[4]      (one byte per child)
//   word 2: q_hi.y[4] | q_hi.z[4] | child[0] | child[1]
//   word 3: child[2] | child[3] | unused | unused
// child box = origin + q * scale per axis, q_lo rounded down and q_hi rounded up (and verified in double), so the
// decoded box always contains the exact padded box.
// child reference: inner node -> node index; leaf -> 0x80000000 | count << 28 | first triangle (count 1..4);
// empty slot -> reference 0xffffffff (its box is (+inf, -inf)); traversal skips it by reference.
// Boxes are padded by `pad` so that every triangle the exhaustive test would report is reached by traversal
// (the triangle test itself has an absolute slop of a few ulp(t); see DESIGN.md "BVH conservativeness").
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

namespace akr {

namespace {
struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::numeric_limits<float>::infinity();
            hi[a] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float* b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], b[a]);
            hi[a] = std::max(hi[a], b[3 + a]);
        }
    }
    void grow(const Box& o) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], o.lo[a]);
            hi[a] = std::max(hi[a], o.hi[a]);
        }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};
struct BinNode {
    Box box;
    int32_t left = -1, right = -1;  // children (binary) or -1
    uint32_t first = 0, count = 0;  // leaf range in `order`
};

constexpr int kBins = 16;
constexpr uint32_t kLeafMax = 4;

struct Builder {
    const float* bounds;
    std::vector<uint32_t>& order;
    std::vector<float> centroid;  // 3 / tri
    std::vector<BinNode> nodes;

    Builder(const float* b, uint32_t n, std::vector<uint32_t>& ord) : bounds(b), order(ord) {
        order.resize(n);
        std::iota(order.begin(), order.end(), 0u);
        centroid.resize(3ull * n);
        for (uint32_t i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) centroid[3ull * i + a] = 0.5f * (b[6ull * i + a] + b[6ull * i + 3 + a]);
        nodes.reserve(2ull * n / 2 + 16);
    }

    int32_t build(uint32_t first, uint32_t count) {
        // iterative to survive degenerate inputs; explicit stack of (node, first, count)
        struct Item { int32_t node; uint32_t first, count; };
        std::vector<Item> stack;
        nodes.emplace_back();
        stack.push_back({0, first, count});
        while (!stack.empty()) {
            Item it = stack.back();
            stack.pop_back();
            Box box, cbox;
            box.reset();
            cbox.reset();
            for (uint32_t i = it.first; i < it.first + it.count; i++) {
                box.grow(bounds + 6ull * order[i]);
                const float* c = &centroid[3ull * order[i]];
                for (int a = 0; a < 3; a++) {
                    cbox.lo[a] = std::min(cbox.lo[a], c[a]);
                    cbox.hi[a] = std::max(cbox.hi[a], c[a]);
                }
            }
            nodes[it.node].box = box;
            nodes[it.node].first = it.first;
            nodes[it.node].count = it.count;
            if (it.count <= kLeafMax) continue;
            // binned SAH over the three axes
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_split = -1;
            for (int axis = 0; axis < 3; axis++) {
                float lo = cbox.lo[axis], ext = cbox.hi[axis] - cbox.lo[axis];
                if (!(ext > 0.0f)) continue;
                Box bin_box[kBins];
                uint32_t bin_cnt[kBins] = {0};
                for (auto& b : bin_box) b.reset();
                float scale = (float)kBins / ext;
                for (uint32_t i = it.first; i < it.first + it.count; i++) {
                    int b = (int)((centroid[3ull * order[i] + axis] - lo) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    bin_cnt[b]++;
                    bin_box[b].grow(bounds + 6ull * order[i]);
                }
                float right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    acc.grow(bin_box[b]);
                    cnt += bin_cnt[b];
                    right_area[b] = acc.half_area();
                    right_cnt[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.grow(bin_box[b]);
                    cnt += bin_cnt[b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = axis;
                        best_split = b;
                    }
                }
            }
            uint32_t mid;
            if (best_axis < 0) {
                mid = it.first + it.count / 2;  // all centroids coincide: split by index
            } else {
                float lo = cbox.lo[best_axis], ext = cbox.hi[best_axis] - cbox.lo[best_axis];
                float scale = (float)kBins / ext;
                auto* beg = order.data() + it.first;
                auto* end = beg + it.count;
                auto* m = std::partition(beg, end, [&](uint32_t t) {
                    int b = (int)((centroid[3ull * t + best_axis] - lo) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    return b <= best_split;
                });
                mid = it.first + (uint32_t)(m - beg);
                if (mid == it.first || mid == it.first + it.count) mid = it.first + it.count / 2;
            }
            int32_t l = (int32_t)nodes.size();
            nodes.emplace_back();
            int32_t r = (int32_t)nodes.size();
            nodes.emplace_back();
            nodes[it.node].left = l;
            nodes[it.node].right = r;
            stack.push_back({l, it.first, mid - it.first});
            stack.push_back({r, mid, it.first + it.count - mid});
        }
        return 0;
    }
};
}  // namespace

void build_bvh4(const std::vector<float>& tri_bounds, uint32_t n_tris, float pad, std::vector<uint32_t>& order, std::vector<float>& out_nodes) {
    Builder b(tri_bounds.data(), n_tris, order);
    b.build(0, n_tris);
    const auto& bn = b.nodes;
    // collapse: each BVH4 node adopts up to 4 descendants of a binary node, always opening the child with the
    // largest surface area first
    struct Pending { int32_t bin; uint32_t out; };
    std::vector<Pending> queue;
    out_nodes.clear();
    out_nodes.resize(16, 0.0f);  // 16 words per node
    queue.push_back({0, 0});
    auto put_u32 = [](float* p, uint32_t v) { std::memcpy(p, &v, 4); };
    // depth-first emission (LIFO): a node's inner children get consecutive slots right after the nodes emitted so far and
    // each subtree is laid out before its siblings' subtrees, so a ray that descends stays within a few DRAM pages / L2
    // lines instead of jumping level by level through the array as a breadth-first layout would make it do
    while (!queue.empty()) {
        Pending pe = queue.back();
        queue.pop_back();
        int32_t kids[4];
        int nk = 0;
        const BinNode& root = bn[pe.bin];
        if (root.left < 0) {
            kids[nk++] = pe.bin;  // the whole tree is a single leaf
        } else {
            kids[nk++] = root.left;
            kids[nk++] = root.right;
            while (nk < 4) {
                int pick = -1;
                float best = -1.0f;
                for (int i = 0; i < nk; i++) {
                    if (bn[kids[i]].left < 0) continue;
                    float a = bn[kids[i]].box.half_area();
                    if (a > best) { best = a; pick = i; }
                }
                if (pick < 0) break;
                int32_t k = kids[pick];
                kids[pick] = bn[k].left;
                kids[nk++] = bn[k].right;
            }
        }
        // padded child boxes and their union (= this node's frame)
        float clo[4][3], chi[4][3], origin[3], top[3];
        for (int a = 0; a < 3; a++) { origin[a] = std::numeric_limits<float>::infinity(); top[a] = -origin[a]; }
        for (int i = 0; i < nk; i++)
            for (int a = 0; a < 3; a++) {
                clo[i][a] = bn[kids[i]].box.lo[a] - pad;
                chi[i][a] = bn[kids[i]].box.hi[a] + pad;
                origin[a] = std::min(origin[a], clo[i][a]);
                top[a] = std::max(top[a], chi[i][a]);
            }
        uint32_t ebits[3];
        double scale[3];
        for (int a = 0; a < 3; a++) {
            double ext = (double)top[a] - (double)origin[a];
            int e = -100;
            if (ext > 0.0) {
                e = (int)std::ceil(std::log2(ext / 255.0));
                while (std::ldexp(255.0, e) < ext) e++;  // guard against log2 rounding
            }
            e = std::max(-126, std::min(127, e));
            ebits[a] = (uint32_t)(e + 127);
            scale[a] = std::ldexp(1.0, e);
        }
        uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0}, refs[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        for (int i = 0; i < 4; i++) {
            uint32_t lo_q[3] = {255, 255, 255}, hi_q[3] = {0, 0, 0};  // empty slot: inverted
            if (i < nk) {
                const BinNode& c = bn[kids[i]];
                for (int a = 0; a < 3; a++) {
                    double l = std::floor(((double)clo[i][a] - (double)origin[a]) / scale[a]);
                    double h = std::ceil(((double)chi[i][a] - (double)origin[a]) / scale[a]);
                    l = std::max(0.0, std::min(255.0, l));
                    h = std::max(0.0, std::min(255.0, h));
                    // the f32 decode origin + q * scale rounds to nearest: step outwards until it is conservative
                    while (l > 0.0 && (float)((double)origin[a] + l * scale[a]) > clo[i][a]) l -= 1.0;
                    while (h < 255.0 && (float)((double)origin[a] + h * scale[a]) < chi[i][a]) h += 1.0;
                    lo_q[a] = (uint32_t)l;
                    hi_q[a] = (uint32_t)h;
                }
                if (c.left < 0) {
                    refs[i] = 0x80000000u | (c.count << 28) | c.first;
                } else {
                    uint32_t idx = (uint32_t)(out_nodes.size() / 16);
                    out_nodes.resize(out_nodes.size() + 16, 0.0f);
                    queue.push_back({kids[i], idx});
                    refs[i] = idx;
                }
            }
            for (int a = 0; a < 3; a++) {
                qlo[a] |= lo_q[a] << (8 * i);
                qhi[a] |= hi_q[a] << (8 * i);
            }
        }
        float* n = &out_nodes[16ull * pe.out];
        n[0] = origin[0]; n[1] = origin[1]; n[2] = origin[2];
        put_u32(&n[3], ebits[0] | (ebits[1] << 8) | (ebits[2] << 16));
        put_u32(&n[4], qlo[0]); put_u32(&n[5], qlo[1]); put_u32(&n[6], qlo[2]); put_u32(&n[7], qhi[0]);
        put_u32(&n[8], qhi[1]); put_u32(&n[9], qhi[2]); put_u32(&n[10], refs[0]); put_u32(&n[11], refs[1]);
        put_u32(&n[12], refs[2]); put_u32(&n[13], refs[3]); put_u32(&n[14], 0); put_u32(&n[15], 0);
    }
}

}  // namespace akr
