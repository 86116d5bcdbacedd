// bvh.cpp -- host BVH4 builder (replaces the BLAS/TLAS build the reference delegates to LuisaCompute,
// crates/akari_render/src/mesh.rs:288-294,331-333).
//
// Binned-SAH binary build over world-space triangle boxes (instances are flattened: every triangle is stored
// once per instance, which is exact for the rigid/affine instance transforms of the scene graph), collapsed to
// a 4-wide tree. Node = 128 B = 8 x float4, child boxes in SoA so one node is seven 16-byte loads:
//   row 0/1: lo.x[4] / hi.x[4]   row 2/3: lo.y[4] / hi.y[4]   row 4/5: lo.z[4] / hi.z[4]
//   row 6  : child reference [4] (u32 bits)      row 7: unused
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
    out_nodes.resize(32, 0.0f);
    queue.push_back({0, 0});
    size_t qi = 0;
    const float inf = std::numeric_limits<float>::infinity();
    auto put_u32 = [](float* p, uint32_t v) { std::memcpy(p, &v, 4); };
    while (qi < queue.size()) {
        Pending pe = queue[qi++];
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
        for (int i = 0; i < 4; i++) {
            float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
            uint32_t ref = 0xffffffffu;
            if (i < nk) {
                const BinNode& c = bn[kids[i]];
                for (int a = 0; a < 3; a++) { lo[a] = c.box.lo[a] - pad; hi[a] = c.box.hi[a] + pad; }
                if (c.left < 0) {
                    ref = 0x80000000u | (c.count << 28) | c.first;
                } else {
                    uint32_t idx = (uint32_t)(out_nodes.size() / 32);
                    out_nodes.resize(out_nodes.size() + 32, 0.0f);
                    queue.push_back({kids[i], idx});
                    ref = idx;
                }
            }
            float* n = &out_nodes[32ull * pe.out];
            n[0 + i] = lo[0]; n[4 + i] = hi[0];
            n[8 + i] = lo[1]; n[12 + i] = hi[1];
            n[16 + i] = lo[2]; n[20 + i] = hi[2];
            put_u32(&n[24 + i], ref);
        }
    }
}

}  // namespace akr
