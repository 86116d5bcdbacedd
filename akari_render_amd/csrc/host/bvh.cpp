// bvh.cpp -- host builder of the compressed wide BVH (replaces the BLAS/TLAS build the reference delegates to
// LuisaCompute, crates/akari_render/src/mesh.rs:288-294,331-333).
//
// Binned-SAH binary build over world-space triangle boxes (instances are flattened: every triangle is stored once per
// instance, which is exact for the rigid/affine instance transforms of the scene graph), leaves of at most 3 triangles,
// collapsed to a wide tree in a compressed layout after Ylitie, Karras & Laine, "Efficient Incoherent Ray Traversal on GPUs
// Through Compressed Wide BVHs" (HPG 2017), with two changes: a node's inner children live at child_base + octant position
// (unused positions of the block are holes), so that a group of pending children is ONE 32-bit stack entry (24-bit base | 8 hit
// bits); and a node has SIX children in its eight octant positions, which makes it 64 bytes -- one memory sector, like a triangle
// record. (Round 2's 8-wide node was 80 bytes: two sectors wherever it sat. The hall's rays visit 10 % more of the narrower
// nodes and fetch 39 % fewer bytes: tools/bvh_sim.cpp; the traversal is bound by the memory system)
//
// Node = 64 bytes = 16 words (device/disect.h trav_step reads it as four 16-byte loads):
//   word 0-2 : p.xyz (f32, the node's own padded lower corner)
//   word 3   : exponent bytes ex, ey, ez (scale_a = 2^(e_a - 127)) | child_base bits 0-7
//   word 4   : child_base bits 8-23 | meta[4] | meta[5]
//   word 5   : meta[0..3]
//   word 6   : tri_base
//   word 7-12: entries 0..3, one word per plane: q_lo.x, q_lo.y, q_lo.z, q_hi.x, q_hi.y, q_hi.z (one byte per entry)
//   word 13-15: entries 4, 5, one word per axis: q_lo[4], q_lo[5], q_hi[4], q_hi[5]
// child box = p + q * scale per axis, q_lo rounded down and q_hi rounded up (verified in double), so the decoded box always
// contains the exact padded box. Entries are stored in ascending octant position. meta byte of an entry: 0 = empty; inner child:
// 0x20 | (24 + position) -- its node is child_base + position; leaf: (unary triangle count 1 / 3 / 7) << 5 | offset -- its
// triangles are tri_base + offset .. (at most 18 triangles under one node). A child's position is the octant of the node it
// lies in (bit a = the side of the node's centre along axis a): a ray then visits the positions in the order position ^ octant,
// near to far, without sorting distances.
// Triangles are re-ordered so that every node's leaf triangles are contiguous (`order`).
// Boxes are padded by `pad` so that every triangle the exhaustive test would report is reached by traversal
// (the triangle test itself has an absolute slop of a few ulp(t); see DESIGN.md "BVH conservativeness").
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <vector>

#include "host_parallel.h"

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
constexpr size_t kBvhTopSlots = 1024;  // node slots laid out breadth-first (the first four levels of a full tree: 1 + 8 + 64 + 512)
#ifndef AKR_BVH_LEAF_MAX
#define AKR_BVH_LEAF_MAX 3  // triangles per leaf (at most 3: the unary count of a node entry)
#endif
constexpr uint32_t kLeafMax = AKR_BVH_LEAF_MAX;
constexpr int kWide = 6;  // children per node (in 8 octant positions)

struct Builder {
    const float* bounds;
    bool balanced = false;  // object-median splits only: depth = ceil(log2(n / leaf)), whatever the geometry
    std::vector<uint32_t>& order;
    std::vector<float> centroid;  // 3 / tri
    std::vector<BinNode> nodes;
    unsigned threads = 1;
    // A node of more than kWideMin triangles is split with its passes over the triangles (bounds, SAH bins) divided over the
    // threads; the subtrees below kTaskMax triangles are built one per thread. Every quantity a split depends on is a minimum, a
    // maximum or an integer count, and the partition of `order` is the sequential std::partition in both cases: the tree and the
    // triangle order are the same for every thread count (tests/test_host.py compares builds).
    static constexpr uint32_t kWideMin = 1u << 17, kTaskMax = 1u << 16;

    Builder(const float* b, uint32_t n, std::vector<uint32_t>& ord) : bounds(b), order(ord) {
        order.resize(n);
        std::iota(order.begin(), order.end(), 0u);
        centroid.resize(3ull * n);
        threads = host_threads();
        const unsigned chunks = n > kWideMin ? threads * 4 : 1;
        parallel_chunks(chunks, threads, [&](unsigned c) {
            const uint64_t lo = (uint64_t)n * c / chunks, hi = (uint64_t)n * (c + 1) / chunks;
            for (uint64_t i = lo; i < hi; i++)
                for (int a = 0; a < 3; a++) centroid[3ull * i + a] = 0.5f * (b[6ull * i + a] + b[6ull * i + 3 + a]);
        });
        nodes.reserve(2ull * n / 2 + 16);
    }

    struct Item { int32_t node; uint32_t first, count; };
    struct Bins {
        Box box[3][kBins];
        uint32_t cnt[3][kBins];
    };

    // Bounds, the split decision and the partition of one node's triangles. Returns false for a leaf, else `mid`.
    bool split(const Item& it, Box& box_out, uint32_t& mid, unsigned th) {
        const unsigned chunks = (th > 1 && it.count > kWideMin) ? th * 2 : 1;
        auto range = [&](unsigned c, uint32_t& lo, uint32_t& hi) {
            lo = it.first + (uint32_t)((uint64_t)it.count * c / chunks);
            hi = it.first + (uint32_t)((uint64_t)it.count * (c + 1) / chunks);
        };
        Box box, cbox;
        box.reset();
        cbox.reset();
        {
            Box pb1, pc1;
            std::vector<Box> pbv, pcv;
            if (chunks > 1) { pbv.resize(chunks); pcv.resize(chunks); }
            Box* pb = chunks > 1 ? pbv.data() : &pb1;
            Box* pc = chunks > 1 ? pcv.data() : &pc1;
            parallel_chunks(chunks, th, [&](unsigned c) {
                uint32_t lo, hi;
                range(c, lo, hi);
                Box b1, c1;
                b1.reset();
                c1.reset();
                for (uint32_t i = lo; i < hi; i++) {
                    b1.grow(bounds + 6ull * order[i]);
                    const float* cc = &centroid[3ull * order[i]];
                    for (int a = 0; a < 3; a++) {
                        c1.lo[a] = std::min(c1.lo[a], cc[a]);
                        c1.hi[a] = std::max(c1.hi[a], cc[a]);
                    }
                }
                pb[c] = b1;
                pc[c] = c1;
            });
            for (unsigned c = 0; c < chunks; c++) {
                box.grow(pb[c]);
                cbox.grow(pc[c]);
            }
        }
        box_out = box;
        if (it.count <= kLeafMax) return false;
        if (balanced) {  // median of the centroids along their widest axis
            int axis = 0;
            for (int a = 1; a < 3; a++)
                if (cbox.hi[a] - cbox.lo[a] > cbox.hi[axis] - cbox.lo[axis]) axis = a;
            auto* beg = order.data() + it.first;
            std::nth_element(beg, beg + it.count / 2, beg + it.count,
                             [&](uint32_t x, uint32_t y) { return centroid[3ull * x + axis] < centroid[3ull * y + axis]; });
            mid = it.first + it.count / 2;
            return true;
        }
        // binned SAH over the three axes
        Bins part1;
        std::vector<Bins> partv;
        if (chunks > 1) partv.resize(chunks);
        Bins* part = chunks > 1 ? partv.data() : &part1;
        parallel_chunks(chunks, th, [&](unsigned c) {
            uint32_t lo, hi;
            range(c, lo, hi);
            Bins& bn = part[c];
            for (int axis = 0; axis < 3; axis++)
                for (int k = 0; k < kBins; k++) {
                    bn.box[axis][k].reset();
                    bn.cnt[axis][k] = 0;
                }
            for (int axis = 0; axis < 3; axis++) {
                const float clo = cbox.lo[axis], ext = cbox.hi[axis] - cbox.lo[axis];
                if (!(ext > 0.0f)) continue;
                const float scale = (float)kBins / ext;
                for (uint32_t i = lo; i < hi; i++) {
                    int k = (int)((centroid[3ull * order[i] + axis] - clo) * scale);
                    k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                    bn.cnt[axis][k]++;
                    bn.box[axis][k].grow(bounds + 6ull * order[i]);
                }
            }
        });
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_split = -1;
        for (int axis = 0; axis < 3; axis++) {
            const float ext = cbox.hi[axis] - cbox.lo[axis];
            if (!(ext > 0.0f)) continue;
            Box bin_box[kBins];
            uint32_t bin_cnt[kBins];
            for (int k = 0; k < kBins; k++) {
                bin_box[k].reset();
                bin_cnt[k] = 0;
                for (unsigned c = 0; c < chunks; c++) {
                    if (part[c].cnt[axis][k] == 0) continue;
                    bin_box[k].grow(part[c].box[axis][k]);
                    bin_cnt[k] += part[c].cnt[axis][k];
                }
            }
            float right_area[kBins];
            uint32_t right_cnt[kBins];
            Box acc;
            acc.reset();
            uint32_t cnt = 0;
            for (int k = kBins - 1; k > 0; k--) {
                acc.grow(bin_box[k]);
                cnt += bin_cnt[k];
                right_area[k] = acc.half_area();
                right_cnt[k] = cnt;
            }
            acc.reset();
            cnt = 0;
            for (int k = 0; k < kBins - 1; k++) {
                acc.grow(bin_box[k]);
                cnt += bin_cnt[k];
                if (cnt == 0 || right_cnt[k + 1] == 0) continue;
                float cost = acc.half_area() * (float)cnt + right_area[k + 1] * (float)right_cnt[k + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_split = k;
                }
            }
        }
        if (best_axis < 0) {
            mid = it.first + it.count / 2;  // all centroids coincide: split by index
        } else {
            float lo = cbox.lo[best_axis], ext = cbox.hi[best_axis] - cbox.lo[best_axis];
            float scale = (float)kBins / ext;
            auto* beg = order.data() + it.first;
            auto* end = beg + it.count;
            auto* m = std::partition(beg, end, [&](uint32_t t) {
                int k = (int)((centroid[3ull * t + best_axis] - lo) * scale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                return k <= best_split;
            });
            mid = it.first + (uint32_t)(m - beg);
            if (mid == it.first || mid == it.first + it.count) mid = it.first + it.count / 2;
        }
        return true;
    }

    // the subtree of `root` into `out` (out[0] = its root; children are indices into `out`), one thread
    void build_subtree(const Item& root, std::vector<BinNode>& out) {
        std::vector<Item> stack;  // iterative to survive degenerate inputs
        out.emplace_back();
        stack.push_back({0, root.first, root.count});
        while (!stack.empty()) {
            Item it = stack.back();
            stack.pop_back();
            Box box;
            uint32_t mid = 0;
            const bool inner = split(it, box, mid, 1);
            out[it.node].box = box;
            out[it.node].first = it.first;
            out[it.node].count = it.count;
            if (!inner) continue;
            const int32_t l = (int32_t)out.size();
            out.emplace_back();
            const int32_t r = (int32_t)out.size();
            out.emplace_back();
            out[it.node].left = l;
            out[it.node].right = r;
            stack.push_back({l, it.first, mid - it.first});
            stack.push_back({r, mid, it.first + it.count - mid});
        }
    }

    int32_t build(uint32_t first, uint32_t count) {
        nodes.emplace_back();
        std::vector<Item> stack, tasks;
        stack.push_back({0, first, count});
        // the top of the tree, node by node with every thread on the node's triangles, down to subtrees small enough to hand out whole
        while (!stack.empty()) {
            Item it = stack.back();
            stack.pop_back();
            if (threads > 1 && it.count <= kTaskMax) {
                tasks.push_back(it);
                continue;
            }
            if (threads <= 1) {  // one thread: the whole tree is one "subtree"
                tasks.push_back(it);
                continue;
            }
            Box box;
            uint32_t mid = 0;
            const bool inner = split(it, box, mid, threads);
            nodes[it.node].box = box;
            nodes[it.node].first = it.first;
            nodes[it.node].count = it.count;
            if (!inner) continue;
            const int32_t l = (int32_t)nodes.size();
            nodes.emplace_back();
            const int32_t r = (int32_t)nodes.size();
            nodes.emplace_back();
            nodes[it.node].left = l;
            nodes[it.node].right = r;
            stack.push_back({l, it.first, mid - it.first});
            stack.push_back({r, mid, it.first + it.count - mid});
        }
        const bool timing = std::getenv("AKR_TIMING") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!timing) return;
            auto n = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[akari_hip] bvh build (%u threads): %-22s %.3f s\n", threads, what, std::chrono::duration<double>(n - t0).count());
            t0 = n;
        };
        std::vector<std::vector<BinNode>> sub(tasks.size());
        // the largest subtrees first: the tail of the parallel region is then made of small ones
        std::vector<unsigned> by_size(tasks.size());
        std::iota(by_size.begin(), by_size.end(), 0u);
        std::sort(by_size.begin(), by_size.end(), [&](unsigned x, unsigned y) { return tasks[x].count > tasks[y].count; });
        parallel_chunks((unsigned)tasks.size(), threads, [&](unsigned t) { build_subtree(tasks[by_size[t]], sub[by_size[t]]); });
        lap("subtrees");
        for (size_t t = 0; t < tasks.size(); t++) {  // splice: local index i > 0 becomes base + i - 1, local 0 is the task's node
            const int32_t base = (int32_t)nodes.size();
            auto fix = [&](int32_t c) { return c < 0 ? c : base + c - 1; };
            const std::vector<BinNode>& sn = sub[t];
            BinNode root = sn[0];
            root.left = fix(root.left);
            root.right = fix(root.right);
            nodes[tasks[t].node] = root;
            for (size_t i = 1; i < sn.size(); i++) {
                BinNode n = sn[i];
                n.left = fix(n.left);
                n.right = fix(n.right);
                nodes.push_back(n);
            }
            std::vector<BinNode>().swap(sub[t]);
        }
        lap("splice");
        return 0;
    }
};
}  // namespace

// Result of the build. nodes: `stride` words per node (20 used). order[k] = source triangle of traversal-order triangle k.
// depth = levels of the wide tree (root = 1) = the most stack entries a traversal can need (device/disect.h).
// balanced = true: median splits and widest-subtree-first collapse instead of SAH: a tree of depth ~ log8(n), the fallback for
// geometry whose SAH tree would be deeper than the traversal stack.
void build_bvh8(const std::vector<float>& tri_bounds, uint32_t n_tris, float pad, uint32_t stride, bool balanced, std::vector<uint32_t>& order_out,
                std::vector<uint32_t>& out_nodes, uint32_t& depth_out) {
    std::vector<uint32_t> order;
    Builder b(tri_bounds.data(), n_tris, order);
    b.balanced = balanced;
    const bool timing = std::getenv("AKR_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    b.build(0, n_tris);
    if (timing) std::fprintf(stderr, "[akari_hip] build_bvh8: binary SAH build %.3f s (%zu nodes)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), b.nodes.size());
    const auto& bn = b.nodes;
    struct Pending { int32_t bin; uint32_t out; uint32_t depth; };
    std::deque<Pending> queue;
    out_nodes.assign(stride, 0u);
    order_out.clear();
    order_out.reserve(n_tris);
    depth_out = 0;
    queue.push_back({0, 0, 1});
    auto fbits = [](float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; };
    // Emission order = memory order. The top of the tree breadth-first -- level by level until kBvhTopSlots node slots exist --
    // so that the nodes every ray visits first are the nodes 0 .. n of the array: the kernels keep a prefix of it in LDS
    // (device/disect.h: node tile). Below that depth-first: a node's child block is allocated when the node is emitted and each
    // subtree is laid out before its siblings' subtrees, so a ray that descends stays within a few DRAM pages / L2 lines.
    while (!queue.empty()) {
        Pending pe;
        if (out_nodes.size() / stride < kBvhTopSlots) { pe = queue.front(); queue.pop_front(); }
        else { pe = queue.back(); queue.pop_back(); }
        depth_out = std::max(depth_out, pe.depth);
        int32_t kids[8];
        int nk = 0;
        const BinNode& root = bn[pe.bin];
        if (root.left < 0) {
            kids[nk++] = pe.bin;  // the whole (sub)tree is a single leaf
        } else {
            kids[nk++] = root.left;
            kids[nk++] = root.right;
            while (nk < kWide) {  // open the inner child with the largest surface area (balanced: with the most triangles)
                int pick = -1;
                float best = -1.0f;
                for (int i = 0; i < nk; i++) {
                    if (bn[kids[i]].left < 0) continue;
                    float a = balanced ? (float)bn[kids[i]].count : bn[kids[i]].box.half_area();
                    if (a > best) { best = a; pick = i; }
                }
                if (pick < 0) break;
                int32_t k = kids[pick];
                kids[pick] = bn[k].left;
                kids[nk++] = bn[k].right;
            }
        }
        // padded child boxes and their union (= this node's frame)
        float clo[8][3], chi[8][3], origin[3], top[3];
        for (int a = 0; a < 3; a++) { origin[a] = std::numeric_limits<float>::infinity(); top[a] = -origin[a]; }
        for (int i = 0; i < nk; i++)
            for (int a = 0; a < 3; a++) {
                clo[i][a] = bn[kids[i]].box.lo[a] - pad;
                chi[i][a] = bn[kids[i]].box.hi[a] + pad;
                origin[a] = std::min(origin[a], clo[i][a]);
                top[a] = std::max(top[a], chi[i][a]);
            }
        // slot assignment: child c -> slot s maximising sum_a (s_a ? +1 : -1) * (centre_c - centre_node)_a, greedily by
        // the largest remaining gain (the pairing the paper's reference implementation uses as well)
        int slot_of[8], child_in[8];
        for (int i = 0; i < 8; i++) { slot_of[i] = -1; child_in[i] = -1; }
        {
            float cost[8][8];
            for (int c = 0; c < nk; c++)
                for (int sl = 0; sl < 8; sl++) {
                    float v = 0.0f;
                    for (int a = 0; a < 3; a++) {
                        float ext = top[a] - origin[a];
                        float rel = ext > 0.0f ? (0.5f * (clo[c][a] + chi[c][a]) - 0.5f * (origin[a] + top[a])) / ext : 0.0f;
                        v += ((sl >> a) & 1) ? rel : -rel;
                    }
                    cost[c][sl] = v;
                }
            for (int it = 0; it < nk; it++) {
                int bc = -1, bs = -1;
                float bv = -std::numeric_limits<float>::infinity();
                for (int c = 0; c < nk; c++) {
                    if (slot_of[c] >= 0) continue;
                    for (int sl = 0; sl < 8; sl++) {
                        if (child_in[sl] >= 0) continue;
                        // `!(cost <= bv)` also takes a NaN cost (boxes with non-finite corners are refused by compile_scene, but a
                        // pairing must come out of this loop whatever the numbers are: bc / bs index the stack arrays below)
                        if (bc < 0 || !(cost[c][sl] <= bv)) { bv = cost[c][sl]; bc = c; bs = sl; }
                    }
                }
                slot_of[bc] = bs;
                child_in[bs] = bc;
            }
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
        // children block: the inner child at octant position s lives at child_base + s
        int max_inner_slot = -1;
        for (int sl = 0; sl < 8; sl++)
            if (child_in[sl] >= 0 && bn[kids[child_in[sl]]].left >= 0) max_inner_slot = sl;
        uint32_t child_base = 0;
        if (max_inner_slot >= 0) {
            child_base = (uint32_t)(out_nodes.size() / stride);
            out_nodes.resize(out_nodes.size() + (size_t)stride * (size_t)(max_inner_slot + 1), 0u);
            if (out_nodes.size() / stride > (1u << 24)) throw std::runtime_error("unsupported: scene needs more than 2^24 BVH node slots");
        }
        const uint32_t tri_base = (uint32_t)order_out.size();
        // storage order = ascending octant position; entry e of the node holds the child at position pos_of[e]
        uint8_t meta[kWide], q[6][kWide];
        for (int e = 0; e < kWide; e++) {
            meta[e] = 0;
            for (int k = 0; k < 6; k++) q[k][e] = k < 3 ? 255 : 0;  // empty entry: inverted box
        }
        int e = 0;
        for (int sl = 0; sl < 8; sl++) {
            const int c = child_in[sl];
            if (c < 0) continue;
            const BinNode& cn = bn[kids[c]];
            for (int a = 0; a < 3; a++) {
                double l = std::floor(((double)clo[c][a] - (double)origin[a]) / scale[a]);
                double h = std::ceil(((double)chi[c][a] - (double)origin[a]) / scale[a]);
                l = std::max(0.0, std::min(255.0, l));
                h = std::max(0.0, std::min(255.0, h));
                // the f32 decode origin + q * scale rounds to nearest: step outwards until it is conservative
                while (l > 0.0 && (float)((double)origin[a] + l * scale[a]) > clo[c][a]) l -= 1.0;
                while (h < 255.0 && (float)((double)origin[a] + h * scale[a]) < chi[c][a]) h += 1.0;
                q[a][e] = (uint8_t)l;
                q[3 + a][e] = (uint8_t)h;
            }
            if (cn.left < 0) {  // leaf: its triangles follow the node's earlier leaves
                const uint32_t offset = (uint32_t)order_out.size() - tri_base;
                for (uint32_t t = 0; t < cn.count; t++) order_out.push_back(order[cn.first + t]);
                const uint32_t unary = cn.count >= 3 ? 7u : (cn.count == 2 ? 3u : 1u);
                meta[e] = (uint8_t)((unary << 5) | offset);
            } else {
                meta[e] = (uint8_t)(0x20u | (24u + (uint32_t)sl));
                queue.push_back({kids[c], child_base + (uint32_t)sl, pe.depth + 1});
            }
            e++;
        }
        uint32_t* n = &out_nodes[(size_t)stride * pe.out];
        n[0] = fbits(origin[0]); n[1] = fbits(origin[1]); n[2] = fbits(origin[2]);
        n[3] = ebits[0] | (ebits[1] << 8) | (ebits[2] << 16) | ((child_base & 0xffu) << 24);
        n[4] = ((child_base >> 8) & 0xffffu) | ((uint32_t)meta[4] << 16) | ((uint32_t)meta[5] << 24);
        n[5] = (uint32_t)meta[0] | ((uint32_t)meta[1] << 8) | ((uint32_t)meta[2] << 16) | ((uint32_t)meta[3] << 24);
        n[6] = tri_base;
        for (int k = 0; k < 6; k++)  // entries 0..3: one word per plane
            n[7 + k] = (uint32_t)q[k][0] | ((uint32_t)q[k][1] << 8) | ((uint32_t)q[k][2] << 16) | ((uint32_t)q[k][3] << 24);
        for (int a = 0; a < 3; a++)  // entries 4, 5: one word per axis = lo[4] lo[5] hi[4] hi[5]
            n[13 + a] = (uint32_t)q[a][4] | ((uint32_t)q[a][5] << 8) | ((uint32_t)q[3 + a][4] << 16) | ((uint32_t)q[3 + a][5] << 24);
    }
}

}  // namespace akr
