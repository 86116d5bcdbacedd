// scene_inst.cpp -- a scene kept as meshes + instances: one BLAS per mesh in object space, a TLAS over the instances' world boxes.
//
// The reference builds exactly that (crates/akari_render/src/mesh.rs:259-348: `push_mesh(mesh, transform, ..)` per instance into a
// LuisaCompute accel) and evaluates everything per hit from object-space buffers and the instance transform (mesh.rs:487-654). The
// flattening compiler (scene_build.cpp) stores 192 bytes per instance-triangle instead -- fine for the Cornell box and a 10 M-triangle
// hall, impossible for a thousand instances of a 100 k-triangle plant. Here nothing is stored per instance-triangle:
//   * the two box levels only CULL (they must be conservative, nothing more): the TLAS in world space over boxes of the exactly
//     transformed vertices, a BLAS in object space, entered with the ray taken through the instance's inverse -- its boxes padded for
//     the round-off of that;
//   * every accept / reject of a triangle and every shaded value is the flattened arithmetic, computed at the candidate from the
//     object-space triangle and the instance transform by the code the flattening compiler runs (device/dinst.h): f32-transformed
//     vertices, Woop rows in f64, the coplanar-neighbour rule, the per-triangle frame and area. Films are the oracle's bit for bit
//     (the oracle flattens).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "../device/dinst.h"
#include "host_parallel.h"
#include "scene_build.h"

namespace akr {

void build_bvh8(const std::vector<float>& tri_bounds, uint32_t n_tris, float pad, uint32_t stride, bool balanced, std::vector<uint32_t>& order,
                std::vector<uint32_t>& nodes, uint32_t& depth);

namespace {
vec3 ld3(const std::vector<float>& v, size_t i) { return mk3(v[3 * i], v[3 * i + 1], v[3 * i + 2]); }
// flattened records of a scene: 64 B traversal + 128 B shade per instance-triangle (+ 96 B with corner normals), ~0.3 nodes of 64 B
constexpr uint64_t kFlatBytesPerTriangle = 64 + 128 + 20;
constexpr uint64_t kFlatBudgetBytes = 8ull << 30;   // automatic mode: flatten while the records stay below this
constexpr uint64_t kFlatMaxTriangles = 48u << 20;   // ... and the flattened tree below its 2^24 node slots (~0.29 slots per triangle)

// largest and smallest singular value of the linear part of an instance transform (column-major 4x4): square roots of the extreme
// eigenvalues of L^T L, by the trigonometric solution of the symmetric 3x3 characteristic polynomial
void singular_range(const float* m, double& smax, double& smin) {
    double a[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            a[i][j] = 0.0;
            for (int k = 0; k < 3; k++) a[i][j] += (double)m[4 * i + k] * (double)m[4 * j + k];  // (columns i, j of L)
        }
    const double p1 = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double q = (a[0][0] + a[1][1] + a[2][2]) / 3.0;
    const double p2 = (a[0][0] - q) * (a[0][0] - q) + (a[1][1] - q) * (a[1][1] - q) + (a[2][2] - q) * (a[2][2] - q) + 2.0 * p1;
    double e0, e2;
    if (!(p2 > 0.0)) {
        e0 = e2 = q;
    } else {
        const double p = std::sqrt(p2 / 6.0);
        double b[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) b[i][j] = (a[i][j] - (i == j ? q : 0.0)) / p;
        double r = 0.5 * (b[0][0] * (b[1][1] * b[2][2] - b[1][2] * b[2][1]) - b[0][1] * (b[1][0] * b[2][2] - b[1][2] * b[2][0]) +
                          b[0][2] * (b[1][0] * b[2][1] - b[1][1] * b[2][0]));
        r = std::max(-1.0, std::min(1.0, r));
        const double phi = std::acos(r) / 3.0;
        e0 = q + 2.0 * p * std::cos(phi);                             // largest
        e2 = q + 2.0 * p * std::cos(phi + 2.0 * 3.14159265358979323846 / 3.0);  // smallest
    }
    smax = std::sqrt(std::max(e0, 0.0));
    // the smallest eigenvalue comes out of a cancellation: never report it larger than |det| / smax^2 allows, nor smaller than what is representable
    smin = std::sqrt(std::max(e2, 0.0));
    const double det = (double)m[0] * ((double)m[5] * m[10] - (double)m[9] * m[6]) - (double)m[4] * ((double)m[1] * m[10] - (double)m[9] * m[2]) +
                       (double)m[8] * ((double)m[1] * m[6] - (double)m[5] * m[2]);
    if (smax > 0.0) smin = std::min(smin > 0.0 ? smin : 1e300, std::fabs(det) / (smax * smax));  // s0 s1 s2 = |det|, s1 <= s0  =>  s2 >= |det| / s0^2
}

// ---- a per-mesh tree as build_bvh8 wrote it, decoded for the host (bvh.cpp: node layout)
struct TreeNode {
    float lo[3], hi[3];        // object-space box of everything below (from the padded triangle boxes the tree was built over)
    uint32_t inner[6];         // inner children: node indices relative to the tree
    uint32_t tri_first[6];     // leaf children: first triangle (tree order), count
    uint8_t tri_count[6];
    uint8_t n_inner = 0, n_leaf = 0;
};
void decode_tree_nodes(const std::vector<uint32_t>& words, const std::vector<uint32_t>& order, const std::vector<float>& bounds, std::vector<TreeNode>& out) {
    const size_t n_nodes = words.size() / kBvhNodeWords;
    out.assign(n_nodes, TreeNode());
    std::vector<uint8_t> reached(n_nodes, 0);
    // children first: walk from the root, remember the visiting order, fold the boxes in reverse
    std::vector<uint32_t> visit;
    visit.reserve(n_nodes);
    visit.push_back(0);
    reached[0] = 1;
    for (size_t v = 0; v < visit.size(); v++) {
        const uint32_t idx = visit[v];
        const uint32_t* n = &words[(size_t)kBvhNodeWords * idx];
        TreeNode& t = out[idx];
        const uint32_t child_base = (n[3] >> 24) | ((n[4] & 0xffffu) << 8), tri_base = n[6];
        for (int e = 0; e < 6; e++) {
            const uint32_t meta = e < 4 ? (n[5] >> (8 * e)) & 0xffu : (n[4] >> (16 + 8 * (e - 4))) & 0xffu;
            if (meta == 0) continue;
            if ((meta >> 5) == 1u && (meta & 0x1fu) >= 24u) {
                const uint32_t c = child_base + (meta & 0x1fu) - 24u;
                if (c >= n_nodes || reached[c]) throw std::runtime_error("internal: malformed per-mesh tree");
                reached[c] = 1;
                t.inner[t.n_inner++] = c;
                visit.push_back(c);
            } else {
                t.tri_first[t.n_leaf] = tri_base + (meta & 0x1fu);
                t.tri_count[t.n_leaf] = (uint8_t)((meta >> 5) == 7u ? 3 : ((meta >> 5) == 3u ? 2 : 1));
                t.n_leaf++;
            }
        }
    }
    for (size_t v = visit.size(); v-- > 0;) {
        TreeNode& t = out[visit[v]];
        for (int a = 0; a < 3; a++) { t.lo[a] = INFINITY; t.hi[a] = -INFINITY; }
        for (int l = 0; l < t.n_leaf; l++)
            for (uint32_t k = 0; k < t.tri_count[l]; k++) {
                const float* bb = &bounds[6ull * order[t.tri_first[l] + k]];
                for (int a = 0; a < 3; a++) { t.lo[a] = min_f(t.lo[a], bb[a]); t.hi[a] = max_f(t.hi[a], bb[3 + a]); }
            }
        for (int c = 0; c < t.n_inner; c++) {
            const TreeNode& ch = out[t.inner[c]];
            for (int a = 0; a < 3; a++) { t.lo[a] = min_f(t.lo[a], ch.lo[a]); t.hi[a] = max_f(t.hi[a], ch.hi[a]); }
        }
    }
}
// half the surface area of the world box of an object-space box under an instance transform (a heuristic only: which subtree to open)
float world_half_area(const InstXf& x, const float* lo, const float* hi) {
    float wl[3] = {INFINITY, INFINITY, INFINITY}, wh[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int c = 0; c < 8; c++) {
        const vec3 p = xf_point(x.c0, x.c1, x.c2, x.t, mk3((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]));
        wl[0] = min_f(wl[0], p.x); wl[1] = min_f(wl[1], p.y); wl[2] = min_f(wl[2], p.z);
        wh[0] = max_f(wh[0], p.x); wh[1] = max_f(wh[1], p.y); wh[2] = max_f(wh[2], p.z);
    }
    const float dx = wh[0] - wl[0], dy = wh[1] - wl[1], dz = wh[2] - wl[2];
    if (!(dx >= 0.0f) || !is_finite(dx) || !is_finite(dy) || !is_finite(dz)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}
}  // namespace

bool want_instancing(const FlatScene& flat) {
    const TuningOptions t = tuning();
    if (t.instancing == 0) return false;
    std::vector<uint32_t> uses(flat.meshes.size(), 0);
    uint64_t flat_tris = 0;
    bool shared = false;
    for (const HostInstance& in : flat.instances) {
        if (in.mesh >= flat.meshes.size()) return false;  // (compile_scene reports it)
        shared = shared || ++uses[in.mesh] > 1;
        flat_tris += flat.meshes[in.mesh].n_triangles();
        // a singular transform has no inverse to take the ray through: such a scene is flattened (its triangles are degenerate there)
        const float* m = in.transform;
        const double det = (double)m[0] * ((double)m[5] * m[10] - (double)m[9] * m[6]) - (double)m[4] * ((double)m[1] * m[10] - (double)m[9] * m[2]) +
                           (double)m[8] * ((double)m[1] * m[6] - (double)m[5] * m[2]);
        if (!(std::fabs(det) > 1e-30) || !std::isfinite(det)) return false;
    }
    if (!shared) return false;
    if (t.instancing == 1) return true;
    return flat_tris * kFlatBytesPerTriangle > kFlatBudgetBytes || flat_tris > kFlatMaxTriangles;
}

void compile_instanced_geometry(const FlatScene& flat, const std::vector<InstXf>& xf, const std::vector<akr_material_desc>& descs, CompiledScene& out) {
    CompiledScene::Instanced& is = out.instanced;
    is = CompiledScene::Instanced();
    is.on = true;
    const size_t n_inst = flat.instances.size(), n_src = flat.meshes.size();
    const TexScene host_tex{out.tex_nodes.data(), out.images.data(), out.texels.data(), out.mat_inputs.data(), 0, 0};
    // ---- instance world boxes from the exactly transformed vertices (what the flattened triangles' boxes would span), scene box
    std::vector<float> inst_bounds(6 * n_inst);
    std::vector<std::string> errors(n_inst);
    parallel_chunks((unsigned)n_inst, n_inst > 16 ? host_threads() : 1u, [&](unsigned i) {
        const HostMesh& g = flat.meshes[flat.instances[i].mesh];
        const InstXf& x = xf[i];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool finite = true;
        for (size_t k = 0; k < g.indices.size(); k++) {  // (every corner that some triangle uses)
            const vec3 p = xf_point(x.c0, x.c1, x.c2, x.t, ld3(g.vertices, g.indices[k]));
            finite = finite && is_finite(p.x) && is_finite(p.y) && is_finite(p.z);
            lo[0] = min_f(lo[0], p.x); lo[1] = min_f(lo[1], p.y); lo[2] = min_f(lo[2], p.z);
            hi[0] = max_f(hi[0], p.x); hi[1] = max_f(hi[1], p.y); hi[2] = max_f(hi[2], p.z);
        }
        if (!finite) errors[i] = "instance " + std::to_string(i) + ": non-finite vertex position after the instance transform";
        for (int a = 0; a < 3; a++) { inst_bounds[6 * i + a] = lo[a]; inst_bounds[6 * i + 3 + a] = hi[a]; }
    });
    for (int a = 0; a < 3; a++) { out.scene_lo[a] = INFINITY; out.scene_hi[a] = -INFINITY; }
    for (size_t i = 0; i < n_inst; i++) {
        if (!errors[i].empty()) throw std::invalid_argument(errors[i]);
        if (flat.meshes[flat.instances[i].mesh].n_triangles() == 0) continue;
        for (int a = 0; a < 3; a++) {
            out.scene_lo[a] = min_f(out.scene_lo[a], inst_bounds[6 * i + a]);
            out.scene_hi[a] = max_f(out.scene_hi[a], inst_bounds[6 * i + 3 + a]);
        }
    }
    const float pad_scale = 0.01f * (float)tuning().pad_percent;  // (test hook: 100)
    const float pad_world = pad_scale * bvh_box_padding(out.scene_lo, out.scene_hi, flat.camera.c2w);  // the flattened tree's padding
    float scene_mag;  // 2-norm of the largest coordinates a ray origin or a hit point can have
    {
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) { lo[a] = min_f(out.scene_lo[a], flat.camera.c2w[12 + a]); hi[a] = max_f(out.scene_hi[a], flat.camera.c2w[12 + a]); }
        scene_mag = box_magnitude(lo, hi);
    }
    // ---- per instance: the inverse transform (double -> f32; used for culling only) and its norm
    std::vector<double> inv(12 * n_inst);
    std::vector<float> inv_norm(n_inst, 0.0f);
    std::vector<float> inst_cond(n_inst, 1.0f);         // condition number of the instance's linear part (2-norm)
    std::vector<float> inst_back_reach(n_inst, 0.0f);   // |M^-1| x (|M| x the mesh's own coordinates + |translation|)
    std::vector<float> inst_back_mag(n_inst, 0.0f);     // |M^-1|_2 x (2-norm of the largest world coordinates of the instance)
    std::vector<float> src_obj_reach(n_src, 0.0f);      // sum over the axes of the largest |object-space coordinate| of a mesh
    {
        std::vector<uint8_t> used_src(n_src, 0);
        for (const HostInstance& in : flat.instances) used_src[in.mesh] = 1;
        for (size_t m = 0; m < n_src; m++) {
            if (!used_src[m]) continue;
            const HostMesh& g = flat.meshes[m];
            float mx[3] = {0.0f, 0.0f, 0.0f};
            for (size_t k = 0; k < g.indices.size(); k++)
                for (int a = 0; a < 3; a++) mx[a] = max_f(mx[a], abs_f(g.vertices[3ull * g.indices[k] + a]));
            src_obj_reach[m] = (mx[0] + mx[1]) + mx[2];
        }
    }
    for (size_t i = 0; i < n_inst; i++) {
        const float* m = flat.instances[i].transform;
        const double a[3][3] = {{m[0], m[4], m[8]}, {m[1], m[5], m[9]}, {m[2], m[6], m[10]}};  // row-major 3x3 of the column-major 4x4
        const double t[3] = {m[12], m[13], m[14]};
        const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                           a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
        double r[3][3];
        r[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / det; r[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det; r[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
        r[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / det; r[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det; r[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
        r[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / det; r[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det; r[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
        float norm = 0.0f;
        for (int row = 0; row < 3; row++) {
            double* o = &inv[12 * i + 4 * row];
            o[0] = r[row][0]; o[1] = r[row][1]; o[2] = r[row][2];
            o[3] = -(r[row][0] * t[0] + r[row][1] * t[1] + r[row][2] * t[2]);
            norm = max_f(norm, (float)(std::fabs(o[0]) + std::fabs(o[1]) + std::fabs(o[2])));
        }
        {   // 2-norms: |M^-1|_2 = 1 / (smallest singular value); the padding formulas below take whichever norm is larger
            double smax, smin;
            singular_range(m, smax, smin);
            if (smin > 0.0) {
                norm = max_f(norm, (float)std::min(1.0001 / smin, 3e38));
                inst_cond[i] = (float)std::min(1.0001 * smax / smin, 3e38);
            }
        }
        inv_norm[i] = norm;
        // how large the numbers are that cancel when a ray goes through M^-1 and a vertex through M: a mesh modelled far from its own
        // origin and moved back by the instance's translation has world coordinates ~ 1 and both of these ~ 1e4
        float fwd = 0.0f;  // largest absolute row sum of the linear part
        for (int row = 0; row < 3; row++) fwd = max_f(fwd, (float)(std::fabs(a[row][0]) + std::fabs(a[row][1]) + std::fabs(a[row][2])));
        const float tl = (float)(std::fabs(t[0]) + std::fabs(t[1]) + std::fabs(t[2]));
        inst_back_reach[i] = norm * (fwd * src_obj_reach[flat.instances[i].mesh] + tl);
        inst_back_mag[i] = norm * box_magnitude(&inst_bounds[6 * i], &inst_bounds[6 * i + 3]);
    }
    // ---- padding classes: a mesh's tree is padded for the WORST of its instances -- |M^-1| scales every term -- so one tiny
    // (or far away) copy among a thousand ordinary ones used to inflate the boxes of all of them until the tree stopped culling: correct, and
    // a performance cliff nothing reported. The instances of a mesh are therefore sorted into up to four classes by what their own
    // transform asks for (within 8 x, 64 x, 512 x the mesh's median, beyond), and each class that occurs gets a tree -- and a copy of the
    // mesh's triangle records in that tree's order -- of its own: a "virtual mesh". Ordinary scenes have one class per mesh and compile to
    // the bytes they always did.
    std::vector<uint32_t> vmesh_of(n_inst, 0), vsrc;  // instance -> virtual mesh; virtual mesh -> the mesh whose geometry it is
    {
        std::vector<std::vector<uint32_t>> by_mesh(n_src);
        for (size_t i = 0; i < n_inst; i++) by_mesh[flat.instances[i].mesh].push_back((uint32_t)i);
        auto ask = [&](uint32_t i) { return inv_norm[i] * scene_mag + inst_back_reach[i] + inst_back_mag[i]; };
        for (size_t m = 0; m < n_src; m++) {
            if (by_mesh[m].empty()) continue;
            std::vector<float> asks;
            for (uint32_t i : by_mesh[m]) asks.push_back(ask(i));
            std::nth_element(asks.begin(), asks.begin() + (asks.size() - 1) / 2, asks.end());
            const float med = asks[(asks.size() - 1) / 2];  // (the lower median: of two copies the ordinary one sets the scale)
            int vid[4] = {-1, -1, -1, -1};
            for (int c = 0; c < 4; c++)
                for (uint32_t i : by_mesh[m]) {
                    const float q = ask(i);
                    const int cls = !(q > 8.0f * med) ? 0 : (!(q > 64.0f * med) ? 1 : (!(q > 512.0f * med) ? 2 : 3));
                    if (cls != c) continue;
                    if (vid[c] < 0) { vid[c] = (int)vsrc.size(); vsrc.push_back((uint32_t)m); }
                    vmesh_of[i] = (uint32_t)vid[c];
                }
        }
    }
    const size_t n_mesh = vsrc.size();
    is.n_padding_classes = (uint32_t)n_mesh;
    std::vector<uint8_t> used(n_mesh, 1);
    std::vector<uint32_t> mesh_base(n_mesh, 0);
    uint32_t n_mesh_tris = 0;
    bool any_normals = false;
    for (size_t m = 0; m < n_mesh; m++) {
        mesh_base[m] = n_mesh_tris;
        const uint64_t total = (uint64_t)n_mesh_tris + flat.meshes[vsrc[m]].n_triangles();
        if (total > 0xffffffffull) throw std::runtime_error("unsupported: more than 2^32 mesh triangles");
        if (flat.meshes[vsrc[m]].n_triangles() > kMeshPrimMask) throw std::runtime_error("unsupported: a mesh of more than 2^31 triangles");
        n_mesh_tris = (uint32_t)total;
        if (!flat.meshes[vsrc[m]].normals.empty() || !flat.meshes[vsrc[m]].tangents.empty()) any_normals = true;
    }
    is.n_mesh_tris = n_mesh_tris;
    std::vector<float> mesh_inv_norm(n_mesh, 0.0f), mesh_back_reach(n_mesh, 0.0f), mesh_back_mag(n_mesh, 0.0f), mesh_obj_reach(n_mesh, 0.0f);  // maxima over a virtual mesh's instances
    for (size_t m = 0; m < n_mesh; m++) mesh_obj_reach[m] = src_obj_reach[vsrc[m]];
    for (size_t i = 0; i < n_inst; i++) {
        const uint32_t mesh = vmesh_of[i];
        mesh_inv_norm[mesh] = max_f(mesh_inv_norm[mesh], inv_norm[i]);
        mesh_back_reach[mesh] = max_f(mesh_back_reach[mesh], inst_back_reach[i]);
        mesh_back_mag[mesh] = max_f(mesh_back_mag[mesh], inst_back_mag[i]);
    }
    // ---- per mesh: BLAS over object-space boxes, triangles in BLAS order, lookup by prim
    is.mesh_tris.assign(16ull * n_mesh_tris, 0.0f);
    is.mesh_pos.assign(n_mesh_tris, 0u);
    is.mesh_meta.assign(n_mesh_tris, 0u);
    if (any_normals) is.mesh_normals.assign(24ull * n_mesh_tris, 0.0f);
    std::vector<uint32_t> blas_node_off(n_mesh, 0);
    std::vector<std::vector<uint32_t>> blas_nodes(n_mesh);
    std::vector<uint32_t> blas_depth(n_mesh, 0);
    std::vector<float> mesh_size(n_mesh, 0.0f);
    std::vector<float> mesh_k2max(n_mesh, 0.0f);  // worst conditioning (isotropic: 2 |e1||e2| / |n|) of a mesh's triangles, scene_build.h
    const TuningOptions tune = tuning();
    const bool rebraid = tune.rebraid > 1;
    std::vector<std::vector<uint32_t>> mesh_order(n_mesh);   // (re-braiding only) tree order -> prim, and the decoded tree
    std::vector<std::vector<TreeNode>> mesh_tree(n_mesh);
    for (size_t m = 0; m < n_mesh; m++) {
        if (!used[m]) continue;
        const HostMesh& g = flat.meshes[vsrc[m]];
        const uint32_t nt = g.n_triangles();
        if (nt == 0) continue;
        std::vector<float> bounds(6ull * nt);
        float olo[3] = {INFINITY, INFINITY, INFINITY}, ohi[3] = {-INFINITY, -INFINITY, -INFINITY};
        double max_n2 = 0.0;
        for (uint32_t prim = 0; prim < nt; prim++) {
            const vec3 v0 = ld3(g.vertices, g.indices[3 * prim]), v1 = ld3(g.vertices, g.indices[3 * prim + 1]), v2 = ld3(g.vertices, g.indices[3 * prim + 2]);
            {
                const double ax = (double)v1.x - v0.x, ay = (double)v1.y - v0.y, az = (double)v1.z - v0.z, bx = (double)v2.x - v0.x, by = (double)v2.y - v0.y, bz = (double)v2.z - v0.z;
                const double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
                max_n2 = std::max(max_n2, nx * nx + ny * ny + nz * nz);
            }
            float* bb = &bounds[6ull * prim];
            bb[0] = min_f(min_f(v0.x, v1.x), v2.x); bb[1] = min_f(min_f(v0.y, v1.y), v2.y); bb[2] = min_f(min_f(v0.z, v1.z), v2.z);
            bb[3] = max_f(max_f(v0.x, v1.x), v2.x); bb[4] = max_f(max_f(v0.y, v1.y), v2.y); bb[5] = max_f(max_f(v0.z, v1.z), v2.z);
            for (int a = 0; a < 3; a++) { olo[a] = min_f(olo[a], bb[a]); ohi[a] = max_f(ohi[a], bb[3 + a]); }
            // A needle's ill-conditioned inside test (scene_build.h tri_conditioning). The world-space triangle the test sees is displaced
            // by du e1_w + dv e2_w with |du| <= eps x reach x |r0_w|, |r0_w| <= |M^-1|_2 |r0|; taken back through M^-1 that is
            // du e1 + dv e2 of the OBJECT-space edges: per axis eps x reach x |M^-1| x k[a], whatever the instance does to the shape.
            float k[3];
            {
                const double Ad[3] = {v0.x, v0.y, v0.z}, Bd[3] = {v1.x, v1.y, v1.z}, Cd[3] = {v2.x, v2.y, v2.z};
                tri_conditioning(Ad, Bd, Cd, k);
            }
            mesh_k2max[m] = max_f(mesh_k2max[m], (k[0] + k[1]) + k[2]);  // (>= the 2-norm of the per-axis values)
            for (int a = 0; a < 3; a++) {
                // worst instance of the mesh: |M^-1| x how far from the origin its copy of the triangle can be
                const float extra = pad_scale * kTriCondEps * mesh_back_mag[m] * k[a];  // (in full: the flat padding below holds no allowance for it)
                bb[a] -= extra; bb[3 + a] += extra;
            }
        }
        mesh_size[m] = (float)(std::sqrt(std::sqrt(max_n2)) * 1.0001);  // sqrt(|n|) of the mesh's largest triangle
        // Flat padding of the object-space boxes (round 6: derived term by term instead of stacked 4e-6's -- the forest's per-mesh trees carried
        // a padding of a quarter of their triangles' size). u = 2^-24; |R| = the instance's inverse (the larger of its infinity- and 2-norm),
        // Ms = magnitude of the scene's coordinates (2-norm of the largest per axis, camera included), back = |R| (|M| V + |t|), V = the mesh's own
        // coordinates. A pair the world-space test accepts has its hit point p on the world ray, within 16 u Ms of the triangle's plane-and-edges
        // (plane solve: 4 roundings on sums of size 2 |r2| max(|o|, |A|), the rows' own rounding, the division; p = fma(t, d, o): 1 more) plus
        // the needle term that is added per triangle above. Into object space: x |R|. The object-space ray: origin off by 4 u (|R| Ms + |c|), |c| <=
        // back (rounded rows, three products and three sums), direction by 4 u |R||d|, i.e. 8 u |R| Ms at the far end; v_rcp_f32's ulp scales
        // every slab distance: 2 u |R| Ms. The vertices themselves: fl(M v + t) is off by 3 u (|M| V + |t|), back through R: 3 u back. The slab
        // test's own fma roundings: 6 u (V + |R| Ms + |c|). Sum: u (36 |R| Ms + 13 back + 6 V), doubled.
        const float kU = 5.9604645e-8f;
        float pad_obj = 2.0f * kU * (36.0f * mesh_inv_norm[m] * scene_mag + 13.0f * mesh_back_reach[m] + 6.0f * mesh_obj_reach[m]);
        pad_obj *= pad_scale;
        std::vector<uint32_t> order;
        build_bvh8(bounds, nt, pad_obj, kBvhNodeWords, tune.bvh_balanced != 0, order, blas_nodes[m], blas_depth[m]);
        if (blas_depth[m] > kBvhStackDepth) build_bvh8(bounds, nt, pad_obj, kBvhNodeWords, true, order, blas_nodes[m], blas_depth[m]);
        is.blas_depth = std::max(is.blas_depth, blas_depth[m]);
        if (rebraid) {
            decode_tree_nodes(blas_nodes[m], order, bounds, mesh_tree[m]);
            mesh_order[m] = order;
        }
        const uint32_t base = mesh_base[m];
        for (uint32_t k = 0; k < nt; k++) {
            const uint32_t prim = order[k];
            is.mesh_pos[base + prim] = k;
            const vec3 v0 = ld3(g.vertices, g.indices[3 * prim]), v1 = ld3(g.vertices, g.indices[3 * prim + 1]), v2 = ld3(g.vertices, g.indices[3 * prim + 2]);
            vec2 uv0, uv1, uv2;
            tri_default_uvs(uv0, uv1, uv2);
            if (!g.uvs.empty()) {
                uv0 = mk2(g.uvs[6 * prim + 0], g.uvs[6 * prim + 1]);
                uv1 = mk2(g.uvs[6 * prim + 2], g.uvs[6 * prim + 3]);
                uv2 = mk2(g.uvs[6 * prim + 4], g.uvs[6 * prim + 5]);
            }
            float* r = &is.mesh_tris[16ull * (base + k)];
            r[0] = v0.x; r[1] = v0.y; r[2] = v0.z; r[3] = uv0.x;
            r[4] = v1.x; r[5] = v1.y; r[6] = v1.z; r[7] = uv0.y;
            r[8] = v2.x; r[9] = v2.y; r[10] = v2.z; r[11] = uv1.x;
            r[12] = uv1.y; r[13] = uv2.x; r[14] = uv2.y; r[15] = u2f(prim);
        }
        for (uint32_t prim = 0; prim < nt; prim++) {
            uint32_t tri_flags = 0;
            bool tangents_ok = false;
            if (!g.tangents.empty()) {  // mesh.rs:557-571: per-corner tangents are used only if all nine are finite
                tangents_ok = true;
                for (int k = 0; k < 9; k++) tangents_ok = tangents_ok && is_finite(g.tangents[9 * prim + k]);
            }
            if (!g.normals.empty()) tri_flags |= TRI_HAS_NORMALS;
            if (tangents_ok) tri_flags |= TRI_HAS_TANGENTS;
            const uint32_t slot = (g.slots.size() > 1) ? g.slots[prim] : 0;
            if (slot >= (1u << 30)) throw std::invalid_argument("material slot out of range for instance");
            is.mesh_meta[base + prim] = slot | (tri_flags << 30);
            if (any_normals) {
                float* nr = &is.mesh_normals[24ull * (base + prim)];
                const vec3 v0 = ld3(g.vertices, g.indices[3 * prim]), v1 = ld3(g.vertices, g.indices[3 * prim + 1]), v2 = ld3(g.vertices, g.indices[3 * prim + 2]);
                const vec3 ngc = cross(v1 - v0, v2 - v0);
                const vec3 ng_local = div_s(ngc, length(ngc));  // = TriWorld.ng_local
                for (int k = 0; k < 3; k++) {
                    vec3 nk = g.normals.empty() ? ng_local : mk3(g.normals[9 * prim + 3 * k], g.normals[9 * prim + 3 * k + 1], g.normals[9 * prim + 3 * k + 2]);
                    nr[4 * k] = nk.x; nr[4 * k + 1] = nk.y; nr[4 * k + 2] = nk.z;
                    if (tangents_ok) {
                        nr[12 + 4 * k] = g.tangents[9 * prim + 3 * k]; nr[12 + 4 * k + 1] = g.tangents[9 * prim + 3 * k + 1];
                        nr[12 + 4 * k + 2] = g.tangents[9 * prim + 3 * k + 2];
                    }
                }
            }
        }
    }
    // ---- what the TLAS is built over: (instance, entry node of its mesh's tree) pairs. One pair per instance -- the root -- unless
    // option rebraid > 1: then the pairs with the largest world boxes are OPENED (replaced by the children of their node) until there are
    // rebraid x as many as instances (after Benthin, Woop, Wald, Afra, "Improved Two-Level BVHs using Partial Re-Braiding", HPG 2017:
    // instances whose boxes overlap -- trees of a forest -- make every ray that crosses the overlap descend all of them; the top levels
    // of their trees, taken into the world-space tree, separate what the instance boxes cannot). A node is opened only if all its
    // children are nodes (triangles are reached through some node of the mesh's tree: a leaf record names a node to start at).
    struct Prim { uint32_t inst, node; };
    std::vector<Prim> prims;
    for (size_t i = 0; i < n_inst; i++)
        if (flat.meshes[flat.instances[i].mesh].n_triangles() != 0) prims.push_back({(uint32_t)i, 0u});
    if (prims.empty()) throw std::invalid_argument("instanced scene without triangles");
    if (rebraid) {
        struct Cand { float sa; uint32_t inst, node; };
        auto lower = [](const Cand& a, const Cand& b) {  // (a strict order: the choice does not depend on how the heap breaks ties)
            if (a.sa != b.sa) return a.sa < b.sa;
            if (a.inst != b.inst) return a.inst > b.inst;
            return a.node > b.node;
        };
        std::vector<Cand> heap;
        for (const Prim& pr : prims) {
            const TreeNode& t = mesh_tree[vmesh_of[pr.inst]][0];
            heap.push_back({world_half_area(xf[pr.inst], t.lo, t.hi), pr.inst, 0u});
        }
        std::make_heap(heap.begin(), heap.end(), lower);
        const uint64_t budget = std::min<uint64_t>((uint64_t)prims.size() * (uint64_t)tune.rebraid, 1ull << 22);
        uint64_t count = prims.size();
        prims.clear();
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), lower);
            const Cand c = heap.back();
            heap.pop_back();
            const std::vector<TreeNode>& tree = mesh_tree[vmesh_of[c.inst]];
            const TreeNode& t = tree[c.node];
            if (t.n_leaf == 0 && t.n_inner > 0 && count + t.n_inner - 1u <= budget) {
                count += t.n_inner - 1u;
                for (int k = 0; k < t.n_inner; k++) {
                    const TreeNode& ch = tree[t.inner[k]];
                    heap.push_back({world_half_area(xf[c.inst], ch.lo, ch.hi), c.inst, t.inner[k]});
                    std::push_heap(heap.begin(), heap.end(), lower);
                }
            } else {
                prims.push_back({c.inst, c.node});
            }
        }
        std::sort(prims.begin(), prims.end(), [](const Prim& a, const Prim& b) { return a.inst != b.inst ? a.inst < b.inst : a.node < b.node; });
    }
    // ---- TLAS over the world boxes of the pairs: boxes of the exactly transformed vertices of the triangles below the entry node
    std::vector<uint32_t> tlas_order;
    {
        std::vector<float> tb(6 * prims.size());
        std::vector<uint32_t> first_of(n_inst + 1, 0);  // prims of instance i: [first_of[i], first_of[i + 1])
        for (const Prim& pr : prims) first_of[pr.inst + 1]++;
        for (size_t i = 0; i < n_inst; i++) first_of[i + 1] += first_of[i];
        parallel_chunks((unsigned)n_inst, n_inst > 16 ? host_threads() : 1u, [&](unsigned i) {
            const uint32_t p0 = first_of[i], p1 = first_of[i + 1];
            if (p0 == p1) return;
            const uint32_t mesh = vmesh_of[i];
            if (p1 - p0 == 1 && prims[p0].node == 0) {
                for (int a = 0; a < 6; a++) tb[6ull * p0 + a] = inst_bounds[6 * i + a];
            } else {
                const HostMesh& g = flat.meshes[vsrc[mesh]];
                const InstXf& x = xf[i];
                const std::vector<TreeNode>& tree = mesh_tree[mesh];
                const std::vector<uint32_t>& ord = mesh_order[mesh];
                std::vector<uint32_t> stack;
                for (uint32_t p = p0; p < p1; p++) {
                    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                    stack.assign(1, prims[p].node);
                    while (!stack.empty()) {
                        const TreeNode& t = tree[stack.back()];
                        stack.pop_back();
                        for (int c = 0; c < t.n_inner; c++) stack.push_back(t.inner[c]);
                        for (int l = 0; l < t.n_leaf; l++)
                            for (uint32_t k = 0; k < t.tri_count[l]; k++) {
                                const uint32_t prim = ord[t.tri_first[l] + k];
                                for (int c = 0; c < 3; c++) {
                                    const vec3 q = xf_point(x.c0, x.c1, x.c2, x.t, ld3(g.vertices, g.indices[3 * prim + c]));
                                    lo[0] = min_f(lo[0], q.x); lo[1] = min_f(lo[1], q.y); lo[2] = min_f(lo[2], q.z);
                                    hi[0] = max_f(hi[0], q.x); hi[1] = max_f(hi[1], q.y); hi[2] = max_f(hi[2], q.z);
                                }
                            }
                    }
                    for (int a = 0; a < 3; a++) { tb[6ull * p + a] = lo[a]; tb[6ull * p + 3 + a] = hi[a]; }
                }
            }
            // the instance's needles in world space: conditioning at most cond(M) x the mesh's worst (the instance's own value for each of its pairs)
            const float extra = pad_scale * tri_cond_extra(inst_cond[i] * mesh_k2max[mesh], box_magnitude(&inst_bounds[6 * i], &inst_bounds[6 * i + 3]), pad_world / pad_scale);
            for (uint32_t p = p0; p < p1; p++)
                for (int a = 0; a < 3; a++) { tb[6ull * p + a] -= extra; tb[6ull * p + 3 + a] += extra; }
        });
        build_bvh8(tb, (uint32_t)prims.size(), pad_world, kBvhNodeWords, tune.bvh_balanced != 0, tlas_order, is.nodes, is.tlas_depth);
        if (is.tlas_depth > kBvhStackDepth) build_bvh8(tb, (uint32_t)prims.size(), pad_world, kBvhNodeWords, true, tlas_order, is.nodes, is.tlas_depth);
        is.tlas_nodes = (uint32_t)(is.nodes.size() / kBvhNodeWords);
    }
    for (size_t m = 0; m < n_mesh; m++) {
        if (blas_nodes[m].empty()) continue;
        blas_node_off[m] = (uint32_t)(is.nodes.size() / kBvhNodeWords);
        is.nodes.insert(is.nodes.end(), blas_nodes[m].begin(), blas_nodes[m].end());
        std::vector<uint32_t>().swap(blas_nodes[m]);
    }
    // one stack entry per level of either tree + the three words that remember where the TLAS traversal stood (disect.h)
    out.bvh_depth = is.tlas_depth + is.blas_depth + 3;
    if (out.bvh_depth > 40)
        throw std::runtime_error("unsupported: two-level BVH depth " + std::to_string(out.bvh_depth) + " exceeds the traversal stack (40 levels)");
    // ---- TLAS leaf records, instance material lists
    is.tlas_leaves.assign(16ull * prims.size(), 0.0f);
    std::vector<uint32_t> mat_base(n_inst, 0);
    for (size_t i = 0; i < n_inst; i++) {
        mat_base[i] = (uint32_t)is.inst_mats.size();
        for (uint32_t mi : flat.instances[i].materials) is.inst_mats.push_back(mi);
    }
    for (size_t k = 0; k < prims.size(); k++) {
        const uint32_t i = prims[tlas_order[k]].inst;
        const HostInstance& in = flat.instances[i];
        float* r = &is.tlas_leaves[16ull * k];
        for (int row = 0; row < 3; row++)
            for (int c = 0; c < 4; c++) r[4 * row + c] = (float)inv[12 * i + 4 * row + c];
        r[12] = u2f(blas_node_off[vmesh_of[i]]);
        r[13] = u2f(mesh_base[vmesh_of[i]]);
        r[14] = u2f(i);
        r[15] = u2f(prims[tlas_order[k]].node);  // where in the mesh's tree this record starts (0 = the root)
    }
    // ---- lights (load.rs:345-444), per emissive instance only
    is.inst_light.assign(n_inst, 0xffffffffu);
    std::vector<float> light_weights;
    for (size_t i = 0; i < n_inst; i++) {
        const HostInstance& in = flat.instances[i];
        if (!instance_may_emit(out, descs, in)) continue;
        const HostMesh& g = flat.meshes[in.mesh];
        const uint32_t count = g.n_triangles();
        std::vector<float> powers(count, 0.0f);
        std::atomic<bool> bad_slot{false};
        parallel_chunks(std::max(1u, count >> 14), count > (1u << 15) ? host_threads() : 1u, [&](unsigned c) {
            const unsigned nc = std::max(1u, count >> 14);
            const uint32_t lo = (uint32_t)((uint64_t)count * c / nc), hi = (uint32_t)((uint64_t)count * (c + 1) / nc);
            for (uint32_t prim = lo; prim < hi; prim++) {
                const uint32_t slot = (g.slots.size() > 1) ? g.slots[prim] : 0;
                if (slot >= in.materials.size()) { bad_slot = true; return; }
                const vec3 v0 = ld3(g.vertices, g.indices[3 * prim]), v1 = ld3(g.vertices, g.indices[3 * prim + 1]), v2 = ld3(g.vertices, g.indices[3 * prim + 2]);
                vec2 uv0, uv1, uv2;
                tri_default_uvs(uv0, uv1, uv2);
                if (!g.uvs.empty()) {
                    uv0 = mk2(g.uvs[6 * prim + 0], g.uvs[6 * prim + 1]);
                    uv1 = mk2(g.uvs[6 * prim + 2], g.uvs[6 * prim + 3]);
                    uv2 = mk2(g.uvs[6 * prim + 4], g.uvs[6 * prim + 5]);
                }
                const TriWorld tw = tri_world(xf[i], v0, v1, v2, uv0, uv1, uv2);
                powers[prim] = triangle_emission_power(out, host_tex, in.materials[slot], prim, uv0, uv1, uv2, tw.area);
            }
        });
        if (bad_slot) throw std::invalid_argument("material slot out of range for instance");
        float total = 0.0f;
        for (float pw : powers) total += pw;
        if (total > 1e-4f) {
            const uint32_t light_id = (uint32_t)out.light_inst.size();
            out.light_inst.push_back((uint32_t)i);
            out.light_power.push_back(total);
            light_weights.push_back(total);
            std::vector<AliasEntry> ent;
            std::vector<float> pdf;
            build_alias_table(powers, ent, pdf);
            out.light_tri_offset.push_back((uint32_t)out.area_entries.size());
            out.light_n_tris.push_back(count);
            out.area_entries.insert(out.area_entries.end(), ent.begin(), ent.end());
            out.area_pdf.insert(out.area_pdf.end(), pdf.begin(), pdf.end());
            is.inst_light[i] = light_id;
        }
    }
    out.n_lights = (uint32_t)out.light_inst.size();
    if (out.n_lights > 0) build_alias_table(light_weights, out.light_entries, out.light_pdf);
    // every instance's material slots must be in range even when it does not emit (the flattening compiler checks per triangle)
    for (size_t i = 0; i < n_inst; i++) {
        const HostMesh& g = flat.meshes[flat.instances[i].mesh];
        uint32_t max_slot = 0;
        if (g.slots.size() > 1)
            for (uint32_t sl : g.slots) max_slot = std::max(max_slot, sl);
        if (g.n_triangles() && max_slot >= flat.instances[i].materials.size()) throw std::invalid_argument("material slot out of range for instance");
    }
    // the instance records' spare words: where the device finds an instance's mesh, materials, light and global ids (dinst.h)
    for (size_t i = 0; i < n_inst; i++) {
        float* r = &out.inst[32 * i];
        r[7] = u2f(mesh_base[vmesh_of[i]]);
        r[11] = u2f(mat_base[i]);
        r[15] = u2f(is.inst_light[i]);
        r[23] = u2f(out.inst_tri_offset[i]);
        r[27] = u2f((uint32_t)flat.instances[i].materials.size());
        // sqrt(|n|) of a world-space triangle of this instance is at most |M|_F x the mesh's (cof(M) has singular values <= |M|_F^2):
        // what tri_may_hit (dinst.h) allows for a plane row shared with the even neighbour
        const float* tm = flat.instances[i].transform;
        double f2 = 0.0;
        for (int c = 0; c < 3; c++)
            for (int rr = 0; rr < 3; rr++) f2 += (double)tm[4 * c + rr] * tm[4 * c + rr];
        r[28] = (float)(std::sqrt(f2) * 1.0001) * mesh_size[vmesh_of[i]];
    }
}

}  // namespace akr
