// dscene.h -- the scene as it lives in HBM, and the hit -> SurfaceInteraction reconstruction.
//
// Layout (all arrays 16-byte aligned, read with 16-byte vector loads):
//   woop      exhaustive path: 3 x float4 per triangle (the ray-triangle test's 48 B record), global order;
//             BVH path: 4 x float4 per triangle in traversal order = the 48 B record | global triangle id | 12 B unused
//             (one 64-byte fetch per test; global id = inst_tri_offset[inst] + prim)
//   bvh_nodes 6-wide compressed nodes of 64 B (host/bvh.cpp)
//   shade     8 x float4 per triangle, indexed by global id    : everything surface_interaction needs (128 B)
//   inst      8 x float4 per instance                          : object->world matrix and its cofactors (128 B)
//   materials DMaterial[ ]                                     : folded shader graphs (256 B)
//   lights    alias tables (8 B entries + 4 B pdfs)            : LightAggregate + per-light triangle samplers
// The reference gathers the same information through five dependent bindless-buffer reads per hit
// (crates/akari_render/src/mesh.rs:499-653); here the per-triangle part is folded on the host.
#pragma once
#include "dtex.h"

namespace akr {

constexpr uint32_t kInvalid = 0xffffffffu;

// BVH record geometry shared by the host builder (host/bvh.cpp, scene_build.cpp) and the traversal (disect.h)
constexpr uint32_t kBvhNodeWords = 16;                  // one node = 64 bytes = one sector, like a triangle record
constexpr uint32_t kBvhTriWords = 16;                   // BVH path: 12 words Woop record + global id + 3 unused
// Traversal stack entries per lane. A pending group of sibling nodes is ONE entry and a traversal holds at most one group
// per tree level, so a tree of depth <= kBvhStackDepth can never overflow; scene_build.cpp rejects deeper trees.
// kBvhStackDepth is the deepest tree a scene may have; a launch sizes its stacks from the tree the scene actually got
// (DScene.bvh_stack_depth: 10 levels for the 10 M-triangle hall -- 10 KB of LDS per workgroup instead of 24).
constexpr uint32_t kBvhStackDepth = 24;

// shade record rows (float4 each):
//  0: v0.xyz | uv0.x     1: v1.xyz | uv0.y     2: v2.xyz | uv1.x     (object-space vertices)
//  3: ng.xyz | uv1.y     4: frame.t.xyz | uv2.x   5: frame.s.xyz | uv2.y   (world-space, flat-shaded frame)
//  6: prim_area | material (u32) | instance (u32) | light id (i32, -1 = none)
//  7: tt.xyz (world dpdu, for meshes with shading normals) | flags
enum : uint32_t { SHADE_ROWS = 8, INST_ROWS = 8 };
enum : uint32_t { TRI_HAS_NORMALS = 1u, TRI_HAS_TANGENTS = 2u };

struct LightRec {  // one light = one emissive instance: where its triangles sit in the area table and in the global triangle order
    uint32_t tri_offset, n_tris, first_gid, inst;
};

// A scene kept as meshes + instances (host/scene_inst.cpp; two-level traversal: disect.h trav_step_inst). When `on`, the per
// instance-triangle arrays of DScene (woop, shade, normals, tri_gid) are null and bvh_nodes holds the TLAS followed by the BLASes.
// the last word of a mesh_tris record: the triangle's index in its mesh, and (set by k_inst_share_bits once the scene is on the device) whether
// SOME instance of the mesh gives it its even neighbour's plane row -- share_bits then says which
constexpr uint32_t kMeshPrimMask = 0x7fffffffu, kMeshTriShares = 0x80000000u;
struct DInst {
    const uint4* __restrict__ tlas_leaves;     // 64 B per top-level leaf entry (one or more per instance, TLAS order): world->object rows | BLAS node offset, mesh triangle base, instance, entry node
    const float4* __restrict__ mesh_tris;      // 64 B per mesh triangle in BLAS order: v0 | uv0.x, v1 | uv0.y, v2 | uv1.x, uv1.y uv2.x uv2.y | prim (| kMeshTriShares, set on the device)
    const uint32_t* __restrict__ mesh_pos;     // mesh order -> position in mesh_tris (relative to the mesh's base)
    const uint32_t* __restrict__ mesh_meta;    // mesh order: material slot | TRI_HAS_* << 30
    const float4* __restrict__ mesh_normals;   // 6 x float4 per mesh triangle, mesh order, or nullptr
    const uint32_t* __restrict__ inst_mats;    // the instances' material lists
    const uint32_t* __restrict__ share_bits;   // one bit per instance-triangle (global id): it takes its even neighbour's plane row (dinst.h share_plane_row; k_inst_share_bits)
    uint32_t on, n_instances;
};

struct DScene {
    const float4* __restrict__ woop;        // 3 float4 per triangle (exhaustive path) or 4 (BVH path: + global id)
    const uint32_t* __restrict__ tri_gid;   // traversal order -> global id (host-side tests; the BVH path reads the id from the record)
    const float4* __restrict__ shade;
    const float4* __restrict__ normals;     // 6 x float4 per global triangle (per-corner normals, tangents) or nullptr
    const float4* __restrict__ inst;
    const DMaterial* __restrict__ materials;
    const float* __restrict__ ggx_table;
    const AliasEntry* __restrict__ light_entries;
    const float* __restrict__ light_pdf;
    const uint32_t* __restrict__ light_inst;        // light id -> instance
    const uint32_t* __restrict__ light_tri_offset;  // light id -> first entry in area_entries / area_pdf
    const uint32_t* __restrict__ light_n_tris;
    const AliasEntry* __restrict__ area_entries;
    const float* __restrict__ area_pdf;
    const uint32_t* __restrict__ inst_tri_offset;   // instance -> first global triangle id
    const AliasPacked* __restrict__ light_alias;    // light_entries + light_pdf, packed
    const AliasPacked* __restrict__ area_alias;     // area_entries + area_pdf, packed
    const LightRec* __restrict__ lights;            // light_tri_offset + light_n_tris + light_inst (+ inst_tri_offset), packed
    const uint4* __restrict__ bvh_nodes;            // nullptr on the exhaustive path
    uint32_t n_tris, n_lights, n_nodes, has_alpha;
    uint32_t bvh_stack_depth;                       // BVH path: traversal stack entries per lane in LDS = depth of this scene's tree
    uint32_t bvh_tile_nodes;                        // nodes 0 .. n-1 (the top of the tree, laid out breadth-first) a launch keeps in LDS; 0 = none
    uint64_t plane_share_mask;                      // exhaustive path: bit k = record k carries the plane row of record k-1
    TexScene tex;                                   // textures + shader-graph node lists (all nullptr without textures)
    DInst in2;                                      // meshes + instances kept as they are (in2.on; else all zero)
};

struct SurfacePoint {  // interaction.rs:15-48, minus what this path never reads
    Frame frame;
    vec3 p, ng;
    float prim_area;
    uint32_t material;
    int32_t light;
    uint32_t inst;
    vec2 uv;
};

AKR_HD vec3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }
// TriangleInterpolate: (1 - u - v) a + u b + v c
AKR_HD vec3 interp3(vec2 b, vec3 a0, vec3 a1, vec3 a2) {
    float w = 1.0f - b.x - b.y;
    return (a0 * w + a1 * b.x) + a2 * b.y;
}
AKR_HD vec3 xf_point(vec3 c0, vec3 c1, vec3 c2, vec3 t, vec3 p) { return ((c0 * p.x + c1 * p.y) + c2 * p.z) + t; }
AKR_HD vec3 xf_vector(vec3 c0, vec3 c1, vec3 c2, vec3 v) { return (c0 * v.x + c1 * v.y) + c2 * v.z; }

// MeshAggregate::surface_interaction (mesh.rs:487-654) from the eight rows of a shade record (q7: dpdu | flags), the instance's
// record `m` and the triangle's corner normals / tangents `nr` (read only when the flags say so).
AKR_D SurfacePoint surface_interaction_rows(const float4* m, float4 q0, float4 q1, float4 q2, float4 q3, float4 q4, float4 q5, float4 q6, float4 q7, bool has_normals,
                                            const float4* nr, vec2 bary) {
    SurfacePoint s;
    s.prim_area = q6.x;
    s.material = f2u(q6.y);
    s.inst = f2u(q6.z);
    s.light = (int32_t)f2u(q6.w);
    float4 c0 = m[0], c1 = m[1], c2 = m[2], t = m[3];
    vec3 p_local = interp3(bary, xyz(q0), xyz(q1), xyz(q2));
    s.p = xf_point(xyz(c0), xyz(c1), xyz(c2), xyz(t), p_local);
    s.ng = xyz(q3);
    s.frame = Frame{s.ng, xyz(q4), xyz(q5)};
    {  // uv = TriangleInterpolate(uv0, uv1, uv2), mesh.rs:527-546
        float w = 1.0f - bary.x - bary.y;
        s.uv = mk2((q0.w * w + q2.w * bary.x) + q4.w * bary.y, (q1.w * w + q3.w * bary.x) + q5.w * bary.y);
    }
    if (has_normals) {
        const uint32_t tf = f2u(q7.w);
        if (tf & (TRI_HAS_NORMALS | TRI_HAS_TANGENTS)) {  // mesh.rs:557-571, 591-602, 619-640
            vec3 ns = s.ng;
            if (tf & TRI_HAS_NORMALS) {
                vec3 ns_local = interp3(bary, xyz(nr[0]), xyz(nr[1]), xyz(nr[2]));
                float4 k0 = m[4], k1 = m[5], k2 = m[6];
                vec3 rr = (xyz(k0) * ns_local.x + xyz(k1) * ns_local.y) + xyz(k2) * ns_local.z;
                ns = normalize(rr * k0.w);  // k0.w = 1 / det
            }
            vec3 tt = xyz(q7);
            if (tf & TRI_HAS_TANGENTS) {
                vec3 tt_local = normalize(interp3(bary, xyz(nr[3]), xyz(nr[4]), xyz(nr[5])));
                tt = xf_vector(xyz(c0), xyz(c1), xyz(c2), tt_local);
            }
            s.frame = (tt.x != 0.0f || tt.y != 0.0f || tt.z != 0.0f) ? frame_from_n_t(ns, tt) : frame_from_n(ns);
        }
    }
    return s;
}
// ... from the folded records of a flattened scene
AKR_D SurfacePoint surface_interaction(const DScene& sc, uint32_t gid, vec2 bary) {
    const float4* r = sc.shade + (size_t)gid * SHADE_ROWS;
    float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3], q4 = r[4], q5 = r[5], q6 = r[6];
    const float4* m = sc.inst + (size_t)f2u(q6.z) * INST_ROWS;
    const bool has_normals = sc.normals != nullptr;
    float4 q7 = make_float4(0, 0, 0, 0);
    if (has_normals) q7 = r[7];
    return surface_interaction_rows(m, q0, q1, q2, q3, q4, q5, q6, q7, has_normals, has_normals ? sc.normals + (size_t)gid * 6 : nullptr, bary);
}

}  // namespace akr
