// scene_build.h -- host side of the scene: the flat description (what load.rs resolves the scene graph to)
// and its compiled, device-ready form.
#pragma once
#include <cmath>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../../include/akari_hip.h"
#include "../kernels.h"

namespace akr {

struct HostMesh {
    std::vector<float> vertices;     // 3 / vertex
    std::vector<uint32_t> indices;   // 3 / triangle
    std::vector<float> uvs;          // 6 / triangle or empty
    std::vector<float> normals;      // 9 / triangle or empty
    std::vector<float> tangents;     // 9 / triangle or empty
    std::vector<uint32_t> slots;     // 1 / triangle or empty
    uint32_t n_triangles() const { return (uint32_t)(indices.size() / 3); }
};
struct HostInstance {
    uint32_t mesh = 0;
    std::vector<uint32_t> materials;
    float transform[16];
};
struct HostImage {
    uint32_t width = 0, height = 0, format = 0, filter = 0, address = 0;
    std::vector<uint32_t> words;  // RGBA8: one word per texel; RGBA32F: four
};
struct HostGraph {
    std::vector<akr_shader_node> nodes;  // empty = constant material
    uint32_t input[AKR_IN_COUNT];
    HostGraph() { for (uint32_t& i : input) i = AKR_NODE_NONE; }
};
// Owning copy of an akr_scene_desc.
struct FlatScene {
    std::vector<HostMesh> meshes;
    std::vector<HostInstance> instances;
    std::vector<akr_material_desc> materials;
    akr_camera_desc camera;
    std::vector<float> ggx_table;  // 4096 or empty
    std::vector<HostImage> images;
    std::vector<HostGraph> graphs;  // empty, or one per material
    static FlatScene from_desc(const akr_scene_desc& d);
};

// Output of the host-side scene compiler: plain arrays ready for hipMemcpy.
struct CompiledScene {
    uint32_t n_tris = 0, n_lights = 0;
    std::vector<float> woop;             // exhaustive path: 12 / triangle; BVH path: kBvhTriWords / triangle (record | gid | pad), traversal order
    std::vector<uint32_t> tri_gid;       // BVH path: traversal order -> global id; empty = identity
    std::vector<float> shade;            // 32 / triangle (8 float4), by global id
    std::vector<float> normals;          // 12 / triangle (3 float4) or empty
    std::vector<float> inst;             // 32 / instance
    std::vector<DMaterial> materials;
    std::vector<uint32_t> inst_tri_offset;
    // lights
    std::vector<AliasEntry> light_entries;
    std::vector<float> light_pdf, light_power;
    std::vector<uint32_t> light_inst, light_tri_offset, light_n_tris;
    std::vector<AliasEntry> area_entries;
    std::vector<float> area_pdf;
    // 6-wide compressed BVH nodes (kBvhNodeWords = 16 words = 64 bytes each, host/bvh.cpp) or empty for the exhaustive path
    std::vector<uint32_t> bvh_nodes;
    uint32_t bvh_depth = 0;              // levels of the wide tree = the most stack entries a traversal can need
    // textures: pruned node lists of the materials with texture-fed inputs, image headers, texel words, raw inputs
    std::vector<DNode> tex_nodes;
    std::vector<DImage> images;
    std::vector<uint32_t> texels;
    std::vector<MatInputs> mat_inputs;   // one per material when any material is textured, else empty
    uint32_t tex_slots = 0;              // value slots per lane the widest node list needs (device/dtex.h), 0 without textures
    // The same node lists before slot allocation (arguments name nodes of the material's own list), element for element with
    // tex_nodes, and the scene's shader kinds: materials whose lists have the same shape share a kind (svm/compiler.rs:16-76); the
    // kind of a material is also in DMaterial.tex_n_nodes >> 16. What host/specialise.cpp turns into per-scene kernel code.
    std::vector<akr_shader_node> tex_nodes_ssa;
    struct ShaderKind {
        std::string signature;            // everything that shapes the code: surface kind, operations, argument topology, modes, image formats, fed inputs
        uint32_t mat_kind = 0, n_nodes = 0;
        std::vector<uint32_t> materials;  // the materials of this kind, ascending
    };
    std::vector<ShaderKind> shader_kinds;
    uint32_t absent = 0;                  // lobes no material of the scene can have (device/dbsdf.h AB_*)
    // Two-level mode (scene_inst.cpp; option `instancing`): the scene is kept as meshes + instances. woop / shade / normals / tri_gid /
    // bvh_nodes above stay EMPTY; nothing is stored per instance-triangle. What the device computes at a hit from these arrays is
    // the flattened record bit for bit (device/dinst.h).
    struct Instanced {
        bool on = false;
        std::vector<uint32_t> nodes;        // TLAS nodes (root = slot 0) followed by every mesh's BLAS nodes; child_base / tri_base are
                                            // relative to the start of their own tree
        std::vector<float> tlas_leaves;     // 16 words per top-level leaf entry (one per instance; more with option rebraid) in TLAS order: world->object
                                            // rows (3 x float4) | BLAS node offset, mesh triangle base, instance id, the node of the mesh's tree to start at
        std::vector<float> mesh_tris;       // 16 words per mesh triangle in BLAS order: v0 | uv0.x, v1 | uv0.y, v2 | uv1.x, uv1.y uv2.x uv2.y | prim
        std::vector<uint32_t> mesh_pos;     // per mesh triangle in MESH order: its position in mesh_tris (relative to the mesh's base)
        std::vector<uint32_t> mesh_meta;    // per mesh triangle in mesh order: material slot | TRI_HAS_* flags << 30
        std::vector<float> mesh_normals;    // 24 words per mesh triangle in mesh order (corner normals, corner tangents) or empty
        std::vector<uint32_t> inst_mats;    // the instances' material lists, concatenated
        std::vector<uint32_t> inst_light;   // light id per instance or 0xffffffff
        uint32_t tlas_nodes = 0, tlas_depth = 0, blas_depth = 0, n_mesh_tris = 0;
        uint32_t n_padding_classes = 0;    // per-mesh trees built: one per (mesh, padding class of its instances), scene_inst.cpp
    } instanced;
    bool has_textures = false;
    bool has_alpha = false;
    bool needs_ggx_table = false;
    float scene_lo[3], scene_hi[3];
};

// Padding of the acceleration structure's boxes. It covers the round-off of the slab test and of the triangle test, and both scale
// with the MAGNITUDE of the coordinates involved -- ray origins lie on the scene's surfaces or at the camera -- not with the scene's
// size alone: a building modelled 10 km from the origin loses hits with a padding that looks at its diagonal only (round 5:
// tools/offset_check.py, scenes/cbox shifted by 1000 differed from the oracle in 12 film floats, by 10 000 in 3 621).
inline float bvh_box_padding(const float lo[3], const float hi[3], const float* c2w /* column-major 4x4 */) {
    float diag2 = 0.0f, reach = 0.0f;
    for (int a = 0; a < 3; a++) {
        diag2 += (hi[a] - lo[a]) * (hi[a] - lo[a]);
        const float m = std::fabs(lo[a]) > std::fabs(hi[a]) ? std::fabs(lo[a]) : std::fabs(hi[a]);
        const float c = std::fabs(c2w[12 + a]);
        reach += m > c ? m : c;
    }
    const float diag = __builtin_sqrtf(diag2);
    return 4e-6f * (diag > reach ? diag : reach);
}

// Conditioning of a triangle's (u, v) parametrisation. The inside test computes u = r0 . p + c0 with |r0| = |e2| / |n| (v likewise with
// |r1| = |e1| / |n|): two of its three fma roundings act on partial sums of size S = |r0||p| + |c0| <= 2 |r0||p|, the rounded row and the
// rounded hit point add one unit each -- |du| <= 8 x 2^-24 x |r0| x |p|, p a point of the triangle -- and the triangle the test "sees" is
// displaced by du e1 + dv e2. For a well-shaped triangle that is a few ulp of its coordinates, what bvh_box_padding allows for; but
// |r0||e1| = 1 / sin(angle at the first vertex): a needle whose first vertex holds its small angle is displaced along its long axis by 1 / sin
// times as much, and the exhaustive loop accepts hits that far outside it (round 6: tests/bvh_model.py found such pairs in 9 of 150
// extreme scenes, among them the one film pixel of HISTORY R5.7). k[a] = |r0||e1_a| + |r1||e2_a| per axis (2 for a right angle at the first
// vertex, 2.31 for 60 degrees); a box must reach kTriCondEps x |p| x k[a] beyond the triangle along axis a. kTriCondFree of the flat
// padding is counted towards that (the rest of it covers the slab test's own round-off and the plane's); what exceeds it is added to the
// triangle's own box.
constexpr float kTriCondEps = 10.0f * 5.9604645e-8f;
constexpr float kTriCondFree = 0.64f;
inline void tri_conditioning(const double A[3], const double B[3], const double C[3], float k[3]) {
    const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
    for (int a = 0; a < 3; a++) {
        const double v = nl > 0.0 ? (l2 * std::fabs(e1[a]) + l1 * std::fabs(e2[a])) / nl : 0.0;  // (a degenerate triangle has an all-zero record: never accepted)
        k[a] = v < 1e30 ? (float)(v * 1.0001) : 1e30f;
    }
}
// what a box of a triangle with conditioning k needs along an axis beyond the flat padding `pad`; magnitude = |p| (2-norm) of the triangle's points
inline float tri_cond_extra(float k, float magnitude, float pad) {
    const float need = kTriCondEps * magnitude * k, have = kTriCondFree * pad;
    return need > have ? need - have : 0.0f;
}
// 2-norm of the largest coordinates a box holds
inline float box_magnitude(const float lo[3], const float hi[3]) {
    float m2 = 0.0f;
    for (int a = 0; a < 3; a++) {
        const float m = std::fabs(lo[a]) > std::fabs(hi[a]) ? std::fabs(lo[a]) : std::fabs(hi[a]);
        m2 += m * m;
    }
    return __builtin_sqrtf(m2) * 1.0001f;
}
DMaterial fold_material(const akr_material_desc& m, uint32_t color = 0);
// materials / node lists / raw inputs under the colour pipeline `color` (scene_build.cpp); fills out.materials, out.tex_nodes,
// out.mat_inputs only
void compile_materials(const FlatScene& flat, uint32_t color, CompiledScene& out, std::vector<akr_material_desc>& descs);
void build_alias_table(const std::vector<float>& weights, std::vector<AliasEntry>& entries, std::vector<float>& pdf);
// Process-wide tuning switches and test hooks (akr_option_set / akr_option_get of the C ABI). Each starts from its environment
// variable, read ONCE when the first of them is looked at; after that only akr_option_set changes it. No launch path calls getenv.
//   force_bvh    AKR_FORCE_BVH=1          scenes of <= 64 triangles get a BVH too (both intersectors on one scene)
//   bvh_balanced AKR_BVH_BALANCED=1       the median-split fallback builder instead of SAH
//   defer_metal  AKR_PT_DEFER_METAL=<m>   -1 = the library decides (default); 0 = off; m > 0 = iterations with (i & m) != 0 put conductor hits off
//   wavefront    AKR_PT_MODE=wavefront|megakernel   the wavefront schedule (wf_kernels.hip) instead of the megakernel: 1 = wherever it can run, 0 = never,
//                                         -1 = the library decides (default: pt sessions of >= 0.5 M ... 2 M pixels, by mesh size, on scenes kept as meshes + instances)
//   simple_kernels AKR_PT_SIMPLE=0        0 = never use the SIMPLE instantiations (scenes without coat / transmission / normal map / glass)
//   defer_on     (no environment hook)   BVH kernels of textured scenes: which hits the deferral puts off (0 / 1 conductor lobe, 2 texture-fed, 3 both)
//   specialise   AKR_SPECIALISE=<v>       per-scene kernels for scenes with texture-fed materials (host/specialise.cpp): -1 = the library decides (a cached
//                                         kernel always; a compile for renders of at least kSpecAutoSamples samples), 0 = never (the interpreter), 1 = always
//   specialise_waves AKR_SPECIALISE_WAVES=<n>  waves per SIMD a per-scene kernel is compiled for: 0 = the library's choice, else 2..4
//   instancing   AKR_INSTANCING=<v>       meshes + instances kept as they are (BLAS per mesh, TLAS over instances; scene_inst.cpp): -1 auto, 0 never, 1 always
//   rebraid      AKR_REBRAID=<k>          scenes kept as meshes + instances: the top-level tree is built over k x as many (instance, subtree) pairs as
//                                         there are instances, the largest instance boxes opened first (scene_inst.cpp); 1 = one pair per instance
//   arith        AKR_ARITH=1              pt megakernel in the relaxed arithmetic tier (flattened scenes; precompiled kernels): hardware rcp / sqrt /
//                                         sin / cos / log / exp and contraction instead of the bit-exact contract
//   pad_percent  (no environment hook)    test hook: box padding in percent of the derived value (100)
//   wf_sort      AKR_WF_SORT=1            wavefront schedule: ray queues sorted by origin cell + direction octant before each trace launch
//   wf_groups    AKR_WF_GROUPS=<g>        wavefront schedule: the slots run as g groups with queues and streams of their own (api_pt.cpp wf_run); 0 = the library decides
//   wf_carry     AKR_WF_CARRY=0           wavefront schedule: 0 = every trace launch traces its rays to the end (1, default: a wave that finds the queue empty and
//                                         has few lanes left hands their traversals to the next launch -- wf_kernels.hip; launches of >= 65 536 rays only;
//                                         a value n > 1 = test hook: launches of >= n rays, and waves hand over after 4 steps with up to 56 lanes left)
//   sched_trial  AKR_SCHED_TRIAL=<v>      flattened scenes, option wavefront = -1: -1 (default) = a long render of a large frame of a large untextured scene starts with two passes
//                                         under each schedule and goes on with the faster; 0 = never (the megakernel); 1 = every pt session on a scene with a tree (tests)
//   max_fused_passes (no environment hook)     most passes akr_pt_passes fuses into one launch: 0 = adaptive (16, up to 64 once a pass has been timed), else 1..64
struct TuningOptions {
    int force_bvh = 0, bvh_balanced = 0, defer_metal = -1, wavefront = -1, simple_kernels = 1;
    int defer_on = 0;  // BVH kernels of textured scenes: which hits the deferral puts off -- 0 / 1 = the conductor lobe (default), 2 = texture-fed materials, 3 = both
    int specialise = -1, specialise_waves = 0;
    int max_fused_passes = 0;
    int instancing = -1;  // two-level acceleration structure for scenes whose meshes are instanced: -1 the library decides (flattening is the
                          // default while its records fit a budget), 0 never, 1 whenever a mesh has more than one instance
    int rebraid = 1;  // kept scenes: (instance, subtree) pairs of the top-level tree per instance, on average (partial re-braiding); 1 = off
    int arith = 0;    // arithmetic tier of the pt megakernel: 0 = the AKR-F32 contract (bit-exact with the oracle), 1 = relaxed (pt_kernels_relaxed.hip:
                      // films within north_star's relRMSE < 1e-3 of the oracle, not identical to it)
    int pad_percent = 100;  // test hook: the padding of the acceleration structures' boxes (flat part and needle part) in percent of what the compiler derives --
                            // tests/test_bvh_conservative.py shows with it how far the derived padding is from the first lost hit
    int wf_groups = 0;  // wavefront schedule: slot groups whose init / trace / shade chains run side by side on streams of their own (1 = one chain, 0 = the library decides)
    int wf_carry = 1;  // wavefront schedule: 1 = the last rays of a trace launch are carried into the next one (wf_kernels.hip), 0 = every launch traces to the end
    int sched_trial = -1;  // flattened scenes under option wavefront = -1: a timed trial of both schedules at the start of a long render (api_pt.cpp schedule_trial):
                           // -1 = for the sessions it can pay for (large frame, large scene, many passes), 0 = never, 1 = every pt session on a scene with a tree (tests)
    int wf_sort = 0;  // wavefront schedule: 1 = the ray queues are sorted by (Morton code of the origin, octant) before every trace launch (wf_sort.hip)
};
constexpr uint64_t kSpecAutoSamples = 1ull << 31;  // option specialise = -1: a first-use compile (about a second; 20-30 % of the render to win) has to be worth it
TuningOptions tuning();                          // a snapshot (thread-safe)
bool tuning_set(const char* name, int value);    // false: unknown name
bool tuning_get(const char* name, int* value);

void compile_scene(const FlatScene& flat, CompiledScene& out);
float triangle_emission_power(const CompiledScene& out, const TexScene& host_tex, uint32_t material, uint32_t prim, vec2 uv0, vec2 uv1, vec2 uv2, float area);
bool instance_may_emit(const CompiledScene& out, const std::vector<akr_material_desc>& descs, const HostInstance& in);
// scene_inst.cpp: does this scene take the two-level route, and its geometry + light tables if so (materials and the instance table
// of `out` must be filled: compile_scene calls it)
bool want_instancing(const FlatScene& flat);
struct InstXf;
void compile_instanced_geometry(const FlatScene& flat, const std::vector<InstXf>& xf, const std::vector<akr_material_desc>& descs, CompiledScene& out);
// PerspectiveCameraData::new (camera/mod.rs:119-153)
void camera_matrices(const akr_camera_desc& cam, float r2c[16], float c2w[16], uint32_t* c2w_identity);
PcgStartConsts pcg_start_constants();

// scene.json / method.json readers (host/scene_json.cpp); throw std::runtime_error on failure
FlatScene load_scene_json(const std::string& path);
void parse_method_json(const std::string& text, akr_pt_config* cfg, std::string* film_out);
// all tasks of a RenderTask file (Single | Multi), lib.rs:103-109; allow_sampler_override: pmj02bn -> independent
// most LDS a workgroup of the exhaustive path tracer kernels spends on staged scene tables (4 workgroups per CU, 160 KB of LDS)
constexpr size_t kStageMaxBytes = 32 * 1024;
// the same for the BVH path, whose workgroups already hold 24 KB of traversal stacks each
constexpr size_t kStageMaxBytesBvh = 12 * 1024;

struct ParsedTask {
    bool is_aov = false;     // Method::NormalVis instead of Method::PathTracer
    akr_pt_config cfg;
    akr_aov_config aov;
    bool is_gpt = false;     // Method::GradientPathTracer
    akr_gpt_config gpt;
    bool is_mcmc = false;    // Method::McmcOpt
    akr_mcmc_config mcmc;
    std::string film_out;
};
std::vector<ParsedTask> parse_render_tasks(const std::string& text, bool allow_sampler_override);
// tables of the pmj02bn sampler (host/pmj_tables.cpp): 5 x 65536 x 2 u32 points; 48 x 128 x 128 u16 blue-noise arrays
void make_pmj02_sets(std::vector<uint32_t>& out);
void load_bluenoise(std::vector<uint16_t>& out);
// image writers (host/image_io.cpp)
void write_image(const std::string& path, const float* rgb, uint32_t w, uint32_t h);
// PNG -> RGBA8 in file order (image crate `decode().to_rgba8()` conventions); throws std::runtime_error
void decode_png(const uint8_t* data, size_t n, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba);
// JPEG (baseline + progressive Huffman, 8 bit, 1 or 3 components) -> RGBA8 in file order
void decode_jpeg(const uint8_t* data, size_t n, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba);
// OpenEXR (single-part scanline; none / RLE / ZIPS / ZIP; half / float / uint channels) -> RGBA f32 in file order
void decode_exr(const uint8_t* data, size_t n, uint32_t& w, uint32_t& h, std::vector<float>& rgba);
// TIFF (classic; strips / tiles; 8 / 16-bit and float samples; none / LZW / deflate / PackBits; predictor) and DDS (DXT1 / 3 / 5)
// -> RGBA8 in file order (host/image_formats.cpp)
void decode_tiff(const uint8_t* data, size_t n, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba);
void decode_dds(const uint8_t* data, size_t n, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba);
std::vector<uint8_t> inflate_zlib_stream(const uint8_t* data, size_t n);  // the PNG reader's inflate (zlib framing)

}  // namespace akr
