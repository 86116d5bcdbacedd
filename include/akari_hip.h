/* akari_hip.h -- C ABI of libakari_hip.so: an MI355X (gfx950) implementation of akari_render's `pt`
 * path-tracing integrator.
 *
 * This is the boundary a Rust `impl Integrator for HipPathTracer` would bind (INTEGRATION.md shows the
 * `extern "C"` block). The reference has no C ABI for integrators; the path sits behind
 *   trait Integrator::render(&self, scene, sampler, color_pipeline, film, session)
 *                                              crates/akari_integrator/src/lib.rs:38-47
 *   pt::render(device, scene, sampler, color_pipeline, film, &Config, &RenderSession)
 *                                              crates/akari_integrator/src/pt.rs:1161-1172
 * and each entry point below names the reference item it replaces. Conventions follow the reference's
 * own FFI precedent (crates/akari_api/src/lib.rs:6-26: plain pointers + lengths, paired create/free):
 *   - every function returns an int32 status (AKR_OK or a negative akr_status); nothing throws or aborts
 *     across the ABI; akr_last_error() returns a thread-local, library-owned message;
 *   - the caller owns every input array (the library copies during *_create) and every output buffer;
 *   - handles are opaque and owned by the library until the matching *_destroy;
 *   - one context = one HIP device + one stream; calls on one context must be serialised by the caller,
 *     distinct contexts may be used from distinct threads / processes (one process per GPU).
 * No PyTorch / C++ types appear in any signature.
 */
#ifndef AKARI_HIP_H
#define AKARI_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define AKR_API
#else
#define AKR_API __attribute__((visibility("default")))
#endif

typedef enum {
    AKR_OK = 0,
    AKR_ERR_INVALID_ARGUMENT = -1,
    AKR_ERR_HIP = -2,          /* a HIP runtime call failed (message has hipGetErrorString) */
    AKR_ERR_NO_DEVICE = -3,    /* no gfx950 device / HIP runtime unavailable */
    AKR_ERR_IO = -4,           /* scene / method file could not be read */
    AKR_ERR_PARSE = -5,        /* malformed JSON or unsupported node */
    AKR_ERR_UNSUPPORTED = -6,  /* feature of the reference that this build does not cover */
    AKR_ERR_OUT_OF_MEMORY = -7,
    AKR_ERR_RENDER = -8        /* the device ran but the result is not a valid render (e.g. mcmc bootstrap found no light path) */
} akr_status;

typedef struct akr_context akr_context;
typedef struct akr_scene akr_scene;
typedef struct akr_film akr_film;
typedef struct akr_pt_session akr_pt_session;

/* ---------------------------------------------------------------------------------------------------
 * Scene description: the flat form of the reference's scene graph after load.rs has resolved buffers.
 * ------------------------------------------------------------------------------------------------- */

/* One `Mesh` (crates/akari_render/src/mesh.rs:14-25; on disk crates/akari_scenegraph/src/scene.rs:333-340). */
typedef struct {
    uint32_t n_vertices, n_triangles;
    const float *vertices;          /* 3 * n_vertices, object space */
    const uint32_t *indices;        /* 3 * n_triangles */
    const float *uvs;               /* 2 * 3 * n_triangles, per corner, or NULL */
    const float *normals;           /* 3 * 3 * n_triangles, per corner, or NULL */
    const float *tangents;          /* 3 * 3 * n_triangles, per corner, or NULL */
    const uint32_t *material_slots; /* n_triangles, or NULL (= slot 0) */
} akr_mesh_desc;

/* One `MeshInstanceHost` (mesh.rs:197-204); instance index = order in this array
 * (the reference iterates its BTreeMap, i.e. lexicographic node-id order, load.rs:287-292). */
typedef struct {
    uint32_t mesh;
    uint32_t n_materials;
    const uint32_t *materials;      /* indices into akr_scene_desc.materials, one per material slot */
    float transform[16];            /* column-major object->world (AffineTransform.m, geometry.rs:203-209) */
} akr_instance_desc;

/* A surface shader graph whose inputs are constants (all of scenes/cbox), already folded:
 * svm/compiler.rs:116-337 + svm/eval.rs:97-269 applied on the host. Colours are linear RGB in the colour space their
 * Rgb node declares (akari_scenegraph ColorSpace: srgb, or ACEScg when the AKR_MAT_CS_* bit of `kind` is set); the render's
 * ColorPipeline converts them (svm/texture/mod.rs:9-43): Rgb node space -> rgb_colorspace, then spectral_uplift ->
 * the space of color_repr. */
typedef enum {
    AKR_MAT_PRINCIPLED = 0,  /* ShaderNode::PrincipledBsdf, svm/surface/principled.rs */
    AKR_MAT_DIFFUSE = 1,     /* ShaderNode::DiffuseBsdf,    svm/surface/diffuse.rs:83-104 */
    AKR_MAT_GLASS = 2,       /* ShaderNode::GlassBsdf,      svm/surface/glass.rs */
    AKR_MAT_EMISSION = 3,    /* ShaderNode::Emission,       svm/mod.rs:114-123 */
    AKR_MAT_KIND_MASK = 0xff,
    /* OR-ed into akr_material_desc.kind: this constant colour input is given in ACEScg instead of sRGB primaries */
    AKR_MAT_CS_BASE_COLOR = 0x100, AKR_MAT_CS_SPECULAR_TINT = 0x200, AKR_MAT_CS_COAT_TINT = 0x400, AKR_MAT_CS_EMISSION_COLOR = 0x800
} akr_material_kind;

typedef struct {
    uint32_t kind;
    float base_color[3];
    float base_alpha;
    float metallic, roughness, ior, specular_ior_level;
    float specular_tint[3];
    float transmission_weight;
    float coat_weight, coat_roughness, coat_ior;
    float coat_tint[3];
    float emission_color[3];
    float emission_strength;
    float normal[3];
} akr_material_desc;

/* ---- Shader graphs with non-constant inputs (textures) -------------------------------------------------------
 * A material whose inputs are all constants is fully described by akr_material_desc. When an input of the surface
 * node is fed by a texture expression, the material additionally carries its node list (the reference's compiled
 * bytecode, svm/compiler.rs:116-337: one entry per node in topological order, arguments refer to EARLIER entries)
 * and, per input of the surface node, the index of the node that feeds it (AKR_NODE_NONE = use the constant in
 * akr_material_desc). Every node value is a float4; narrower values are zero-extended, so the reference's
 * eval_float*_auto_convert rules (svm/eval.rs:301-349) are plain .x / .xy / .xyz reads. */
#define AKR_NODE_NONE 0xffffffffu
typedef enum {
    AKR_NODE_CONST = 0,          /* Float / Float3: (k0,k1,k2,0)                           eval.rs:109-116 */
    AKR_NODE_RGB = 1,            /* Rgb: (k0,k1,k2,1) converted from its colour space (arg0: 0 / NONE = srgb,
                                    1 = ACEScg) to the pipeline's rgb_colorspace           eval.rs:123-133 */
    AKR_NODE_TEXCOORDS = 2,      /* si.uv: (u,v,0,0)                                       eval.rs:219-226 */
    AKR_NODE_IMAGE = 3,          /* arg0 image, arg1 uv node | NONE, arg2 1 = sRGB decode  eval.rs:134-154 */
    AKR_NODE_MAPPING = 4,        /* arg0 vector, arg1 location, arg2 scale, arg3 akr_mapping_type (rotation is
                                    ignored by the reference)                              eval.rs:192-207 */
    AKR_NODE_CHECKERBOARD = 5,   /* arg0 vector | NONE, arg1 scale, arg2 color1, arg3 color2   eval.rs:227-240 */
    AKR_NODE_SPECTRAL_UPLIFT = 6,/* arg0 rgb: rgb_colorspace -> the space of color_repr    eval.rs:155-175 */
    AKR_NODE_SEPARATE_COLOR = 7, /* arg0 colour (carries the value for AKR_NODE_EXTRACT)   eval.rs:241-256 */
    AKR_NODE_EXTRACT = 8,        /* arg0 node, arg1 akr_extract_field                      eval.rs:208-218 */
    AKR_NODE_NORMAL_MAP = 9      /* arg0 normal, arg1 strength (tangent space)             eval.rs:176-191 */
} akr_node_op;
typedef enum { AKR_MAPPING_POINT = 0, AKR_MAPPING_TEXTURE = 1 } akr_mapping_type;
typedef enum { AKR_FIELD_RED = 0, AKR_FIELD_GREEN = 1, AKR_FIELD_BLUE = 2, AKR_FIELD_UV = 3 } akr_extract_field;
typedef struct {
    uint32_t op;
    uint32_t arg[4];
    float k[3];
} akr_shader_node;   /* 32 bytes */

/* Inputs of the surface node a texture expression may feed (order of akr_material_graph.input). Diffuse / Glass /
 * Emission use BASE_COLOR for `color`; Glass ROUGHNESS, IOR; Emission EMISSION_STRENGTH for `strength` and
 * EMISSION_COLOR for `color`. */
typedef enum {
    AKR_IN_BASE_COLOR = 0, AKR_IN_METALLIC, AKR_IN_ROUGHNESS, AKR_IN_IOR, AKR_IN_SPECULAR_IOR_LEVEL, AKR_IN_SPECULAR_TINT,
    AKR_IN_TRANSMISSION_WEIGHT, AKR_IN_COAT_WEIGHT, AKR_IN_COAT_ROUGHNESS, AKR_IN_COAT_IOR, AKR_IN_COAT_TINT,
    AKR_IN_EMISSION_COLOR, AKR_IN_EMISSION_STRENGTH, AKR_IN_NORMAL, AKR_IN_COUNT
} akr_material_input;
typedef struct {
    uint32_t n_nodes;                 /* 0 = constant material */
    uint32_t _pad;
    const akr_shader_node *nodes;
    uint32_t input[AKR_IN_COUNT];     /* node index or AKR_NODE_NONE */
} akr_material_graph;

/* One (texture, sampler) pair of the reference's bindless heap (load.rs:477-489,680-702): texels are what ends up in
 * the Tex2d -- row 0 is v = 0 (encoded images are flipped vertically on load, load.rs:596; raw float images are not),
 * always 4 channels (load.rs:552-569). */
typedef enum { AKR_IMAGE_RGBA8 = 0 /* PixelStorage::Byte4 */, AKR_IMAGE_RGBA32F = 1 /* PixelStorage::Float4 */ } akr_image_format;
typedef enum { AKR_TEX_FILTER_NEAREST = 0, AKR_TEX_FILTER_LINEAR = 1 } akr_tex_filter;       /* load.rs:690-699 */
typedef enum { AKR_TEX_REPEAT = 0, AKR_TEX_CLIP = 1 /* zero */, AKR_TEX_MIRROR = 2, AKR_TEX_EXTEND = 3 /* edge */ } akr_tex_address;
typedef struct {
    uint32_t width, height;
    uint32_t format, filter, address, _pad;
    const void *texels;               /* 4 * width * height bytes (RGBA8) or floats (RGBA32F) */
} akr_image_desc;

/* PerspectiveCamera (camera/mod.rs:15-66): only the fields generate_ray uses. */
typedef struct {
    float c2w[16];           /* column-major camera->world */
    float fov;               /* radians; spans the larger image side */
    uint32_t width, height;
} akr_camera_desc;

typedef struct {
    uint32_t n_meshes, n_instances, n_materials, _pad;
    const akr_mesh_desc *meshes;
    const akr_instance_desc *instances;
    const akr_material_desc *materials;
    akr_camera_desc camera;
    /* Optional 16x16x16 f32 "ggx_dielectric_s" table (svm/surface/precompute.rs:133-145). NULL = the
     * library computes it on the GPU the first time a material needs it (same definition, 2^20 samples). */
    const float *ggx_dielectric_table;
    /* Textures (optional): n_images = 0 and material_graphs = NULL for scenes with constant materials. */
    uint32_t n_images, _pad2;
    const akr_image_desc *images;
    const akr_material_graph *material_graphs;   /* n_materials entries, or NULL */
} akr_scene_desc;

/* ---------------------------------------------------------------------------------------------------
 * Render configuration = pt::Config (pt.rs:916-944) + RenderConfig.sampler / .film.filter (lib.rs:75-102).
 * ------------------------------------------------------------------------------------------------- */
typedef enum { AKR_FILTER_BOX = 0, AKR_FILTER_GAUSSIAN = 1 } akr_filter_type;   /* film.rs:22-54 */
/* sampler/mod.rs:282-295. PMJ02BN runs on REGENERATED tables (the reference's copies of pbrt-v4's are not in its tree):
 * same algorithm, different point sets / blue-noise arrays, so its images are not bit-comparable with the reference's. */
/* SOBOL: the reference ships only the data stub of a Sobol' sampler (akari_data/src/lib.rs:13,19: sobolmat, never used by a
 * Sampler impl); this build's "sobol" is an Owen-scrambled, padded Sobol' (0,2)-sequence sampler with the same state and
 * interface as Pmj02BnSampler (index-based, per-dimension Kensler permutation of the sample index) that needs no tables. */
typedef enum { AKR_SAMPLER_INDEPENDENT = 0, AKR_SAMPLER_PMJ02BN = 1, AKR_SAMPLER_SOBOL = 2 } akr_sampler_type;
/* ColorPipeline (color.rs:663-676) as bits of akr_pt_config.color; 0 = the default {color_repr: Rgb(SRgb), rgb_colorspace:
 * SRgb}. Shading happens in the space of color_repr; the film is always sRGB-primaries linear (film.rs:176-229 converts
 * every sample with color.to_rgb(SRgb), color.rs:262-275). Spectral rendering is todo!() in the reference. */
typedef enum { AKR_COLOR_REPR_ACESCG = 1, AKR_COLOR_RGB_ACESCG = 2 } akr_color_pipeline_bits;

typedef struct {
    uint32_t spp, max_depth, spp_per_pass, rr_depth;
    uint32_t use_nee, indirect_only, force_diffuse;
    int32_t pixel_offset[2];
    int32_t debug_depth;      /* -1 = None */
    uint32_t filter_type;
    float filter_radius;
    uint32_t sampler_type;
    uint32_t color;           /* akr_color_pipeline_bits; 0 = sRGB / sRGB */
    uint64_t sampler_seed;
    /* Multi-GPU sharding (no reference counterpart): rank r of n renders the pixel tiles (tx, ty) with
     * morton(tx, ty) % n == r -- tiles of tile_w x tile_h (0 = 32) dealt along the Z-order curve (x in the even bits of the code), so that
     * 8 ranks each own one tile of every aligned 4 x 2 block of tiles. shard_count <= 1 renders all.
     * Pixels a rank does not own are left untouched in its film, so a sum-reduce assembles the frame. */
    uint32_t shard_rank, shard_count, tile_w, tile_h;
    /* Sample-range split (SURVEY.md 8b "sample range", 8e "sample-range split"; no reference counterpart): the session renders
     * samples [sample_begin, sample_begin + sample_count) of every pixel it owns, out of the `spp` samples of the whole render;
     * sample_count = 0 means "all of them" (sample_begin must then be 0). GPU g of n takes [g * spp / n, (g + 1) * spp / n) of
     * EVERY pixel -- perfectly balanced at any resolution -- and the films' sum-reduce (akr_film_reduce) gives the frame.
     * Only for the index-based samplers (PMJ02BN, SOBOL), whose sample s of a pixel is a pure function of (pixel, s, seed,
     * spp): Pmj02BnState.sample_index simply starts at sample_begin - 1 (sampler/mod.rs:451-466, 650-663). With the
     * INDEPENDENT sampler a range is refused with AKR_ERR_UNSUPPORTED: start() advances the pixel's PCG stream from wherever
     * the previous sample stopped (sampler/mod.rs:115-131, 192-203), so sample s cannot be drawn without drawing 0 .. s-1. */
    uint32_t sample_begin, sample_count;
} akr_pt_config;

/* Device counters of one render call (the n_* of the algorithmic-bytes model). */
typedef struct {
    uint64_t n_samples;       /* camera paths */
    uint64_t n_closest;       /* closest-hit queries (scene.rs:131-153) */
    uint64_t n_shadow;        /* any-hit queries (scene.rs:155-185) */
    uint64_t n_shaded;        /* path vertices that ran NEE + BSDF sampling (pt.rs:471-513) */
    uint64_t n_node_visits;   /* BVH nodes fetched (0 on the exhaustive small-scene path) */
    uint64_t n_tri_tests;     /* ray-triangle tests */
    double kernel_ms;         /* sum of the path-tracing kernels' durations, HIP events on the context stream */
    uint32_t n_launches;
    uint32_t _pad;
} akr_pt_stats;

/* ---------------------------------------------------------------------------------------------------
 * Entry points
 * ------------------------------------------------------------------------------------------------- */

/* Thread-local description of the last failure on this thread ("" if none). Never NULL. */
AKR_API const char *akr_last_error(void);

/* Replaces luisa::Context::create_device(-d ...) (akari_api/src/bin/akari_cli.rs:61-62).
 * `device` is the HIP device ordinal. Fails with AKR_ERR_NO_DEVICE when no GPU is visible:
 * there is no CPU fallback in this library. */
AKR_API int32_t akr_context_create(int32_t device, akr_context **out);
AKR_API int32_t akr_context_destroy(akr_context *ctx);
/* Blocks until all work queued on the context's stream is done. */
AKR_API int32_t akr_context_synchronize(akr_context *ctx);
/* Number of HIP devices visible to the process (0 without a GPU; never fails for that reason). */
AKR_API int32_t akr_device_count(int32_t *count);
/* The HIP device ordinal the context was created on. */
AKR_API int32_t akr_context_device_ordinal(akr_context *ctx, int32_t *device);
/* Device name / compute units / HBM bytes of the context's GPU. `name` gets at most name_len-1 chars. */
AKR_API int32_t akr_context_device_info(akr_context *ctx, char *name, uint32_t name_len, uint32_t *compute_units,
                                        uint64_t *hbm_bytes);

/* Replaces SceneLoader::do_load after buffers are resolved (load.rs:238-456): uploads geometry, builds the
 * BVH (replacing rtx::Accel, mesh.rs:288-294,331-333), folds materials, runs the emission-power estimate
 * and builds the light alias tables (load.rs:308-444). */
AKR_API int32_t akr_scene_create(akr_context *ctx, const akr_scene_desc *desc, akr_scene **out);
/* Both scene constructors accept ctx == NULL: the scene is then compiled on the host only (flattening, material
 * folding, light tables, BVH) and can be inspected with the akr_scene_get_* calls but not rendered. */
/* Replaces akari_render::load::load_from_path (load.rs:63-72 -> MmapScene::open, scenegraph scene.rs:603-647):
 * parses the reference's scene.json (+ its binary buffers, resolved relative to the JSON's directory, falling
 * back to the basename for the absolute Windows paths found in scenes/cbox). width/height override the
 * camera's sensor resolution when non-zero (Camera::set_resolution, camera/mod.rs:54-65). */
AKR_API int32_t akr_scene_load(akr_context *ctx, const char *scene_json_path, uint32_t width, uint32_t height,
                               akr_scene **out);
AKR_API int32_t akr_scene_destroy(akr_scene *scene);
/* Camera::set_resolution (camera/mod.rs:54-65). */
AKR_API int32_t akr_scene_set_resolution(akr_scene *scene, uint32_t width, uint32_t height);

typedef struct {
    uint32_t width, height;
    uint32_t n_instances, n_triangles, n_materials, n_lights;
    uint32_t n_bvh_nodes;     /* 0 when the scene uses the exhaustive small-scene intersector */
    uint32_t uses_bvh;        /* 0 exhaustive small-scene intersector, 1 one tree over the flattened triangles, 2 meshes + instances
                               * (a tree over the instances, one per mesh; option "instancing") */
    uint64_t device_bytes;    /* HBM held by the scene (a host-only scene: what its upload would take) */
    uint32_t node_bytes;      /* bytes read per BVH node visit (64: 6-wide compressed node = one sector), 0 without a BVH */
    uint32_t node_stride_bytes; /* distance between nodes in memory */
    uint32_t tri_bytes;       /* bytes read per triangle test: 48 (exhaustive path) or 64 (BVH path: record + id) */
    uint32_t bvh_depth;       /* levels of the wide tree = most traversal-stack entries a ray can need */
} akr_scene_info;
AKR_API int32_t akr_scene_get_info(const akr_scene *scene, akr_scene_info *info);
/* Light `light` of LightAggregate (light/mod.rs:87-98): owning instance, total power, selection pdf. */
AKR_API int32_t akr_scene_get_light(const akr_scene *scene, uint32_t light, uint32_t *instance, float *power, float *pdf);
/* The folded ggx_dielectric_s table in use (4096 floats), for comparison with a golden copy. */
AKR_API int32_t akr_scene_get_ggx_table(const akr_scene *scene, float *dst4096);
/* Host-side copy of the flattened description akr_scene_load produced (for loader tests):
 * counts first, then the caller sizes its buffers and asks for the arrays. */
AKR_API int32_t akr_scene_get_desc_counts(const akr_scene *scene, uint32_t *n_meshes, uint32_t *n_instances, uint32_t *n_materials);
AKR_API int32_t akr_scene_get_mesh(const akr_scene *scene, uint32_t mesh, akr_mesh_desc *out /* pointers owned by scene */);
AKR_API int32_t akr_scene_get_instance(const akr_scene *scene, uint32_t instance, akr_instance_desc *out);
AKR_API int32_t akr_scene_get_material(const akr_scene *scene, uint32_t material, akr_material_desc *out);
AKR_API int32_t akr_scene_get_camera(const akr_scene *scene, akr_camera_desc *out);
AKR_API int32_t akr_scene_get_image_count(const akr_scene *scene, uint32_t *n_images);
AKR_API int32_t akr_scene_get_image(const akr_scene *scene, uint32_t image, akr_image_desc *out /* texels owned by scene */);
AKR_API int32_t akr_scene_get_material_graph(const akr_scene *scene, uint32_t material, akr_material_graph *out /* nodes owned by scene */);
/* Read-only views of the compiled, device-ready arrays (owned by the scene): see akari_render_amd/csrc/device/dscene.h
 * for the record layouts. */
typedef enum {
    AKR_ARRAY_WOOP = 0,          /* ray-triangle records in traversal order: f32[12 * n_tris] (exhaustive path) or, with a BVH,
                                  * 16 words per triangle = the 12-float record | global triangle id (u32) | 3 unused */
    AKR_ARRAY_TRI_GID = 1,       /* u32[n_tris]       traversal order -> global triangle id (empty = identity) */
    AKR_ARRAY_SHADE = 2,         /* f32[32 * n_tris]  shading records by global triangle id */
    AKR_ARRAY_INSTANCES = 3,     /* f32[32 * n_instances] */
    AKR_ARRAY_MATERIALS = 4,     /* folded materials, 256 B each */
    AKR_ARRAY_BVH_NODES = 5,     /* u32[(node_stride_bytes / 4) * n_bvh_nodes]: 64-byte compressed nodes of six children (csrc/host/bvh.cpp) */
    AKR_ARRAY_LIGHT_ENTRIES = 6, /* {u32 j, f32 t}[n_lights] */
    AKR_ARRAY_LIGHT_PDF = 7,     /* f32[n_lights] */
    AKR_ARRAY_AREA_ENTRIES = 8,  /* {u32 j, f32 t}[sum of light triangle counts] */
    AKR_ARRAY_AREA_PDF = 9,      /* f32[sum of light triangle counts] */
    AKR_ARRAY_INST_TRI_OFFSET = 10, /* u32[n_instances + 1] */
    AKR_ARRAY_R2C = 11,          /* f32[16] raster->camera, column-major (camera/mod.rs:119-153) */
    AKR_ARRAY_C2W = 12,          /* f32[16] */
    /* texture-fed materials (default colour pipeline; empty without them): csrc/device/dtex.h, dbsdf.h for the layouts */
    AKR_ARRAY_TEX_NODES = 13,    /* pruned, slot-allocated node lists, 32 B each */
    AKR_ARRAY_TEX_IMAGES = 14,   /* image headers, 32 B each */
    AKR_ARRAY_TEX_TEXELS = 15,   /* u32[] texel words of all images */
    AKR_ARRAY_MAT_INPUTS = 16,   /* raw inputs per material = akr_material_desc, 104 B each */
    /* a scene kept as meshes + instances (akr_scene_info.uses_bvh == 2; csrc/host/scene_inst.cpp, empty otherwise). AKR_ARRAY_BVH_NODES
     * then holds the top-level tree over the instances followed by every mesh's own tree; WOOP / TRI_GID / SHADE are empty */
    AKR_ARRAY_INST_LEAVES = 17,  /* f32[16 * top-level leaf entries] (one or more per instance with triangles): world->object rows | tree, mesh, instance, entry node */
    AKR_ARRAY_MESH_TRIS = 18,    /* f32[16 * mesh triangles] object-space vertices and uvs in each mesh's traversal order (the host's copy: on the device the top
                                    bit of a record's last word says that some instance gives the triangle its even neighbour's plane row) */
    AKR_ARRAY_MESH_POS = 19,     /* u32[mesh triangles] mesh order -> position in MESH_TRIS */
    AKR_ARRAY_MESH_META = 20,    /* u32[mesh triangles] material slot | flags << 30 */
    AKR_ARRAY_MESH_NORMALS = 21  /* f32[24 * mesh triangles] corner normals / tangents (empty if no mesh has any) */
} akr_array_id;
AKR_API int32_t akr_scene_get_array(const akr_scene *scene, int32_t which, const void **ptr, uint64_t *bytes);

/* Replaces Film::new (film.rs:93-151): f32[(1 + 2*3) * W * H] on the device, laid out exactly as the
 * reference's buffer [rgb * N | splat * N | weight * N] (film.rs:69, 85-90), zero-initialised. */
AKR_API int32_t akr_film_create(akr_context *ctx, uint32_t width, uint32_t height, akr_film **out);
AKR_API int32_t akr_film_destroy(akr_film *film);
AKR_API int32_t akr_film_clear(akr_film *film);                       /* Film::clear, film.rs:231-233 */
/* Copies the raw accumulator (7 * W * H floats, reference layout) to host memory. */
AKR_API int32_t akr_film_read(akr_film *film, float *dst);
/* Overwrites the raw accumulator from host memory (7 * W * H floats). */
AKR_API int32_t akr_film_write(akr_film *film, const float *src);
/* Film resolve = the copy_to_rgba_image kernel with hdr = true (film.rs:120-148): rgb / (w == 0 ? 1 : w)
 * + splat * splat_scale; writes 3 * W * H floats of linear RGB to host memory. */
AKR_API int32_t akr_film_resolve(akr_film *film, float *dst_rgb);
/* Film::set_splat_scale (film.rs:152-154): the factor of the splat channels in the resolve; 1 after akr_film_create
 * (akr_film_clear leaves it alone, as Film::clear does). The gpt integrator sets 1 / spp like the reference (gpt.rs:463-466). */
AKR_API int32_t akr_film_set_splat_scale(akr_film *film, float scale);
AKR_API int32_t akr_film_get_splat_scale(const akr_film *film, float *scale);
/* Wraps caller-owned device memory (7 * W * H floats, reference layout, e.g. a torch tensor that a
 * torch.distributed/RCCL reduce will run on) as a film; akr_film_destroy then leaves the memory alone. The memory must
 * be a plain device allocation (hipMalloc) on the context's GPU: host-mapped, managed or fine-grained memory is rejected
 * with AKR_ERR_INVALID_ARGUMENT, because the splatting integrators (gpt, mcmc_opt) use hardware float atomics, which
 * such memory silently drops. */
AKR_API int32_t akr_film_wrap(akr_context *ctx, uint32_t width, uint32_t height, void *device_ptr, akr_film **out);
/* Device pointer + byte size of the accumulator, for an RCCL reduce issued by the host application
 * (one process per GPU). The pointer stays valid until akr_film_destroy. */
AKR_API int32_t akr_film_device_ptr(akr_film *film, void **ptr, uint64_t *bytes);

/* ---- Multi-GPU: the one exchange step of the path (the reference has no counterpart) ----------------------
 * One process per GPU renders its pixel tiles (akr_pt_config.shard_rank / shard_count) into a full-frame film that is zero
 * elsewhere; akr_film_reduce sums the films over RCCL (xGMI) in place: onto rank `root`, or onto every rank with root = -1.
 * Disjoint tiles make the sum exact. The collective is enqueued on the context's stream, after the render that filled the
 * film; `blocking` != 0 waits for it. Bootstrap like NCCL: rank 0 calls akr_comm_unique_id and hands the 128 bytes to the
 * other ranks by whatever channel the host has (MPI, a file, torch.distributed ...), then every rank calls akr_comm_create.
 * A host that already has an ncclComm_t on the context's device passes it to akr_comm_wrap instead (not destroyed by
 * akr_comm_destroy). librccl.so is loaded on first use (an RCCL the process already holds -- e.g. the copy PyTorch ships -- is
 * reused: a host that also uses PyTorch imports it first); AKR_ERR_UNSUPPORTED if there is none. */
#define AKR_COMM_ID_BYTES 128
typedef struct akr_comm akr_comm;
AKR_API int32_t akr_comm_unique_id(uint8_t id[AKR_COMM_ID_BYTES]);
AKR_API int32_t akr_comm_create(akr_context *ctx, const uint8_t id[AKR_COMM_ID_BYTES], int32_t rank, int32_t world, akr_comm **out);
AKR_API int32_t akr_comm_wrap(akr_context *ctx, void *nccl_comm, int32_t rank, int32_t world, akr_comm **out);
AKR_API int32_t akr_comm_destroy(akr_comm *comm);
AKR_API int32_t akr_film_reduce(akr_film *film, akr_comm *comm, int32_t root, int32_t blocking);
/* The same for a subset of the accumulator's planes [rgb 3N | splat 3N | weight N]: a pt / aov film is rgb + weight only (its
 * splat plane stays zero), i.e. 4 N floats -- SURVEY.md 8(e)'s count -- as two collectives of one RCCL group; gpt / mcmc_opt films
 * need AKR_FILM_PLANES_ALL, which is what akr_film_reduce passes. Every rank must pass the same mask. */
enum { AKR_FILM_PLANE_RGB = 1, AKR_FILM_PLANE_SPLAT = 2, AKR_FILM_PLANE_WEIGHT = 4, AKR_FILM_PLANES_PT = 5, AKR_FILM_PLANES_ALL = 7 };
AKR_API int32_t akr_film_reduce_planes(akr_film *film, akr_comm *comm, int32_t root, int32_t blocking, uint32_t planes);
AKR_API int32_t akr_gpt_reduce(struct akr_gpt_session *se, akr_comm *comm, int32_t root, int32_t blocking);   /* see akr_gpt_begin */

/* Fills *cfg with pt::Config::default() (pt.rs:930-944), the default filter (film.rs:50-54) and sampler
 * (sampler/mod.rs:290-294). */
AKR_API int32_t akr_pt_config_default(akr_pt_config *cfg);
/* Parses a reference method file (RenderTask / RenderConfig JSON, akari_integrator/src/lib.rs:93-109;
 * example scenes/cbox/pt.json). Only "type":"pt" is accepted. `film_out` (may be NULL) receives film.out. */
AKR_API int32_t akr_pt_config_from_json(const char *json_text, akr_pt_config *cfg, char *film_out, uint32_t film_out_len);

/* Replaces pt::render / PathTracer::render (pt.rs:1056-1172): initialises the per-pixel PCG32 states from
 * cfg->sampler_seed (init_pcg32_buffer_with_seed, sampler/mod.rs:148-160), then runs
 * ceil(spp / spp_per_pass) passes, accumulating into `film`. Blocks until the film is complete on the device.
 * `stats` may be NULL. */
AKR_API int32_t akr_pt_render(akr_context *ctx, akr_scene *scene, const akr_pt_config *cfg, akr_film *film,
                              akr_pt_stats *stats);

/* The same render split at the reference's own pass granularity (the body of the `while cnt < spp` loop,
 * pt.rs:1126-1149), so a host can interleave its progress bar / intermediate saves / benchmark timing:
 *   begin  -> sampler state buffer created and seeded;
 *   passes -> runs up to n_passes further passes (fewer if spp is reached), asynchronously on the context
 *             stream unless `blocking` is non-zero; *spp_done receives the cumulative sample count. Passes of
 *             one call may share a kernel launch (up to 16; up to 64 once earlier launches of the session have completed and been timed):
 *             results are those of separate launches;
 *   end    -> waits, returns accumulated counters, frees the session. */
AKR_API int32_t akr_pt_begin(akr_context *ctx, akr_scene *scene, const akr_pt_config *cfg, akr_film *film,
                             akr_pt_session **out);
AKR_API int32_t akr_pt_passes(akr_pt_session *session, uint32_t n_passes, int32_t blocking, uint32_t *spp_done);
AKR_API int32_t akr_pt_end(akr_pt_session *session, akr_pt_stats *stats);
/* Waits for the queued passes and returns the counters accumulated so far (the session stays open). */
AKR_API int32_t akr_pt_get_stats(akr_pt_session *session, akr_pt_stats *stats);

/* Per-scene kernels. The reference JIT-compiles every scene's shader graphs into its kernel: graphs of the same shape share a
 * `shader_kind` (crates/akari_render/src/svm/compiler.rs:16-76) and the kernel switches over the kind into straight-line node code
 * (svm/eval.rs:428-467). This library's precompiled kernels interpret a pruned node list per textured hit; with option
 * "specialise" a pt session on a scene with texture-fed materials instead gets a kernel compiled for the scene at akr_pt_begin
 * (hiprtc; code objects cached in $AKR_KERNEL_CACHE, default ~/.cache/akari_hip, keyed by library sources + generated code +
 * kernel flags + compiler options). Films are bit-identical either way; when hiprtc is missing or the compile fails the session
 * silently uses the interpreter and `status` says why. struct_size = sizeof(akr_kernel_info), set by the caller. */
typedef struct {
    uint32_t struct_size;     /* in: the caller's sizeof(akr_kernel_info) */
    uint32_t specialised;     /* 1 = the session launches a per-scene kernel */
    uint32_t cache_hit;       /* 1 = no compile: the code object came from the disk cache or an earlier session of this process */
    uint32_t n_shader_kinds;  /* distinct graph shapes of the scene */
    uint32_t absent_mask;     /* lobes no material of the scene can have: 1 coat, 2 transmission, 4 normal map, 8 glass, 16 conductor */
    uint32_t min_waves;       /* waves per SIMD the kernel was compiled for */
    uint32_t vgprs, scratch_bytes;
    uint32_t kernel_flags;    /* the instantiation: bit 0 BVH intersector, 1 index-based sampler, 2 tables staged in LDS, 3 deferral, 4 relaxed arithmetic tier (option arith) */
    uint32_t _pad;
    double compile_ms;        /* hiprtc compile at akr_pt_begin (0 on a cache hit) */
    double load_ms;           /* cache lookup + module load */
    char status[256];         /* "ok", or why the session uses the interpreter */
} akr_kernel_info;
AKR_API int32_t akr_pt_kernel_info(akr_pt_session *session, akr_kernel_info *info);
/* The kernel text generated for the scene's shader kinds (what device/dtex.h includes in a per-scene kernel): length without the
 * terminator in *length; dst may be NULL to ask for the length only. Empty = no texture-fed material. Works on host-only scenes. */
AKR_API int32_t akr_scene_spec_source(akr_scene *scene, char *dst, uint64_t capacity, uint64_t *length);
/* Compiles the scene's per-scene kernel for `arch` ("gfx950" when NULL) without a device (hiprtc cross-compiles): flags bit 0 BVH,
 * 1 index-based sampler, 2 staged tables, 3 deferral. For tests and tools; sessions compile through the cache. */
AKR_API int32_t akr_host_spec_compile(akr_scene *scene, uint32_t flags, uint32_t min_waves, const char *arch, uint64_t *code_bytes, char *log, uint32_t log_len);
/* The same from the generated text, written to a file: what the library's helper process runs (akari-cli --spec-compile; sessions
 * compile their kernel there so that a host application's own copies of the ROCm compiler libraries cannot change the code). */
AKR_API int32_t akr_host_spec_compile_text(const char *spec_header, uint32_t flags, uint32_t min_waves, const char *arch, const char *out_path);
/* Copies the session's Pcg32 state buffer (2 x u64 per pixel: state, inc) to the host. */
AKR_API int32_t akr_pt_read_sampler_states(akr_pt_session *session, uint64_t *dst);

/* ---------------------------------------------------------------------------------------------------
 * Render driver = akari_integrator::render / render_single (akari_integrator/src/lib.rs:111-207) with its
 * RenderSession (lib.rs:8-23): runs every task of a method file and writes film.out; with save_intermediate it writes
 * "{name}-{spp}.exr" after every pass (pt.rs:1138-1147), with save_stats "{name}.json" = RenderStats (lib.rs:24-37,
 * pt.rs:1150-1155). `override_sampler_independent` != 0 replaces an unsupported sampler (pmj02bn: tables absent from
 * the reference tree) by {independent, same seed} instead of failing.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t save_intermediate;
    int32_t save_stats;
    const char *name;                      /* NULL = "default" */
    int32_t override_sampler_independent;
    int32_t verbose;                       /* log to stderr */
} akr_render_session;
AKR_API int32_t akr_render_task(akr_context *ctx, akr_scene *scene, const char *method_json_text, const akr_render_session *session,
                                akr_pt_stats *stats_of_last_task);
/* ---------------------------------------------------------------------------------------------------
 * `aov` integrator (Method::NormalVis, akari_integrator/src/aov.rs:57-173; "type": "aov" in a method file): per pixel
 * `spp` camera rays, one attribute of the first hit accumulated in the film like a radiance sample.
 * ------------------------------------------------------------------------------------------------- */
typedef enum {
    AKR_AOV_NS = 0,         /* "ns"        closure.ns(), remapped v * 0.5 + 0.5 when `remap`   aov.rs:103-113 */
    AKR_AOV_NG = 1,         /* "ng"        si.ng                                                aov.rs:114-118 */
    AKR_AOV_TANGENT = 2,    /* "tangent"   si.frame.t                                           aov.rs:119-123 */
    AKR_AOV_BITANGENT = 3,  /* "bitangent" si.frame.s                                           aov.rs:124-128 */
    AKR_AOV_ALBEDO = 4,     /* "albedo"    closure.albedo(wo) + closure.emission(wo)            aov.rs:129-143 */
    AKR_AOV_ROUGHNESS = 5   /* "roughness" closure.roughness(wo, sampler.next_1d())             aov.rs:145-158 */
} akr_aov_kind;
typedef struct {
    uint32_t spp;            /* aov::Config::default: 256 */
    uint32_t aov;            /* akr_aov_kind, default ns */
    uint32_t remap;          /* default 1 */
    uint32_t filter_type;    /* film.filter, as in akr_pt_config */
    float filter_radius;
    uint32_t sampler_type;
    uint64_t sampler_seed;
    uint32_t shard_rank, shard_count, tile_w, tile_h;
    uint32_t color, _pad;    /* ColorPipeline bits as in akr_pt_config.color (aov.rs:62: the values go through the film like colours) */
} akr_aov_config;
AKR_API int32_t akr_aov_config_default(akr_aov_config *cfg);
/* Renders into `film` (accumulates: clear it first for a fresh image). stats: n_samples = n_closest = camera rays. */
AKR_API int32_t akr_aov_render(akr_context *ctx, akr_scene *scene, const akr_aov_config *cfg, akr_film *film, akr_pt_stats *stats);

/* ---------------------------------------------------------------------------------------------------
 * `gpt` integrator (Method::GradientPathTracer, akari_integrator/src/gpt.rs; "type": "gpt"): gradient-domain path tracing.
 * Per sample one base path and four offset paths through the neighbouring pixels (stride apart, mirrored at the border)
 * on the same random numbers; the offset paths rejoin the base path through the reconnection shift mapping of
 * run_pt_hybrid_shift_mapping (pt.rs:329-900: min_dist 0.03, min_roughness 0.2). reconstruction = none: the five paths are
 * MIS-combined into the film's splat channels (resolve scale 1 / spp); uniform / weighted: primal + gradient images are
 * accumulated and `reconstruction_iter` Jacobi sweeps of the screened Poisson problem write the film (gpt.rs:495-606).
 * The reference lets float atomics order a pixel's splats; here the order is fixed (own terms, then neighbours 0..3).
 * Independent sampler only (the reference's Pmj02BnSampler::clone_box is todo!()). akr_gpt_render = one GPU per frame; the session
 * calls below (akr_gpt_begin with an akr_shard, akr_gpt_reduce) shard a frame over several.
 * ------------------------------------------------------------------------------------------------- */
typedef enum { AKR_GPT_RECON_NONE = 0, AKR_GPT_RECON_UNIFORM = 1, AKR_GPT_RECON_WEIGHTED = 2 } akr_gpt_reconstruction;
typedef struct {
    uint32_t spp, max_depth, rr_depth, spp_per_pass;        /* gpt::Config::default: 256, 7, 5, 64  (gpt.rs:48-65) */
    uint32_t use_nee, indirect_only, reconnect, stride;     /* 1, 0, 1, 1 */
    uint32_t separate_weights, reconstruction, reconstruction_iter, filter_type; /* 0, none, 30; film.filter */
    float filter_radius;
    uint32_t sampler_type;
    uint64_t sampler_seed;
    uint64_t seed;                                          /* gpt::Config.seed: carried, never read by the reference either */
    uint32_t color, _pad;                                   /* ColorPipeline bits as in akr_pt_config.color (gpt.rs:96) */
} akr_gpt_config;
AKR_API int32_t akr_gpt_config_default(akr_gpt_config *cfg);
/* Renders into `film` (clear it first; sets its splat scale). aux (host memory, optional, reconstruction != none):
 * [primal 3 N | Gx 3 (W+1)(H+1) | Gy 3 (W+1)(H+1)] floats, the sums the reference writes / spp to output/gpt_*.exr. */
AKR_API int32_t akr_gpt_render(akr_context *ctx, akr_scene *scene, const akr_gpt_config *cfg, akr_film *film, float *aux, akr_pt_stats *stats);
/* The same render in steps, so that the GPUs of a node can share it (no reference counterpart; gpt.rs:381-640 is one device):
 *   akr_gpt_begin(.., shard, ..)   shard = NULL: the whole frame. Otherwise rank r of n folds the pixels of its tiles (the rule of
 *                                  akr_pt_config.shard_*) and samples those pixels plus their halo: the pixels one of whose four
 *                                  offset paths (`stride` away, mirrored at the border, gpt.rs:118-142) lands in an owned tile --
 *                                  with a reconstruction, the left and upper neighbours whose gradients the update reads. Every
 *                                  rank keeps the whole frame's sampler states; a halo pixel draws the same numbers everywhere.
 *   akr_gpt_sample(se, n, block)   n more samples per pixel (0 = all that are left of cfg.spp)
 *   akr_gpt_reduce(se, comm, root) the exchange: reconstruction none -> akr_film_reduce of the film (splat channels of disjoint
 *                                  tiles); otherwise ncclReduce of the primal / gradient sums [6 N + 12 (W+1)(H+1) floats].
 *                                  akr_gpt_sums / _read / _write expose the sums to hosts with their own collective.
 *   akr_gpt_finish(se, aux, stats) splat scale or the reconstruction sweeps (gpt.rs:495-606) on the (reduced) sums; frees se.
 * The summed result is the one-GPU result bit for bit: every film / sum entry has exactly one rank that writes it. */
typedef struct { uint32_t shard_rank, shard_count, tile_w, tile_h; } akr_shard;   /* tile sizes: multiples of 8; 0 = 32 */
typedef struct akr_gpt_session akr_gpt_session;
AKR_API int32_t akr_gpt_begin(akr_context *ctx, akr_scene *scene, const akr_gpt_config *cfg, const akr_shard *shard, akr_film *film, akr_gpt_session **out);
AKR_API int32_t akr_gpt_sample(akr_gpt_session *se, uint32_t n_samples, int32_t blocking);
AKR_API int32_t akr_gpt_sums(akr_gpt_session *se, float **device_ptr, uint64_t *n_floats);
AKR_API int32_t akr_gpt_sums_read(akr_gpt_session *se, float *dst);
AKR_API int32_t akr_gpt_sums_write(akr_gpt_session *se, const float *src);
AKR_API int32_t akr_gpt_finish(akr_gpt_session *se, float *aux, akr_pt_stats *stats);
/* Frees the session without akr_gpt_finish's reconstruction: for the ranks that are not the root of akr_gpt_reduce (only the root
 * holds the whole frame's sums) and for abandoning a render. The film keeps what the samples / the reduce left in it. */
AKR_API int32_t akr_gpt_abort(akr_gpt_session *se, akr_pt_stats *stats);

/* ---------------------------------------------------------------------------------------------------
 * `mcmc_opt` integrator (Method::McmcOpt, akari_integrator/src/mcmc_opt.rs + mcmc.rs:8-80; "type": "mcmc_opt"): primary-sample-
 * space Metropolis light transport. An optional direct-illumination pass of the path tracer (direct_spp > 0: max_depth 1
 * into the film's rgb / weight channels; the chains then render indirect light only), n_bootstrap independent paths to
 * estimate the normalisation and seed n_chains Markov chains, then W * H * spp mutations spread over the chains (Kelemen
 * large / small steps on lazily mutated sample vectors of 5 + 7 (1 + mcmc_depth) dimensions), every mutation splatting the
 * proposed and the current path with their acceptance weights; resolve scale b / spp.
 * Splats are float atomics as in the reference: each chain is reproducible bit for bit, the summation order per pixel is not.
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t spp, max_depth, rr_depth, spp_per_pass;        /* mcmc::Config::default = pt defaults: 256, 7, 5, 64 (mcmc.rs:60-79) */
    uint32_t use_nee;                                       /* 1 */
    uint32_t mcmc_depth;                                    /* 0xffffffff = None = max_depth */
    uint32_t n_chains, n_bootstrap;                         /* 512, 100000 (the reference's defaults). One lane per chain: 512 chains are 8 waves
                                                             * on a chip with 1024 SIMDs (2 M mutations/s); a GPU wants n_chains >= 1e5 (262144: 380 M/s) */
    int32_t direct_spp;                                     /* 64; 0 = no direct pass but indirect-only chains; < 0 = chains render everything */
    uint32_t exponential_mutation;                          /* Method::Kelemen (mcmc.rs:10-32): 1 */
    float small_sigma, large_step_prob, image_mutation_prob;/* 0.01, 0.1, 0 */
    float image_mutation_size;                              /* <= 0 = None */
    uint32_t adaptive, wis;                                 /* carried, unused by mcmc_opt.rs */
    uint64_t seed;                                          /* 0 */
    uint32_t filter_type;
    float filter_radius;
    uint32_t sampler_type, color;                           /* sampler of the direct pass; ColorPipeline bits as in akr_pt_config.color */
    uint64_t sampler_seed;
} akr_mcmc_config;
typedef struct {
    double normalization;      /* b = (sum of bootstrap + large-step contributions) / their count */
    double acceptance_rate;    /* accepted / proposed small steps */
    float splat_scale;         /* (float)b / (float)spp, also set on the film */
    float contribution;        /* weight of one mutation */
    uint64_t n_mutations;      /* mutations executed, all chains */
    uint32_t sample_dimension, _pad;
} akr_mcmc_result;
/* What the normalisation (reconstruct, mcmc_opt.rs:587-611) needs from one rank of a sharded render. */
typedef struct {
    double bootstrap_sum;      /* sum of the n_bootstrap bootstrap contributions: the same on every rank */
    double b_sum;              /* this rank's chains: sum of their large-step contributions (MarkovState.b) */
    uint64_t n_bootstrap;
    uint64_t b_cnt;            /* ... and their count */
    uint64_t n_accepted, n_mutations;   /* small steps accepted / proposed by this rank's chains */
    uint64_t n_executed;       /* mutations this rank executed (small and large) */
    uint32_t spp;              /* samples per pixel of the render (cfg.spp) */
    float contribution;        /* weight of one mutation (from the global chain count) */
} akr_mcmc_partial;
AKR_API int32_t akr_mcmc_config_default(akr_mcmc_config *cfg);
/* Renders into `film` (clear it first; sets its splat scale). chain_states (host memory, optional): n_chains records of 10
 * u32 = MarkovState {cur_pixel[2], chain_id, cur_f, b, b_cnt, n_accepted, n_mutations, cur_iter, last_large_iter}. */
AKR_API int32_t akr_mcmc_render(akr_context *ctx, akr_scene *scene, const akr_mcmc_config *cfg, akr_film *film, akr_mcmc_result *result,
                                uint32_t *chain_states, akr_pt_stats *stats);
/* mcmc_opt over several GPUs (no reference counterpart): the chains are independent, so rank r of n runs chains
 * [r c / n, (r + 1) c / n) of the c = cfg.n_chains. Everything that defines a chain -- the bootstrap path it starts from (every rank runs
 * the same bootstrap and resampling), its sampler stream, the mutations per chain and the weight of a mutation -- comes from the chain's
 * GLOBAL index and the global count: the union of the ranks' chain sets is the one-GPU chain set, chain for chain (chain_states:
 * n_chains records, only this rank's are filled, the others zero). The direct-lighting pass (direct_spp > 0) renders the rank's pixel
 * tiles. Afterwards the films are summed (all planes) and the normalisation is made from every rank's sums:
 *   akr_mcmc_combine(film, comm, root, &mine, &result)   akr_film_reduce + an all-reduce of the sums over RCCL; sets the film's splat scale
 *   akr_mcmc_combine_host(film, partials, n, &result)    the same arithmetic from partial sums the caller gathered itself */
AKR_API int32_t akr_mcmc_render_shard(akr_context *ctx, akr_scene *scene, const akr_mcmc_config *cfg, uint32_t shard_rank, uint32_t shard_count,
                                      akr_film *film, akr_mcmc_partial *partial, uint32_t *chain_states, akr_pt_stats *stats);
AKR_API int32_t akr_mcmc_combine_host(akr_film *film, const akr_mcmc_partial *partials, uint32_t n, akr_mcmc_result *result);
AKR_API int32_t akr_mcmc_combine(akr_film *film, akr_comm *comm, int32_t root, const akr_mcmc_partial *mine, akr_mcmc_result *result);

/* util::write_image (akari_render/src/util/mod.rs:57-127): ".exr" -> linear RGB f32 OpenEXR (uncompressed scanlines),
 * ".png" -> 8-bit sRGB. rgb = 3 * W * H floats, row-major, top row first. Creates parent directories. */
AKR_API int32_t akr_image_write(const char *path, const float *rgb, uint32_t width, uint32_t height);



/* Library / build identification: "akari_hip <version> gfx950". */
AKR_API const char *akr_version(void);
/* sizeof() of the structs of this header as the LIBRARY was built, so that a caller (or a binding generated from another version of the
 * header) can check its own before passing one: 0 = unknown id. The bindings in akari_render_amd/capi.py check every one at load. */
typedef enum {
    AKR_STRUCT_MESH_DESC = 1, AKR_STRUCT_INSTANCE_DESC, AKR_STRUCT_MATERIAL_DESC, AKR_STRUCT_CAMERA_DESC, AKR_STRUCT_SCENE_DESC,
    AKR_STRUCT_PT_CONFIG, AKR_STRUCT_PT_STATS, AKR_STRUCT_SCENE_INFO, AKR_STRUCT_KERNEL_INFO, AKR_STRUCT_AOV_CONFIG, AKR_STRUCT_GPT_CONFIG,
    AKR_STRUCT_MCMC_CONFIG, AKR_STRUCT_MCMC_RESULT, AKR_STRUCT_MCMC_PARTIAL
} akr_struct_id;
AKR_API uint32_t akr_struct_size(int32_t which);
/* Process-wide tuning switches and test hooks (no reference counterpart). Each starts from its environment variable, read once;
 * afterwards only akr_option_set changes it, and it applies to scenes / sessions created after the call:
 *   "force_bvh"    (AKR_FORCE_BVH=1)        scenes of <= 64 triangles get a BVH as well
 *   "bvh_balanced" (AKR_BVH_BALANCED=1)     median-split fallback builder instead of binned SAH
 *   "defer_metal"  (AKR_PT_DEFER_METAL=m)   -1 the library decides; 0 off; m > 0: conductor hits shaded when (iteration & m) == 0
 *   "wavefront"    (AKR_PT_MODE=wavefront|megakernel|auto)  pt sessions on scenes with a tree: 1 = the wavefront schedule, 0 = the megakernel,
 *                                           -1 (default) = the library decides (wavefront for sessions of >= 0.5 M ... 2 M pixels, by mesh size, on scenes kept
 *                                           as meshes + instances, else the megakernel). Films are the same bit for bit either way.
 *   "wf_groups"    (AKR_WF_GROUPS=g)        wavefront schedule: the path slots run as g groups with queues and streams of their own; 0 = the
 *                                           library decides
 *   "wf_carry"     (AKR_WF_CARRY=0)         wavefront schedule: 1 (default) = the last rays of a trace launch -- the few lanes a wave has left once the
 *                                           queue is empty -- are carried into the next launch instead of being waited for; 0 = every launch
 *                                           traces all its rays to the end. Films are the same bit for bit either way.
 *   "sched_trial"  (AKR_SCHED_TRIAL=v)      flattened scenes under "wavefront" = -1: -1 (default) = a long render (>= 16 passes) of a large frame (>= 1 M pixels) of
 *                                           a large untextured scene (>= 256 MB on the device) starts with two passes under each schedule and goes on with the
 *                                           faster one (skipped for closed scenes, where the megakernel always measured faster); 0 = never; 1 = every pt
 *                                           session on a scene with a tree (tests). akr_pt_kernel_info's status names the outcome. Same film either way.
 *   "simple_kernels" (AKR_PT_SIMPLE=0)      0 = never pick the kernels specialised for scenes without coat / transmission / normal map / glass
 *   "defer_on"     (no environment hook)    BVH kernels of scenes with textures: which hits "defer_metal" puts off -- 0 / 1 the conductor
 *                                           lobe (default), 2 texture-fed materials, 3 both
 *   "specialise"   (AKR_SPECIALISE=v)       per-scene kernels for pt sessions on scenes with texture-fed materials (below): -1 the library
 *                                           decides (a cached kernel always, a compile for renders of >= 2^31 samples), 0 never, 1 always
 *   "specialise_waves" (AKR_SPECIALISE_WAVES=n)  waves per SIMD a per-scene kernel is compiled for: 0 the library's choice, else 2..4
 *   "max_fused_passes" (no environment hook)     most passes one launch of akr_pt_passes fuses: 0 adaptive, else 1..64
 *   "wf_sort"      (AKR_WF_SORT=1)          wavefront schedule: the ray queues are sorted by (Morton code of the origin, octant of the
 *                                           direction) before every trace launch (films unchanged; measurement in DESIGN.md)
 *   "instancing"   (AKR_INSTANCING=v)       scenes in which a mesh has several instances: -1 the library decides (kept as meshes +
 *                                           instances -- a tree over the instances and one per mesh in object space, one BIT stored per
 *                                           instance-triangle -- when the flattened records would pass 8 GB or 48 M triangles), 0 always
 *                                           flattened, 1 kept as meshes + instances whenever a mesh is shared. Films are the same bit for
 *                                           bit either way, for every integrator and schedule. A singular instance transform means
 *                                           flattening whatever the option says.
 *   "rebraid"      (AKR_REBRAID=k)          scenes kept as meshes + instances: the tree over the instances is built over k x as many (instance,
 *                                           subtree of its mesh's tree) pairs, the largest boxes opened first; 1 (default) = one pair per instance
 *   "arith"        (AKR_ARITH=1)            pt megakernel of flattened scenes in the relaxed arithmetic tier (hardware rcp / sqrt / sin / cos /
 *                                           log / exp, contraction): faster, NOT bit-identical to the reference arithmetic
 *   "pad_percent"  (no environment hook)    test hook: padding of the acceleration structures' boxes in percent of the derived value
 * Values out of an option's range fail with AKR_ERR_INVALID_ARGUMENT.
 * A session reads the options once, when it begins (akr_pt_begin / akr_gpt_begin / ...): a later akr_option_set does not change it.
 * "wavefront" = 1 on a scene without a BVH renders with the megakernel. Unknown names fail with AKR_ERR_INVALID_ARGUMENT. */
AKR_API int32_t akr_option_set(const char *name, int32_t value);
AKR_API int32_t akr_option_get(const char *name, int32_t *value);


#ifdef __cplusplus
}
#endif
#endif /* AKARI_HIP_H */
