/* or_api.h -- input/output structs of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The scene/material/config structs are layout-identical to the product's public C ABI
 * (include/akari_hip.h: akr_mesh_desc, akr_instance_desc, akr_material_desc, akr_camera_desc,
 * akr_scene_desc, akr_pt_config) so a test can hand the very same ctypes objects to both sides.
 * The oracle deliberately re-declares them instead of including the product header: the two
 * implementations share no code.
 */
#ifndef OR_API_H
#define OR_API_H
#include <stdint.h>

typedef struct {
    uint32_t n_vertices, n_triangles;
    const float *vertices;          /* 3 * n_vertices, object space (mesh.rs:14-25) */
    const uint32_t *indices;        /* 3 * n_triangles */
    const float *uvs;               /* 2 * 3 * n_triangles (per corner) or NULL */
    const float *normals;           /* 3 * 3 * n_triangles (per corner) or NULL */
    const float *tangents;          /* 3 * 3 * n_triangles (per corner) or NULL */
    const uint32_t *material_slots; /* n_triangles or NULL (= slot 0) */
} or_mesh_desc;

typedef struct {
    uint32_t mesh;           /* index into meshes */
    uint32_t n_materials;
    const uint32_t *materials; /* indices into materials, one per slot */
    float transform[16];     /* column-major object->world (glam Mat4 / AffineTransform.m) */
} or_instance_desc;

enum { OR_MAT_PRINCIPLED = 0, OR_MAT_DIFFUSE = 1, OR_MAT_GLASS = 2, OR_MAT_EMISSION = 3, OR_MAT_KIND_MASK = 0xff,
       /* OR-ed into kind: this constant colour input is given in ACEScg (akari_scenegraph ColorSpace "aces") instead of sRGB */
       OR_MAT_CS_BASE_COLOR = 0x100, OR_MAT_CS_SPECULAR_TINT = 0x200, OR_MAT_CS_COAT_TINT = 0x400, OR_MAT_CS_EMISSION_COLOR = 0x800 };
/* ColorPipeline (color.rs:663-676) as bits of or_pt_config.color: color_repr = Rgb(ACEScg), rgb_colorspace = ACEScg */
enum { OR_COLOR_REPR_ACES = 1, OR_COLOR_RGB_ACES = 2 };
typedef struct {
    uint32_t kind;
    float base_color[3];     /* principled base_color / diffuse color / glass color (linear, target RGB space) */
    float base_alpha;        /* alpha of the base-colour node (1 for constants, svm/eval.rs:125-135) */
    float metallic, roughness, ior, specular_ior_level;
    float specular_tint[3];
    float transmission_weight;
    float coat_weight, coat_roughness, coat_ior;
    float coat_tint[3];
    float emission_color[3];
    float emission_strength;
    float normal[3];         /* principled `normal` socket (0,0,0 = unperturbed) */
} or_material_desc;

/* Shader graphs with texture-fed inputs and the images they sample: layout-identical to akr_shader_node,
 * akr_material_graph, akr_image_desc (include/akari_hip.h). Node semantics: svm/eval.rs:97-269. */
#define OR_NODE_NONE 0xffffffffu
enum { OR_NODE_CONST = 0, OR_NODE_RGB, OR_NODE_TEXCOORDS, OR_NODE_IMAGE, OR_NODE_MAPPING, OR_NODE_CHECKERBOARD,
       OR_NODE_SPECTRAL_UPLIFT, OR_NODE_SEPARATE_COLOR, OR_NODE_EXTRACT, OR_NODE_NORMAL_MAP };
enum { OR_IN_BASE_COLOR = 0, OR_IN_METALLIC, OR_IN_ROUGHNESS, OR_IN_IOR, OR_IN_SPECULAR_IOR_LEVEL, OR_IN_SPECULAR_TINT,
       OR_IN_TRANSMISSION_WEIGHT, OR_IN_COAT_WEIGHT, OR_IN_COAT_ROUGHNESS, OR_IN_COAT_IOR, OR_IN_COAT_TINT,
       OR_IN_EMISSION_COLOR, OR_IN_EMISSION_STRENGTH, OR_IN_NORMAL, OR_IN_COUNT };
typedef struct { uint32_t op; uint32_t arg[4]; float k[3]; } or_shader_node;
typedef struct { uint32_t n_nodes, _pad; const or_shader_node *nodes; uint32_t input[OR_IN_COUNT]; } or_material_graph;
enum { OR_IMAGE_RGBA8 = 0, OR_IMAGE_RGBA32F = 1 };
enum { OR_TEX_NEAREST = 0, OR_TEX_LINEAR = 1 };
enum { OR_TEX_REPEAT = 0, OR_TEX_CLIP = 1, OR_TEX_MIRROR = 2, OR_TEX_EXTEND = 3 };
typedef struct { uint32_t width, height, format, filter, address, _pad; const void *texels; } or_image_desc;

typedef struct {
    float c2w[16];           /* column-major camera->world (load.rs:129-171 applied to the camera TRS) */
    float fov;               /* radians, spans the larger image side (camera/mod.rs:135-140) */
    uint32_t width, height;
} or_camera_desc;

typedef struct {
    uint32_t n_meshes, n_instances, n_materials, _pad;
    const or_mesh_desc *meshes;
    const or_instance_desc *instances;
    const or_material_desc *materials;
    or_camera_desc camera;
    const float *ggx_dielectric_table; /* 16^3 f32 ("ggx_dielectric_s", precompute.rs:133-145) or NULL */
    uint32_t n_images, _pad2;
    const or_image_desc *images;
    const or_material_graph *material_graphs; /* n_materials entries or NULL */
} or_scene_desc;

enum { OR_FILTER_BOX = 0, OR_FILTER_GAUSSIAN = 1 };
enum { OR_SAMPLER_INDEPENDENT = 0 };
typedef struct {
    /* pt::Config, pt.rs:916-944 */
    uint32_t spp, max_depth, spp_per_pass, rr_depth;
    uint32_t use_nee, indirect_only, force_diffuse;
    int32_t pixel_offset[2];
    int32_t debug_depth;     /* -1 = None */
    /* film.filter (film.rs:22-54) and sampler (sampler/mod.rs:282-295) of RenderConfig */
    uint32_t filter_type;
    float filter_radius;
    uint32_t sampler_type;
    uint32_t color;          /* OR_COLOR_* bits; 0 = the default sRGB / sRGB pipeline */
    uint64_t sampler_seed;
    /* pixel-tile sharding (rank r of n renders tiles t with t % n == r); n = 1 renders everything */
    uint32_t shard_rank, shard_count, tile_w, tile_h;
    /* samples [sample_begin, sample_begin + sample_count) of the spp of the whole render; 0 count = all (index-based samplers only) */
    uint32_t sample_begin, sample_count;
} or_pt_config;

typedef struct {
    uint64_t n_samples, n_closest, n_shadow, n_shaded, n_tri_tests;
} or_stats;
#endif
