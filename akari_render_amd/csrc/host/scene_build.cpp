// scene_build.cpp -- host-side scene compiler: FlatScene -> CompiledScene.
//
// Mirrors what the reference does between "buffers resolved" and "kernel can run":
//   MeshAggregate::new            crates/akari_render/src/mesh.rs:259-348   (instance table, transform_det)
//   surface_interaction, per-triangle part   mesh.rs:499-653                (folded here, once per triangle)
//   svm compile + constant eval   svm/compiler.rs:116-337, svm/eval.rs:97-269, svm/surface/principled.rs:23-131
//   light discovery               load.rs:308-444  (power estimate, alias tables, light list)
//   AliasTable::new               util/distribution.rs:35-78
// All f32 arithmetic here uses the same akr:: helpers as the kernels (this file is compiled with
// -ffp-contract=off too), so a quantity folded on the host has the bits the device would have computed.
#include "scene_build.h"
#include "../device/dinst.h"
#include "host_parallel.h"

#include <chrono>
#include <cstdio>
#include <mutex>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <stdexcept>

#include "../device/dgeom.h"

namespace akr {

FlatScene FlatScene::from_desc(const akr_scene_desc& d) {
    FlatScene s;
    if ((d.n_meshes && !d.meshes) || (d.n_instances && !d.instances) || (d.n_materials && !d.materials))
        throw std::invalid_argument("akr_scene_desc: null array with non-zero count");
    s.meshes.resize(d.n_meshes);
    for (uint32_t i = 0; i < d.n_meshes; i++) {
        const akr_mesh_desc& m = d.meshes[i];
        if (!m.vertices || !m.indices || m.n_triangles == 0) throw std::invalid_argument("akr_mesh_desc: empty mesh");
        HostMesh& h = s.meshes[i];
        h.vertices.assign(m.vertices, m.vertices + 3ull * m.n_vertices);
        h.indices.assign(m.indices, m.indices + 3ull * m.n_triangles);
        for (uint32_t idx : h.indices)
            if (idx >= m.n_vertices) throw std::invalid_argument("akr_mesh_desc: vertex index out of range");
        if (m.uvs) h.uvs.assign(m.uvs, m.uvs + 6ull * m.n_triangles);
        if (m.normals) h.normals.assign(m.normals, m.normals + 9ull * m.n_triangles);
        if (m.tangents) h.tangents.assign(m.tangents, m.tangents + 9ull * m.n_triangles);
        if (m.material_slots) h.slots.assign(m.material_slots, m.material_slots + m.n_triangles);
    }
    s.materials.assign(d.materials, d.materials + d.n_materials);
    s.instances.resize(d.n_instances);
    for (uint32_t i = 0; i < d.n_instances; i++) {
        const akr_instance_desc& in = d.instances[i];
        if (in.mesh >= d.n_meshes) throw std::invalid_argument("akr_instance_desc: mesh index out of range");
        if (in.n_materials == 0 || !in.materials) throw std::invalid_argument("akr_instance_desc: no materials");
        HostInstance& h = s.instances[i];
        h.mesh = in.mesh;
        h.materials.assign(in.materials, in.materials + in.n_materials);
        for (uint32_t m : h.materials)
            if (m >= d.n_materials) throw std::invalid_argument("akr_instance_desc: material index out of range");
        std::memcpy(h.transform, in.transform, sizeof h.transform);
    }
    s.camera = d.camera;
    if (d.camera.width == 0 || d.camera.height == 0) throw std::invalid_argument("akr_camera_desc: zero resolution");
    if (d.ggx_dielectric_table) s.ggx_table.assign(d.ggx_dielectric_table, d.ggx_dielectric_table + 4096);
    if (d.n_images && !d.images) throw std::invalid_argument("akr_scene_desc: null image array with non-zero count");
    s.images.resize(d.n_images);
    for (uint32_t i = 0; i < d.n_images; i++) {
        const akr_image_desc& im = d.images[i];
        if (im.width == 0 || im.height == 0 || !im.texels) throw std::invalid_argument("akr_image_desc: empty image");
        if (im.format > AKR_IMAGE_RGBA32F || im.filter > AKR_TEX_FILTER_LINEAR || im.address > AKR_TEX_EXTEND)
            throw std::invalid_argument("akr_image_desc: unknown format / filter / address mode");
        if ((uint64_t)im.width * im.height > (1ull << 30)) throw std::invalid_argument("akr_image_desc: image too large");
        HostImage& h = s.images[i];
        h.width = im.width; h.height = im.height; h.format = im.format; h.filter = im.filter; h.address = im.address;
        size_t words = (size_t)im.width * im.height * (im.format == AKR_IMAGE_RGBA32F ? 4 : 1);
        h.words.resize(words);
        std::memcpy(h.words.data(), im.texels, words * 4);
    }
    if (d.material_graphs) {
        s.graphs.resize(d.n_materials);
        for (uint32_t i = 0; i < d.n_materials; i++) {
            const akr_material_graph& g = d.material_graphs[i];
            if (g.n_nodes && !g.nodes) throw std::invalid_argument("akr_material_graph: null node array with non-zero count");
            s.graphs[i].nodes.assign(g.nodes, g.nodes + g.n_nodes);
            std::memcpy(s.graphs[i].input, g.input, sizeof g.input);
        }
    }
    return s;
}

// ------------------------------------------------------------------------------------------------ materials
static vec3 v3(const float* p) { return mk3(p[0], p[1], p[2]); }

static_assert(sizeof(MatInputs) == sizeof(akr_material_desc), "MatInputs mirrors akr_material_desc");
DMaterial fold_material(const akr_material_desc& m, uint32_t color) {
    DMaterial d;
    std::memset(&d, 0, sizeof d);
    MatInputs in;
    std::memcpy(&in, &m, sizeof in);
    const bool fed[4] = {false, false, false, false};
    convert_color_inputs(in, m.kind, color, fed);  // Rgb node space -> rgb_colorspace -> repr space (texture/mod.rs:9-43)
    in.kind = m.kind & MAT_KIND_MASK;
    if (!fold_inputs(in, d)) throw std::invalid_argument("akr_material_desc: unknown kind");
    for (uint32_t i = 0; i < 14; i++) d.tex_input[i] = 0xffffffffu;
    return d;
}

// ------------------------------------------------------------------------------------------------ shader graphs
static_assert(sizeof(DNode) == sizeof(akr_shader_node), "DNode mirrors akr_shader_node");
static uint32_t node_n_args(uint32_t op) {
    switch (op) {
        case AKR_NODE_CONST: case AKR_NODE_RGB: case AKR_NODE_TEXCOORDS: return 0;
        case AKR_NODE_IMAGE: return 2;  // arg0 is an image id, arg1 the optional uv node
        case AKR_NODE_MAPPING: return 3;
        case AKR_NODE_CHECKERBOARD: return 4;
        case AKR_NODE_SPECTRAL_UPLIFT: case AKR_NODE_SEPARATE_COLOR: case AKR_NODE_EXTRACT: return 1;
        case AKR_NODE_NORMAL_MAP: return 2;
        default: throw std::invalid_argument("akr_shader_node: unknown op");
    }
}
// node argument slots that refer to other nodes (the rest are immediates)
static void node_refs(const akr_shader_node& n, uint32_t out[4], uint32_t& count) {
    count = 0;
    uint32_t na = node_n_args(n.op);
    for (uint32_t a = 0; a < na; a++) {
        if (n.op == AKR_NODE_IMAGE && a == 0) continue;
        if (n.arg[a] != AKR_NODE_NONE) out[count++] = n.arg[a];
    }
}
// Splits a material's graph into the inputs that are constant after all (folded into `desc`, the way the reference
// would evaluate them to the same value at every point) and the texture-fed rest (pruned node list appended to
// `out.tex_nodes`, input map into `dm`).
// Is the alpha (w) of this node's value 1 at every point? Only then may the any-hit alpha test skip the graph: constant
// colours carry alpha 1 (svm/eval.rs:123-133), an image carries its alpha channel (bilinear filtering of all-ones is
// exactly one; Zero addressing returns 0 outside), checkerboards pick one of their colours, uplift / separate pass it on.
static bool alpha_is_one(const HostGraph& g, const std::vector<HostImage>& images, uint32_t node) {
    if (node == AKR_NODE_NONE) return false;
    const akr_shader_node& n = g.nodes[node];
    switch (n.op) {
        case AKR_NODE_RGB: return true;
        case AKR_NODE_IMAGE: {
            const HostImage& im = images[n.arg[0]];
            if (im.address == AKR_TEX_CLIP) return false;
            const size_t nt = (size_t)im.width * im.height;
            if (im.format == AKR_IMAGE_RGBA8) {
                for (size_t t = 0; t < nt; t++)
                    if ((im.words[t] >> 24) != 0xffu) return false;
            } else {
                for (size_t t = 0; t < nt; t++)
                    if (im.words[4 * t + 3] != 0x3f800000u) return false;
            }
            return true;
        }
        case AKR_NODE_SPECTRAL_UPLIFT: case AKR_NODE_SEPARATE_COLOR: return alpha_is_one(g, images, n.arg[0]);
        case AKR_NODE_CHECKERBOARD: return alpha_is_one(g, images, n.arg[2]) && alpha_is_one(g, images, n.arg[3]);
        default: return false;  // float / float3 / texcoords / mapping / extract / normal_map values have w = 0
    }
}
static void compile_graph(const HostGraph& g, const std::vector<HostImage>& images, uint32_t color, uint32_t material_index, akr_material_desc& desc, DMaterial& dm,
                          CompiledScene& out) {
    const uint32_t n_images = (uint32_t)images.size();
    uint32_t shader_kind = 0;
    const uint32_t n = (uint32_t)g.nodes.size();
    std::vector<uint8_t> varying(n, 0);
    for (uint32_t i = 0; i < n; i++) {
        const akr_shader_node& nd = g.nodes[i];
        uint32_t refs[4], nr;
        node_refs(nd, refs, nr);
        for (uint32_t r = 0; r < nr; r++)
            if (refs[r] >= i) throw std::invalid_argument("akr_shader_node: arguments must refer to earlier nodes");
        if (nd.op == AKR_NODE_IMAGE && nd.arg[0] >= n_images) throw std::invalid_argument("akr_shader_node: image index out of range");
        if (nd.op == AKR_NODE_MAPPING && (nd.arg[0] == AKR_NODE_NONE || nd.arg[1] == AKR_NODE_NONE || nd.arg[2] == AKR_NODE_NONE || nd.arg[3] > 1))
            throw std::invalid_argument("akr_shader_node: mapping needs vector, location, scale and a mapping type");
        if (nd.op == AKR_NODE_CHECKERBOARD && (nd.arg[1] == AKR_NODE_NONE || nd.arg[2] == AKR_NODE_NONE || nd.arg[3] == AKR_NODE_NONE))
            throw std::invalid_argument("akr_shader_node: checkerboard needs scale, color1, color2");
        if ((nd.op == AKR_NODE_SPECTRAL_UPLIFT || nd.op == AKR_NODE_SEPARATE_COLOR || nd.op == AKR_NODE_EXTRACT) && nd.arg[0] == AKR_NODE_NONE)
            throw std::invalid_argument("akr_shader_node: missing argument");
        if (nd.op == AKR_NODE_EXTRACT && nd.arg[1] > AKR_FIELD_UV) throw std::invalid_argument("akr_shader_node: unknown extract field");
        if (nd.op == AKR_NODE_NORMAL_MAP && (nd.arg[0] == AKR_NODE_NONE || nd.arg[1] == AKR_NODE_NONE))
            throw std::invalid_argument("akr_shader_node: normal_map needs normal and strength");
        bool v = nd.op == AKR_NODE_TEXCOORDS || nd.op == AKR_NODE_IMAGE || (nd.op == AKR_NODE_CHECKERBOARD && nd.arg[0] == AKR_NODE_NONE);
        for (uint32_t r = 0; r < nr; r++) v = v || varying[refs[r]];
        varying[i] = v;
    }
    // constant inputs: evaluate the whole list once (the varying nodes see uv = 0 and are not read)
    std::vector<TexVal> val(n ? n : 1);
    {
        // images are not available here; constant inputs never read an image node
        std::vector<akr_shader_node> tmp(g.nodes);
        for (auto& nd : tmp)
            if (nd.op == AKR_NODE_IMAGE) nd.op = AKR_NODE_CONST;
        TexScene ts{reinterpret_cast<const DNode*>(tmp.data()), nullptr, nullptr, nullptr, color, 0};
        eval_graph(ts, 0, n, mk2(0, 0), val.data());
    }
    uint32_t map[AKR_IN_COUNT];
    bool any_varying = false;
    MatInputs in;
    std::memcpy(&in, &desc, sizeof in);
    {
        uint32_t cmap[AKR_IN_COUNT];
        for (uint32_t k = 0; k < AKR_IN_COUNT; k++) {
            uint32_t node = g.input[k];
            if (node != AKR_NODE_NONE && node >= n) throw std::invalid_argument("akr_material_graph: input node out of range");
            bool v = node != AKR_NODE_NONE && varying[node];
            cmap[k] = (node != AKR_NODE_NONE && !v) ? node : AKR_NODE_NONE;
            map[k] = v ? node : AKR_NODE_NONE;
            any_varying = any_varying || v;
        }
        apply_inputs(cmap, val.data(), in);
        // constants no node feeds go through the pipeline here; node-fed ones went through it inside the graph (Rgb / uplift nodes)
        const bool fed[4] = {g.input[AKR_IN_BASE_COLOR] != AKR_NODE_NONE, g.input[AKR_IN_SPECULAR_TINT] != AKR_NODE_NONE,
                             g.input[AKR_IN_COAT_TINT] != AKR_NODE_NONE, g.input[AKR_IN_EMISSION_COLOR] != AKR_NODE_NONE};
        convert_color_inputs(in, desc.kind, color, fed);
        in.kind = desc.kind & MAT_KIND_MASK;
        std::memcpy(&desc, &in, sizeof in);
    }
    dm = fold_material(desc, 0);  // desc is in the pipeline's space now
    if (!any_varying) return;
    // prune: nodes reachable from the varying inputs, in index order
    std::vector<uint8_t> keep(n, 0);
    for (uint32_t k = 0; k < AKR_IN_COUNT; k++)
        if (map[k] != AKR_NODE_NONE) keep[map[k]] = 1;
    for (uint32_t i = n; i-- > 0;) {
        if (!keep[i]) continue;
        uint32_t refs[4], nr;
        node_refs(g.nodes[i], refs, nr);
        for (uint32_t r = 0; r < nr; r++) keep[refs[r]] = 1;
    }
    std::vector<uint32_t> remap(n, AKR_NODE_NONE);
    const uint32_t first = (uint32_t)out.tex_nodes.size();
    uint32_t count = 0;
    std::vector<akr_shader_node> pruned;
    for (uint32_t i = 0; i < n; i++) {
        if (!keep[i]) continue;
        remap[i] = count++;
        akr_shader_node nd = g.nodes[i];
        uint32_t na = node_n_args(nd.op);
        for (uint32_t a = 0; a < na; a++) {
            if (nd.op == AKR_NODE_IMAGE && a == 0) continue;
            if (nd.arg[a] != AKR_NODE_NONE) nd.arg[a] = remap[nd.arg[a]];
        }
        pruned.push_back(nd);
    }
    if (count > kMaxGraphNodes)
        throw std::invalid_argument("unsupported: shader graph needs more than " + std::to_string(kMaxGraphNodes) + " texture nodes");
    // Register allocation of the node values (device/dtex.h: kTexMaxSlots value slots per lane, in LDS): a node's slot is free
    // again after its last consumer; a node reads its arguments before it writes, so it may take over the slot of an argument
    // whose last consumer it is. Inputs of the surface node are written the moment the node that feeds them has its value
    // (`feeds` mask), so they pin nothing.
    {
        std::vector<uint32_t> last_use(count), feeds(count, 0), slot(count, kTexNoSlot);
        for (uint32_t i = 0; i < count; i++) last_use[i] = i;
        for (uint32_t i = 0; i < count; i++) {
            uint32_t refs[4], nr;
            node_refs(pruned[i], refs, nr);
            for (uint32_t r = 0; r < nr; r++) last_use[refs[r]] = std::max(last_use[refs[r]], i);
        }
        for (uint32_t k = 0; k < AKR_IN_COUNT; k++)
            if (map[k] != AKR_NODE_NONE) feeds[remap[map[k]]] |= 1u << k;
        uint32_t owner[kTexMaxSlots], used = 0;
        for (uint32_t& o : owner) o = AKR_NODE_NONE;
        for (uint32_t i = 0; i < count; i++) {
            for (uint32_t sl = 0; sl < kTexMaxSlots; sl++)
                if (owner[sl] != AKR_NODE_NONE && last_use[owner[sl]] <= i) owner[sl] = AKR_NODE_NONE;  // read by node i at the latest
            if (last_use[i] == i) continue;  // nobody reads this value back: no slot
            uint32_t sl = 0;
            while (sl < kTexMaxSlots && owner[sl] != AKR_NODE_NONE) sl++;
            if (sl == kTexMaxSlots)
                throw std::invalid_argument("unsupported: shader graph keeps more than " + std::to_string(kTexMaxSlots) + " values alive at once");
            owner[sl] = i;
            slot[i] = sl;
            used = std::max(used, sl + 1);
        }
        out.tex_slots = std::max(out.tex_slots, std::max(used, 1u));
        {   // the list as the per-scene code generator wants it, and its shape
            std::string sig = "k" + std::to_string(dm.kind) + ";";
            for (uint32_t i = 0; i < count; i++) {
                const akr_shader_node& nd = pruned[i];
                out.tex_nodes_ssa.push_back(nd);
                sig += std::to_string(nd.op) + "(";
                const uint32_t na = node_n_args(nd.op);
                for (uint32_t a = 0; a < na; a++) {
                    if (nd.op == AKR_NODE_IMAGE && a == 0) {
                        const HostImage& im = images[nd.arg[0]];
                        sig += "i" + std::to_string(im.format) + "." + std::to_string(im.filter) + "." + std::to_string(im.address) + ",";
                    } else {
                        sig += (nd.arg[a] == AKR_NODE_NONE ? std::string("-") : std::to_string(nd.arg[a])) + ",";
                    }
                }
                // immediates that select code: the Rgb node's colour space, the sRGB decode of an image, mapping type, extract field
                if (nd.op == AKR_NODE_RGB) sig += "t" + std::to_string(nd.arg[0] == 1u ? 1 : 0);
                if (nd.op == AKR_NODE_IMAGE) sig += "s" + std::to_string(nd.arg[2] != 0 ? 1 : 0);
                if (nd.op == AKR_NODE_MAPPING) sig += "m" + std::to_string(nd.arg[3]);
                if (nd.op == AKR_NODE_EXTRACT) sig += "f" + std::to_string(nd.arg[1]);
                sig += ")" + std::to_string(feeds[i]) + ";";
            }
            uint32_t kind = 0;
            while (kind < out.shader_kinds.size() && out.shader_kinds[kind].signature != sig) kind++;
            if (kind == out.shader_kinds.size()) {
                CompiledScene::ShaderKind sk;
                sk.signature = sig;
                sk.mat_kind = dm.kind;
                sk.n_nodes = count;
                out.shader_kinds.push_back(sk);
            }
            if (kind > 0xffffu) throw std::invalid_argument("unsupported: more than 65536 distinct shader graph shapes");
            out.shader_kinds[kind].materials.push_back(material_index);
            shader_kind = kind;
        }
        for (uint32_t i = 0; i < count; i++) {
            akr_shader_node nd = pruned[i];
            uint32_t na = node_n_args(nd.op);
            for (uint32_t a = 0; a < na; a++) {
                if (nd.op == AKR_NODE_IMAGE && a == 0) continue;
                if (nd.arg[a] != AKR_NODE_NONE) nd.arg[a] = slot[nd.arg[a]];  // consumers name slots
            }
            nd.op = (nd.op & 0xffu) | (slot[i] << 8) | (feeds[i] << 16);
            DNode dn;
            std::memcpy(&dn, &nd, sizeof dn);
            out.tex_nodes.push_back(dn);
        }
    }
    dm.flags |= MF_TEXTURED;
    if (map[AKR_IN_BASE_COLOR] != AKR_NODE_NONE && (dm.kind == MAT_PRINCIPLED || dm.kind == MAT_DIFFUSE)) {
        // the alpha of a node-fed base colour is the node's, never the constant of the description: either the graph is evaluated
        // per candidate hit, or -- the node's alpha is 1 everywhere -- the folded record says so (found by tools/soak.py: a constant
        // base_alpha below 1 next to an opaque texture-fed base colour made the alpha test read the constant)
        if (!alpha_is_one(g, images, map[AKR_IN_BASE_COLOR])) dm.flags |= MF_ALPHA_TEXTURED;
        else dm.base_alpha = 1.0f;
    }
    dm.tex_first_node = first;
    dm.tex_n_nodes = count | (shader_kind << kTexKindShift);
    for (uint32_t k = 0; k < AKR_IN_COUNT; k++) dm.tex_input[k] = map[k] == AKR_NODE_NONE ? AKR_NODE_NONE : remap[map[k]];
    out.has_textures = true;
}

// ------------------------------------------------------------------------------------------------ alias table
void build_alias_table(const std::vector<float>& weights, std::vector<AliasEntry>& table, std::vector<float>& pdf) {
    const size_t n = weights.size();
    if (n == 0) throw std::invalid_argument("alias table needs at least one weight");
    float sum = 0.0f;
    for (float w : weights) sum += w;
    std::vector<float> prob(n);
    for (size_t i = 0; i < n; i++) prob[i] = weights[i] / sum * (float)n;
    std::deque<uint32_t> small, large;
    for (size_t i = 0; i < n; i++) (prob[i] >= 1.0f ? large : small).push_back((uint32_t)i);
    table.assign(n, AliasEntry{0, 0.0f});
    while (!small.empty() && !large.empty()) {
        uint32_t l = small.front(), g = large.front();
        small.pop_front();
        large.pop_front();
        table[l].t = prob[l];
        table[l].j = g;
        prob[g] = (prob[g] + prob[l]) - 1.0f;
        (prob[g] < 1.0f ? small : large).push_back(g);
    }
    while (!large.empty()) {
        uint32_t g = large.front();
        large.pop_front();
        table[g] = AliasEntry{g, 1.0f};
    }
    while (!small.empty()) {
        uint32_t l = small.front();
        small.pop_front();
        table[l] = AliasEntry{l, 1.0f};
    }
    pdf.resize(n);
    for (size_t i = 0; i < n; i++) pdf[i] = weights[i] / sum;
}

// ------------------------------------------------------------------------------------------------ camera
static void m4_mul(const float* a, const float* b, float* out) {  // out = a * b, column-major, un-fused
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 4; i++)
            r[c * 4 + i] = ((a[0 * 4 + i] * b[c * 4 + 0] + a[1 * 4 + i] * b[c * 4 + 1]) + a[2 * 4 + i] * b[c * 4 + 2]) + a[3 * 4 + i] * b[c * 4 + 3];
    std::memcpy(out, r, sizeof r);
}
static void m4_scale(float x, float y, float z, float* m) {
    std::memset(m, 0, 64);
    m[0] = x; m[5] = y; m[10] = z; m[15] = 1.0f;
}
static void m4_translate(float x, float y, float z, float* m) {
    m4_scale(1, 1, 1, m);
    m[12] = x; m[13] = y; m[14] = z;
}
void camera_matrices(const akr_camera_desc& cam, float r2c[16], float c2w[16], uint32_t* c2w_identity) {
    float m[16], s[16];
    float fw = (float)cam.width, fh = (float)cam.height;
    m4_scale(1, 1, 1, m);
    m4_scale(1.0f / fw, 1.0f / fh, 1.0f, s); m4_mul(s, m, m);
    m4_scale(2.0f, 2.0f, 1.0f, s); m4_mul(s, m, m);
    m4_translate(-1.0f, -1.0f, 0.0f, s); m4_mul(s, m, m);
    m4_scale(1.0f, -1.0f, 1.0f, s); m4_mul(s, m, m);
    float t = tanf(cam.fov / 2.0f);
    if (cam.width > cam.height) m4_scale(t, t * fh / fw, 1.0f, s); else m4_scale(t * fw / fh, t, 1.0f, s);
    m4_mul(s, m, m);
    m4_translate(0.0f, 0.0f, -1.0f, s); m4_mul(s, m, m);
    std::memcpy(r2c, m, 64);
    std::memcpy(c2w, cam.c2w, 64);
    uint32_t ident = 1;  // glam abs_diff_eq(IDENTITY, 1e-4), geometry.rs:212-218
    for (int i = 0; i < 16; i++) {
        float id = (i % 5 == 0) ? 1.0f : 0.0f;
        if (!(std::fabs(cam.c2w[i] - id) <= 1e-4f)) ident = 0;
    }
    *c2w_identity = ident;
}

PcgStartConsts pcg_start_constants() {
    // run the reference's advance() loop (sampler/mod.rs:115-131) symbolically for delta = 16384 = 1 << 14:
    // state' = acc_mult * state + acc_plus with acc_mult = cur_mult_14, acc_plus = cur_mult_14 + cur_plus_14,
    // cur_plus_14 = inc * prod_{k<14}(cur_mult_k + 1)
    uint64_t cur_mult = kPcgMult, c = 1;
    for (int k = 0; k < 14; k++) {
        c = (cur_mult + 1) * c;
        cur_mult *= cur_mult;
    }
    return PcgStartConsts{cur_mult, c};
}

// ------------------------------------------------------------------------------------------------ geometry
static InstXf make_xform(const float* m) {
    InstXf x;
    x.c0 = mk3(m[0], m[1], m[2]);
    x.c1 = mk3(m[4], m[5], m[6]);
    x.c2 = mk3(m[8], m[9], m[10]);
    x.t = mk3(m[12], m[13], m[14]);
    x.k0 = cross(x.c1, x.c2);
    x.k1 = cross(x.c2, x.c0);
    x.k2 = cross(x.c0, x.c1);
    x.det = dot(x.c0, cross(x.c1, x.c2));  // MeshInstance.transform_det, mesh.rs:309-310
    x.inv_det = 1.0f / x.det;
    return x;
}
static vec3 ld3(const std::vector<float>& v, size_t i) { return mk3(v[3 * i], v[3 * i + 1], v[3 * i + 2]); }

// (woop_precompute, share_plane_row, tri_world: device/dinst.h -- shared with the two-level traversal, which computes them at the hit)
void build_bvh8(const std::vector<float>& tri_bounds, uint32_t n_tris, float pad, uint32_t stride, bool balanced, std::vector<uint32_t>& order,
                std::vector<uint32_t>& nodes, uint32_t& depth);

// The material part of the scene under the colour pipeline `color` (ColorPipeline bits): folded records, pruned node lists of
// the texture-fed materials, raw inputs. `out.images` must be filled already. The light tables always come from the default
// pipeline (load.rs:316-319 fixes sRGB for the power kernel), i.e. from compile_scene's own call with color = 0.
void compile_materials(const FlatScene& flat, uint32_t color, CompiledScene& out, std::vector<akr_material_desc>& descs) {
    out.materials.clear();
    out.tex_nodes.clear();
    out.tex_nodes_ssa.clear();
    out.shader_kinds.clear();
    out.mat_inputs.clear();
    out.has_textures = false;
    out.tex_slots = 0;
    descs = flat.materials;
    for (size_t mi = 0; mi < flat.materials.size(); mi++) {
        DMaterial d;
        if (!flat.graphs.empty() && !flat.graphs[mi].nodes.empty()) {
            compile_graph(flat.graphs[mi], flat.images, color, (uint32_t)mi, descs[mi], d, out);
        } else {
            d = fold_material(descs[mi], color);
            MatInputs in;  // keep the converted constants: mat_inputs is what the device re-folds textured materials from
            std::memcpy(&in, &descs[mi], sizeof in);
            const bool fed[4] = {false, false, false, false};
            convert_color_inputs(in, descs[mi].kind, color, fed);
            in.kind = descs[mi].kind & MAT_KIND_MASK;
            std::memcpy(&descs[mi], &in, sizeof in);
        }
        const bool tex = (d.flags & MF_TEXTURED) != 0;
        if ((d.kind == MAT_PRINCIPLED || d.kind == MAT_DIFFUSE) && (d.base_alpha < 1.0f || (d.flags & MF_ALPHA_TEXTURED))) out.has_alpha = true;
        // a textured material may switch the specular / coat layers on at any point
        if (d.kind == MAT_PRINCIPLED && ((d.flags & (MF_SPEC | MF_COAT)) || tex)) out.needs_ggx_table = true;
        out.materials.push_back(d);
    }
    if (out.has_textures) {
        out.mat_inputs.resize(descs.size());
        std::memcpy(out.mat_inputs.data(), descs.data(), descs.size() * sizeof(MatInputs));
    }
    // Lobes no material can have (device/dbsdf.h AB_*): the switching VALUE is exactly zero everywhere and nothing feeds it.
    uint32_t absent = AB_COAT | AB_TRANSMISSION | AB_NORMAL_MAP | AB_GLASS | AB_METAL;
    for (const DMaterial& m : out.materials) {
        auto fed = [&](uint32_t k) { return (m.flags & MF_TEXTURED) && m.tex_input[k] != kNodeNone; };
        if (m.kind == MAT_GLASS) absent &= ~(uint32_t)AB_GLASS;
        if (m.kind != MAT_PRINCIPLED) continue;
        if (m.coat_weight != 0.0f || (m.flags & MF_COAT) || fed(IN_COAT_WEIGHT)) absent &= ~(uint32_t)AB_COAT;
        if (m.transmission != 0.0f || (m.flags & MF_EVAL_DIEL) || fed(IN_TRANSMISSION_WEIGHT)) absent &= ~(uint32_t)AB_TRANSMISSION;
        if ((m.flags & MF_NORMAL_MAP) || fed(IN_NORMAL)) absent &= ~(uint32_t)AB_NORMAL_MAP;
        if (m.metallic != 0.0f || (m.flags & MF_EVAL_METAL) || fed(IN_METALLIC)) absent &= ~(uint32_t)AB_METAL;
    }
    out.absent = absent;
}

// Emission power estimate of one triangle, load.rs:312-343: 16 x (max(emission) * prim_area) / 16. For the folded (constant)
// emitters the emission does not depend on the sampled point or direction, so the RNG of the reference kernel does not
// influence the value; the f32 accumulation is kept. Texture-fed emission: the kernel of load.rs:312-343 as is --
// Pcg32::new_seq(prim), per sample next_2d -> barycentrics, next_2d -> wo (drawn, unused by the emission of these closures).
float triangle_emission_power(const CompiledScene& out, const TexScene& host_tex, uint32_t material, uint32_t prim, vec2 uv0, vec2 uv1, vec2 uv2, float area) {
    const DMaterial& dm = out.materials[material];
    vec3 e = (dm.kind == MAT_PRINCIPLED || dm.kind == MAT_EMISSION) ? dm.emission : mk3(0, 0, 0);
    float acc = 0.0f;
    const bool tex_emission = (dm.flags & MF_TEXTURED) && (dm.kind == MAT_PRINCIPLED || dm.kind == MAT_EMISSION) &&
                              (dm.tex_input[IN_EMISSION_COLOR] != kNodeNone || dm.tex_input[IN_EMISSION_STRENGTH] != kNodeNone);
    if (tex_emission) {
        Pcg32 rng = pcg_new_seq((uint64_t)prim);
        for (int k = 0; k < 16; k++) {
            float u0 = pcg_next_1d(rng), u1 = pcg_next_1d(rng);
            vec2 bary = uniform_sample_triangle(mk2(u0, u1));
            (void)pcg_next_1d(rng);
            (void)pcg_next_1d(rng);
            float w = 1.0f - bary.x - bary.y;
            vec2 uv = mk2((uv0.x * w + uv1.x * bary.x) + uv2.x * bary.y, (uv0.y * w + uv1.y * bary.x) + uv2.y * bary.y);
            DMaterial at = dm;
            material_at(host_tex, material, uv, at);
            acc += max3(at.emission) * area;
        }
    } else {
        for (int k = 0; k < 16; k++) acc += max3(e) * area;
    }
    return acc / 16.0f;
}
// has_potential_surface_emission, load.rs:94-127, over an instance's material list
bool instance_may_emit(const CompiledScene& out, const std::vector<akr_material_desc>& descs, const HostInstance& in) {
    bool any = false;
    for (uint32_t mi : in.materials) {
        const akr_material_desc& m = descs[mi];
        if (m.kind != AKR_MAT_PRINCIPLED && m.kind != AKR_MAT_EMISSION) { any = true; continue; }
        const DMaterial& dm = out.materials[mi];
        if ((dm.flags & MF_TEXTURED) && (dm.tex_input[IN_EMISSION_COLOR] != kNodeNone || dm.tex_input[IN_EMISSION_STRENGTH] != kNodeNone)) {
            any = true;  // estimate_emission_tex_intensity_fast gives None for texture nodes (load.rs:76-92)
            continue;
        }
        float power = max_f(max_f(m.emission_color[0], m.emission_color[1]), m.emission_color[2]);
        if (!(power * m.emission_strength == 0.0f)) any = true;
    }
    return any;
}

namespace {
// AKR_TIMING=1: wall clock of the phases of a scene compile on stderr (host-side diagnostics; no effect on results)
struct PhaseTimer {
    bool on = std::getenv("AKR_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[akari_hip] compile_scene: %-28s %.3f s\n", what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};
}  // namespace
void compile_scene(const FlatScene& flat, CompiledScene& out) {
    PhaseTimer phase;
    const size_t n_inst = flat.instances.size();
    out = CompiledScene();
    out.materials.reserve(flat.materials.size());
    if (!flat.graphs.empty() && flat.graphs.size() != flat.materials.size())
        throw std::invalid_argument("material graphs: one per material expected");
    // images: headers + one texel buffer (float4 images 16-byte aligned)
    for (const HostImage& im : flat.images) {
        while (im.format == AKR_IMAGE_RGBA32F && (out.texels.size() & 3u)) out.texels.push_back(0u);
        DImage d{};
        uint64_t off = out.texels.size();
        d.offset_lo = (uint32_t)off; d.offset_hi = (uint32_t)(off >> 32);
        d.width = im.width; d.height = im.height; d.format = im.format; d.filter = im.filter; d.address = im.address;
        out.images.push_back(d);
        out.texels.insert(out.texels.end(), im.words.begin(), im.words.end());
    }
    std::vector<akr_material_desc> descs;  // constants after folding the constant graph inputs, in the (default) pipeline's space
    compile_materials(flat, 0, out, descs);
    const TexScene host_tex{out.tex_nodes.data(), out.images.data(), out.texels.data(), out.mat_inputs.data(), 0, 0};
    // instance table
    std::vector<InstXf> xf(n_inst);
    out.inst.assign(32 * n_inst, 0.0f);
    out.inst_tri_offset.resize(n_inst + 1);
    uint32_t n_tris = 0;
    bool any_normals = false;
    {   // global triangle ids are 32 bits (0xffffffff = none): a scene kept as meshes + instances costs nothing per instance-triangle,
        // so nothing else would stop one that has more of them than that
        uint64_t total = 0;
        for (const HostInstance& in : flat.instances)
            if (in.mesh < flat.meshes.size()) total += flat.meshes[in.mesh].n_triangles();
        if (total >= 0xffffffffull)
            throw std::runtime_error("unsupported: " + std::to_string(total) + " instance-triangles; global triangle ids are 32 bits");
    }
    for (size_t i = 0; i < n_inst; i++) {
        const HostInstance& in = flat.instances[i];
        xf[i] = make_xform(in.transform);
        const InstXf& x = xf[i];
        float* r = &out.inst[32 * i];
        r[0] = x.c0.x; r[1] = x.c0.y; r[2] = x.c0.z; r[3] = x.det;
        r[4] = x.c1.x; r[5] = x.c1.y; r[6] = x.c1.z;
        r[8] = x.c2.x; r[9] = x.c2.y; r[10] = x.c2.z;
        r[12] = x.t.x; r[13] = x.t.y; r[14] = x.t.z;
        r[16] = x.k0.x; r[17] = x.k0.y; r[18] = x.k0.z; r[19] = x.inv_det;
        r[20] = x.k1.x; r[21] = x.k1.y; r[22] = x.k1.z;
        r[24] = x.k2.x; r[25] = x.k2.y; r[26] = x.k2.z;
        out.inst_tri_offset[i] = n_tris;
        n_tris += flat.meshes[in.mesh].n_triangles();
        if (!flat.meshes[in.mesh].normals.empty() || !flat.meshes[in.mesh].tangents.empty()) any_normals = true;
    }
    out.inst_tri_offset[n_inst] = n_tris;
    out.n_tris = n_tris;
    if (want_instancing(flat)) {  // meshes + instances as they are: nothing per instance-triangle (scene_inst.cpp)
        compile_instanced_geometry(flat, xf, descs, out);
        phase.lap("instanced geometry");
        return;
    }
    out.woop.assign(12ull * n_tris, 0.0f);
    out.shade.assign(32ull * n_tris, 0.0f);
    if (any_normals) out.normals.assign(24ull * n_tris, 0.0f);  // 3 float4 normals + 3 float4 tangents per triangle
    std::vector<float> bounds(6ull * n_tris);
    std::vector<float> tri_power(n_tris, 0.0f);
    std::vector<float> tri_cond(3ull * n_tris, 0.0f);  // conditioning of each triangle's (u, v) parametrisation per axis (scene_build.h)
    for (int a = 0; a < 3; a++) { out.scene_lo[a] = INFINITY; out.scene_hi[a] = -INFINITY; }

    // Per-triangle records, chunk by chunk on the host's threads (a 10 M-triangle mesh: 0.9 s on one). A chunk starts at an even
    // triangle: share_plane_row looks at the pair (prim - 1, prim). What the triangles share -- the scene box and the first error --
    // is kept per chunk and merged in chunk order afterwards, so results and error messages do not depend on the thread count.
    struct TriChunk { size_t inst; uint32_t first, last; float lo[3], hi[3]; std::string error; bool bad_slot = false; };
    std::vector<TriChunk> chunks;
    {
        const uint32_t kChunk = 1u << 16;
        for (size_t i = 0; i < n_inst; i++) {
            const uint32_t nt = flat.meshes[flat.instances[i].mesh].n_triangles();
            for (uint32_t f = 0; f < nt; f += kChunk) {
                TriChunk c;
                c.inst = i; c.first = f; c.last = std::min(nt, f + kChunk);
                for (int a = 0; a < 3; a++) { c.lo[a] = INFINITY; c.hi[a] = -INFINITY; }
                chunks.push_back(c);
            }
        }
    }
    parallel_chunks((unsigned)chunks.size(), n_tris > (1u << 17) ? host_threads() : 1u, [&](unsigned ci) {
        TriChunk& chunk = chunks[ci];
        const size_t i = chunk.inst;
        const HostInstance& in = flat.instances[i];
        const HostMesh& g = flat.meshes[in.mesh];
        const InstXf& x = xf[i];
        for (uint32_t prim = chunk.first; prim < chunk.last; prim++) {
            const uint32_t gid = out.inst_tri_offset[i] + prim;
            // material: mats[slots[prim]] when the slot buffer has more than one entry, else mats[0] (mesh.rs:508-521)
            uint32_t slot = (g.slots.size() > 1) ? g.slots[prim] : 0;
            if (slot >= in.materials.size()) { chunk.bad_slot = true; return; }
            uint32_t material = in.materials[slot];
            vec3 v0 = ld3(g.vertices, g.indices[3 * prim]), v1 = ld3(g.vertices, g.indices[3 * prim + 1]), v2 = ld3(g.vertices, g.indices[3 * prim + 2]);
            vec2 uv0, uv1, uv2;
            tri_default_uvs(uv0, uv1, uv2);  // mesh.rs:541-546
            if (!g.uvs.empty()) {
                uv0 = mk2(g.uvs[6 * prim + 0], g.uvs[6 * prim + 1]);
                uv1 = mk2(g.uvs[6 * prim + 2], g.uvs[6 * prim + 3]);
                uv2 = mk2(g.uvs[6 * prim + 4], g.uvs[6 * prim + 5]);
            }
            const TriWorld tw = tri_world(x, v0, v1, v2, uv0, uv1, uv2);  // device/dinst.h: mesh.rs:527-535, 572-589, 608-635
            const vec3 ng_local = tw.ng_local, tt = tw.tt, ng = tw.ng;
            const float area = tw.area;
            const Frame fr = tw.frame;
            uint32_t tri_flags = 0;
            bool tangents_ok = false;
            if (!g.tangents.empty()) {  // mesh.rs:557-571: per-corner tangents are used only if all nine are finite
                tangents_ok = true;
                for (int k = 0; k < 9; k++) tangents_ok = tangents_ok && is_finite(g.tangents[9 * prim + k]);
            }
            if (!g.normals.empty()) tri_flags |= TRI_HAS_NORMALS;
            if (tangents_ok) tri_flags |= 2u;  // TRI_HAS_TANGENTS
            float* r = &out.shade[32ull * gid];
            r[0] = v0.x; r[1] = v0.y; r[2] = v0.z; r[3] = uv0.x;
            r[4] = v1.x; r[5] = v1.y; r[6] = v1.z; r[7] = uv0.y;
            r[8] = v2.x; r[9] = v2.y; r[10] = v2.z; r[11] = uv1.x;
            r[12] = ng.x; r[13] = ng.y; r[14] = ng.z; r[15] = uv1.y;
            r[16] = fr.t.x; r[17] = fr.t.y; r[18] = fr.t.z; r[19] = uv2.x;
            r[20] = fr.s.x; r[21] = fr.s.y; r[22] = fr.s.z; r[23] = uv2.y;
            r[24] = area; r[25] = u2f(material); r[26] = u2f((uint32_t)i); r[27] = u2f(0xffffffffu);
            r[28] = tt.x; r[29] = tt.y; r[30] = tt.z; r[31] = u2f(tri_flags);
            if (any_normals) {
                float* nr = &out.normals[24ull * gid];
                for (int k = 0; k < 3; k++) {
                    vec3 nk = g.normals.empty() ? ng_local : mk3(g.normals[9 * prim + 3 * k], g.normals[9 * prim + 3 * k + 1], g.normals[9 * prim + 3 * k + 2]);
                    nr[4 * k] = nk.x; nr[4 * k + 1] = nk.y; nr[4 * k + 2] = nk.z;
                    if (tangents_ok) {
                        nr[12 + 4 * k] = g.tangents[9 * prim + 3 * k]; nr[12 + 4 * k + 1] = g.tangents[9 * prim + 3 * k + 1];
                        nr[12 + 4 * k + 2] = g.tangents[9 * prim + 3 * k + 2];
                    }
                }
            }
            // world-space triangle for the intersector
            vec3 A = xf_point(x.c0, x.c1, x.c2, x.t, v0), B = xf_point(x.c0, x.c1, x.c2, x.t, v1), C = xf_point(x.c0, x.c1, x.c2, x.t, v2);
            woop_precompute(A, B, C, &out.woop[12ull * gid]);
            if (prim & 1u) {
                const vec3 vb[3] = {A, B, C};
                share_plane_row(&out.woop[12ull * (gid - 1) + 8], &out.woop[12ull * gid + 8], vb);
            }
            float* bb = &bounds[6ull * gid];
            bb[0] = min_f(min_f(A.x, B.x), C.x); bb[1] = min_f(min_f(A.y, B.y), C.y); bb[2] = min_f(min_f(A.z, B.z), C.z);
            bb[3] = max_f(max_f(A.x, B.x), C.x); bb[4] = max_f(max_f(A.y, B.y), C.y); bb[5] = max_f(max_f(A.z, B.z), C.z);
            {
                const double Ad[3] = {A.x, A.y, A.z}, Bd[3] = {B.x, B.y, B.z}, Cd[3] = {C.x, C.y, C.z};
                tri_conditioning(Ad, Bd, Cd, &tri_cond[3ull * gid]);
            }
            // a non-finite corner (inf / NaN vertex, or a transform that produces one) has no place in a box hierarchy: the
            // builder's costs become NaN and geometry would silently go missing. Refused here, for every scene size.
            if (!(is_finite(A.x) && is_finite(A.y) && is_finite(A.z) && is_finite(B.x) && is_finite(B.y) && is_finite(B.z) && is_finite(C.x) &&
                  is_finite(C.y) && is_finite(C.z)))
            {
                chunk.error = "instance " + std::to_string(i) + ", triangle " + std::to_string(prim) + ": non-finite vertex position after the instance transform";
                return;
            }
            for (int a = 0; a < 3; a++) {
                chunk.lo[a] = min_f(chunk.lo[a], bb[a]);
                chunk.hi[a] = max_f(chunk.hi[a], bb[3 + a]);
            }
            tri_power[gid] = triangle_emission_power(out, host_tex, material, prim, uv0, uv1, uv2, area);
        }
    });
    for (const TriChunk& c : chunks) {  // in chunk order = in triangle order: the first error is the one the serial loop would raise
        if (c.bad_slot) throw std::invalid_argument("material slot out of range for instance");
        if (!c.error.empty()) throw std::invalid_argument(c.error);
        for (int a = 0; a < 3; a++) {
            out.scene_lo[a] = min_f(out.scene_lo[a], c.lo[a]);
            out.scene_hi[a] = max_f(out.scene_hi[a], c.hi[a]);
        }
    }
    phase.lap("triangle records");
    // lights, load.rs:345-444
    std::vector<float> light_weights;
    for (size_t i = 0; i < n_inst; i++) {
        const HostInstance& in = flat.instances[i];
        const bool any = instance_may_emit(out, descs, in);
        if (!any) continue;
        uint32_t first = out.inst_tri_offset[i], count = out.inst_tri_offset[i + 1] - first;
        std::vector<float> powers(tri_power.begin() + first, tri_power.begin() + first + count);
        float total = 0.0f;
        for (float pw : powers) total += pw;
        if (total > 1e-4f) {
            uint32_t light_id = (uint32_t)out.light_inst.size();
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
            for (uint32_t k = 0; k < count; k++) out.shade[32ull * (first + k) + 27] = u2f(light_id);
        }
    }
    out.n_lights = (uint32_t)out.light_inst.size();
    if (out.n_lights > 0) build_alias_table(light_weights, out.light_entries, out.light_pdf);

    phase.lap("light tables");
    // acceleration structure: tiny scenes are intersected exhaustively (records from LDS or the scalar cache), others get the compressed wide BVH of host/bvh.cpp
    // (AKR_FORCE_BVH=1 builds the BVH for tiny scenes too: lets the tests run both intersectors on scenes/cbox)
    const TuningOptions tune = tuning();
    const uint32_t kExhaustiveMax = tune.force_bvh ? 0u : 64u;
    // the exhaustive kernels stage the shading tables in LDS (pt_kernels.hip): a tiny mesh with a huge material list goes the BVH way
    size_t stage = 0;
    for (size_t b : {out.shade.size() * 4, out.normals.size() * 4, out.inst.size() * 4, out.materials.size() * sizeof(DMaterial),
                     (size_t)out.n_lights * 32, out.area_entries.size() * 16, out.light_pdf.size() * 4, out.area_pdf.size() * 4,
                     out.tex_nodes.size() * sizeof(DNode), out.images.size() * sizeof(DImage), out.mat_inputs.size() * sizeof(MatInputs)})
        stage += (b + 15) & ~(size_t)15;
    if (n_tris > kExhaustiveMax || stage > kStageMaxBytes) {
        const float pad_scale = 0.01f * (float)tune.pad_percent;  // (test hook: 100)
        const float pad = pad_scale * bvh_box_padding(out.scene_lo, out.scene_hi, flat.camera.c2w);
        {   // needles: what their ill-conditioned inside test reaches beyond the flat padding, per triangle and axis (scene_build.h)
            const unsigned nc = n_tris > (1u << 17) ? host_threads() * 4 : 1;
            parallel_chunks(nc, host_threads(), [&](unsigned c) {
                const uint32_t lo = (uint32_t)((uint64_t)n_tris * c / nc), hi = (uint32_t)((uint64_t)n_tris * (c + 1) / nc);
                for (uint32_t g = lo; g < hi; g++) {
                    const float* k = &tri_cond[3ull * g];
                    float* bb = &bounds[6ull * g];
                    const float mag = box_magnitude(bb, bb + 3);
                    for (int a = 0; a < 3; a++) {
                        const float extra = pad_scale * tri_cond_extra(k[a], mag, pad / pad_scale);
                        if (extra > 0.0f) {
                            bb[a] -= extra;
                            bb[3 + a] += extra;
                        }
                    }
                }
            });
        }
        std::vector<float>().swap(tri_cond);
        std::vector<uint32_t> order;
        const bool force_balanced = tune.bvh_balanced != 0;  // test hook: take the fallback builder
        build_bvh8(bounds, n_tris, pad, kBvhNodeWords, force_balanced, order, out.bvh_nodes, out.bvh_depth);
        // A traversal keeps at most one stack entry per tree level (device/disect.h): a tree that fits the stack cannot
        // overflow it. An SAH tree deeper than that (pathological geometry) is replaced by a median-split tree of depth
        // ~log8(n); if even that does not fit the scene is refused rather than rendered wrongly.
        if (out.bvh_depth > kBvhStackDepth) build_bvh8(bounds, n_tris, pad, kBvhNodeWords, true, order, out.bvh_nodes, out.bvh_depth);
        if (out.bvh_depth > kBvhStackDepth)
            throw std::runtime_error("unsupported: BVH depth " + std::to_string(out.bvh_depth) + " exceeds the traversal stack (" +
                                     std::to_string(kBvhStackDepth) + " levels)");
        phase.lap("BVH build");
        // triangle records in traversal order: the 48-byte test record, then the global id (one 64-byte fetch per test)
        std::vector<float> rec((size_t)kBvhTriWords * n_tris, 0.0f);
        {
            const unsigned nc = n_tris > (1u << 17) ? host_threads() * 4 : 1;
            parallel_chunks(nc, host_threads(), [&](unsigned c) {
                const uint32_t lo = (uint32_t)((uint64_t)n_tris * c / nc), hi = (uint32_t)((uint64_t)n_tris * (c + 1) / nc);
                for (uint32_t k = lo; k < hi; k++) {
                    std::memcpy(&rec[(size_t)kBvhTriWords * k], &out.woop[12ull * order[k]], 48);
                    rec[(size_t)kBvhTriWords * k + 12] = u2f(order[k]);
                }
            });
        }
        out.woop.swap(rec);
        out.tri_gid = order;
        phase.lap("records in traversal order");
    }
    // two all-zero records of padding: the exhaustive intersectors prefetch up to record n + 1
    out.woop.resize(out.woop.size() + 32, 0.0f);
}

namespace {
std::mutex g_tuning_mutex;
bool g_tuning_init = false;
TuningOptions g_tuning;
void tuning_init_locked() {
    if (g_tuning_init) return;
    g_tuning_init = true;
    auto flag = [](const char* name) { const char* e = std::getenv(name); return e && e[0] == '1' ? 1 : 0; };
    g_tuning.force_bvh = flag("AKR_FORCE_BVH");
    g_tuning.bvh_balanced = flag("AKR_BVH_BALANCED");
    if (const char* e = std::getenv("AKR_PT_DEFER_METAL")) g_tuning.defer_metal = std::atoi(e);
    if (const char* e = std::getenv("AKR_PT_MODE")) g_tuning.wavefront = std::string(e) == "wavefront" ? 1 : (std::string(e) == "auto" ? -1 : 0);
    if (const char* e = std::getenv("AKR_PT_SIMPLE")) g_tuning.simple_kernels = std::atoi(e) != 0 ? 1 : 0;
    if (const char* e = std::getenv("AKR_SPECIALISE")) g_tuning.specialise = std::atoi(e);
    if (const char* e = std::getenv("AKR_SPECIALISE_WAVES")) g_tuning.specialise_waves = std::atoi(e);
    if (const char* e = std::getenv("AKR_WF_SORT")) g_tuning.wf_sort = std::atoi(e) != 0 ? 1 : 0;
    if (const char* e = std::getenv("AKR_INSTANCING")) g_tuning.instancing = std::atoi(e);
    if (const char* e = std::getenv("AKR_WF_GROUPS")) g_tuning.wf_groups = std::max(0, std::min(32, std::atoi(e)));
    if (const char* e = std::getenv("AKR_WF_CARRY")) g_tuning.wf_carry = std::max(0, std::min(1 << 30, std::atoi(e)));
    if (const char* e = std::getenv("AKR_SCHED_TRIAL")) g_tuning.sched_trial = std::max(-1, std::min(1, std::atoi(e)));
    if (const char* e = std::getenv("AKR_REBRAID")) g_tuning.rebraid = std::max(1, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("AKR_ARITH")) g_tuning.arith = std::atoi(e) != 0 ? 1 : 0;
}
int* tuning_field(const char* name) {
    const std::string n = name ? name : "";
    if (n == "force_bvh") return &g_tuning.force_bvh;
    if (n == "bvh_balanced") return &g_tuning.bvh_balanced;
    if (n == "defer_metal") return &g_tuning.defer_metal;
    if (n == "wavefront") return &g_tuning.wavefront;
    if (n == "simple_kernels") return &g_tuning.simple_kernels;
    if (n == "defer_on") return &g_tuning.defer_on;
    if (n == "specialise") return &g_tuning.specialise;
    if (n == "specialise_waves") return &g_tuning.specialise_waves;
    if (n == "max_fused_passes") return &g_tuning.max_fused_passes;
    if (n == "wf_sort") return &g_tuning.wf_sort;
    if (n == "instancing") return &g_tuning.instancing;
    if (n == "arith") return &g_tuning.arith;
    if (n == "rebraid") return &g_tuning.rebraid;
    if (n == "wf_groups") return &g_tuning.wf_groups;
    if (n == "wf_carry") return &g_tuning.wf_carry;
    if (n == "sched_trial") return &g_tuning.sched_trial;
    if (n == "pad_percent") return &g_tuning.pad_percent;
    return nullptr;
}
}  // namespace
TuningOptions tuning() {
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    tuning_init_locked();
    return g_tuning;
}
bool tuning_set(const char* name, int value) {
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    tuning_init_locked();
    int* f = tuning_field(name);
    if (!f) return false;
    if (f == &g_tuning.defer_on && (value < 0 || value > 3)) return false;
    if (f == &g_tuning.specialise && (value < -1 || value > 1)) return false;
    if (f == &g_tuning.specialise_waves && value != 0 && (value < 2 || value > 4)) return false;
    if (f == &g_tuning.max_fused_passes && (value < 0 || value > 64)) return false;
    if (f == &g_tuning.wf_sort && (value < 0 || value > 1)) return false;
    if (f == &g_tuning.instancing && (value < -1 || value > 1)) return false;
    if (f == &g_tuning.arith && (value < 0 || value > 1)) return false;
    if (f == &g_tuning.rebraid && (value < 1 || value > 64)) return false;
    if (f == &g_tuning.wf_groups && (value < 0 || value > 32)) return false;
    if (f == &g_tuning.wf_carry && (value < 0 || value > (1 << 30))) return false;
    if (f == &g_tuning.sched_trial && (value < -1 || value > 1)) return false;
    if (f == &g_tuning.wavefront && (value < -1 || value > 1)) return false;
    if (f == &g_tuning.pad_percent && (value < 1 || value > 10000)) return false;
    *f = value;
    return true;
}
bool tuning_get(const char* name, int* value) {
    std::lock_guard<std::mutex> lock(g_tuning_mutex);
    tuning_init_locked();
    int* f = tuning_field(name);
    if (!f) return false;
    *value = *f;
    return true;
}

}  // namespace akr
