// api_scene.cpp -- scenes: compile + upload, inspection, per-scene kernel text (C ABI of libakari_hip.so, include/akari_hip.h; shared internals: api_internal.h)
#include "api_internal.h"

static void ensure_ggx_table(akr_scene* s) {
    akr_context* ctx = s->ctx;
    if (!s->flat.ggx_table.empty()) {
        s->ggx_host = s->flat.ggx_table;
    } else if (s->cs.needs_ggx_table) {
        std::lock_guard<std::mutex> lock(ctx->ggx_mutex);
        if (!ctx->ggx_cache.empty()) {  // computed for an earlier scene of this context
            s->ggx_host = ctx->ggx_cache;
            s->ggx_table.upload(s->ggx_host);
            return;
        }
        // PreComputedTables::init (svm/surface/precompute.rs:133-145): seeds = StdRng(0) stream, one per entry
        std::vector<uint64_t> seeds(4096);
        StdRng rng(0);
        for (auto& v : seeds) v = rng.next_u64();
        DevBuf dseeds;
        dseeds.upload(seeds);
        s->ggx_table.alloc(4096 * sizeof(float));
        HIP_CHECK(launch_ggx_table(dseeds.as<uint64_t>(), s->ggx_table.as<float>(), 1u << 20, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        s->ggx_host.resize(4096);
        HIP_CHECK(hipMemcpy(s->ggx_host.data(), s->ggx_table.p, 4096 * sizeof(float), hipMemcpyDeviceToHost));
        ctx->ggx_cache = s->ggx_host;
        return;
    } else {
        s->ggx_host.assign(4096, 0.0f);  // never read with a non-zero weight
    }
    s->ggx_table.upload(s->ggx_host);
}

void akr_api::scene_finish(akr_scene* s) {
    akr_context* ctx = s->ctx;
    compile_scene(s->flat, s->cs);
    CompiledScene& cs = s->cs;
    camera_matrices(s->flat.camera, s->r2c, s->c2w, &s->c2w_identity);
    if (!ctx) {  // host-only scene: inspectable, not renderable
        s->ggx_host = s->flat.ggx_table.empty() ? std::vector<float>(4096, 0.0f) : s->flat.ggx_table;
        std::memset(&s->dscene, 0, sizeof s->dscene);
        return;
    }
    ctx->bind();
    s->woop.upload(cs.woop);
    s->tri_gid.upload(cs.tri_gid);
    s->shade.upload(cs.shade);
    s->normals.upload(cs.normals);
    s->inst.upload(cs.inst);
    s->materials.upload(cs.materials);
    s->light_entries.upload(cs.light_entries);
    s->light_pdf.upload(cs.light_pdf);
    s->light_inst.upload(cs.light_inst);
    s->light_tri_offset.upload(cs.light_tri_offset);
    s->light_n_tris.upload(cs.light_n_tris);
    s->area_entries.upload(cs.area_entries);
    s->area_pdf.upload(cs.area_pdf);
    s->inst_tri_offset.upload(cs.inst_tri_offset);
    {  // the light tables once more, packed so that each level of light sampling is ONE gather (device/dgeom.h, dscene.h)
        auto pack = [](const std::vector<AliasEntry>& e, const std::vector<float>& pdf, size_t first, size_t n, std::vector<AliasPacked>& out) {
            for (size_t i = 0; i < n; i++) out.push_back(AliasPacked{e[first + i].j, e[first + i].t, pdf[first + i], pdf[first + e[first + i].j]});
        };
        std::vector<AliasPacked> la, aa;
        std::vector<LightRec> lr;
        pack(cs.light_entries, cs.light_pdf, 0, cs.n_lights, la);
        for (uint32_t l = 0; l < cs.n_lights; l++) {
            pack(cs.area_entries, cs.area_pdf, cs.light_tri_offset[l], cs.light_n_tris[l], aa);
            lr.push_back(LightRec{cs.light_tri_offset[l], cs.light_n_tris[l], cs.inst_tri_offset[cs.light_inst[l]], cs.light_inst[l]});
        }
        s->light_alias.upload(la);
        s->area_alias.upload(aa);
        s->lights.upload(lr);
    }
    s->bvh_nodes.upload(cs.instanced.on ? cs.instanced.nodes : cs.bvh_nodes);
    if (cs.instanced.on) {
        s->in2_tlas_leaves.upload(cs.instanced.tlas_leaves);
        s->in2_mesh_tris.upload(cs.instanced.mesh_tris);
        s->in2_mesh_pos.upload(cs.instanced.mesh_pos);
        s->in2_mesh_meta.upload(cs.instanced.mesh_meta);
        s->in2_mesh_normals.upload(cs.instanced.mesh_normals);
        s->in2_inst_mats.upload(cs.instanced.inst_mats);
    }
    if (cs.has_textures) {
        s->tex_nodes.upload(cs.tex_nodes);
        s->tex_images.upload(cs.images);
        s->tex_texels.upload(cs.texels);
        s->tex_mat_inputs.upload(cs.mat_inputs);
    }
    ensure_ggx_table(s);
    DScene& d = s->dscene;
    std::memset(&d, 0, sizeof d);
    d.woop = s->woop.as<float4>();
    d.tri_gid = s->tri_gid.as<uint32_t>();
    d.shade = s->shade.as<float4>();
    d.normals = s->normals.as<float4>();
    d.inst = s->inst.as<float4>();
    d.materials = s->materials.as<DMaterial>();
    d.ggx_table = s->ggx_table.as<float>();
    d.light_entries = s->light_entries.as<AliasEntry>();
    d.light_pdf = s->light_pdf.as<float>();
    d.light_inst = s->light_inst.as<uint32_t>();
    d.light_tri_offset = s->light_tri_offset.as<uint32_t>();
    d.light_n_tris = s->light_n_tris.as<uint32_t>();
    d.area_entries = s->area_entries.as<AliasEntry>();
    d.area_pdf = s->area_pdf.as<float>();
    d.inst_tri_offset = s->inst_tri_offset.as<uint32_t>();
    d.light_alias = s->light_alias.as<AliasPacked>();
    d.area_alias = s->area_alias.as<AliasPacked>();
    d.lights = s->lights.as<LightRec>();
    d.bvh_nodes = s->bvh_nodes.as<uint4>();
    d.n_tris = cs.n_tris;
    d.n_lights = cs.n_lights;
    d.n_nodes = (uint32_t)((cs.instanced.on ? cs.instanced.nodes.size() : cs.bvh_nodes.size()) / kBvhNodeWords);
    d.has_alpha = cs.has_alpha ? 1u : 0u;
    d.bvh_stack_depth = std::max(1u, std::min(cs.bvh_depth, kBvhStackDepth));  // one pending group per tree level at most (disect.h)
    if (cs.instanced.on) {  // two levels + the three words that remember the TLAS position (dinst_trav.h); scene_inst.cpp checked the bound
        d.bvh_stack_depth = cs.bvh_depth;
        d.in2.tlas_leaves = s->in2_tlas_leaves.as<uint4>();
        d.in2.mesh_tris = s->in2_mesh_tris.as<float4>();
        d.in2.mesh_pos = s->in2_mesh_pos.as<uint32_t>();
        d.in2.mesh_meta = s->in2_mesh_meta.as<uint32_t>();
        d.in2.mesh_normals = s->in2_mesh_normals.as<float4>();
        d.in2.inst_mats = s->in2_inst_mats.as<uint32_t>();
        d.in2.on = 1u;
        d.in2.n_instances = (uint32_t)s->flat.instances.size();
    }
    d.plane_share_mask = 0;
    if (cs.bvh_nodes.empty() && !cs.instanced.on)  // exhaustive path (<= 64 triangles): which records repeat their predecessor's plane row
        for (uint32_t k = 1; k < d.n_tris && k < 64; k++)
            if (std::memcmp(&cs.woop[12ull * k + 8], &cs.woop[12ull * (k - 1) + 8], 16) == 0) d.plane_share_mask |= 1ull << k;
    if (cs.has_textures) {
        d.tex.nodes = s->tex_nodes.as<DNode>();
        d.tex.images = s->tex_images.as<DImage>();
        d.tex.texels = s->tex_texels.as<uint32_t>();
        d.tex.mat_inputs = s->tex_mat_inputs.as<MatInputs>();
    }
    if (cs.instanced.on) {  // which instance-triangles take their even neighbour's plane row: one bit each, decided on the device (dinst_trav.h resolve_pending)
        const size_t words = ((size_t)cs.n_tris + 31u) / 32u;
        s->in2_share_bits.alloc(std::max<size_t>(words, 1u) * 4u);
        HIP_CHECK(hipMemsetAsync(s->in2_share_bits.p, 0, s->in2_share_bits.bytes, ctx->stream));
        HIP_CHECK(launch_inst_share_bits(d, s->in2_share_bits.as<uint32_t>(), s->in2_mesh_tris.as<uint32_t>(), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        d.in2.share_bits = s->in2_share_bits.as<uint32_t>();
    }
    s->device_bytes = 0;
    for (const DevBuf* b : {&s->woop, &s->tri_gid, &s->shade, &s->normals, &s->inst, &s->materials, &s->ggx_table, &s->light_entries,
                            &s->light_pdf, &s->light_inst, &s->light_tri_offset, &s->light_n_tris, &s->area_entries, &s->area_pdf,
                            &s->inst_tri_offset, &s->light_alias, &s->area_alias, &s->lights, &s->bvh_nodes, &s->tex_nodes, &s->tex_images, &s->tex_texels, &s->tex_mat_inputs,
                            &s->in2_tlas_leaves, &s->in2_mesh_tris, &s->in2_mesh_pos, &s->in2_mesh_meta, &s->in2_mesh_normals, &s->in2_inst_mats, &s->in2_share_bits})
        s->device_bytes += b->bytes;
}

// the arrays scene_finish uploads, in bytes (the packed light tables repeat the alias tables: 16 B per entry)
static uint64_t compiled_scene_bytes(const CompiledScene& cs) {
    auto b = [](const auto& v) { return (uint64_t)v.size() * sizeof(v[0]); };
    uint64_t n = b(cs.woop) + b(cs.tri_gid) + b(cs.shade) + b(cs.normals) + b(cs.inst) + b(cs.materials) + b(cs.light_entries) + b(cs.light_pdf) +
                 b(cs.light_inst) + b(cs.light_tri_offset) + b(cs.light_n_tris) + b(cs.area_entries) + b(cs.area_pdf) + b(cs.inst_tri_offset) +
                 16ull * (cs.light_entries.size() + cs.area_entries.size() + cs.n_lights) + b(cs.bvh_nodes) + b(cs.tex_nodes) + b(cs.images) + b(cs.texels) +
                 b(cs.mat_inputs);
    const CompiledScene::Instanced& is = cs.instanced;
    return n + b(is.nodes) + b(is.tlas_leaves) + b(is.mesh_tris) + b(is.mesh_pos) + b(is.mesh_meta) + b(is.mesh_normals) + b(is.inst_mats) + (is.on ? std::max<uint64_t>(((uint64_t)cs.n_tris + 31u) / 32u, 1u) * 4u : 0u);
}

extern "C" {

AKR_API int32_t akr_scene_create(akr_context* ctx, const akr_scene_desc* desc, akr_scene** out) {
    if (!desc || !out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_create: NULL argument");
    *out = nullptr;
    return guarded([&] {
        auto s = std::make_unique<akr_scene>();
        s->ctx = ctx;
        s->flat = FlatScene::from_desc(*desc);
        scene_finish(s.get());
        *out = s.release();
    });
}
AKR_API int32_t akr_scene_load(akr_context* ctx, const char* path, uint32_t width, uint32_t height, akr_scene** out) {
    if (!path || !out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_load: NULL argument");
    *out = nullptr;
    return guarded([&] {
        auto s = std::make_unique<akr_scene>();
        s->ctx = ctx;
        s->flat = load_scene_json(path);
        if (width && height) {
            s->flat.camera.width = width;
            s->flat.camera.height = height;
        }
        scene_finish(s.get());
        *out = s.release();
    });
}
AKR_API int32_t akr_scene_destroy(akr_scene* scene) {
    if (!scene) return AKR_OK;
    return guarded([&] {
        if (scene->ctx) (void)hipSetDevice(scene->ctx->device);
        delete scene;
    });
}
AKR_API int32_t akr_scene_set_resolution(akr_scene* s, uint32_t width, uint32_t height) {
    if (!s || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_set_resolution: bad argument");
    s->flat.camera.width = width;
    s->flat.camera.height = height;
    camera_matrices(s->flat.camera, s->r2c, s->c2w, &s->c2w_identity);
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_info(const akr_scene* s, akr_scene_info* info) {
    if (!s || !info) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_info: NULL argument");
    info->width = s->flat.camera.width;
    info->height = s->flat.camera.height;
    info->n_instances = (uint32_t)s->flat.instances.size();
    info->n_triangles = s->cs.n_tris;
    info->n_materials = (uint32_t)s->flat.materials.size();
    info->n_lights = s->cs.n_lights;
    info->n_bvh_nodes = (uint32_t)((s->cs.instanced.on ? s->cs.instanced.nodes.size() : s->cs.bvh_nodes.size()) / kBvhNodeWords);
    info->uses_bvh = (s->cs.bvh_nodes.empty() && !s->cs.instanced.on) ? 0u : (s->cs.instanced.on ? 2u : 1u);  // 2 = two-level (meshes + instances)
    info->device_bytes = s->device_bytes ? s->device_bytes : compiled_scene_bytes(s->cs);  // (a host-only scene: what an upload would take)
    info->node_bytes = (s->cs.bvh_nodes.empty() && !s->cs.instanced.on) ? 0u : kBvhNodeWords * 4u;   // bytes a traversal reads per node visit
    info->node_stride_bytes = (s->cs.bvh_nodes.empty() && !s->cs.instanced.on) ? 0u : kBvhNodeWords * 4u;
    info->tri_bytes = (s->cs.bvh_nodes.empty() && !s->cs.instanced.on) ? 48u : kBvhTriWords * 4u;
    info->bvh_depth = s->cs.bvh_depth;
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_light(const akr_scene* s, uint32_t light, uint32_t* instance, float* power, float* pdf) {
    if (!s || light >= s->cs.n_lights) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_light: bad argument");
    if (instance) *instance = s->cs.light_inst[light];
    if (power) *power = s->cs.light_power[light];
    if (pdf) *pdf = s->cs.light_pdf[light];
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_ggx_table(const akr_scene* s, float* dst) {
    if (!s || !dst) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_ggx_table: NULL argument");
    std::memcpy(dst, s->ggx_host.data(), 4096 * sizeof(float));
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_desc_counts(const akr_scene* s, uint32_t* n_meshes, uint32_t* n_instances, uint32_t* n_materials) {
    if (!s) return fail(AKR_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (n_meshes) *n_meshes = (uint32_t)s->flat.meshes.size();
    if (n_instances) *n_instances = (uint32_t)s->flat.instances.size();
    if (n_materials) *n_materials = (uint32_t)s->flat.materials.size();
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_mesh(const akr_scene* s, uint32_t i, akr_mesh_desc* out) {
    if (!s || !out || i >= s->flat.meshes.size()) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_mesh: bad argument");
    const HostMesh& m = s->flat.meshes[i];
    out->n_vertices = (uint32_t)(m.vertices.size() / 3);
    out->n_triangles = m.n_triangles();
    out->vertices = m.vertices.data();
    out->indices = m.indices.data();
    out->uvs = m.uvs.empty() ? nullptr : m.uvs.data();
    out->normals = m.normals.empty() ? nullptr : m.normals.data();
    out->tangents = m.tangents.empty() ? nullptr : m.tangents.data();
    out->material_slots = m.slots.empty() ? nullptr : m.slots.data();
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_instance(const akr_scene* s, uint32_t i, akr_instance_desc* out) {
    if (!s || !out || i >= s->flat.instances.size()) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_instance: bad argument");
    const HostInstance& h = s->flat.instances[i];
    out->mesh = h.mesh;
    out->n_materials = (uint32_t)h.materials.size();
    out->materials = h.materials.data();
    std::memcpy(out->transform, h.transform, sizeof out->transform);
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_material(const akr_scene* s, uint32_t i, akr_material_desc* out) {
    if (!s || !out || i >= s->flat.materials.size()) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_material: bad argument");
    *out = s->flat.materials[i];
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_image_count(const akr_scene* s, uint32_t* n) {
    if (!s || !n) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_image_count: NULL argument");
    *n = (uint32_t)s->flat.images.size();
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_image(const akr_scene* s, uint32_t i, akr_image_desc* out) {
    if (!s || !out || i >= s->flat.images.size()) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_image: bad argument");
    const HostImage& h = s->flat.images[i];
    out->width = h.width; out->height = h.height; out->format = h.format; out->filter = h.filter; out->address = h.address; out->_pad = 0;
    out->texels = h.words.data();
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_material_graph(const akr_scene* s, uint32_t i, akr_material_graph* out) {
    if (!s || !out || i >= s->flat.materials.size()) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_material_graph: bad argument");
    std::memset(out, 0, sizeof *out);
    for (uint32_t& k : out->input) k = AKR_NODE_NONE;
    if (i < s->flat.graphs.size()) {
        const HostGraph& g = s->flat.graphs[i];
        out->n_nodes = (uint32_t)g.nodes.size();
        out->nodes = g.nodes.empty() ? nullptr : g.nodes.data();
        std::memcpy(out->input, g.input, sizeof out->input);
    }
    return AKR_OK;
}
AKR_API int32_t akr_scene_get_camera(const akr_scene* s, akr_camera_desc* out) {
    if (!s || !out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_camera: NULL argument");
    *out = s->flat.camera;
    return AKR_OK;
}

AKR_API int32_t akr_scene_get_array(const akr_scene* s, int32_t which, const void** ptr, uint64_t* bytes) {
    if (!s || !ptr || !bytes) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_array: NULL argument");
    const CompiledScene& cs = s->cs;
    auto set = [&](const void* p, size_t n) { *ptr = n ? p : nullptr; *bytes = n; };
    switch (which) {
        case AKR_ARRAY_WOOP: set(cs.woop.data(), cs.woop.size() * 4); break;
        case AKR_ARRAY_TRI_GID: set(cs.tri_gid.data(), cs.tri_gid.size() * 4); break;
        case AKR_ARRAY_SHADE: set(cs.shade.data(), cs.shade.size() * 4); break;
        case AKR_ARRAY_INSTANCES: set(cs.inst.data(), cs.inst.size() * 4); break;
        case AKR_ARRAY_MATERIALS: set(cs.materials.data(), cs.materials.size() * sizeof(DMaterial)); break;
        case AKR_ARRAY_BVH_NODES:
            if (cs.instanced.on) set(cs.instanced.nodes.data(), cs.instanced.nodes.size() * 4);
            else set(cs.bvh_nodes.data(), cs.bvh_nodes.size() * 4);
            break;
        case AKR_ARRAY_LIGHT_ENTRIES: set(cs.light_entries.data(), cs.light_entries.size() * sizeof(AliasEntry)); break;
        case AKR_ARRAY_LIGHT_PDF: set(cs.light_pdf.data(), cs.light_pdf.size() * 4); break;
        case AKR_ARRAY_AREA_ENTRIES: set(cs.area_entries.data(), cs.area_entries.size() * sizeof(AliasEntry)); break;
        case AKR_ARRAY_AREA_PDF: set(cs.area_pdf.data(), cs.area_pdf.size() * 4); break;
        case AKR_ARRAY_INST_TRI_OFFSET: set(cs.inst_tri_offset.data(), cs.inst_tri_offset.size() * 4); break;
        case AKR_ARRAY_R2C: set(s->r2c, 64); break;
        case AKR_ARRAY_C2W: set(s->c2w, 64); break;
        case AKR_ARRAY_TEX_NODES: set(cs.tex_nodes.data(), cs.tex_nodes.size() * sizeof(DNode)); break;
        case AKR_ARRAY_TEX_IMAGES: set(cs.images.data(), cs.images.size() * sizeof(DImage)); break;
        case AKR_ARRAY_TEX_TEXELS: set(cs.texels.data(), cs.texels.size() * 4); break;
        case AKR_ARRAY_MAT_INPUTS: set(cs.mat_inputs.data(), cs.mat_inputs.size() * sizeof(MatInputs)); break;
        case AKR_ARRAY_INST_LEAVES: set(cs.instanced.tlas_leaves.data(), cs.instanced.tlas_leaves.size() * 4); break;
        case AKR_ARRAY_MESH_TRIS: set(cs.instanced.mesh_tris.data(), cs.instanced.mesh_tris.size() * 4); break;
        case AKR_ARRAY_MESH_POS: set(cs.instanced.mesh_pos.data(), cs.instanced.mesh_pos.size() * 4); break;
        case AKR_ARRAY_MESH_META: set(cs.instanced.mesh_meta.data(), cs.instanced.mesh_meta.size() * 4); break;
        case AKR_ARRAY_MESH_NORMALS: set(cs.instanced.mesh_normals.data(), cs.instanced.mesh_normals.size() * 4); break;
        default: return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_array: unknown array id");
    }
    return AKR_OK;
}

AKR_API int32_t akr_scene_spec_source(akr_scene* scene, char* dst, uint64_t capacity, uint64_t* length) {
    if (!scene || !length) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_spec_source: NULL argument");
    return guarded([&] {
        std::lock_guard<std::mutex> lock(scene->spec_mutex);
        if (!scene->spec_header_made) {
            scene->spec_header = generate_scene_spec(scene->cs);
            scene->spec_header_made = true;
        }
        *length = scene->spec_header.size();
        if (dst && capacity) {
            const size_t n = std::min<size_t>(capacity - 1, scene->spec_header.size());
            std::memcpy(dst, scene->spec_header.data(), n);
            dst[n] = 0;
        }
    });
}
AKR_API int32_t akr_host_spec_compile(akr_scene* scene, uint32_t flags, uint32_t min_waves, const char* arch, uint64_t* code_bytes, char* log, uint32_t log_len) {
    if (!scene || !code_bytes) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_spec_compile: NULL argument");
    return guarded([&] {
        std::string header;
        {
            std::lock_guard<std::mutex> lock(scene->spec_mutex);
            if (!scene->spec_header_made) {
                scene->spec_header = generate_scene_spec(scene->cs);
                scene->spec_header_made = true;
            }
            header = scene->spec_header;
        }
        if (header.empty()) throw Unsupported("unsupported: the scene has no per-scene code (no texture-fed material, or too many shader kinds)");
        SpecRequest rq;
        rq.bvh = flags & 1u; rq.pmj = flags & 2u; rq.stage = flags & 4u; rq.defer = flags & 8u; rq.inst = flags & 16u;
        rq.min_waves = (int)min_waves;
        std::vector<char> code;
        std::string text;
        const bool ok = spec_compile(header, rq, arch && *arch ? arch : "gfx950", code, text);
        if (log && log_len) std::snprintf(log, log_len, "%s", text.c_str());
        if (!ok) throw RenderError("per-scene kernel did not compile: " + text.substr(0, 1500));
        *code_bytes = code.size();
    });
}
// the helper process's entry (akari-cli --spec-compile): generated text in, code object file out; always this process's hiprtc
AKR_API int32_t akr_host_spec_compile_text(const char* spec_header, uint32_t flags, uint32_t min_waves, const char* arch, const char* out_path) {
    if (!spec_header || !arch || !out_path) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_spec_compile_text: NULL argument");
    return guarded([&] {
        SpecRequest rq;
        rq.bvh = flags & 1u; rq.pmj = flags & 2u; rq.stage = flags & 4u; rq.defer = flags & 8u; rq.inst = flags & 16u;
        rq.min_waves = (int)min_waves;
        std::vector<char> code;
        std::string log;
        if (!spec_compile(spec_header, rq, arch, code, log, /*in_process=*/true)) throw RenderError("per-scene kernel did not compile: " + log.substr(0, 3000));
        FILE* f = std::fopen(out_path, "wb");
        if (!f) throw IoError(std::string("cannot open ") + out_path);
        const size_t n = std::fwrite(code.data(), 1, code.size(), f);
        std::fclose(f);
        if (n != code.size()) throw IoError(std::string("short write to ") + out_path);
    });
}
}  // extern "C"
