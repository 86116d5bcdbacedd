// api.cpp -- the C ABI of libakari_hip.so (include/akari_hip.h).
//
// Every entry point catches C++ exceptions and HIP errors and turns them into an akr_status plus a thread-local
// message; nothing throws or aborts across the boundary. There is no CPU path here: without a GPU
// akr_context_create fails with AKR_ERR_NO_DEVICE.
#include <mutex>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene_build.h"
#include "specialise.h"
#include "stdrng.h"

using namespace akr;

namespace {

thread_local std::string g_last_error;

struct HipError : std::runtime_error {
    explicit HipError(const std::string& s) : std::runtime_error(s) {}
};
struct Unsupported : std::runtime_error {
    explicit Unsupported(const std::string& s) : std::runtime_error(s) {}
};
struct IoError : std::runtime_error {
    explicit IoError(const std::string& s) : std::runtime_error(s) {}
};
struct RenderError : std::runtime_error {  // the device ran, but the result is not a valid render (not an input-file problem)
    explicit RenderError(const std::string& s) : std::runtime_error(s) {}
};

#define HIP_CHECK(expr)                                                                                         \
    do {                                                                                                        \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess) throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e));                \
    } while (0)

int32_t fail(int32_t code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

template <typename F>
int32_t guarded(F&& f) {
    try {
        g_last_error.clear();
        f();
        return AKR_OK;
    } catch (const HipError& e) {
        return fail(AKR_ERR_HIP, e.what());
    } catch (const Unsupported& e) {
        return fail(AKR_ERR_UNSUPPORTED, e.what());
    } catch (const IoError& e) {
        return fail(AKR_ERR_IO, e.what());
    } catch (const RenderError& e) {
        return fail(AKR_ERR_RENDER, e.what());
    } catch (const std::invalid_argument& e) {
        return fail(AKR_ERR_INVALID_ARGUMENT, e.what());
    } catch (const std::bad_alloc&) {
        return fail(AKR_ERR_OUT_OF_MEMORY, "out of host memory");
    } catch (const std::exception& e) {
        std::string w = e.what();
        if (w.rfind("unsupported", 0) == 0 || w.find("unsupported:") != std::string::npos) return fail(AKR_ERR_UNSUPPORTED, w);
        if (w.rfind("cannot open", 0) == 0) return fail(AKR_ERR_IO, w);
        return fail(AKR_ERR_PARSE, w);
    } catch (...) {
        return fail(AKR_ERR_INVALID_ARGUMENT, "unknown error");
    }
}

// RAII device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    void alloc(size_t n) {
        release();
        if (n == 0) return;
        HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
    }
    template <typename T>
    void upload(const std::vector<T>& v) {
        alloc(v.size() * sizeof(T));
        if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    template <typename T>
    T* as() const { return (T*)p; }
};

}  // namespace

struct akr_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t props;
    void bind() const { HIP_CHECK(hipSetDevice(device)); }
    SpecCache spec_cache;  // per-scene kernels loaded on this device (host/specialise.cpp)
    // tables of the pmj02bn sampler, uploaded when the first session asks for it
    DevBuf pmj_sets, bluenoise;
    void ensure_pmj_tables() {
        if (pmj_sets.p && bluenoise.p) return;
        std::vector<uint32_t> sets;
        std::vector<uint16_t> bn;
        make_pmj02_sets(sets);
        load_bluenoise(bn);
        pmj_sets.upload(sets);
        bluenoise.upload(bn);
    }
};

struct akr_scene {
    akr_context* ctx = nullptr;
    FlatScene flat;
    CompiledScene cs;
    DevBuf light_alias, area_alias, lights;
    DevBuf woop, tri_gid, shade, normals, inst, materials, ggx_table, light_entries, light_pdf, light_inst, light_tri_offset,
        light_n_tris, area_entries, area_pdf, inst_tri_offset, bvh_nodes, tex_nodes, tex_images, tex_texels, tex_mat_inputs;
    std::vector<float> ggx_host;
    // materials / node lists / raw inputs re-compiled for a non-default colour pipeline (akr_pt_config.color), by pipeline
    struct ColorSet {
        DevBuf materials, tex_nodes, mat_inputs;
    };
    std::map<uint32_t, std::unique_ptr<ColorSet>> color_sets;
    std::mutex color_sets_mutex;  // sessions of several host threads may begin on one scene; entries are never removed before the scene dies
    // the scene's shader kinds as kernel text (host/specialise.cpp), made when the first session asks for a per-scene kernel
    std::string spec_header;
    bool spec_header_made = false;
    std::mutex spec_mutex;
    DScene dscene;
    float r2c[16], c2w[16];
    uint32_t c2w_identity = 0;
    uint64_t device_bytes = 0;
};

struct akr_film {
    akr_context* ctx = nullptr;
    uint32_t width = 0, height = 0;
    DevBuf own;
    float* data = nullptr;  // 7 * W * H floats
    float splat_scale = 1.0f;  // Film.splat_scale, film.rs:73,117
    size_t n_floats() const { return 7ull * width * height; }
};

struct akr_pt_session {
    akr_context* ctx = nullptr;
    akr_scene* scene = nullptr;
    akr_film* film = nullptr;
    akr_pt_config cfg;
    DevBuf states, counters;
    // wavefront schedule (wf_kernels.hip): path state SoA + ray queues
    bool wavefront = false;
    DevBuf wf_state, wf_queues, wf_ctrl;
    // option wf_sort: keys of the queue entries, the sorted copies the trace kernel reads, rocPRIM's scratch
    bool wf_sort = false;
    DevBuf wf_keys, wf_sorted, wf_sort_tmp;
    uint32_t *wf_sorted_closest = nullptr, *wf_sorted_shadow = nullptr, *wf_sorted_keys = nullptr;
    WfBuffers wf;
    uint32_t wf_slots = 0, wf_trace_blocks = 0;
    uint32_t spp_done = 0, n_launches = 0;
    uint64_t passes_launched = 0;  // passes of all akr_pt_passes launches so far (kernel_ms / passes_launched = what a pass costs)
    uint32_t pmj_spp = 1;  // the spp the pmj02bn sampler stratifies for (the method's total spp)
    const akr_scene::ColorSet* color_set = nullptr;  // the scene's tables for cfg.color != 0 (looked up under the scene's lock by akr_pt_begin)
    // the process-wide tuning options as they were when the session began (akr_pt_begin): an akr_option_set from another thread
    // cannot change the kernel of a running session
    int defer_metal_option = -1;
    int simple_kernels_option = 1;
    int defer_on_option = 0;
    int max_fused_option = 0;
    // per-scene kernel (host/specialise.cpp): set by akr_pt_begin when the options ask for one and the compile succeeded; the
    // precompiled interpreter kernel otherwise. spec_active also shapes fill_params (no value slots in LDS, the kernel's own LDS budget).
    bool spec_active = false;
    int spec_waves = 3;
    std::shared_ptr<SpecKernel> spec;
    std::string spec_status = "not requested";
    // timed regions on the context's stream: pairs still in flight, and the elapsed time of the completed ones (folded in and
    // destroyed as they complete, so a long progressive session holds a bounded number of events)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double kernel_ms = 0.0;
    PtParams params;
    void fold_events(bool all) {  // all: the stream has been synchronised
        size_t keep = 0;
        for (size_t i = 0; i < pending.size(); i++) {
            auto& ev = pending[i];
            if (all || hipEventQuery(ev.second) == hipSuccess) {
                float t = 0.0f;
                if (hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) kernel_ms += t;
                (void)hipEventDestroy(ev.first);
                (void)hipEventDestroy(ev.second);
            } else {
                pending[keep++] = ev;
            }
        }
        pending.resize(keep);
    }
    ~akr_pt_session() {
        for (auto& ev : pending) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    }
};

namespace {
// One timed region on a session's stream. The event pair is handed to the session by stop(); if the region is left by an
// exception the pair is destroyed here.
struct LaunchTimer {
    akr_pt_session* se;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    explicit LaunchTimer(akr_pt_session* s) : se(s) {
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, se->ctx->stream));
    }
    LaunchTimer(const LaunchTimer&) = delete;
    LaunchTimer& operator=(const LaunchTimer&) = delete;
    void stop() {
        HIP_CHECK(hipEventRecord(e1, se->ctx->stream));
        se->pending.emplace_back(e0, e1);
        e0 = e1 = nullptr;
        if (se->pending.size() > 16) se->fold_events(false);
    }
    ~LaunchTimer() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};
}  // namespace

namespace akr {
// for host/comm.cpp (the RCCL film reduce): what it needs to know about a film, and the shared error slot
int32_t film_device_view(akr_film* film, int* device, hipStream_t* stream, float** data, size_t* n_floats) {
    if (!film || !film->ctx) return fail(AKR_ERR_INVALID_ARGUMENT, "film is NULL");
    *device = film->ctx->device;
    *stream = film->ctx->stream;
    *data = film->data;
    *n_floats = film->n_floats();
    return AKR_OK;
}
int32_t api_fail(int32_t code, const std::string& msg) { return fail(code, msg); }
}  // namespace akr

static void ensure_ggx_table(akr_scene* s) {
    akr_context* ctx = s->ctx;
    if (!s->flat.ggx_table.empty()) {
        s->ggx_host = s->flat.ggx_table;
    } else if (s->cs.needs_ggx_table) {
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
        return;
    } else {
        s->ggx_host.assign(4096, 0.0f);  // never read with a non-zero weight
    }
    s->ggx_table.upload(s->ggx_host);
}

static void scene_finish(akr_scene* s) {
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
    s->bvh_nodes.upload(cs.bvh_nodes);
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
    d.n_nodes = (uint32_t)(cs.bvh_nodes.size() / kBvhNodeWords);
    d.has_alpha = cs.has_alpha ? 1u : 0u;
    d.bvh_stack_depth = std::max(1u, std::min(cs.bvh_depth, kBvhStackDepth));  // one pending group per tree level at most (disect.h)
    d.plane_share_mask = 0;
    if (cs.bvh_nodes.empty())  // exhaustive path (<= 64 triangles): which records repeat their predecessor's plane row
        for (uint32_t k = 1; k < d.n_tris && k < 64; k++)
            if (std::memcmp(&cs.woop[12ull * k + 8], &cs.woop[12ull * (k - 1) + 8], 16) == 0) d.plane_share_mask |= 1ull << k;
    if (cs.has_textures) {
        d.tex.nodes = s->tex_nodes.as<DNode>();
        d.tex.images = s->tex_images.as<DImage>();
        d.tex.texels = s->tex_texels.as<uint32_t>();
        d.tex.mat_inputs = s->tex_mat_inputs.as<MatInputs>();
    }
    s->device_bytes = 0;
    for (const DevBuf* b : {&s->woop, &s->tri_gid, &s->shade, &s->normals, &s->inst, &s->materials, &s->ggx_table, &s->light_entries,
                            &s->light_pdf, &s->light_inst, &s->light_tri_offset, &s->light_n_tris, &s->area_entries, &s->area_pdf,
                            &s->inst_tri_offset, &s->light_alias, &s->area_alias, &s->lights, &s->bvh_nodes, &s->tex_nodes, &s->tex_images, &s->tex_texels, &s->tex_mat_inputs})
        s->device_bytes += b->bytes;
}

static void fill_params(akr_pt_session* se, uint32_t n_passes, uint32_t last_pass_spp) {
    PtParams& p = se->params;
    const akr_scene* s = se->scene;
    const akr_pt_config& c = se->cfg;
    std::memset(&p, 0, sizeof p);
    p.sc = s->dscene;
    std::memcpy(p.r2c, s->r2c, 64);
    std::memcpy(p.c2w, s->c2w, 64);
    p.c2w_identity = s->c2w_identity;
    p.width = s->flat.camera.width;
    p.height = s->flat.camera.height;
    p.max_depth = c.max_depth;
    p.rr_depth = c.rr_depth;
    p.use_nee = c.use_nee;
    p.indirect_only = c.indirect_only;
    p.force_diffuse = c.force_diffuse;
    p.debug_depth = c.debug_depth;
    p.pixel_offset[0] = c.pixel_offset[0];
    p.pixel_offset[1] = c.pixel_offset[1];
    p.filter_type = c.filter_type;
    p.filter_radius = c.filter_radius;
    p.pass_spp = c.spp_per_pass;
    p.n_passes = n_passes;
    p.last_pass_spp = last_pass_spp;
    p.start = pcg_start_constants();
    p.states = se->states.as<Pcg32>();
    p.film = se->film->data;
    p.counters = se->counters.as<uint64_t>();
    p.color = c.color;
    p.sc.tex.color = c.color;
    if (c.color != 0) {  // the material tables of this pipeline (created by akr_pt_begin)
        const auto& set = *se->color_set;
        p.sc.materials = set.materials.as<DMaterial>();
        if (s->cs.has_textures) {
            p.sc.tex.nodes = set.tex_nodes.as<DNode>();
            p.sc.tex.mat_inputs = set.mat_inputs.as<MatInputs>();
        }
    }
    p.sampler = c.sampler_type;
    if (c.sampler_type == AKR_SAMPLER_PMJ02BN || c.sampler_type == AKR_SAMPLER_SOBOL) {  // Pmj02BnSamplerCreator::new (sampler/mod.rs:376-395)
        p.smp_seed = (uint32_t)c.sampler_seed;
        p.smp_spp = se->pmj_spp;
        uint32_t w = se->pmj_spp - 1;
        w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
        p.smp_w = w;
        p.smp_mod_magic = fastmod_magic(se->pmj_spp);
        if (c.sampler_type == AKR_SAMPLER_PMJ02BN) {  // the sobol sampler computes its points, no tables
            p.pmj_sets = se->ctx->pmj_sets.as<uint32_t>();
            p.bluenoise = se->ctx->bluenoise.as<uint16_t>();
        }
    }
    {  // LDS staging of the tables the shading phase gathers from (pt_kernels.hip: STAGE)
        const CompiledScene& cs = s->cs;
        const bool bvh = !cs.bvh_nodes.empty();
        // exhaustive path: everything, per-triangle records included (scene_build.cpp guarantees the fit);
        // BVH path: the per-scene tables only, if they fit beside the traversal stacks
        size_t bytes[13] = {bvh ? 0 : cs.shade.size() * 4, bvh ? 0 : cs.normals.size() * 4, cs.inst.size() * 4, cs.materials.size() * sizeof(DMaterial),
                            (size_t)cs.n_lights * sizeof(AliasPacked), cs.area_entries.size() * sizeof(AliasPacked), (size_t)cs.n_lights * sizeof(LightRec),
                            cs.light_pdf.size() * 4, cs.area_pdf.size() * 4, 0, 0, 0, 0};
        if (cs.has_textures) {  // the node lists have the same size in every colour pipeline
            bytes[9] = cs.tex_nodes.size() * sizeof(DNode);
            bytes[10] = cs.images.size() * sizeof(DImage);
            bytes[11] = cs.mat_inputs.size() * sizeof(MatInputs);
        }
        size_t total = 0;
        for (int i = 0; i < 12; i++) total += (bytes[i] + 15) & ~(size_t)15;
        std::memset(p.stage_bytes, 0, sizeof p.stage_bytes);
        p.stage_total = 0;
        p.tex_slots = (cs.has_textures && !se->spec_active) ? cs.tex_slots : 0;  // a per-scene kernel keeps node values in registers
        // one workgroup's dynamic LDS stays within 64 KB: traversal stacks + staged tables + the graph evaluation's value slots
        // what the launch keeps in LDS besides the staged tables: traversal stacks, graph values, and the columns / records of pt_lds_plan
        const PtLdsPlan plan = pt_lds_plan(bvh, c.force_diffuse != 0, cs.has_textures, /*defer: the larger park block*/ true, cs.n_tris);
        const size_t other = (bvh ? (size_t)p.sc.bvh_stack_depth * 256 * 4 : 0) + (size_t)p.tex_slots * kTexValStride * sizeof(TexVal) + plan.recs_bytes +
                             plan.park_bytes + plan.carry_bytes;
        const size_t lds_budget = (se->spec_active && se->spec_waves >= 4) ? pt_lds_budget(false) : pt_lds_budget(cs.has_textures);
        if (total <= (bvh ? kStageMaxBytesBvh : kStageMaxBytes) && (!bvh || other + total <= lds_budget)) {  // all of it or nothing (a TEX kernel reads its tables through LDS addresses)
            // the albedo table as well for the full-graph exhaustive kernel of a textured scene (stage_scene_tables: GGX), if three
            // workgroups per CU still fit (AKR_PT_MIN_WAVES_TEX = 3: 160 KB / 3)
            const size_t ggx_bytes = 4096 * sizeof(float);
            if (!bvh && cs.has_textures && !c.force_diffuse && other + total + ggx_bytes <= lds_budget) {
                bytes[12] = ggx_bytes;
                total += ggx_bytes;
            }
            for (int i = 0; i < 13; i++) p.stage_bytes[i] = (uint32_t)bytes[i];
            p.stage_total = (uint32_t)std::max<size_t>(total, 16);
        }
    }
    {   // SIMPLE instantiations (dbsdf.h principled_eval): the reference traces its kernel from the scene's shader graphs, so a scene
        // without coat / transmission / normal map / glass runs a kernel without that code there too. The conditions are on the
        // folded VALUES (coat_weight and transmission exactly 0), which is what makes dropping the branches exact.
        const CompiledScene& cs = s->cs;
        bool simple = !cs.has_textures;
        for (const DMaterial& m : cs.materials) {
            if (m.kind == MAT_GLASS) simple = false;
            if (m.kind == MAT_PRINCIPLED && ((m.flags & (MF_COAT | MF_EVAL_DIEL | MF_NORMAL_MAP)) != 0 || m.transmission != 0.0f || m.coat_weight != 0.0f)) simple = false;
        }
        p.simple_scene = (simple && se->simple_kernels_option) ? 1u : 0u;
    }
    {   // hits on "expensive" materials on even iterations only (pt_kernels.hip: DEFER): pays when SOME materials are expensive and
        // most hits are not. Expensive = the conductor lobe; in the BVH kernels of scenes with textures (option defer_on) also /
        // instead a shader graph to evaluate at the hit.
        const CompiledScene& cs = s->cs;
        const bool bvh = !cs.bvh_nodes.empty();
        uint32_t flags = MF_EVAL_METAL;
        // (measured on the textured room, BVH kernel: conductor hits deferred 591 Msamples/s, textured hits 573, both 573, none 544)
        if (bvh && cs.has_textures) flags = se->defer_on_option == 2 ? MF_TEXTURED : (se->defer_on_option == 3 ? (MF_EVAL_METAL | MF_TEXTURED) : MF_EVAL_METAL);
        size_t n_dear = 0, n_surface = 0;
        for (const DMaterial& m : cs.materials) {
            if (m.kind == MAT_EMISSION) continue;
            n_surface++;
            if (m.flags & flags) n_dear++;  // the kernel's own test (pt_pass.h: DEFER), whatever the material's kind
        }
        bool want = n_dear > 0 && 2 * n_dear <= n_surface;
        uint32_t mask = 1u;  // iterations with (iteration & mask) != 0 put those hits off
        if (se->defer_metal_option >= 0) { mask = (uint32_t)se->defer_metal_option; want = mask != 0; }  // akr_option_set("defer_metal"): measurements / tests
        p.defer_metal = (want && (!bvh || cs.has_textures) && !c.force_diffuse) ? mask : 0u;
        p.defer_flags = flags;
    }
    p.wf_sort = se->wf_sort ? 1u : 0u;
    for (int a = 0; a < 3; a++) {  // the sort key's grid: 128 cells per axis over the scene's box
        const float lo = s->cs.scene_lo[a], ext = s->cs.scene_hi[a] - s->cs.scene_lo[a];
        p.sort_lo[a] = lo;
        p.sort_scale[a] = ext > 0.0f ? 128.0f / ext : 0.0f;
    }
    p.shard_rank = c.shard_count > 1 ? c.shard_rank : 0;
    p.shard_count = c.shard_count > 1 ? c.shard_count : 1;
    p.tile_w = c.tile_w ? c.tile_w : 32;
    p.tile_h = c.tile_h ? c.tile_h : 32;
    p.tiles_x = (p.width + p.tile_w - 1) / p.tile_w;
    p.tiles_y = (p.height + p.tile_h - 1) / p.tile_h;
    uint32_t n_tiles = p.tiles_x * p.tiles_y;
    uint32_t owned = p.shard_rank < n_tiles ? (n_tiles - p.shard_rank + p.shard_count - 1) / p.shard_count : 0;
    p.n_items = owned * p.tile_w * p.tile_h;
}

// Which schedule renders this session: "mega" = persistent-lane megakernel (pt_kernels.hip), "wavefront" = trace /
// shade kernels with the path state in HBM (wf_kernels.hip; needs a BVH scene). AKR_PT_MODE selects; the default is the
// megakernel, which measured faster on every configuration so far (on the 10 M-triangle hall both
// schedules trace 3.3 - 3.7 G rays/s -- the traversal is bound by the memory system's rate for random 64-byte records, not by
// occupancy -- and the wavefront schedule pays for streaming the path state and for its per-iteration tail on top).
// The option is process-wide and aov / gpt / mcmc_opt sessions come through here too: a scene without a BVH (64 triangles or
// fewer, no force_bvh) has no wavefront kernels and renders with the megakernel whatever the option says.
static bool choose_wavefront(const akr_scene* scene) {
    if (!tuning().wavefront) return false;
    return !scene->cs.bvh_nodes.empty();
}

static void wf_allocate(akr_pt_session* se, uint32_t n_slots) {
    se->wf_slots = n_slots;
    const size_t n = n_slots ? n_slots : 1;
    se->wf_state.alloc(12 * n * 16);  // 10 float4 + 2 uint4 arrays
    char* base = (char*)se->wf_state.p;
    auto take = [&](size_t k) { void* p = base + k * n * 16; return p; };
    WfBuffers& w = se->wf;
    w.ray_o = (float4*)take(0); w.ray_d = (float4*)take(1); w.sh_o = (float4*)take(2); w.sh_d = (float4*)take(3);
    w.sh_c = (float4*)take(4); w.hit = (float4*)take(5); w.beta = (float4*)take(6); w.rad = (float4*)take(7);
    w.base = (float4*)take(8); w.film = (float4*)take(9); w.rng = (uint4*)take(10); w.misc = (uint4*)take(11);
    se->wf_queues.alloc(4 * n * sizeof(uint32_t));
    uint32_t* q = (uint32_t*)se->wf_queues.p;
    w.queue_closest[0] = q; w.queue_closest[1] = q + n; w.queue_shadow[0] = q + 2 * n; w.queue_shadow[1] = q + 3 * n;
    w.key_closest[0] = w.key_closest[1] = w.key_shadow[0] = w.key_shadow[1] = nullptr;
    if (se->wf_sort) {
        se->wf_keys.alloc(4 * n * sizeof(uint32_t));
        uint32_t* k = (uint32_t*)se->wf_keys.p;
        w.key_closest[0] = k; w.key_closest[1] = k + n; w.key_shadow[0] = k + 2 * n; w.key_shadow[1] = k + 3 * n;
        se->wf_sorted.alloc(3 * n * sizeof(uint32_t));
        se->wf_sorted_closest = (uint32_t*)se->wf_sorted.p;
        se->wf_sorted_shadow = se->wf_sorted_closest + n;
        se->wf_sorted_keys = se->wf_sorted_closest + 2 * n;
        se->wf_sort_tmp.alloc(wf_sort_temp_bytes((uint32_t)n));
    }
    se->wf_ctrl.alloc(8 * sizeof(uint32_t));
    uint32_t* c = (uint32_t*)se->wf_ctrl.p;
    w.qcount = c; w.qhead = c + 4; w.n_active = c + 5;
    // persistent trace kernel: as many workgroups as the CUs hold at once (occupancy API: registers + this tree's LDS stacks)
    se->wf_trace_blocks = (uint32_t)se->ctx->props.multiProcessorCount * wf_trace_blocks_per_cu(se->params);
}

// One launch group of the wavefront schedule = `fused` passes for every slot: init, then trace/shade iterations until
// no slot is active. The host only looks at the device every kCheckEvery iterations.
static void wf_run(akr_pt_session* se) {
    hipStream_t st = se->ctx->stream;
    const PtParams& p = se->params;
    uint32_t* ctrl = (uint32_t*)se->wf_ctrl.p;
    HIP_CHECK(hipMemsetAsync(ctrl, 0, 8 * sizeof(uint32_t), st));
    HIP_CHECK(launch_wf_init(p, se->wf, st));
    const int kCheckEvery = 16;
    uint32_t q = 0;
    for (uint64_t iter = 0;; iter++) {
        // option wf_sort: the sizes of the queue this iteration traces (rocPRIM wants the element count on the host): one small read-back
        // per iteration, before the counters are reset -- it also ends the loop the moment the last path has finished
        uint32_t nc = 0, ns = 0;
        const bool sort_now = se->wf_sort && iter > 0;  // (the first iteration's camera rays are in pixel order: coherent as they are)
        if (sort_now) {
            uint32_t counts[6];
            HIP_CHECK(hipMemcpyAsync(counts, ctrl, sizeof counts, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (counts[5] == 0) break;  // n_active after the last shade
            nc = counts[2 * q];
            ns = counts[2 * q + 1];
        }
        // queue q holds the rays to trace; reset the head, the other queue's counts and the active counter
        HIP_CHECK(hipMemsetAsync(ctrl + 2 * (1 - q), 0, 2 * sizeof(uint32_t), st));
        HIP_CHECK(hipMemsetAsync(ctrl + 4, 0, 2 * sizeof(uint32_t), st));
        if (sort_now) {
            WfBuffers sorted = se->wf;
            HIP_CHECK(wf_sort_pairs(se->wf_sort_tmp.p, se->wf_sort_tmp.bytes, se->wf.key_closest[q], se->wf_sorted_keys, se->wf.queue_closest[q], se->wf_sorted_closest, nc, st));
            HIP_CHECK(wf_sort_pairs(se->wf_sort_tmp.p, se->wf_sort_tmp.bytes, se->wf.key_shadow[q], se->wf_sorted_keys, se->wf.queue_shadow[q], se->wf_sorted_shadow, ns, st));
            sorted.queue_closest[q] = se->wf_sorted_closest;
            sorted.queue_shadow[q] = se->wf_sorted_shadow;
            HIP_CHECK(launch_wf_trace(p, sorted, q, se->wf_trace_blocks, st));
        } else {
            HIP_CHECK(launch_wf_trace(p, se->wf, q, se->wf_trace_blocks, st));
        }
        HIP_CHECK(launch_wf_shade(p, se->wf, 1 - q, st));
        q = 1 - q;
        if (!se->wf_sort && (iter + 1) % kCheckEvery == 0) {
            uint32_t n_active = 0;
            HIP_CHECK(hipMemcpyAsync(&n_active, ctrl + 5, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (n_active == 0) break;
        }
        if (iter > (1ull << 26)) throw RenderError("wavefront schedule did not terminate");
    }
}

static void validate_config(const akr_pt_config& c) {
    if (c.spp_per_pass == 0) throw std::invalid_argument("akr_pt_config: spp_per_pass must be > 0");
    if (c.filter_type > AKR_FILTER_GAUSSIAN) throw std::invalid_argument("akr_pt_config: unknown filter_type");
    if (c.sampler_type > AKR_SAMPLER_SOBOL) throw std::invalid_argument("akr_pt_config: unknown sampler_type");
    if (c.color > (AKR_COLOR_REPR_ACESCG | AKR_COLOR_RGB_ACESCG)) throw std::invalid_argument("akr_pt_config: unknown colour pipeline bits");
    if (c.sampler_type == AKR_SAMPLER_PMJ02BN && c.spp > 65536u)
        throw std::invalid_argument("Pmj02BnSampler supports up to 65536 spp (sampler/mod.rs:381-387)");
    uint32_t tw = c.tile_w ? c.tile_w : 32, th = c.tile_h ? c.tile_h : 32;
    if ((tw % 8) || (th % 8)) throw std::invalid_argument("akr_pt_config: tile_w and tile_h must be multiples of 8");
    if (c.shard_count > 1 && c.shard_rank >= c.shard_count) throw std::invalid_argument("akr_pt_config: shard_rank >= shard_count");
    if (c.sample_begin != 0 || c.sample_count != 0) {  // sample-range split (akari_hip.h)
        if (c.sampler_type != AKR_SAMPLER_PMJ02BN && c.sampler_type != AKR_SAMPLER_SOBOL)
            throw Unsupported("akr_pt_config: a sample range needs an index-based sampler (pmj02bn, sobol): the independent sampler's start() advances the pixel's "
                              "PCG stream from wherever the previous sample stopped (sampler/mod.rs:115-131,192-203), sample s cannot be drawn without samples 0 .. s-1");
        if (c.sample_count == 0) throw std::invalid_argument("akr_pt_config: sample_begin without sample_count");
        if ((uint64_t)c.sample_begin + c.sample_count > c.spp) throw std::invalid_argument("akr_pt_config: sample range exceeds spp");
    }
}
// samples the session renders: the configured range, or all spp of the render
static uint32_t session_samples(const akr_pt_config& c) { return c.sample_count ? c.sample_count : c.spp; }

extern "C" {

AKR_API const char* akr_last_error(void) { return g_last_error.c_str(); }
AKR_API const char* akr_version(void) { return "akari_hip 0.2.0 gfx950"; }  // 0.2.0: akr_pt_config gained sample_begin / sample_count (88 bytes); akr_kernel_info carries its own size
AKR_API int32_t akr_option_set(const char* name, int32_t value) {
    if (!tuning_set(name, value)) return fail(AKR_ERR_INVALID_ARGUMENT, std::string("akr_option_set: unknown option '") + (name ? name : "(null)") + "' or value out of range");
    return AKR_OK;
}
AKR_API int32_t akr_option_get(const char* name, int32_t* value) {
    int v = 0;
    if (!value || !tuning_get(name, &v)) return fail(AKR_ERR_INVALID_ARGUMENT, std::string("akr_option_get: unknown option '") + (name ? name : "(null)") + "'");
    *value = v;
    return AKR_OK;
}

AKR_API int32_t akr_context_create(int32_t device, akr_context** out) {
    if (!out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_context_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(AKR_ERR_NO_DEVICE, std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "0 devices") +
                                           "); libakari_hip has no CPU path");
    if (device < 0 || device >= count) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_context_create: device ordinal out of range");
    return guarded([&] {
        auto ctx = std::make_unique<akr_context>();
        ctx->device = device;
        ctx->bind();
        HIP_CHECK(hipGetDeviceProperties(&ctx->props, device));
        HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        *out = ctx.release();
    });
}
AKR_API int32_t akr_context_destroy(akr_context* ctx) {
    if (!ctx) return AKR_OK;
    return guarded([&] {
        (void)hipSetDevice(ctx->device);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
    });
}
AKR_API int32_t akr_context_synchronize(akr_context* ctx) {
    if (!ctx) return fail(AKR_ERR_INVALID_ARGUMENT, "context is NULL");
    return guarded([&] {
        ctx->bind();
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}
AKR_API int32_t akr_context_device_ordinal(akr_context* ctx, int32_t* device) {
    if (!ctx || !device) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_context_device_ordinal: NULL argument");
    *device = ctx->device;
    return AKR_OK;
}
AKR_API int32_t akr_context_device_info(akr_context* ctx, char* name, uint32_t name_len, uint32_t* compute_units, uint64_t* hbm_bytes) {
    if (!ctx) return fail(AKR_ERR_INVALID_ARGUMENT, "context is NULL");
    if (name && name_len) {
        std::snprintf(name, name_len, "%s (%s)", ctx->props.name, ctx->props.gcnArchName);
    }
    if (compute_units) *compute_units = (uint32_t)ctx->props.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)ctx->props.totalGlobalMem;
    return AKR_OK;
}

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
    info->n_bvh_nodes = (uint32_t)(s->cs.bvh_nodes.size() / kBvhNodeWords);
    info->uses_bvh = s->cs.bvh_nodes.empty() ? 0u : 1u;
    info->device_bytes = s->device_bytes;
    info->node_bytes = s->cs.bvh_nodes.empty() ? 0u : kBvhNodeWords * 4u;   // bytes a traversal reads per node visit
    info->node_stride_bytes = s->cs.bvh_nodes.empty() ? 0u : kBvhNodeWords * 4u;
    info->tri_bytes = s->cs.bvh_nodes.empty() ? 48u : kBvhTriWords * 4u;
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
        case AKR_ARRAY_BVH_NODES: set(cs.bvh_nodes.data(), cs.bvh_nodes.size() * 4); break;
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
        default: return fail(AKR_ERR_INVALID_ARGUMENT, "akr_scene_get_array: unknown array id");
    }
    return AKR_OK;
}

AKR_API int32_t akr_film_create(akr_context* ctx, uint32_t width, uint32_t height, akr_film** out) {
    if (!ctx || !out || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_create: bad argument");
    *out = nullptr;
    return guarded([&] {
        ctx->bind();
        auto f = std::make_unique<akr_film>();
        f->ctx = ctx;
        f->width = width;
        f->height = height;
        f->own.alloc(f->n_floats() * sizeof(float));
        f->data = f->own.as<float>();
        HIP_CHECK(hipMemsetAsync(f->data, 0, f->n_floats() * sizeof(float), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = f.release();
    });
}
AKR_API int32_t akr_film_wrap(akr_context* ctx, uint32_t width, uint32_t height, void* device_ptr, akr_film** out) {
    if (!ctx || !out || !width || !height || !device_ptr) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_wrap: bad argument");
    *out = nullptr;
    {   // The gpt / mcmc_opt kernels splat with hardware float atomics (global_atomic_add_f32), which CDNA silently drops on
        // host-mapped, managed or fine-grained memory: only plain device allocations (hipMalloc) of this context's GPU pass.
        hipPointerAttribute_t attr;
        hipError_t e = hipPointerGetAttributes(&attr, device_ptr);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_wrap: device_ptr is not a HIP allocation");
        }
        if (attr.type != hipMemoryTypeDevice || attr.isManaged)
            return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_wrap: device_ptr must be plain device memory (hipMalloc), not host-mapped or managed memory");
        if (attr.device != ctx->device) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_wrap: device_ptr belongs to another GPU than the context");
    }
    auto* f = new (std::nothrow) akr_film();
    if (!f) return fail(AKR_ERR_OUT_OF_MEMORY, "out of host memory");
    f->ctx = ctx;
    f->width = width;
    f->height = height;
    f->data = (float*)device_ptr;
    *out = f;
    return AKR_OK;
}
AKR_API int32_t akr_film_destroy(akr_film* film) {
    if (!film) return AKR_OK;
    return guarded([&] {
        (void)hipSetDevice(film->ctx->device);
        delete film;
    });
}
AKR_API int32_t akr_film_clear(akr_film* f) {
    if (!f) return fail(AKR_ERR_INVALID_ARGUMENT, "film is NULL");
    return guarded([&] {
        f->ctx->bind();
        HIP_CHECK(hipMemsetAsync(f->data, 0, f->n_floats() * sizeof(float), f->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    });
}
AKR_API int32_t akr_film_read(akr_film* f, float* dst) {
    if (!f || !dst) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_read: NULL argument");
    return guarded([&] {
        f->ctx->bind();
        HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
        HIP_CHECK(hipMemcpy(dst, f->data, f->n_floats() * sizeof(float), hipMemcpyDeviceToHost));
    });
}
AKR_API int32_t akr_film_write(akr_film* f, const float* src) {
    if (!f || !src) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_write: NULL argument");
    return guarded([&] {
        f->ctx->bind();
        HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
        HIP_CHECK(hipMemcpy(f->data, src, f->n_floats() * sizeof(float), hipMemcpyHostToDevice));
    });
}
AKR_API int32_t akr_film_resolve(akr_film* f, float* dst_rgb) {
    if (!f || !dst_rgb) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_resolve: NULL argument");
    return guarded([&] {
        f->ctx->bind();
        uint64_t n = (uint64_t)f->width * f->height;
        DevBuf tmp;
        tmp.alloc(3 * n * sizeof(float));
        HIP_CHECK(launch_film_resolve(f->data, n, f->splat_scale, tmp.as<float>(), f->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
        HIP_CHECK(hipMemcpy(dst_rgb, tmp.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    });
}
AKR_API int32_t akr_film_set_splat_scale(akr_film* f, float scale) {  // film.rs:152-154
    if (!f) return fail(AKR_ERR_INVALID_ARGUMENT, "film is NULL");
    f->splat_scale = scale;
    return AKR_OK;
}
AKR_API int32_t akr_film_get_splat_scale(const akr_film* f, float* scale) {
    if (!f || !scale) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_film_get_splat_scale: NULL argument");
    *scale = f->splat_scale;
    return AKR_OK;
}
AKR_API int32_t akr_film_device_ptr(akr_film* f, void** ptr, uint64_t* bytes) {
    if (!f) return fail(AKR_ERR_INVALID_ARGUMENT, "film is NULL");
    if (ptr) *ptr = f->data;
    if (bytes) *bytes = f->n_floats() * sizeof(float);
    return AKR_OK;
}

AKR_API int32_t akr_pt_config_default(akr_pt_config* c) {
    if (!c) return fail(AKR_ERR_INVALID_ARGUMENT, "config is NULL");
    std::memset(c, 0, sizeof *c);
    c->spp = 256; c->max_depth = 7; c->rr_depth = 5; c->spp_per_pass = 64;  // pt.rs:930-944
    c->use_nee = 1; c->indirect_only = 0; c->force_diffuse = 0;
    c->debug_depth = -1;
    c->filter_type = AKR_FILTER_GAUSSIAN; c->filter_radius = 1.5f;          // film.rs:50-54
    c->sampler_type = AKR_SAMPLER_INDEPENDENT; c->sampler_seed = 0;         // sampler/mod.rs:290-294
    c->shard_rank = 0; c->shard_count = 1; c->tile_w = 32; c->tile_h = 32;
    return AKR_OK;
}
AKR_API int32_t akr_pt_config_from_json(const char* text, akr_pt_config* cfg, char* film_out, uint32_t film_out_len) {
    if (!text || !cfg) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_pt_config_from_json: NULL argument");
    return guarded([&] {
        std::string out;
        parse_method_json(text, cfg, &out);
        if (film_out && film_out_len) std::snprintf(film_out, film_out_len, "%s", out.c_str());
    });
}

AKR_API int32_t akr_pt_begin(akr_context* ctx, akr_scene* scene, const akr_pt_config* cfg, akr_film* film, akr_pt_session** out) {
    if (!ctx || !scene || !cfg || !film || !out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_pt_begin: NULL argument");
    *out = nullptr;
    if (scene->ctx != ctx) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_pt_begin: the scene was not created on this context (host-only scenes cannot render)");
    return guarded([&] {
        validate_config(*cfg);
        if (film->width != scene->flat.camera.width || film->height != scene->flat.camera.height)
            throw std::invalid_argument("film resolution does not match the scene camera (pt.rs:1072-1073)");
        ctx->bind();
        auto se = std::make_unique<akr_pt_session>();
        se->ctx = ctx;
        se->scene = scene;
        se->film = film;
        se->cfg = *cfg;
        std::unique_lock<std::mutex> color_lock(scene->color_sets_mutex);
        if (cfg->color != 0 && !scene->color_sets.count(cfg->color)) {
            // ColorPipeline other than sRGB / sRGB: the scene's constants were folded for the default pipeline; fold them again
            // for this one (svm/texture/mod.rs:9-43 at every Rgb / spectral_uplift node) and keep the tables with the scene
            CompiledScene tmp;
            tmp.images = scene->cs.images;
            std::vector<akr_material_desc> descs;
            compile_materials(scene->flat, cfg->color, tmp, descs);
            auto set = std::make_unique<akr_scene::ColorSet>();
            set->materials.upload(tmp.materials);
            if (tmp.has_textures) {
                set->tex_nodes.upload(tmp.tex_nodes);
                set->mat_inputs.upload(tmp.mat_inputs);
            }
            scene->color_sets[cfg->color] = std::move(set);
        }
        if (cfg->color != 0) se->color_set = scene->color_sets.at(cfg->color).get();  // stable: the map owns it through a unique_ptr
        color_lock.unlock();
        const uint64_t n = (uint64_t)film->width * film->height;
        // init_pcg32_buffer_with_seed (sampler/mod.rs:148-160): host StdRng(seed) u64 per pixel, device new_seq_offset
        if (cfg->sampler_type == AKR_SAMPLER_PMJ02BN || cfg->sampler_type == AKR_SAMPLER_SOBOL) {
            // Pmj02BnState per pixel (sampler/mod.rs:451-466): sample_index = u32::MAX, pixel = (x, y), kept in a Pcg32 slot
            if (cfg->sampler_type == AKR_SAMPLER_PMJ02BN) ctx->ensure_pmj_tables();
            se->pmj_spp = cfg->spp ? cfg->spp : 1;
            std::vector<Pcg32> init(n);
            // a sample range [b, ..) starts with sample_index = b - 1: the next start() makes it b (sampler/mod.rs:650-663)
            const uint64_t first = cfg->sample_begin ? (uint64_t)(cfg->sample_begin - 1u) : 0xffffffffull;
            for (uint64_t i = 0; i < n; i++) init[i] = Pcg32{first, (i % film->width) | ((i / film->width) << 32)};
            se->states.upload(init);
        } else {
            std::vector<uint64_t> seeds(n);
            StdRng rng(cfg->sampler_seed);
            for (auto& v : seeds) v = rng.next_u64();
            DevBuf dseeds;
            dseeds.upload(seeds);
            se->states.alloc(n * sizeof(Pcg32));
            HIP_CHECK(launch_init_pcg32(dseeds.as<uint64_t>(), se->states.p, n, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));  // dseeds goes out of scope
        }
        se->counters.alloc(8 * kStatStripes * sizeof(uint64_t));
        HIP_CHECK(hipMemsetAsync(se->counters.p, 0, se->counters.bytes, ctx->stream));
        se->wavefront = choose_wavefront(scene);
        se->wf_sort = se->wavefront && tuning().wf_sort != 0;
        {
            const TuningOptions t = tuning();
            se->defer_metal_option = t.defer_metal;
            se->simple_kernels_option = t.simple_kernels;
            se->defer_on_option = t.defer_on;
            se->max_fused_option = t.max_fused_passes;
            // A per-scene kernel (host/specialise.cpp) for the megakernel of a scene with texture-fed materials: always / never by
            // option, else when the render is long enough for a first-use compile to pay.
            const uint64_t samples = n * (uint64_t)session_samples(*cfg);
            // (automatic: a kernel that is already cached is used whatever the render's size; a compile -- about a second -- only
            // when the render is long enough to win it back)
            const bool may_compile = t.specialise == 1 || samples >= kSpecAutoSamples;
            if (!scene->cs.has_textures) se->spec_status = "the scene has no texture-fed material";
            else if (t.specialise == 0) se->spec_status = "option specialise = 0";
            else if (cfg->force_diffuse) se->spec_status = "force_diffuse kernels evaluate no surface graphs";
            else if (se->wavefront) se->spec_status = "wavefront schedule";
            else {
                {
                    std::lock_guard<std::mutex> lock(scene->spec_mutex);
                    if (!scene->spec_header_made) {
                        scene->spec_header = generate_scene_spec(scene->cs);
                        scene->spec_header_made = true;
                    }
                }
                se->spec_waves = t.specialise_waves ? t.specialise_waves : 3;
                se->spec_active = true;
                fill_params(se.get(), 1, cfg->spp_per_pass);  // which instantiation the session's launches use
                SpecRequest rq;
                rq.bvh = !scene->cs.bvh_nodes.empty();
                rq.pmj = se->params.sampler != 0;
                rq.stage = se->params.stage_total != 0;
                rq.defer = se->params.defer_metal != 0;
                rq.min_waves = se->spec_waves;
                se->spec = ctx->spec_cache.get(scene->spec_header, rq, ctx->props.gcnArchName, may_compile);
                se->spec_status = se->spec->status;
                if (!se->spec->fn) se->spec_active = false;  // the interpreter kernel renders the same film
            }
        }
        if (se->wavefront) {
            fill_params(se.get(), 1, cfg->spp_per_pass);  // for n_items
            wf_allocate(se.get(), se->params.n_items);
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = se.release();
    });
}
AKR_API int32_t akr_pt_passes(akr_pt_session* se, uint32_t n_passes, int32_t blocking, uint32_t* spp_done) {
    if (!se) return fail(AKR_ERR_INVALID_ARGUMENT, "session is NULL");
    return guarded([&] {
        se->ctx->bind();
        // the passes requested (each min(spp - cnt, spp_per_pass) samples, pt.rs:1127) are fused into launches
        // of at most kMaxFusedPasses passes: 16, or -- once the session knows what a pass costs, i.e. when every earlier launch has
        // completed (a progressive render, a warm-up) -- as many as fit in about three seconds of kernel time, up to 64. The waves of
        // a launch do not finish together; fewer, longer launches spend less of a render in those tails (C2: +1.2 % at 64 passes,
        // profiles/r4_ab_walk.txt). The bound keeps a launch on a heavy scene from running for minutes.
        uint32_t kMaxFusedPasses = 16;
        se->fold_events(false);
        if (se->max_fused_option > 0) {
            kMaxFusedPasses = (uint32_t)se->max_fused_option;  // option max_fused_passes: a fixed bound (deterministic launch counts)
        } else if (blocking && se->pending.empty() && se->passes_launched > 0 && se->kernel_ms > 0.0) {
            // (blocking calls only: a progressive caller that polls between non-blocking calls is not put behind multi-second launches)
            const double per_pass_ms = se->kernel_ms / (double)se->passes_launched;
            const double fit = 3000.0 / per_pass_ms;
            kMaxFusedPasses = fit >= 64.0 ? 64u : (fit <= 16.0 ? 16u : (uint32_t)fit);
        }
        uint32_t left = n_passes;
        const uint32_t total = session_samples(se->cfg);
        while (left > 0 && se->spp_done < total) {
            uint32_t fused = 0, last = 0, done = se->spp_done;
            while (fused < kMaxFusedPasses && fused < left && done < total) {
                last = std::min(total - done, se->cfg.spp_per_pass);
                done += last;
                fused++;
            }
            fill_params(se, fused, last);
            LaunchTimer timer(se);
            if (se->wavefront) wf_run(se);
            else HIP_CHECK(launch_pt_pass(se->params, se->ctx->stream, se->spec_active ? se->spec->fn : nullptr));
            timer.stop();
            se->spp_done = done;
            se->n_launches++;
            se->passes_launched += fused;
            left -= fused;
        }
        if (blocking) HIP_CHECK(hipStreamSynchronize(se->ctx->stream));
        if (spp_done) *spp_done = se->spp_done;
    });
}
AKR_API int32_t akr_pt_read_sampler_states(akr_pt_session* se, uint64_t* dst) {
    if (!se || !dst) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_pt_read_sampler_states: NULL argument");
    return guarded([&] {
        se->ctx->bind();
        HIP_CHECK(hipStreamSynchronize(se->ctx->stream));
        HIP_CHECK(hipMemcpy(dst, se->states.p, se->states.bytes, hipMemcpyDeviceToHost));
    });
}
static void read_stats(akr_pt_session* se, akr_pt_stats* stats) {
    se->ctx->bind();
    HIP_CHECK(hipStreamSynchronize(se->ctx->stream));
    std::vector<uint64_t> stripes(8 * kStatStripes);
    HIP_CHECK(hipMemcpy(stripes.data(), se->counters.p, stripes.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint64_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = 0; k < kStatStripes; k++)
        for (int i = 0; i < 8; i++) c[i] += stripes[8 * k + i];
    se->fold_events(true);
    const double ms = se->kernel_ms;
    if (stats) {
        stats->n_samples = c[0];
        stats->n_closest = c[1];
        stats->n_shadow = c[2];
        stats->n_shaded = c[3];
        stats->n_node_visits = c[4];
        stats->n_tri_tests = c[5];
        stats->kernel_ms = ms;
        stats->n_launches = se->n_launches;
        stats->_pad = (uint32_t)c[6];  // non-zero = a traversal stack overflowed (results invalid)
    }
    if (c[6] != 0) throw RenderError("BVH traversal stack overflow: the render is incomplete");
}
AKR_API int32_t akr_pt_get_stats(akr_pt_session* se, akr_pt_stats* stats) {
    if (!se || !stats) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_pt_get_stats: NULL argument");
    return guarded([&] { read_stats(se, stats); });
}
AKR_API int32_t akr_pt_kernel_info(akr_pt_session* se, akr_kernel_info* info) {
    if (!se || !info) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_pt_kernel_info: NULL argument");
    if (info->struct_size < sizeof(akr_kernel_info)) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_pt_kernel_info: struct_size is smaller than this library's akr_kernel_info (set it to sizeof)");
    return guarded([&] {
        const uint32_t size = info->struct_size;
        std::memset(info, 0, sizeof *info);
        info->struct_size = size;
        info->specialised = se->spec_active ? 1u : 0u;
        info->n_shader_kinds = (uint32_t)se->scene->cs.shader_kinds.size();
        info->kernel_flags = (se->scene->cs.bvh_nodes.empty() ? 0u : 1u) | (se->params.sampler != 0 ? 2u : 0u) | (se->params.stage_total != 0 ? 4u : 0u) |
                             (se->params.defer_metal != 0 ? 8u : 0u);
        info->absent_mask = se->scene->cs.absent;
        if (se->spec) {
            info->cache_hit = se->spec->cache_hit ? 1u : 0u;
            info->min_waves = (uint32_t)se->spec_waves;
            info->vgprs = (uint32_t)se->spec->vgprs;
            info->scratch_bytes = (uint32_t)se->spec->scratch_bytes;
            info->compile_ms = se->spec->compile_ms;
            info->load_ms = se->spec->load_ms;
        }
        std::snprintf(info->status, sizeof info->status, "%s", se->spec_status.c_str());
    });
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
        rq.bvh = flags & 1u; rq.pmj = flags & 2u; rq.stage = flags & 4u; rq.defer = flags & 8u;
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
        rq.bvh = flags & 1u; rq.pmj = flags & 2u; rq.stage = flags & 4u; rq.defer = flags & 8u;
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
AKR_API int32_t akr_pt_end(akr_pt_session* se, akr_pt_stats* stats) {
    if (!se) return AKR_OK;
    int32_t rc = guarded([&] { read_stats(se, stats); });
    (void)hipSetDevice(se->ctx->device);
    delete se;
    return rc;
}
AKR_API int32_t akr_pt_render(akr_context* ctx, akr_scene* scene, const akr_pt_config* cfg, akr_film* film, akr_pt_stats* stats) {
    akr_pt_session* se = nullptr;
    int32_t rc = akr_pt_begin(ctx, scene, cfg, film, &se);
    if (rc != AKR_OK) return rc;
    uint32_t n_passes = (session_samples(*cfg) + cfg->spp_per_pass - 1) / cfg->spp_per_pass;
    rc = akr_pt_passes(se, n_passes, 1, nullptr);
    std::string err = g_last_error;
    int32_t rc2 = akr_pt_end(se, stats);
    if (rc != AKR_OK) {
        g_last_error = err;
        return rc;
    }
    return rc2;
}

// ------------------------------------------------------------------------------------------------ aov integrator
AKR_API int32_t akr_aov_config_default(akr_aov_config* c) {  // aov::Config::default (aov.rs:30-39) + RenderConfig defaults
    if (!c) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_aov_config_default: NULL argument");
    std::memset(c, 0, sizeof *c);
    c->spp = 256; c->aov = AKR_AOV_NS; c->remap = 1;
    c->filter_type = AKR_FILTER_GAUSSIAN; c->filter_radius = 1.5f;
    c->sampler_type = AKR_SAMPLER_INDEPENDENT; c->sampler_seed = 0;
    c->shard_rank = 0; c->shard_count = 1; c->tile_w = 32; c->tile_h = 32;
    return AKR_OK;
}
AKR_API int32_t akr_aov_render(akr_context* ctx, akr_scene* scene, const akr_aov_config* cfg, akr_film* film, akr_pt_stats* stats) {
    if (!ctx || !scene || !cfg || !film) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_aov_render: NULL argument");
    if (cfg->aov > AKR_AOV_ROUGHNESS) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_aov_render: unknown aov");
    // the session machinery of the path tracer provides sampler states, counters, sharding and the kernel parameters
    akr_pt_config pc;
    akr_pt_config_default(&pc);
    pc.spp = cfg->spp; pc.spp_per_pass = cfg->spp ? cfg->spp : 1;
    pc.filter_type = cfg->filter_type; pc.filter_radius = cfg->filter_radius;
    pc.sampler_type = cfg->sampler_type; pc.sampler_seed = cfg->sampler_seed;
    pc.shard_rank = cfg->shard_rank; pc.shard_count = cfg->shard_count; pc.tile_w = cfg->tile_w; pc.tile_h = cfg->tile_h;
    pc.color = cfg->color;
    akr_pt_session* se = nullptr;
    int32_t rc = akr_pt_begin(ctx, scene, &pc, film, &se);
    if (rc != AKR_OK) return rc;
    rc = guarded([&] {
        if (cfg->spp == 0) return;
        fill_params(se, 1, cfg->spp);
        LaunchTimer timer(se);
        HIP_CHECK(launch_aov(se->params, cfg->spp, cfg->aov, cfg->remap ? 1u : 0u, ctx->stream));
        timer.stop();
        se->n_launches++;
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
    std::string err = g_last_error;
    int32_t rc2 = akr_pt_end(se, stats);
    if (rc != AKR_OK) {
        g_last_error = err;
        return rc;
    }
    return rc2;
}

// ------------------------------------------------------------------------------------------------ gpt integrator
AKR_API int32_t akr_gpt_config_default(akr_gpt_config* c) {  // gpt::Config::default (gpt.rs:48-65) + RenderConfig defaults
    if (!c) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_config_default: NULL argument");
    std::memset(c, 0, sizeof *c);
    c->spp = 256; c->max_depth = 7; c->rr_depth = 5; c->spp_per_pass = 64;
    c->use_nee = 1; c->indirect_only = 0; c->reconnect = 1; c->stride = 1;
    c->separate_weights = 0; c->reconstruction = AKR_GPT_RECON_NONE; c->reconstruction_iter = 30;
    c->filter_type = AKR_FILTER_GAUSSIAN; c->filter_radius = 1.5f;
    c->sampler_type = AKR_SAMPLER_INDEPENDENT; c->sampler_seed = 0; c->seed = 0;
    return AKR_OK;
}
// One gpt render in steps, so that several GPUs can share it: begin (with the rank's akr_shard) -> sample -> reduce (the film's
// splat channels with reconstruction none, the primal / gradient sums otherwise) -> finish (the reconstruction sweeps run on the
// reduced sums). akr_gpt_render is begin + sample + finish on the whole frame.
struct akr_gpt_session {
    akr_context* ctx = nullptr;
    akr_scene* scene = nullptr;
    akr_film* film = nullptr;
    akr_gpt_config cfg;
    akr_pt_session* pt = nullptr;  // sampler states, counters, kernel parameters, timing
    DevBuf scratch, sums, item_pixels;
    GptParams g;
    uint32_t W = 0, H = 0, spp_done = 0, n_items = 0;
    bool recon = false;
    size_t n_sums() const { return recon ? 6 * (size_t)W * H + 12 * (size_t)(W + 1) * (H + 1) : 0; }
};
extern "C++" {
namespace akr {
int32_t gpt_reduce_view(akr_gpt_session* se, akr_film** film, int* device, hipStream_t* stream, float** sums, size_t* n_sums) {
    if (!se) return fail(AKR_ERR_INVALID_ARGUMENT, "gpt session is NULL");
    *film = se->film;
    *device = se->ctx->device;
    *stream = se->ctx->stream;
    *sums = se->sums.as<float>();
    *n_sums = se->n_sums();
    return AKR_OK;
}
}  // namespace akr
}  // extern "C++"
static uint32_t gpt_reflect_host(int64_t x, uint32_t r) { return x < 0 ? (uint32_t)(-x) : (x >= (int64_t)r ? r - (uint32_t)(x - r) - 1u : (uint32_t)x); }  // gpt.rs:131-139

AKR_API int32_t akr_gpt_begin(akr_context* ctx, akr_scene* scene, const akr_gpt_config* cfg, const akr_shard* shard, akr_film* film, akr_gpt_session** out) {
    if (!ctx || !scene || !cfg || !film || !out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_begin: NULL argument");
    *out = nullptr;
    const uint32_t W = scene->flat.camera.width, H = scene->flat.camera.height;
    if (cfg->reconstruction > AKR_GPT_RECON_WEIGHTED) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_begin: unknown reconstruction");
    if (cfg->stride < 1 || cfg->stride >= W || cfg->stride >= H) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_begin: stride must be in [1, min(width, height))");
    if (cfg->reconstruction == AKR_GPT_RECON_NONE && !cfg->reconnect)  // shift_mapping.as_ref().unwrap(), gpt.rs:276
        return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_begin: reconstruction 'none' needs reconnect = true (the reference panics)");
    if (cfg->sampler_type != AKR_SAMPLER_INDEPENDENT)  // Pmj02BnSampler::clone_box is todo!(), sampler/mod.rs:677
        return fail(AKR_ERR_UNSUPPORTED, "akr_gpt_begin: gpt needs the independent sampler (the reference's pmj02bn sampler cannot be cloned)");
    if (shard && shard->shard_count > 1 && shard->shard_rank >= shard->shard_count) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_begin: shard_rank >= shard_count");
    akr_pt_config pc;
    akr_pt_config_default(&pc);
    pc.spp = cfg->spp; pc.spp_per_pass = 1; pc.max_depth = cfg->max_depth; pc.rr_depth = cfg->rr_depth;
    pc.use_nee = cfg->use_nee; pc.indirect_only = cfg->indirect_only;
    pc.filter_type = cfg->filter_type; pc.filter_radius = cfg->filter_radius;
    pc.sampler_type = cfg->sampler_type; pc.sampler_seed = cfg->sampler_seed;
    pc.color = cfg->color;
    akr_pt_session* pt = nullptr;
    int32_t rc = akr_pt_begin(ctx, scene, &pc, film, &pt);
    if (rc != AKR_OK) return rc;
    std::unique_ptr<akr_gpt_session> se;
    rc = guarded([&] {
        se = std::make_unique<akr_gpt_session>();  // (inside guarded: a bad_alloc must not cross the C ABI)
        se->ctx = ctx; se->scene = scene; se->film = film; se->cfg = *cfg; se->pt = pt; se->W = W; se->H = H;
        const size_t N = (size_t)W * H, NG = (size_t)(W + 1) * (H + 1);
        se->recon = cfg->reconstruction != AKR_GPT_RECON_NONE;
        se->scratch.alloc(15 * N * sizeof(float));
        // the slots are gathered from neighbours that a sharded render may never write (pixels outside the rank's halo): zero, not garbage
        HIP_CHECK(hipMemsetAsync(se->scratch.p, 0, se->scratch.bytes, ctx->stream));
        GptParams& g = se->g;
        std::memset(&g, 0, sizeof g);
        g.own = se->scratch.as<float>();
        for (int i = 0; i < 4; i++) g.shifted[i] = se->scratch.as<float>() + 3 * N * (size_t)(1 + i);
        g.reconnect = cfg->reconnect ? 1u : 0u; g.stride = cfg->stride; g.separate_weights = cfg->separate_weights ? 1u : 0u;
        g.reconstruction = cfg->reconstruction;
        if (se->recon) {
            se->sums.alloc((6 * N + 12 * NG) * sizeof(float));
            HIP_CHECK(hipMemsetAsync(se->sums.p, 0, se->sums.bytes, ctx->stream));
            float* b = se->sums.as<float>();
            g.acc_p = b; g.sqr_p = b + 3 * N; g.acc_gx = b + 6 * N; g.acc_gy = g.acc_gx + 3 * NG; g.sqr_gx = g.acc_gy + 3 * NG; g.sqr_gy = g.sqr_gx + 3 * NG;
        }
        fill_params(pt, 1, 1);
        se->n_items = pt->params.n_items;
        g.shard_count = 1;
        if (shard && shard->shard_count > 1) {
            // The rank folds (k_gpt_update) the pixels of its own tiles; a pixel's value gathers what its neighbours' offset paths
            // splat onto it, so the rank SAMPLES its own pixels plus the halo of pixels one of whose offset paths lands in an owned
            // tile (reconstruction none: the four pixels `stride` away, mirrored at the borders, gpt.rs:118-142; otherwise the left
            // and the upper neighbour, whose +x / +y gradients the update reads). Every rank keeps the whole frame's sampler states,
            // and a halo pixel draws the same numbers on every rank that samples it. The list is built here, once: own pixels
            // tile by tile in 8x8 blocks (the order of item_to_pixel), then the halo.
            const uint32_t tw = shard->tile_w ? shard->tile_w : 32, th = shard->tile_h ? shard->tile_h : 32;
            if (tw % 8 != 0 || th % 8 != 0) throw std::invalid_argument("akr_shard: tile sizes must be multiples of 8");
            const uint32_t tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th;
            g.shard_rank = shard->shard_rank; g.shard_count = shard->shard_count; g.tile_w = tw; g.tile_h = th; g.tiles_x = tiles_x;
            auto owned = [&](uint32_t x, uint32_t y) { return ((y / th) * tiles_x + x / tw) % shard->shard_count == shard->shard_rank; };
            std::vector<uint32_t> list;
            for (uint32_t t = shard->shard_rank; t < tiles_x * tiles_y; t += shard->shard_count) {
                const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
                for (uint32_t by = 0; by < th / 8; by++)
                    for (uint32_t bx = 0; bx < tw / 8; bx++)
                        for (uint32_t l = 0; l < 64; l++) {
                            const uint32_t x = tx * tw + bx * 8 + (l & 7u), y = ty * th + by * 8 + (l >> 3);
                            if (x < W && y < H) list.push_back(x + y * W);
                        }
            }
            const int64_t st = cfg->stride;
            for (uint32_t y = 0; y < H; y++)
                for (uint32_t x = 0; x < W; x++) {
                    if (owned(x, y)) continue;
                    bool need;
                    if (!se->recon) {
                        need = owned(gpt_reflect_host((int64_t)x + st, W), y) || owned(gpt_reflect_host((int64_t)x - st, W), y) ||
                               owned(x, gpt_reflect_host((int64_t)y + st, H)) || owned(x, gpt_reflect_host((int64_t)y - st, H));
                    } else {
                        need = (x + 1 < W && owned(x + 1, y)) || (y + 1 < H && owned(x, y + 1));
                    }
                    if (need) list.push_back(x + y * W);
                }
            se->item_pixels.upload(list);
            g.item_pixels = se->item_pixels.as<uint32_t>();
            se->n_items = (uint32_t)list.size();
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
    if (rc != AKR_OK) {
        std::string err = g_last_error;
        akr_pt_end(pt, nullptr);
        g_last_error = err;
        return rc;
    }
    *out = se.release();
    return AKR_OK;
}
AKR_API int32_t akr_gpt_sample(akr_gpt_session* se, uint32_t n_samples, int32_t blocking) {
    if (!se) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_sample: session is NULL");
    return guarded([&] {
        se->ctx->bind();
        const uint32_t left = se->cfg.spp - se->spp_done, n = n_samples == 0 ? left : std::min(n_samples, left);
        akr_pt_session* pt = se->pt;
        fill_params(pt, 1, 1);
        pt->params.n_items = se->n_items;
        LaunchTimer timer(pt);
        for (uint32_t s = 0; s < n; s++) {  // gpt.rs:468-485: kernel + update_kernel per sample
            HIP_CHECK(launch_gpt_sample(pt->params, se->g, se->ctx->stream));
            HIP_CHECK(launch_gpt_update(se->g, se->W, se->H, se->film->data, se->ctx->stream));
        }
        timer.stop();
        pt->n_launches += 2 * n;
        se->spp_done += n;
        pt->spp_done = se->spp_done;
        if (blocking) HIP_CHECK(hipStreamSynchronize(se->ctx->stream));
    });
}
// The primal / gradient sums and sums of squares of a reconstructing render ([6 N + 12 (W+1)(H+1)] floats on the device; n = 0
// with reconstruction none, whose sums are the film's splat channels): for hosts that reduce with their own collective.
AKR_API int32_t akr_gpt_sums(akr_gpt_session* se, float** device_ptr, uint64_t* n_floats) {
    if (!se || !device_ptr || !n_floats) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_sums: NULL argument");
    *device_ptr = se->sums.as<float>();
    *n_floats = se->n_sums();
    return AKR_OK;
}
AKR_API int32_t akr_gpt_sums_read(akr_gpt_session* se, float* dst) {
    if (!se || !dst) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_sums_read: NULL argument");
    return guarded([&] {
        se->ctx->bind();
        HIP_CHECK(hipStreamSynchronize(se->ctx->stream));
        if (se->n_sums()) HIP_CHECK(hipMemcpy(dst, se->sums.p, se->n_sums() * sizeof(float), hipMemcpyDeviceToHost));
    });
}
AKR_API int32_t akr_gpt_sums_write(akr_gpt_session* se, const float* src) {
    if (!se || !src) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_gpt_sums_write: NULL argument");
    return guarded([&] {
        se->ctx->bind();
        HIP_CHECK(hipStreamSynchronize(se->ctx->stream));
        if (se->n_sums()) HIP_CHECK(hipMemcpy(se->sums.p, src, se->n_sums() * sizeof(float), hipMemcpyHostToDevice));
    });
}
AKR_API int32_t akr_gpt_finish(akr_gpt_session* se, float* aux, akr_pt_stats* stats) {
    if (!se) return AKR_OK;
    akr_context* ctx = se->ctx;
    akr_film* film = se->film;
    const akr_gpt_config* cfg = &se->cfg;
    const uint32_t W = se->W, H = se->H;
    int32_t rc = guarded([&] {
        ctx->bind();
        const size_t N = (size_t)W * H, NG = (size_t)(W + 1) * (H + 1);
        const GptParams& g = se->g;
        DevBuf old;
        LaunchTimer timer(se->pt);
        if (!se->recon) {
            film->splat_scale = 1.0f / (float)cfg->spp;  // gpt.rs:463-466
        } else if (cfg->spp > 0) {  // gpt.rs:495-606
            const float spp = (float)cfg->spp;
            old.alloc(3 * N * sizeof(float));
            HIP_CHECK(launch_gpt_recon_init(g, W, H, old.as<float>(), spp, ctx->stream));
            std::vector<float> prefix(std::max(cfg->reconstruction_iter, 1u), 1.0f);
            const float eps = 0.01f;
            for (uint32_t i = 1; i < cfg->reconstruction_iter; i++) {
                float p2 = 1.0f;
                for (uint32_t k = 0; k + 1 < i; k++) p2 *= 0.5f;  // 0.5f32.powi(i - 1)
                prefix[i] = prefix[i - 1] * (1.0f / ((eps + 1.0f) + 4.0f * p2));
            }
            float* cur = film->data + 3 * N;
            for (uint32_t it = 0; it < cfg->reconstruction_iter; it++) {
                HIP_CHECK(launch_gpt_recon(g, W, H, old.as<float>(), cur, prefix[it], spp, ctx->stream));
                HIP_CHECK(hipMemcpyAsync(old.p, cur, 3 * N * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        timer.stop();
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (aux && se->recon) {
            HIP_CHECK(hipMemcpy(aux, g.acc_p, 3 * N * sizeof(float), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(aux + 3 * N, g.acc_gx, 6 * NG * sizeof(float), hipMemcpyDeviceToHost));
        }
    });
    std::string err = g_last_error;
    int32_t rc2 = akr_pt_end(se->pt, stats);
    delete se;
    if (rc != AKR_OK) {
        g_last_error = err;
        return rc;
    }
    return rc2;
}
// Ends a session WITHOUT the splat scale / reconstruction sweeps of akr_gpt_finish: the film keeps whatever the samples (and a
// reduce) left in it. For a rank that is not the root of akr_gpt_reduce (its partial sums would reconstruct into garbage) and for
// abandoning a render.
AKR_API int32_t akr_gpt_abort(akr_gpt_session* se, akr_pt_stats* stats) {
    if (!se) return AKR_OK;
    int32_t rc = akr_pt_end(se->pt, stats);
    delete se;
    return rc;
}
AKR_API int32_t akr_gpt_render(akr_context* ctx, akr_scene* scene, const akr_gpt_config* cfg, akr_film* film, float* aux, akr_pt_stats* stats) {
    akr_gpt_session* se = nullptr;
    int32_t rc = akr_gpt_begin(ctx, scene, cfg, nullptr, film, &se);
    if (rc != AKR_OK) return rc;
    rc = akr_gpt_sample(se, 0, 0);
    std::string err = g_last_error;
    int32_t rc2 = akr_gpt_finish(se, aux, stats);
    if (rc != AKR_OK) {
        g_last_error = err;
        return rc;
    }
    return rc2;
}

// ------------------------------------------------------------------------------------------------ mcmc_opt integrator
AKR_API int32_t akr_mcmc_config_default(akr_mcmc_config* c) {  // mcmc::Config::default (mcmc.rs:60-79), Method::default (mcmc.rs:21-32)
    if (!c) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_mcmc_config_default: NULL argument");
    std::memset(c, 0, sizeof *c);
    c->spp = 256; c->max_depth = 7; c->rr_depth = 5; c->spp_per_pass = 64; c->use_nee = 1;
    c->mcmc_depth = 0xffffffffu; c->n_chains = 512; c->n_bootstrap = 100000; c->direct_spp = 64;
    c->exponential_mutation = 1; c->small_sigma = 0.01f; c->large_step_prob = 0.1f; c->image_mutation_prob = 0.0f; c->image_mutation_size = 0.0f;
    c->adaptive = 0; c->wis = 0; c->seed = 0;
    c->filter_type = AKR_FILTER_GAUSSIAN; c->filter_radius = 1.5f;
    c->sampler_type = AKR_SAMPLER_INDEPENDENT; c->sampler_seed = 0;
    return AKR_OK;
}
// on_pass(spp so far, seconds of rendering so far): called after every pass with the film's splat scale already set for that
// many samples (reconstruct(film, cnt), mcmc_opt.rs:644-662); used by akr_render_task for --save-intermediate
// shard_count > 1: this rank's share of the render (akr_mcmc_render_shard) -- chains [rank n / count, (rank + 1) n / count) of the
// n_chains, the direct-lighting pass on the rank's pixel tiles; `partial` receives what the normalisation needs from this rank.
static int32_t mcmc_render_impl(akr_context* ctx, akr_scene* scene, const akr_mcmc_config* cfg, akr_film* film, akr_mcmc_result* result,
                                uint32_t* chain_states, akr_pt_stats* stats, const std::function<void(uint32_t, double)>& on_pass,
                                uint32_t shard_rank = 0, uint32_t shard_count = 1, akr_mcmc_partial* partial = nullptr) {
    if (shard_count == 0 || shard_rank >= shard_count) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_mcmc_render_shard: shard_rank >= shard_count");
    if (!ctx || !scene || !cfg || !film) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_mcmc_render: NULL argument");
    if (cfg->n_chains == 0 || cfg->n_bootstrap == 0) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_mcmc_render: n_chains and n_bootstrap must be positive");
    if (cfg->spp_per_pass == 0) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_mcmc_render: spp_per_pass must be positive");
    const uint32_t W = scene->flat.camera.width, H = scene->flat.camera.height;
    if (cfg->direct_spp > 0) {  // direct illumination by the path tracer, mcmc_opt.rs:704-729
        akr_pt_config d;
        akr_pt_config_default(&d);
        d.max_depth = 1; d.rr_depth = 1; d.spp = (uint32_t)cfg->direct_spp; d.indirect_only = 0; d.spp_per_pass = cfg->spp_per_pass; d.use_nee = cfg->use_nee;
        d.filter_type = cfg->filter_type; d.filter_radius = cfg->filter_radius; d.sampler_type = cfg->sampler_type; d.sampler_seed = cfg->sampler_seed;
        d.color = cfg->color;
        d.shard_rank = shard_rank; d.shard_count = shard_count;  // the direct pass is a pt render: its tiles over the ranks
        int32_t rc = akr_pt_render(ctx, scene, &d, film, nullptr);
        if (rc != AKR_OK) return rc;
    }
    akr_pt_config pc;  // the PathTracer inside McmcOpt::new, mcmc_opt.rs:233-252
    akr_pt_config_default(&pc);
    pc.spp = 1; pc.spp_per_pass = 1; pc.max_depth = cfg->max_depth; pc.rr_depth = cfg->rr_depth; pc.use_nee = cfg->use_nee;
    pc.indirect_only = cfg->direct_spp >= 0 ? 1u : 0u;
    pc.filter_type = cfg->filter_type; pc.filter_radius = cfg->filter_radius;
    pc.color = cfg->color;
    akr_pt_session* se = nullptr;
    int32_t rc = akr_pt_begin(ctx, scene, &pc, film, &se);
    if (rc != AKR_OK) return rc;
    rc = guarded([&] {
        const uint32_t depth = cfg->mcmc_depth == 0xffffffffu ? cfg->max_depth : cfg->mcmc_depth;
        const uint32_t dim = 4 + 1 + (1 + depth) * (3 + 3 + 1);  // sample_dimension, mcmc_opt.rs:230-232
        const uint32_t n_chains = cfg->n_chains, n_boot = cfg->n_bootstrap;
        fill_params(se, 1, 1);
        // init_pcg32_buffer_with_seed(n, seed): the bootstrap seeds and the chains' samplers are prefixes of the same stream
        const size_t n_seeds = std::max(n_chains, n_boot);
        std::vector<Pcg32> seeds(n_seeds);
        {
            StdRng rng(cfg->seed);
            for (size_t i = 0; i < n_seeds; i++) seeds[i] = pcg_new_seq_offset(i, rng.next_u64());
        }
        DevBuf d_seeds, d_fs, d_resampled, d_pss, d_states, d_colors, d_rngs;
        d_seeds.upload(seeds);
        d_fs.alloc(n_boot * sizeof(float));
        d_pss.alloc((size_t)dim * n_chains * sizeof(PssSample));
        d_states.alloc(n_chains * sizeof(MarkovState));
        d_colors.alloc(n_chains * sizeof(float4));
        d_rngs.alloc(n_chains * sizeof(Pcg32));
        HIP_CHECK(hipMemsetAsync(d_states.p, 0, d_states.bytes, ctx->stream));  // (a shard leaves the other ranks' records untouched: zeros)
        HIP_CHECK(hipMemcpyAsync(d_rngs.p, seeds.data(), n_chains * sizeof(Pcg32), hipMemcpyHostToDevice, ctx->stream));
        McmcParams m;
        std::memset(&m, 0, sizeof m);
        m.pss = d_pss.as<PssSample>(); m.states = d_states.as<MarkovState>(); m.cur_colors = d_colors.as<float4>(); m.rngs = d_rngs.as<Pcg32>();
        m.seeds = d_seeds.as<Pcg32>(); m.fs = d_fs.as<float>(); m.film = film->data;
        m.n_chains = n_chains; m.n_bootstrap = n_boot; m.dim = dim; m.width = W; m.height = H;
        // this rank's chains; everything that defines a chain (its bootstrap path, its sampler, the mutations per chain, the weight of a
        // mutation) comes from the GLOBAL chain index and count, so the union of the ranks' chain sets is the one-GPU chain set
        const uint32_t chain_begin = (uint32_t)((uint64_t)shard_rank * n_chains / shard_count);
        const uint32_t chain_end = (uint32_t)((uint64_t)(shard_rank + 1) * n_chains / shard_count);
        m.chain_begin = chain_begin; m.chain_count = chain_end - chain_begin;
        m.exponential_mutation = cfg->exponential_mutation ? 1u : 0u;
        m.small_sigma = cfg->small_sigma; m.large_step_prob = cfg->large_step_prob; m.image_mutation_prob = cfg->image_mutation_prob;
        m.image_mutation_size = cfg->image_mutation_size;
        LaunchTimer timer(se);
        HIP_CHECK(launch_mcmc_bootstrap(se->params, m, ctx->stream));
        std::vector<float> fs(n_boot);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        HIP_CHECK(hipMemcpy(fs.data(), d_fs.p, n_boot * sizeof(float), hipMemcpyDeviceToHost));
        // resample_with_f64 (util/distribution.rs:92-115); the reference sums with rayon, here in index order
        double sum = 0.0;
        for (float f : fs) sum += (double)f;
        if (!(sum > 0.0)) throw RenderError("Bootstrap failed, please retry with more samples (mcmc_opt.rs:352)");
        std::vector<double> cdf(n_boot);
        for (uint32_t i = 0; i < n_boot; i++) {
            double pr = (double)fs[i] / sum;
            cdf[i] = i == 0 ? pr : cdf[i - 1] + pr;
        }
        std::vector<uint32_t> resampled(n_chains);
        {
            StdRng rng(0);
            for (uint32_t k = 0; k < n_chains; k++) {
                double u = (double)(rng.next_u64() >> 11) * (1.0 / 9007199254740992.0);  // rand 0.8.5 Standard f64: 53 random bits
                uint32_t lo = 0, hi = n_boot;  // partition_point(|x| u >= *x)
                while (lo < hi) {
                    uint32_t mid = lo + (hi - lo) / 2;
                    if (u >= cdf[mid]) lo = mid + 1; else hi = mid;
                }
                resampled[k] = std::min(lo, n_boot - 1);
            }
        }
        d_resampled.upload(resampled);
        m.resampled = d_resampled.as<uint32_t>();
        HIP_CHECK(launch_mcmc_init(se->params, m, ctx->stream));
        // render_loop, mcmc_opt.rs:554-683
        const uint64_t npixels = (uint64_t)W * H;
        float contribution;
        {
            const uint64_t n_mut = npixels * (uint64_t)cfg->spp;
            const uint64_t per = std::max<uint64_t>(n_mut / n_chains, 1);
            contribution = (float)((double)n_mut / ((double)per * (double)n_chains));
        }
        // reconstruct(film, spp), mcmc_opt.rs:587-611: normalisation from the bootstrap and the chains' large steps
        std::vector<MarkovState> states(n_chains);
        double b = 0.0;
        uint64_t accepted = 0, mutations = 0;
        auto reconstruct = [&](uint32_t spp_done) {
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            HIP_CHECK(hipMemcpy(states.data(), d_states.p, n_chains * sizeof(MarkovState), hipMemcpyDeviceToHost));
            b = sum;
            uint64_t b_cnt = n_boot;
            accepted = 0; mutations = 0;
            double own_b = 0.0;
            uint64_t own_cnt = 0;
            for (uint32_t k = chain_begin; k < chain_end; k++) {
                const MarkovState& st = states[k];
                if (shard_count == 1) b += (double)st.b;  // (one GPU: the reference's summation order)
                own_b += (double)st.b; own_cnt += st.b_cnt; accepted += st.n_accepted; mutations += st.n_mutations;
            }
            b_cnt += own_cnt;
            if (partial) {
                partial->bootstrap_sum = sum; partial->b_sum = own_b; partial->n_bootstrap = n_boot; partial->b_cnt = own_cnt;
                partial->n_accepted = accepted; partial->n_mutations = mutations;
                partial->spp = spp_done;
            }
            if (shard_count == 1) {
                b = b / (double)b_cnt;
                film->splat_scale = (float)b / (float)spp_done;
            }  // a shard's film gets its scale from akr_mcmc_combine, which knows every rank's sums
        };
        uint32_t cnt = 0;
        uint64_t total_mutations = 0;
        double acc_s = 0.0;
        while (cnt < cfg->spp) {
            const uint32_t cur_pass = std::min(cfg->spp - cnt, cfg->spp_per_pass);
            const uint64_t per = std::max<uint64_t>(npixels * (uint64_t)cur_pass / n_chains, 1);  // (global chain count)
            if (per > 0xffffffffull) throw std::invalid_argument("Number of mutations per chain exceeds u32::MAX, please reduce spp per pass or increase number of chains");
            const auto tic = std::chrono::steady_clock::now();
            HIP_CHECK(launch_mcmc_advance(se->params, m, (uint32_t)per, contribution, ctx->stream));
            total_mutations += per * (chain_end - chain_begin);
            cnt += cur_pass;
            if (on_pass) {
                reconstruct(cnt);
                acc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tic).count();
                on_pass(cnt, acc_s);
            }
        }
        timer.stop();
        se->n_launches += 2 + (cfg->spp + cfg->spp_per_pass - 1) / cfg->spp_per_pass;
        reconstruct(cfg->spp);
        if (partial) { partial->contribution = contribution; partial->n_executed = total_mutations; }
        if (result) {
            result->normalization = b; result->acceptance_rate = (double)accepted / (double)mutations; result->splat_scale = film->splat_scale;
            result->contribution = contribution; result->n_mutations = total_mutations; result->sample_dimension = dim; result->_pad = 0;
        }
        if (chain_states) std::memcpy(chain_states, states.data(), n_chains * sizeof(MarkovState));
    });
    std::string err = g_last_error;
    int32_t rc2 = akr_pt_end(se, stats);
    if (rc != AKR_OK) {
        g_last_error = err;
        return rc;
    }
    return rc2;
}

AKR_API int32_t akr_mcmc_render(akr_context* ctx, akr_scene* scene, const akr_mcmc_config* cfg, akr_film* film, akr_mcmc_result* result,
                                uint32_t* chain_states, akr_pt_stats* stats) {
    return mcmc_render_impl(ctx, scene, cfg, film, result, chain_states, stats, nullptr);
}
AKR_API int32_t akr_mcmc_render_shard(akr_context* ctx, akr_scene* scene, const akr_mcmc_config* cfg, uint32_t shard_rank, uint32_t shard_count, akr_film* film,
                                      akr_mcmc_partial* partial, uint32_t* chain_states, akr_pt_stats* stats) {
    if (!partial) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_mcmc_render_shard: NULL argument");
    std::memset(partial, 0, sizeof *partial);
    return mcmc_render_impl(ctx, scene, cfg, film, nullptr, chain_states, stats, nullptr, shard_rank, shard_count, partial);
}
// reconstruct (mcmc_opt.rs:587-611) from the ranks' sums: b = (bootstrap sum + sum of the chains' large-step contributions) / (their count)
AKR_API int32_t akr_mcmc_combine_host(akr_film* film, const akr_mcmc_partial* partials, uint32_t n, akr_mcmc_result* result) {
    if (!partials || n == 0) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_mcmc_combine_host: no partial sums");
    double b = partials[0].bootstrap_sum;
    uint64_t cnt = partials[0].n_bootstrap, accepted = 0, mutations = 0, executed = 0;
    for (uint32_t r = 0; r < n; r++) {
        if (partials[r].bootstrap_sum != partials[0].bootstrap_sum || partials[r].n_bootstrap != partials[0].n_bootstrap || partials[r].spp != partials[0].spp)
            return fail(AKR_ERR_INVALID_ARGUMENT, "akr_mcmc_combine_host: the partial sums are not of one render (bootstrap or spp differ between ranks)");
        b += partials[r].b_sum; cnt += partials[r].b_cnt; accepted += partials[r].n_accepted; mutations += partials[r].n_mutations;
        executed += partials[r].n_executed;
    }
    b /= (double)cnt;
    const float scale = (float)b / (float)partials[0].spp;
    if (film) film->splat_scale = scale;
    if (result) {
        std::memset(result, 0, sizeof *result);
        result->normalization = b; result->acceptance_rate = mutations ? (double)accepted / (double)mutations : 0.0; result->splat_scale = scale;
        result->contribution = partials[0].contribution; result->n_mutations = executed;
    }
    return AKR_OK;
}

// ------------------------------------------------------------------------------------------------ render driver
AKR_API int32_t akr_image_write(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_image_write: bad argument");
    return guarded([&] { write_image(path, rgb, width, height); });
}

AKR_API int32_t akr_render_task(akr_context* ctx, akr_scene* scene, const char* method_json_text, const akr_render_session* session,
                                akr_pt_stats* stats_out) {
    if (!ctx || !scene || !method_json_text) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_render_task: NULL argument");
    akr_render_session ses{0, 0, nullptr, 0, 0};
    if (session) ses = *session;
    const std::string name = ses.name ? ses.name : "default";
    return guarded([&] {
        std::vector<ParsedTask> tasks = parse_render_tasks(method_json_text, ses.override_sampler_independent != 0);
        const uint32_t w = scene->flat.camera.width, h = scene->flat.camera.height;
        std::vector<float> rgb(3ull * w * h);
        for (size_t ti = 0; ti < tasks.size(); ti++) {  // render_single, lib.rs:112-193
            const ParsedTask& task = tasks[ti];
            if (ses.verbose) std::fprintf(stderr, "[akari_hip] task %zu/%zu (%s): %ux%u, %u spp -> %s\n", ti + 1, tasks.size(), task.is_aov ? "aov" : (task.is_gpt ? "gpt" : (task.is_mcmc ? "mcmc_opt" : "pt")), w, h, task.is_aov ? task.aov.spp : (task.is_gpt ? task.gpt.spp : (task.is_mcmc ? task.mcmc.spp : task.cfg.spp)), task.film_out.c_str());
            akr_film* film = nullptr;
            akr_pt_session* se = nullptr;
            auto check = [&](int32_t rc) { if (rc != AKR_OK) { std::string m = g_last_error; if (se) akr_pt_end(se, nullptr); if (film) akr_film_destroy(film); throw std::runtime_error(m); } };
            check(akr_film_create(ctx, w, h, &film));
            if (task.is_aov) {  // Method::NormalVis: one blocking dispatch, no intermediates (aov.rs:161-171)
                akr_pt_stats st;
                check(akr_aov_render(ctx, scene, &task.aov, film, &st));
                if (ses.verbose) std::fprintf(stderr, "[akari_hip] Rendered in %.2fms\n", st.kernel_ms);
                check(akr_film_resolve(film, rgb.data()));
                akr_film_destroy(film);
                film = nullptr;
                write_image(task.film_out, rgb.data(), w, h);
                if (stats_out) *stats_out = st;
                continue;
            }
            if (task.is_mcmc) {  // McmcOpt::render; --save-intermediate / --save-stats as render_loop does (mcmc_opt.rs:640-676)
                akr_pt_stats st;
                akr_mcmc_result res;
                std::string stats_json = "{\"intermediate\":[";
                bool first = true;
                std::function<void(uint32_t, double)> on_pass;
                if (ses.save_intermediate)
                    on_pass = [&](uint32_t cnt, double time_s) {
                        check(akr_film_resolve(film, rgb.data()));
                        std::string path = name + "-" + std::to_string(cnt) + ".exr";
                        write_image(path, rgb.data(), w, h);
                        char buf[512];
                        std::snprintf(buf, sizeof buf, "%s{\"path\":\"%s\",\"time\":%.9g,\"spp\":%u}", first ? "" : ",", path.c_str(), time_s, cnt);
                        stats_json += buf;
                        first = false;
                    };
                check(mcmc_render_impl(ctx, scene, &task.mcmc, film, &res, nullptr, &st, on_pass));
                stats_json += "]}";
                if (ses.save_stats) {
                    std::string path = name + ".json";
                    FILE* f = std::fopen(path.c_str(), "wb");
                    if (!f) throw std::runtime_error("cannot open '" + path + "' for writing");
                    std::fwrite(stats_json.data(), 1, stats_json.size(), f);
                    std::fclose(f);
                }
                if (ses.verbose)
                    std::fprintf(stderr, "[akari_hip] Normalization factor: %g\n[akari_hip] Acceptance rate: %.2f%%\n[akari_hip] Rendering finished in %.2fs\n",
                                 res.normalization, res.acceptance_rate * 100.0, st.kernel_ms * 1e-3);
                check(akr_film_resolve(film, rgb.data()));
                akr_film_destroy(film);
                film = nullptr;
                write_image(task.film_out, rgb.data(), w, h);
                if (stats_out) *stats_out = st;
                continue;
            }
            if (task.is_gpt) {  // GradientPathTracer::render: no intermediates; with a reconstruction also output/gpt_*.exr (gpt.rs:609-636)
                akr_pt_stats st;
                const bool recon = task.gpt.reconstruction != AKR_GPT_RECON_NONE;
                const size_t N = (size_t)w * h, NG = (size_t)(w + 1) * (h + 1);
                std::vector<float> aux(recon ? 3 * N + 6 * NG : 0);
                check(akr_gpt_render(ctx, scene, &task.gpt, film, recon ? aux.data() : nullptr, &st));
                if (ses.verbose) std::fprintf(stderr, "[akari_hip] Rendering finished in %.2fs\n", st.kernel_ms * 1e-3);
                check(akr_film_resolve(film, rgb.data()));
                akr_film_destroy(film);
                film = nullptr;
                if (recon) {
                    const float scale = 1.0f / (float)task.gpt.spp;  // set_splat_scale(1 / spp) on the accumulators
                    for (float& v : aux) v = v * scale;
                    write_image("output/gpt_primal.exr", aux.data(), w, h);
                    write_image("output/gpt_gx.exr", aux.data() + 3 * N, w + 1, h + 1);
                    write_image("output/gpt_gy.exr", aux.data() + 3 * N + 3 * NG, w + 1, h + 1);
                }
                write_image(task.film_out, rgb.data(), w, h);
                if (stats_out) *stats_out = st;
                continue;
            }
            check(akr_pt_begin(ctx, scene, &task.cfg, film, &se));
            std::string stats_json = "{\"intermediate\":[";
            uint32_t cnt = 0;
            bool first = true;
            while (cnt < task.cfg.spp) {  // pt.rs:1126-1149
                if (ses.save_intermediate) {
                    check(akr_pt_passes(se, 1, 1, &cnt));
                    akr_pt_stats st;
                    check(akr_pt_get_stats(se, &st));
                    check(akr_film_resolve(film, rgb.data()));
                    std::string path = name + "-" + std::to_string(cnt) + ".exr";
                    write_image(path, rgb.data(), w, h);
                    char buf[512];
                    std::snprintf(buf, sizeof buf, "%s{\"path\":\"%s\",\"time\":%.9g,\"spp\":%u}", first ? "" : ",", path.c_str(), st.kernel_ms * 1e-3, cnt);
                    stats_json += buf;
                    first = false;
                } else {
                    check(akr_pt_passes(se, 16, 1, &cnt));
                }
            }
            stats_json += "]}";
            akr_pt_stats st;
            int32_t rc = akr_pt_end(se, &st);
            se = nullptr;
            check(rc);
            if (ses.save_stats) {  // pt.rs:1150-1155
                std::string path = name + ".json";
                FILE* f = std::fopen(path.c_str(), "wb");
                if (!f) throw std::runtime_error("cannot open '" + path + "' for writing");
                std::fwrite(stats_json.data(), 1, stats_json.size(), f);
                std::fclose(f);
            }
            if (ses.verbose) std::fprintf(stderr, "[akari_hip] Rendering finished in %.2fs (%.1f Msamples/s)\n", st.kernel_ms * 1e-3, st.n_samples / (st.kernel_ms * 1e3));
            check(akr_film_resolve(film, rgb.data()));  // film.copy_to_rgba_image(hdr = true), lib.rs:191
            akr_film_destroy(film);
            film = nullptr;
            write_image(task.film_out, rgb.data(), w, h);  // util::write_image(&output_image, &config.film.out), lib.rs:192
            if (stats_out) *stats_out = st;
        }
    });
}

// ------------------------------------------------------------------------------------------------ host KAT hooks
AKR_API int32_t akr_host_stdrng_u64(uint64_t seed, uint32_t n, uint64_t* out) {
    if (!out) return fail(AKR_ERR_INVALID_ARGUMENT, "out is NULL");
    StdRng rng(seed);
    for (uint32_t i = 0; i < n; i++) out[i] = rng.next_u64();
    return AKR_OK;
}
AKR_API int32_t akr_host_chacha_block(const uint32_t* key8, uint64_t counter, uint64_t stream, int32_t rounds, uint32_t* out16) {
    if (!key8 || !out16) return fail(AKR_ERR_INVALID_ARGUMENT, "NULL argument");
    StdRng::chacha_block(key8, counter, stream, rounds, out16);
    return AKR_OK;
}
AKR_API int32_t akr_host_pcg32_states(uint64_t seed, uint64_t n, uint64_t* out2n) {
    if (!out2n) return fail(AKR_ERR_INVALID_ARGUMENT, "out is NULL");
    StdRng rng(seed);
    for (uint64_t i = 0; i < n; i++) {
        Pcg32 p = pcg_new_seq_offset(i, rng.next_u64());
        out2n[2 * i] = p.state;
        out2n[2 * i + 1] = p.inc;
    }
    return AKR_OK;
}
AKR_API int32_t akr_host_pcg_start(uint64_t* state, uint64_t inc) {
    if (!state) return fail(AKR_ERR_INVALID_ARGUMENT, "state is NULL");
    Pcg32 p{*state, inc};
    pcg_start(p, pcg_start_constants());
    *state = p.state;
    return AKR_OK;
}
// device/drng.h on the host: reverse_bits32(sobol_dim1(i)) by the defining loop and by the five-step butterfly the kernels use
AKR_API int32_t akr_host_sobol_dim1(uint32_t n, const uint32_t* index, uint32_t* by_loop, uint32_t* by_butterfly) {
    if (!index || !by_loop || !by_butterfly) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_sobol_dim1: NULL argument");
    for (uint32_t k = 0; k < n; k++) {
        by_loop[k] = reverse_bits32(sobol_dim1(index[k]));
        by_butterfly[k] = sobol_dim1_reversed(index[k]);
    }
    return AKR_OK;
}
// device/drng.h fastmod_u32 on the host: a[k] % d[k] through the precomputed constant
AKR_API int32_t akr_host_fastmod(uint32_t n, const uint32_t* a, const uint32_t* d, uint32_t* out) {
    if (!a || !d || !out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_fastmod: NULL argument");
    for (uint32_t k = 0; k < n; k++) {
        if (d[k] == 0) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_fastmod: divisor 0");
        out[k] = fastmod_u32(a[k], fastmod_magic(d[k]), d[k]);
    }
    return AKR_OK;
}
AKR_API int32_t akr_host_alias_table(const float* weights, uint32_t n, uint32_t* j, float* t, float* pdf) {
    if (!weights || !j || !t || !pdf || n == 0) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_alias_table: bad argument");
    return guarded([&] {
        std::vector<float> w(weights, weights + n), p;
        std::vector<AliasEntry> e;
        build_alias_table(w, e, p);
        for (uint32_t i = 0; i < n; i++) { j[i] = e[i].j; t[i] = e[i].t; pdf[i] = p[i]; }
    });
}

// ------------------------------------------------------------------------------------------------ probes
AKR_API int32_t akr_probe_math(akr_context* ctx, uint32_t n, const float* x, float* s, float* c, float* l) {
    if (!ctx || !x || !s || !c || !l) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_math: NULL argument");
    return guarded([&] {
        ctx->bind();
        DevBuf dx, ds, dc, dl;
        std::vector<float> xv(x, x + n);
        dx.upload(xv);
        ds.alloc(n * 4); dc.alloc(n * 4); dl.alloc(n * 4);
        if (n) HIP_CHECK(launch_probe_math(n, dx.as<float>(), ds.as<float>(), dc.as<float>(), dl.as<float>(), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (n) {
            HIP_CHECK(hipMemcpy(s, ds.p, n * 4, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(c, dc.p, n * 4, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(l, dl.p, n * 4, hipMemcpyDeviceToHost));
        }
    });
}
AKR_API int32_t akr_probe_bsdf(akr_context* ctx, const akr_material_desc* m, const float* table, int32_t mode, const float* wo, uint32_t n,
                               const float* in, float* out) {
    if (!ctx || !m || !wo || !in || !out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_bsdf: NULL argument");
    return guarded([&] {
        ctx->bind();
        std::vector<DMaterial> dm(1, fold_material(*m));
        DevBuf dmat, dtab, din, dout;
        dmat.upload(dm);
        std::vector<float> tab(4096, 0.0f);
        if (table) tab.assign(table, table + 4096);
        dtab.upload(tab);
        std::vector<float> inv(in, in + 3ull * n);
        din.upload(inv);
        size_t out_n = (mode == 0 ? 4ull : 8ull) * n;
        dout.alloc(out_n * 4);
        if (n) HIP_CHECK(launch_probe_bsdf(dmat.as<DMaterial>(), dtab.as<float>(), mode, wo, n, din.as<float>(), dout.as<float>(), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (n) HIP_CHECK(hipMemcpy(out, dout.p, out_n * 4, hipMemcpyDeviceToHost));
    });
}
static PtParams probe_params(akr_scene* s) {
    PtParams p;
    std::memset(&p, 0, sizeof p);
    p.sc = s->dscene;
    p.tex_slots = s->cs.has_textures ? s->cs.tex_slots : 0;
    return p;
}
AKR_API int32_t akr_probe_intersect(akr_context* ctx, akr_scene* scene, uint32_t n, const float* rays, uint32_t* hit_inst_prim, float* bary) {
    if (!ctx || !scene || !rays || !hit_inst_prim || !bary) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_intersect: NULL argument");
    return guarded([&] {
        ctx->bind();
        DevBuf dr, dout, db;
        std::vector<float> rv(rays, rays + 8ull * n);
        dr.upload(rv);
        dout.alloc(3ull * n * 4);
        db.alloc(2ull * n * 4);
        if (n) HIP_CHECK(launch_probe_intersect(probe_params(scene), n, dr.as<float>(), dout.as<uint32_t>(), db.as<float>(), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (n) {
            HIP_CHECK(hipMemcpy(hit_inst_prim, dout.p, 3ull * n * 4, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(bary, db.p, 2ull * n * 4, hipMemcpyDeviceToHost));
        }
    });
}
AKR_API int32_t akr_probe_surface_interaction(akr_context* ctx, akr_scene* scene, uint32_t n, const uint32_t* inst_prim, const float* bary,
                                              float* out) {
    if (!ctx || !scene || !inst_prim || !bary || !out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_surface_interaction: NULL argument");
    return guarded([&] {
        ctx->bind();
        for (uint32_t i = 0; i < n; i++) {
            uint32_t inst = inst_prim[2 * i], prim = inst_prim[2 * i + 1];
            if (inst >= scene->flat.instances.size() || prim >= scene->flat.meshes[scene->flat.instances[inst].mesh].n_triangles())
                throw std::invalid_argument("akr_probe_surface_interaction: (inst, prim) out of range");
        }
        DevBuf dip, db, dout;
        std::vector<uint32_t> ipv(inst_prim, inst_prim + 2ull * n);
        std::vector<float> bv(bary, bary + 2ull * n);
        dip.upload(ipv);
        db.upload(bv);
        dout.alloc(19ull * n * 4);
        if (n) HIP_CHECK(launch_probe_si(probe_params(scene), n, dip.as<uint32_t>(), db.as<float>(), dout.as<float>(), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (n) HIP_CHECK(hipMemcpy(out, dout.p, 19ull * n * 4, hipMemcpyDeviceToHost));
    });
}

AKR_API int32_t akr_host_pmj02bn_tables(uint32_t* sets, uint16_t* bluenoise) {
    return guarded([&] {
        if (sets) {
            std::vector<uint32_t> v;
            make_pmj02_sets(v);
            std::memcpy(sets, v.data(), v.size() * 4);
        }
        if (bluenoise) {
            std::vector<uint16_t> v;
            load_bluenoise(v);
            std::memcpy(bluenoise, v.data(), v.size() * 2);
        }
    });
}
// PNG reader of the scene loader, exposed for tests: rgba == NULL returns the size only.
AKR_API int32_t akr_host_decode_png(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba, uint64_t capacity) {
    if (!data || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_decode_png: NULL argument");
    return guarded([&] {
        std::vector<uint8_t> px;
        decode_png(data, (size_t)len, *width, *height, px);
        if (rgba) {
            if (capacity < px.size()) throw std::invalid_argument("akr_host_decode_png: output buffer too small");
            std::memcpy(rgba, px.data(), px.size());
        }
    });
}
AKR_API int32_t akr_host_decode_jpeg(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba, uint64_t capacity) {
    if (!data || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_decode_jpeg: NULL argument");
    return guarded([&] {
        std::vector<uint8_t> px;
        decode_jpeg(data, (size_t)len, *width, *height, px);
        if (rgba) {
            if (capacity < px.size()) throw std::invalid_argument("akr_host_decode_jpeg: output buffer too small");
            std::memcpy(rgba, px.data(), px.size());
        }
    });
}
AKR_API int32_t akr_host_decode_tiff(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba, uint64_t capacity) {
    if (!data || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_decode_tiff: NULL argument");
    return guarded([&] {
        std::vector<uint8_t> px;
        decode_tiff(data, (size_t)len, *width, *height, px);
        if (rgba) {
            if (capacity < px.size()) throw std::invalid_argument("akr_host_decode_tiff: output buffer too small");
            std::memcpy(rgba, px.data(), px.size());
        }
    });
}
AKR_API int32_t akr_host_decode_dds(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba, uint64_t capacity) {
    if (!data || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_decode_dds: NULL argument");
    return guarded([&] {
        std::vector<uint8_t> px;
        decode_dds(data, (size_t)len, *width, *height, px);
        if (rgba) {
            if (capacity < px.size()) throw std::invalid_argument("akr_host_decode_dds: output buffer too small");
            std::memcpy(rgba, px.data(), px.size());
        }
    });
}
AKR_API int32_t akr_host_decode_exr(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, float* rgba, uint64_t capacity_floats) {
    if (!data || !width || !height) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_decode_exr: NULL argument");
    return guarded([&] {
        std::vector<float> px;
        decode_exr(data, (size_t)len, *width, *height, px);
        if (rgba) {
            if (capacity_floats < px.size()) throw std::invalid_argument("akr_host_decode_exr: output buffer too small");
            std::memcpy(rgba, px.data(), px.size() * sizeof(float));
        }
    });
}
// Evaluated inputs of a material at uv points: on the device (ctx != NULL; needs a scene with textures) or with the
// same code on the host (ctx == NULL).
// The same on the host for an arbitrary colour pipeline: the material tables are compiled for `color` (what akr_pt_begin does
// for a session with akr_pt_config.color != 0) and evaluated with the code the kernels run.
AKR_API int32_t akr_probe_material_inputs_host(akr_scene* scene, uint32_t material, uint32_t color, uint32_t n, const float* uv, float* out26) {
    if (!scene || !uv || !out26) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_material_inputs_host: NULL argument");
    if (material >= scene->flat.materials.size()) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_material_inputs_host: material out of range");
    return guarded([&] {
        CompiledScene tmp;
        tmp.images = scene->cs.images;
        std::vector<akr_material_desc> descs;
        compile_materials(scene->flat, color, tmp, descs);
        const TexScene ts{tmp.tex_nodes.data(), scene->cs.images.data(), scene->cs.texels.data(), tmp.mat_inputs.data(), color, 0};
        const DMaterial& m = tmp.materials[material];
        for (uint32_t i = 0; i < n; i++) {
            MatInputs in;
            std::memcpy(&in, &descs[material], sizeof in);
            if (m.flags & MF_TEXTURED) {
                eval_material_graph(ts, m.tex_first_node, m.tex_n_nodes & kTexCountMask, mk2(uv[2 * i], uv[2 * i + 1]), in);
            }
            std::memcpy(out26 + 26ull * i, &in, sizeof in);
        }
    });
}

// The interpreter's view of a material at n uv points, on the host, default colour pipeline: material_at (the folded record, 64
// words), material_alpha_at and material_emission_inputs_at (device/dtex.h). What a per-scene kernel's generated code must
// reproduce bit for bit (tests/test_specialise.py compiles that text for the host and compares).
AKR_API int32_t akr_probe_material_folded_host(akr_scene* scene, uint32_t material, uint32_t n, const float* uv, uint32_t* out64, float* alpha, float* emission3) {
    if (!scene || !uv || !out64) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_material_folded_host: NULL argument");
    if (material >= scene->cs.materials.size()) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_material_folded_host: material out of range");
    return guarded([&] {
        const CompiledScene& cs = scene->cs;
        const TexScene ts{cs.tex_nodes.data(), cs.images.data(), cs.texels.data(), cs.mat_inputs.data(), 0, 0};
        const DMaterial& folded = cs.materials[material];
        for (uint32_t i = 0; i < n; i++) {
            const vec2 p = mk2(uv[2 * i], uv[2 * i + 1]);
            DMaterial m = folded;
            material_at(ts, material, p, m);
            std::memcpy(out64 + 64ull * i, &m, sizeof m);
            const bool tex = (folded.flags & MF_TEXTURED) != 0;
            if (alpha) alpha[i] = tex ? material_alpha_at(ts, folded, material, p) : folded.base_alpha;
            if (emission3) {
                const vec3 e = tex ? material_emission_inputs_at(ts, folded, material, p) : folded.emission;
                emission3[3 * i] = e.x; emission3[3 * i + 1] = e.y; emission3[3 * i + 2] = e.z;
            }
        }
    });
}

AKR_API int32_t akr_probe_material_inputs(akr_context* ctx, akr_scene* scene, uint32_t material, uint32_t n, const float* uv, float* out26) {
    if (!scene || !uv || !out26) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_material_inputs: NULL argument");
    if (material >= scene->flat.materials.size()) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_probe_material_inputs: material out of range");
    return guarded([&] {
        const CompiledScene& cs = scene->cs;
        if (!ctx) {
            const TexScene ts{cs.tex_nodes.data(), cs.images.data(), cs.texels.data(), cs.mat_inputs.data(), 0, 0};
            const DMaterial& m = cs.materials[material];
            for (uint32_t i = 0; i < n; i++) {
                MatInputs in;
                if (cs.has_textures) in = cs.mat_inputs[material];
                else std::memcpy(&in, &scene->flat.materials[material], sizeof in);
                if (m.flags & MF_TEXTURED) {
                    eval_material_graph(ts, m.tex_first_node, m.tex_n_nodes & kTexCountMask, mk2(uv[2 * i], uv[2 * i + 1]), in);
                }
                std::memcpy(out26 + 26ull * i, &in, sizeof in);
            }
            return;
        }
        if (!cs.has_textures) throw std::invalid_argument("akr_probe_material_inputs: the scene has no textured material");
        if (scene->ctx != ctx) throw std::invalid_argument("akr_probe_material_inputs: scene belongs to another context");
        ctx->bind();
        DevBuf duv, dout;
        std::vector<float> uvv(uv, uv + 2ull * n);
        duv.upload(uvv);
        dout.alloc(26ull * n * 4);
        if (n) HIP_CHECK(launch_probe_material(probe_params(scene), material, n, duv.as<float>(), dout.as<uint32_t>(), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (n) HIP_CHECK(hipMemcpy(out26, dout.p, 26ull * n * 4, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
