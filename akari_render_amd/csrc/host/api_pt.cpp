// api_pt.cpp -- the `pt` integrator: kernel parameters, sessions, both schedules, per-scene kernels (C ABI of libakari_hip.so, include/akari_hip.h; shared internals: api_internal.h)
#include "api_internal.h"

void akr_api::fill_params(akr_pt_session* se, uint32_t n_passes, uint32_t last_pass_spp) {
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
        const bool bvh = !cs.bvh_nodes.empty() || cs.instanced.on;
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
        if (!cs.instanced.on && total <= (bvh ? kStageMaxBytesBvh : kStageMaxBytes) && (!bvh || other + total <= lds_budget)) {  // (the instanced-scene kernels do not stage)  // all of it or nothing (a TEX kernel reads its tables through LDS addresses)
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
        const bool bvh = !cs.bvh_nodes.empty() || cs.instanced.on;
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
        p.defer_metal = (want && (!bvh || cs.has_textures) && !c.force_diffuse && !cs.instanced.on) ? mask : 0u;
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
    if (p.shard_count > 1) {
        if (se->owned_tiles.p == nullptr && se->n_owned_tiles == 0) {  // once per session: the configuration does not change
            const std::vector<uint32_t> list = owned_tiles(p.tiles_x, p.tiles_y, p.shard_rank, p.shard_count);
            se->n_owned_tiles = (uint32_t)list.size();
            if (!list.empty()) se->owned_tiles.upload(list);
        }
        p.owned_tiles = se->owned_tiles.as<uint32_t>();
        p.n_items = se->n_owned_tiles * p.tile_w * p.tile_h;
    } else {
        p.owned_tiles = nullptr;
        p.n_items = p.tiles_x * p.tiles_y * p.tile_w * p.tile_h;
    }
}
std::vector<uint32_t> akr_api::owned_tiles(uint32_t tiles_x, uint32_t tiles_y, uint32_t rank, uint32_t count) {
    std::vector<std::pair<uint32_t, uint32_t>> mine;  // (Morton code, tile)
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++)
            if (tile_owner(tx, ty, count) == rank) mine.emplace_back(tile_morton(tx, ty), ty * tiles_x + tx);
    std::sort(mine.begin(), mine.end());
    std::vector<uint32_t> out;
    out.reserve(mine.size());
    for (const auto& m : mine) out.push_back(m.second);
    return out;
}

// Which schedule renders this session: the persistent-lane megakernel (pt_kernels.hip, pt_inst_kernels.hip) or the wavefront schedule --
// trace / shade kernels with the path state in HBM (wf_kernels.hip; needs a scene with a tree). Option wavefront: 1 = wherever it can run,
// 0 = never, -1 (default) = the library decides:
//   * flattened scenes: the megakernel, which measured faster on every one (on the 10 M-triangle hall both schedules
//     are bound by the memory system's rate for random 64-byte records, and the wavefront schedule pays for streaming the path state on top);
//   * scenes kept as meshes + instances: the WAVEFRONT schedule for pt sessions of at least wf_auto_items() pixels (round 6). The two-level
//     traversal with its exact test spills 172 registers inside the megakernel (one lane = one whole path) and none in k_wf_trace, and the
//     trace kernel refills a wave's idle lanes where the megakernel's wait: 1080p forest 1000 x 10 k triangles 163 -> 228 Msamples/s, 4K
//     177 -> 282 (x 100 k: 113 -> 136 with two slot groups, 4K 125 -> 179). Below that size the persistent trace kernel's 262 k lanes
//     are not filled and the megakernel wins (1024 x 1024, x 100 k: 124 against 85) -- profiles/r6_kept_schedules.txt. Kept scenes with
//     texture-fed materials as well: k_wf_shade interprets the shader graphs where the megakernel has its per-scene kernels, and still
//     the same forest with image-textured leaves and bark renders at 193 against 157 Msamples/s (x 100 k: 123 against 100).
// The option is process-wide and aov / gpt / mcmc_opt sessions come through here too: they render with their own kernels; a scene without a
// tree (64 triangles or fewer, no force_bvh) has no wavefront kernels and renders with the megakernel whatever the option says.
// How large: the persistent trace kernel wants its 262 k lanes refilled many times over, and the larger the meshes the longer a launch's last
// rays take. Measured crossovers (tools/kept_schedules.py at seven frame sizes, profiles/r6_kept_schedules.txt), with a launch's last rays carried
// into the next one (option wf_carry, the default): 0.5 M pixels for the forest of 10 k-triangle meshes (1.7 MB of per-mesh data; 800 x 600: 145 vs
// 147, 1024 x 768: 167 -> 189), 0.8 M for 100 k-triangle ones (17 MB; 1024 x 768: 118 vs 122, 1024 x 1024: 138 -> 150). Without carried rays: 0.7 M
// and 1.45 M. 2 M, the constant of the first version, beyond what was measured.
static uint32_t wf_auto_items(const akr_scene* scene) {
    const uint64_t mesh_bytes = ((uint64_t)scene->cs.instanced.nodes.size() + scene->cs.instanced.mesh_tris.size()) * 4u;
    const bool carry = tuning().wf_carry != 0 && tuning().wf_sort == 0;
    return (uint32_t)std::min<uint64_t>(2000000u, carry ? 500000u + mesh_bytes / 55u : 700000u + mesh_bytes / 18u);
}
static uint32_t session_items(const akr_pt_config& c, uint32_t width, uint32_t height) {  // = fill_params' n_items
    const uint32_t tw = c.tile_w ? c.tile_w : 32, th = c.tile_h ? c.tile_h : 32;
    const uint32_t tiles_x = (width + tw - 1) / tw, tiles_y = (height + th - 1) / th;
    if (c.shard_count <= 1) return tiles_x * tiles_y * tw * th;
    uint32_t n = 0;
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++) n += tile_owner(tx, ty, c.shard_count) == c.shard_rank ? 1u : 0u;
    return n * tw * th;
}
static bool choose_wavefront(const akr_scene* scene, const akr_pt_config& cfg, bool for_pt_kernel) {
    const int opt = tuning().wavefront;
    const bool can = !scene->cs.bvh_nodes.empty() || scene->cs.instanced.on;
    if (opt == 0 || !can) return false;
    if (opt > 0) return true;
    return for_pt_kernel && scene->cs.instanced.on &&
           session_items(cfg, scene->flat.camera.width, scene->flat.camera.height) >= wf_auto_items(scene);
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
    // carried rays (wf_kernels.hip): a record per (kind, slot) = 12 words + this scene's traversal stack. Not with option wf_sort (a carried ray has no key).
    w.pend = nullptr; w.carry = nullptr; w.carry_words = 0; w.n_slots = n_slots;
    // (option values above 1 = test hook: that launch size, and waves hand over early and nearly whole -- small frames carry thousands of rays)
    const bool carry_test = tuning().wf_carry > 1;
    w.carry_queue = carry_test ? (uint32_t)tuning().wf_carry : 65536u;
    w.carry_lanes = carry_test ? 56u : 16u;
    w.carry_steps = carry_test ? 4u : 48u;
    if (tuning().wf_carry != 0 && !se->wf_sort && n_slots > 0) {
        w.carry_words = (12u + se->params.sc.bvh_stack_depth + 3u) & ~3u;  // (16-byte aligned records)
        se->wf_pend.alloc(n * sizeof(uint32_t));
        se->wf_carry.alloc(2 * n * (size_t)w.carry_words * sizeof(uint32_t));
        w.pend = (uint32_t*)se->wf_pend.p;
        w.carry = (uint32_t*)se->wf_carry.p;
    }
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
    // ---- slot groups. A trace launch ends with its slowest rays, a hundred dependent fetches deep, while the rest of the chip idles, and a
    // launch group is a hundred such launches (the slots with the longest paths decide): measured on the kept 1080p forest that tail is
    // 19 ms of 73 ms (1000 x 10 k triangles) and 69 ms of 144 ms (x 100 k) per 8 spp -- tools/kept_schedules.py at four frame sizes,
    // HISTORY R6.6. The slots are therefore divided into groups that run the same init / trace / shade chain on streams of their own:
    // one group's kernels fill the CUs another's tail leaves idle. Nothing a slot computes depends on which group it is in.
    // Measured (profiles/r6_kept_schedules.txt, 1080p): two groups 113 -> 136 Msamples/s on the kept forest of 100 k-triangle meshes (its trace
    // launches wait on memory -- the meshes do not fit the L2s -- and the other group's shade launch streams meanwhile), 228 -> 221 on the
    // 10 k-triangle one, 208 -> 213 on the flattened hall; four groups and more lose everywhere (each group's launches are shorter and their
    // ends no better filled). Option wf_groups: 0 = the library decides (two for a kept scene whose meshes exceed the L2s, else one).
    const akr_scene* sc = se->scene;
    const size_t mesh_bytes = sc->cs.instanced.on ? (sc->cs.instanced.nodes.size() + sc->cs.instanced.mesh_tris.size()) * 4 : 0;
    const int opt_groups = tuning().wf_groups;
    // With carried rays (option wf_carry, the default since the end of round 6) a launch has no such tail and one group wins at every size measured
    // (1080p, x 100 k: 175 against 158 with two; 4K: 220 against 203): the automatic choice is two groups only without them.
    uint32_t groups = se->wf_sort ? 1u : (opt_groups > 0 ? (uint32_t)opt_groups : (mesh_bytes > (16u << 20) && w.carry == nullptr ? 2u : 1u));
    groups = std::min(groups, std::max(1u, n_slots / 65536u));  // (small frames: a group should still be a few waves per CU)
    se->wf_ctrl.alloc((size_t)groups * 8 * sizeof(uint32_t));  // per group: qcount[4], qhead, n_active
    se->wf_group.clear();
    for (uint32_t g = 0; g < groups; g++) {
        WfBuffers wg = w;
        // boundaries on whole 1024-slot tiles
        auto bound = [&](uint32_t k) { return k >= groups ? n_slots : (uint32_t)(((uint64_t)n_slots * k / groups) & ~1023ull); };
        wg.slot_base = bound(g);
        wg.slot_end = bound(g + 1);
        for (int k = 0; k < 2; k++) {  // a group's queues: its share of the session's
            wg.queue_closest[k] = w.queue_closest[k] + wg.slot_base;
            wg.queue_shadow[k] = w.queue_shadow[k] + wg.slot_base;
        }
        uint32_t* c = (uint32_t*)se->wf_ctrl.p + 8 * g;
        wg.qcount = c; wg.qhead = c + 4; wg.n_active = c + 5;
        se->wf_group.push_back(wg);
    }
    for (hipStream_t st : se->wf_streams) (void)hipStreamDestroy(st);
    se->wf_streams.clear();
    if (groups > 1) {
        for (uint32_t g = 0; g < groups; g++) {
            hipStream_t st = nullptr;
            HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            se->wf_streams.push_back(st);
            hipEvent_t ev = nullptr;
            HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            se->wf_join.push_back(ev);
        }
        if (!se->wf_fork) HIP_CHECK(hipEventCreateWithFlags(&se->wf_fork, hipEventDisableTiming));
    }
    // persistent trace kernel: as many workgroups as the CUs hold at once (occupancy API: registers + this tree's LDS stacks)
    // ... but no more than the scene's data can feed: every resident wave is 64 more rays gathering from the tree, and past the L2s and the
    // 256 MB infinity cache more rays in flight only evict each other's nodes (the trace kernel of a flattened scene fits six waves per SIMD,
    // the megakernel runs four). Measured, workgroups per CU 4 / 5 / 6: hall of 0.1 M triangles (21 MB on the device) 337 / 353 / 356 Msamples/s,
    // 1 M (214 MB) 300 / 310 / 307, 3 M (642 MB) 273 / 281 / 265, 10 M (2.1 GB) 253 / 247 / 228, flattened forest of 10 M (2.6 GB) 370 / 366 / 351.
    uint32_t per_cu = wf_trace_blocks_per_cu(se->params);
    const uint64_t scene_bytes = sc->device_bytes;
    per_cu = std::min(per_cu, scene_bytes > (1200ull << 20) ? 4u : (scene_bytes > (400ull << 20) ? 5u : 8u));
    if (const char* e = std::getenv("AKR_WF_TRACE_BLOCKS_PER_CU")) per_cu = (uint32_t)std::max(1, std::min(8, std::atoi(e)));  // (measurement hook)
    se->wf_trace_blocks = (uint32_t)se->ctx->props.multiProcessorCount * per_cu;
}

// One launch group of the wavefront schedule = `fused` passes for every slot: init, then trace/shade iterations until
// no slot is active. The host only looks at the device every 16 iterations (every 4 once few slots are left).
static void wf_run(akr_pt_session* se) {
    hipStream_t main_st = se->ctx->stream;
    const PtParams& p = se->params;
    const uint32_t groups = (uint32_t)se->wf_group.size();
    HIP_CHECK(hipMemsetAsync(se->wf_ctrl.p, 0, se->wf_ctrl.bytes, main_st));
    std::vector<hipStream_t> streams(groups, main_st);
    if (groups > 1) {  // the groups' streams start after everything the session's stream holds so far ...
        HIP_CHECK(hipEventRecord(se->wf_fork, main_st));
        for (uint32_t g = 0; g < groups; g++) {
            streams[g] = se->wf_streams[g];
            HIP_CHECK(hipStreamWaitEvent(streams[g], se->wf_fork, 0));
        }
    }
    struct Join {  // ... and the session's stream goes on after all of them, also when this function is left by an exception
        akr_pt_session* se; std::vector<hipStream_t>& st; hipStream_t main_st;
        ~Join() {
            if (st.size() < 2) return;
            for (size_t g = 0; g < st.size(); g++) {
                (void)hipEventRecord(se->wf_join[g], st[g]);
                (void)hipStreamWaitEvent(main_st, se->wf_join[g], 0);
            }
        }
    } join{se, streams, main_st};
    for (uint32_t g = 0; g < groups; g++) HIP_CHECK(launch_wf_init(p, se->wf_group[g], streams[g]));
    int check_every = 16, since_check = 0;  // iterations between two looks at the device (every look drains the stream)
    uint32_t q = 0;
    std::vector<uint8_t> done(groups, 0);
    for (uint64_t iter = 0;; iter++) {
        // option wf_sort (one group): the sizes of the queue this iteration traces (rocPRIM wants the element count on the host): one small
        // read-back per iteration, before the counters are reset -- it also ends the loop the moment the last path has finished
        uint32_t nc = 0, ns = 0;
        const bool sort_now = se->wf_sort && iter > 0;  // (the first iteration's camera rays are in pixel order: coherent as they are)
        if (sort_now) {
            uint32_t counts[6];
            HIP_CHECK(hipMemcpyAsync(counts, se->wf_ctrl.p, sizeof counts, hipMemcpyDeviceToHost, main_st));
            HIP_CHECK(hipStreamSynchronize(main_st));
            if (counts[5] == 0) break;  // n_active after the last shade
            nc = counts[2 * q];
            ns = counts[2 * q + 1];
        }
        for (uint32_t g = 0; g < groups; g++) {
            if (done[g]) continue;
            const WfBuffers& wg = se->wf_group[g];
            hipStream_t st = streams[g];
            // queue q holds the rays to trace. (The head, the other queue's counts and the active counter are reset by the kernels themselves:
            // k_wf_trace zeroes n_active, k_wf_shade the counts of the queue just traced and the head.)
            if (sort_now) {
                WfBuffers sorted = wg;
                HIP_CHECK(wf_sort_pairs(se->wf_sort_tmp.p, se->wf_sort_tmp.bytes, wg.key_closest[q], se->wf_sorted_keys, wg.queue_closest[q], se->wf_sorted_closest, nc, st));
                HIP_CHECK(wf_sort_pairs(se->wf_sort_tmp.p, se->wf_sort_tmp.bytes, wg.key_shadow[q], se->wf_sorted_keys, wg.queue_shadow[q], se->wf_sorted_shadow, ns, st));
                sorted.queue_closest[q] = se->wf_sorted_closest;
                sorted.queue_shadow[q] = se->wf_sorted_shadow;
                HIP_CHECK(launch_wf_trace(p, sorted, q, se->wf_trace_blocks, st));
            } else {
                HIP_CHECK(launch_wf_trace(p, wg, q, se->wf_trace_blocks, st));
            }
            HIP_CHECK(launch_wf_shade(p, wg, 1 - q, st));
        }
        q = 1 - q;
        if (!se->wf_sort && ++since_check >= check_every) {
            since_check = 0;
            std::vector<uint32_t> n_active(groups, 0);
            for (uint32_t g = 0; g < groups; g++)
                if (!done[g]) HIP_CHECK(hipMemcpyAsync(&n_active[g], se->wf_group[g].n_active, sizeof(uint32_t), hipMemcpyDeviceToHost, streams[g]));
            bool all = true;
            for (uint32_t g = 0; g < groups; g++) {
                if (done[g]) continue;
                HIP_CHECK(hipStreamSynchronize(streams[g]));
                if (n_active[g] == 0) done[g] = 1; else all = false;
            }
            if (all) break;
            // the last slots' paths: launches with next to nothing to do -- look more often, so that fewer of them run for nothing
            uint64_t left = 0;
            for (uint32_t g = 0; g < groups; g++) left += n_active[g];
            check_every = left * 64 < se->wf_slots ? 4 : 16;
        }
        if (iter > (1ull << 26)) throw RenderError("wavefront schedule did not terminate");
    }
}

// Flattened scenes under option wavefront = -1. Which schedule is faster there depends on the scene -- 1080p, 10 M triangles: the closed hall 308
// (megakernel) against 252 Msamples/s (wavefront), the open forest 309 against 369 -- through how much the lengths of a wave's traversals differ, which
// no number the compiler has predicts. So a render that can pay for it MEASURES: the first blocking akr_pt_passes call runs two passes under the
// megakernel, two under the wavefront schedule (HIP-event time per sample of each), and the session goes on with the faster one. Films are the same
// bit for bit under either schedule and under any sequence of them (both start a launch from, and leave behind, the session's sampler states and
// film). "Can pay": a pt session without textures / relaxed tier / ray sort on a scene with a tree of >= 256 MB (small scenes: the megakernel wins
// by up to 2 x), >= 1 M pixels (the persistent trace kernel wants its lanes refilled), >= 16 passes to render; a scene whose paths practically
// never end at a light or in the open (shaded vertices per closest-hit ray >= 0.97 in the megakernel's two passes: the hall 0.999) skips the
// wavefront half -- it has measured slower on every such scene. Option sched_trial: 0 = never, 1 = every session on a scene with a tree (tests).
static bool schedule_trial_eligible(const akr_pt_session* se, bool for_pt_kernel) {
    const TuningOptions t = tuning();
    const akr_scene* sc = se->scene;
    if (t.wavefront != -1 || t.sched_trial == 0 || !for_pt_kernel || se->wavefront || se->arith_relaxed || se->spec_active) return false;
    if (sc->cs.instanced.on || sc->cs.bvh_nodes.empty() || t.wf_sort != 0) return false;
    if (t.sched_trial == 1) return true;
    const uint64_t passes = (session_samples(se->cfg) + se->cfg.spp_per_pass - 1) / se->cfg.spp_per_pass;
    return !sc->cs.has_textures && sc->device_bytes >= (256ull << 20) && passes >= 16 &&
           session_items(se->cfg, sc->flat.camera.width, sc->flat.camera.height) >= 1000000u;
}
static void read_stats(akr_pt_session* se, akr_pt_stats* stats);
static void wf_release(akr_pt_session* se) {
    for (DevBuf* b : {&se->wf_state, &se->wf_queues, &se->wf_ctrl, &se->wf_pend, &se->wf_carry, &se->wf_keys, &se->wf_sorted, &se->wf_sort_tmp}) b->release();
    se->wf_group.clear();
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
uint32_t akr_api::session_samples(const akr_pt_config& c) { return c.sample_count ? c.sample_count : c.spp; }

extern "C" {

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

}  // extern "C"
// for_pt_kernel: the session will launch k_pt_pass (akr_pt_passes). The aov / gpt / mcmc_opt integrators come through here as well for
// sampler states, counters and the kernel parameter block, but launch their own kernels, which interpret shader graphs: they must not
// get a per-scene kernel -- nor its parameter block, which has no graph value slots in LDS (found by the full GPU suite: a cached
// per-scene kernel made a later mcmc_opt render of the same scene evaluate its graphs without value slots).
int32_t akr_api::pt_begin(akr_context* ctx, akr_scene* scene, const akr_pt_config* cfg, akr_film* film, akr_pt_session** out, bool for_pt_kernel) {
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
        se->wavefront = choose_wavefront(scene, *cfg, for_pt_kernel);
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
            // The relaxed arithmetic tier (pt_kernels_relaxed.hip): the precompiled megakernels of flattened scenes. Everything else --
            // kept scenes, the wavefront schedule, aov / gpt / mcmc_opt -- stays on the contract whatever the option says.
            se->arith_relaxed = t.arith == 1 && for_pt_kernel && !scene->cs.instanced.on && !se->wavefront;
            if (!for_pt_kernel) se->spec_status = "not a pt session";
            else if (se->arith_relaxed) se->spec_status = "relaxed arithmetic tier: precompiled kernels";
            else if (se->wavefront) se->spec_status = "wavefront schedule";
            else if (!scene->cs.has_textures) se->spec_status = "the scene has no texture-fed material";
            else if (t.specialise == 0) se->spec_status = "option specialise = 0";
            else if (cfg->force_diffuse) se->spec_status = "force_diffuse kernels evaluate no surface graphs";
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
                rq.bvh = !scene->cs.bvh_nodes.empty() || scene->cs.instanced.on;
                rq.inst = scene->cs.instanced.on;
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
        se->sched_trial = schedule_trial_eligible(se.get(), for_pt_kernel) ? 1 : 0;
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = se.release();
    });
}
// pt_kernels_relaxed.hip: launch_pt_pass of the relaxed arithmetic tier (its PtParams is this one, in another namespace)
extern "C" hipError_t akr_launch_pt_pass_relaxed(const void* params, hipStream_t stream);
extern "C" {
AKR_API int32_t akr_pt_begin(akr_context* ctx, akr_scene* scene, const akr_pt_config* cfg, akr_film* film, akr_pt_session** out) {
    return pt_begin(ctx, scene, cfg, film, out, /*for_pt_kernel=*/true);
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
        auto launch = [&](uint32_t max_fused) {  // one launch (group) of up to max_fused of the passes left
            uint32_t fused = 0, last = 0, done = se->spp_done;
            while (fused < max_fused && fused < left && done < total) {
                last = std::min(total - done, se->cfg.spp_per_pass);
                done += last;
                fused++;
            }
            fill_params(se, fused, last);
            LaunchTimer timer(se);
            if (se->wavefront) wf_run(se);
            else if (se->arith_relaxed) HIP_CHECK(akr_launch_pt_pass_relaxed(&se->params, se->ctx->stream));
            else HIP_CHECK(launch_pt_pass(se->params, se->ctx->stream, se->spec_active ? se->spec->fn : nullptr));
            timer.stop();
            se->spp_done = done;
            se->n_launches++;
            se->passes_launched += fused;
            left -= fused;
        };
        if (se->sched_trial == 1 && blocking && left >= 4 && total - se->spp_done >= 4 * se->cfg.spp_per_pass) {  // (schedule_trial_eligible)
            se->sched_trial = 2;
            auto timed = [&](double& ms_per_sample, double& shaded_per_closest) {  // two passes under the current schedule
                HIP_CHECK(hipStreamSynchronize(se->ctx->stream));
                akr_pt_stats a, b;
                read_stats(se, &a);
                launch(2);
                read_stats(se, &b);
                ms_per_sample = (b.kernel_ms - a.kernel_ms) / (double)std::max<uint64_t>(1, b.n_samples - a.n_samples);
                shaded_per_closest = (double)(b.n_shaded - a.n_shaded) / (double)std::max<uint64_t>(1, b.n_closest - a.n_closest);
            };
            double mk = 0.0, wf = 0.0, ratio = 0.0, unused = 0.0;
            timed(mk, ratio);
            char msg[200];
            if (ratio >= 0.97 && tuning().sched_trial != 1) {
                std::snprintf(msg, sizeof msg, "megakernel (no trial of the other schedule: %.3f shaded vertices per closest-hit ray, a closed scene)", ratio);
            } else {
                se->wavefront = true;
                fill_params(se, 1, se->cfg.spp_per_pass);  // for n_items
                wf_allocate(se, se->params.n_items);
                timed(wf, unused);
                const bool keep = wf < 0.95 * mk;
                // (a session on the megakernel never says "wavefront": callers tell the schedule by that word)
                std::snprintf(msg, sizeof msg, keep ? "wavefront schedule (timed trial, 2 passes each: %.3g ns per sample against the megakernel's %.3g)"
                                                    : "megakernel (timed trial, 2 passes each: %.3g ns per sample against the other schedule's %.3g)",
                              (keep ? wf : mk) * 1e6, (keep ? mk : wf) * 1e6);
                if (!keep) {
                    se->wavefront = false;
                    wf_release(se);
                }
            }
            se->spec_status = msg;
        }
        while (left > 0 && se->spp_done < total) launch(kMaxFusedPasses);
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
                             (se->params.defer_metal != 0 ? 8u : 0u) | (se->arith_relaxed ? 16u : 0u);
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
        if (se->wavefront && se->wf.carry != nullptr && se->counters.p) {  // what the launches so far carried over (wf_kernels.hip; counter 7)
            se->ctx->bind();
            HIP_CHECK(hipStreamSynchronize(se->ctx->stream));
            std::vector<uint64_t> stripes(8 * kStatStripes);
            HIP_CHECK(hipMemcpy(stripes.data(), se->counters.p, stripes.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
            uint64_t carried = 0;
            for (uint32_t k = 0; k < kStatStripes; k++) carried += stripes[8 * k + 7];
            std::snprintf(info->status, sizeof info->status, "%s; %llu rays carried into a later trace launch", se->spec_status.c_str(), (unsigned long long)carried);
        }
    });
}AKR_API int32_t akr_pt_end(akr_pt_session* se, akr_pt_stats* stats) {
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

}  // extern "C"
