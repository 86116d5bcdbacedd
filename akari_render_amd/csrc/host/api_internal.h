// api_internal.h -- what the translation units of the C ABI share (api_common.cpp: errors, options, contexts, films; api_scene.cpp;
// api_pt.cpp: the `pt` sessions; api_aux.cpp: aov / gpt / mcmc_opt; api_task.cpp: akr_render_task; api_probe.cpp: test hooks).
// Every entry point catches C++ exceptions and HIP errors and turns them into an akr_status plus a thread-local message; nothing
// throws or aborts across the boundary.
#pragma once
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


namespace akr_api {
using namespace akr;


extern thread_local std::string g_last_error;  // api_common.cpp

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

inline int32_t fail(int32_t code, const std::string& msg) {
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

}  // namespace akr_api
using namespace akr_api;


struct akr_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t props;
    void bind() const { HIP_CHECK(hipSetDevice(device)); }
    SpecCache spec_cache;  // per-scene kernels loaded on this device (host/specialise.cpp)
    // PreComputedTables (svm/surface/precompute.rs:133-145) as this context's first scene that needed it computed it: the table is
    // a constant of the algorithm (fixed seed stream, 2^20 samples per entry: 1.8 s on an MI355X), not of the scene
    std::mutex ggx_mutex;
    std::vector<float> ggx_cache;
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
    DevBuf in2_tlas_leaves, in2_mesh_tris, in2_mesh_pos, in2_mesh_meta, in2_mesh_normals, in2_inst_mats, in2_share_bits;  // meshes + instances (scene_inst.cpp)
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
    DevBuf owned_tiles;        // shard_count > 1: PtParams.owned_tiles
    uint32_t n_owned_tiles = 0;
    // wavefront schedule (wf_kernels.hip): path state SoA + ray queues
    bool wavefront = false;
    int sched_trial = 0;  // flattened scenes, option wavefront = -1: 1 = the first blocking akr_pt_passes call times both schedules and keeps the faster (api_pt.cpp), 2 = done
    DevBuf wf_state, wf_queues, wf_ctrl, wf_pend, wf_carry;
    // option wf_sort: keys of the queue entries, the sorted copies the trace kernel reads, rocPRIM's scratch
    bool wf_sort = false;
    DevBuf wf_keys, wf_sorted, wf_sort_tmp;
    uint32_t *wf_sorted_closest = nullptr, *wf_sorted_shadow = nullptr, *wf_sorted_keys = nullptr;
    WfBuffers wf;
    uint32_t wf_slots = 0, wf_trace_blocks = 0;
    // slot groups (option wf_groups; api_pt.cpp wf_run): each with its own queues, counters and stream
    std::vector<WfBuffers> wf_group;
    std::vector<hipStream_t> wf_streams;
    std::vector<hipEvent_t> wf_join;
    hipEvent_t wf_fork = nullptr;
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
    bool arith_relaxed = false;  // option arith = 1 and the session is one the relaxed tier covers: launches go to pt_kernels_relaxed.hip
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
        for (hipStream_t st : wf_streams) (void)hipStreamDestroy(st);
        for (hipEvent_t ev : wf_join) (void)hipEventDestroy(ev);
        if (wf_fork) (void)hipEventDestroy(wf_fork);
        for (auto& ev : pending) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    }
};

namespace akr_api {
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

// api_pt.cpp
int32_t pt_begin(akr_context* ctx, akr_scene* scene, const akr_pt_config* cfg, akr_film* film, akr_pt_session** out, bool for_pt_kernel);
void fill_params(akr_pt_session* se, uint32_t n_passes, uint32_t last_pass_spp);
uint32_t session_samples(const akr_pt_config& c);
// the tiles (row-major ids) rank `rank` of `count` owns, in Morton order (kernels.h tile_owner)
std::vector<uint32_t> owned_tiles(uint32_t tiles_x, uint32_t tiles_y, uint32_t rank, uint32_t count);
// api_scene.cpp
void scene_finish(akr_scene* s);
void scene_spec_header(akr_scene* scene, std::string& out);
// api_aux.cpp: shard_count > 1 = this rank's share (akr_mcmc_render_shard); on_pass: akr_render_task's progress hook
int32_t mcmc_render_impl(akr_context* ctx, akr_scene* scene, const akr_mcmc_config* cfg, akr_film* film, akr_mcmc_result* result, uint32_t* chain_states,
                         akr_pt_stats* stats, const std::function<void(uint32_t, double)>& on_pass, uint32_t shard_rank = 0, uint32_t shard_count = 1,
                         akr_mcmc_partial* partial = nullptr);
}  // namespace akr_api
