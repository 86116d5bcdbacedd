// api_aux.cpp -- the aov, gpt and mcmc_opt integrators (C ABI of libakari_hip.so, include/akari_hip.h; shared internals: api_internal.h)
#include "api_internal.h"

extern "C" {

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
    int32_t rc = pt_begin(ctx, scene, &pc, film, &se, /*for_pt_kernel=*/false);
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
    int32_t rc = pt_begin(ctx, scene, &pc, film, &pt, /*for_pt_kernel=*/false);
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
            auto owned = [&](uint32_t x, uint32_t y) { return tile_owner(x / tw, y / th, shard->shard_count) == shard->shard_rank; };
            std::vector<uint32_t> list;
            for (uint32_t t : akr_api::owned_tiles(tiles_x, tiles_y, shard->shard_rank, shard->shard_count)) {
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
}  // extern "C"
int32_t akr_api::mcmc_render_impl(akr_context* ctx, akr_scene* scene, const akr_mcmc_config* cfg, akr_film* film, akr_mcmc_result* result,
                                  uint32_t* chain_states, akr_pt_stats* stats, const std::function<void(uint32_t, double)>& on_pass,
                                  uint32_t shard_rank, uint32_t shard_count, akr_mcmc_partial* partial) {
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
    int32_t rc = pt_begin(ctx, scene, &pc, film, &se, /*for_pt_kernel=*/false);
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

extern "C" {
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

}  // extern "C"
