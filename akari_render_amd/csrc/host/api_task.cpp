// api_task.cpp -- akr_render_task: the reference's render driver (akari_integrator/src/lib.rs:111-207) (C ABI of libakari_hip.so, include/akari_hip.h; shared internals: api_internal.h)
#include "api_internal.h"

extern "C" {

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

}  // extern "C"
