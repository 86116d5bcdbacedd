// api_probe.cpp -- host-side known-answer hooks and device probes for the tests: the TEST HOOKS of libakari_hip.so, declared in
// include/akari_hip_test.h and compiled in only with -DAKR_TEST_HOOKS=1 (the in-tree test build; build.py). Shared internals: api_internal.h
#if defined(AKR_TEST_HOOKS) && AKR_TEST_HOOKS
#include "api_internal.h"
#include "../../../include/akari_hip_test.h"
#include "../device/dinst.h"
#include "../device/disect.h"

extern "C" {

// ------------------------------------------------------------------------------------------------ host KAT hooks
AKR_TEST_API int32_t akr_host_stdrng_u64(uint64_t seed, uint32_t n, uint64_t* out) {
    if (!out) return fail(AKR_ERR_INVALID_ARGUMENT, "out is NULL");
    StdRng rng(seed);
    for (uint32_t i = 0; i < n; i++) out[i] = rng.next_u64();
    return AKR_OK;
}
AKR_TEST_API int32_t akr_host_chacha_block(const uint32_t* key8, uint64_t counter, uint64_t stream, int32_t rounds, uint32_t* out16) {
    if (!key8 || !out16) return fail(AKR_ERR_INVALID_ARGUMENT, "NULL argument");
    StdRng::chacha_block(key8, counter, stream, rounds, out16);
    return AKR_OK;
}
AKR_TEST_API int32_t akr_host_pcg32_states(uint64_t seed, uint64_t n, uint64_t* out2n) {
    if (!out2n) return fail(AKR_ERR_INVALID_ARGUMENT, "out is NULL");
    StdRng rng(seed);
    for (uint64_t i = 0; i < n; i++) {
        Pcg32 p = pcg_new_seq_offset(i, rng.next_u64());
        out2n[2 * i] = p.state;
        out2n[2 * i + 1] = p.inc;
    }
    return AKR_OK;
}
AKR_TEST_API int32_t akr_host_pcg_start(uint64_t* state, uint64_t inc) {
    if (!state) return fail(AKR_ERR_INVALID_ARGUMENT, "state is NULL");
    Pcg32 p{*state, inc};
    pcg_start(p, pcg_start_constants());
    *state = p.state;
    return AKR_OK;
}
// device/drng.h on the host: reverse_bits32(sobol_dim1(i)) by the defining loop and by the five-step butterfly the kernels use
AKR_TEST_API int32_t akr_host_sobol_dim1(uint32_t n, const uint32_t* index, uint32_t* by_loop, uint32_t* by_butterfly) {
    if (!index || !by_loop || !by_butterfly) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_sobol_dim1: NULL argument");
    for (uint32_t k = 0; k < n; k++) {
        by_loop[k] = reverse_bits32(sobol_dim1(index[k]));
        by_butterfly[k] = sobol_dim1_reversed(index[k]);
    }
    return AKR_OK;
}
// device/drng.h fastmod_u32 on the host: a[k] % d[k] through the precomputed constant
AKR_TEST_API int32_t akr_host_fastmod(uint32_t n, const uint32_t* a, const uint32_t* d, uint32_t* out) {
    if (!a || !d || !out) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_fastmod: NULL argument");
    for (uint32_t k = 0; k < n; k++) {
        if (d[k] == 0) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_fastmod: divisor 0");
        out[k] = fastmod_u32(a[k], fastmod_magic(d[k]), d[k]);
    }
    return AKR_OK;
}
// device/dinst.h on the host: the conservative reject of a candidate (tri_may_hit) next to the exact test it stands in front of
// (woop_precompute + tri_test), per item: ray = o.xyz d.xyz tmin tlimit, tri = A B C (world space, f32). exact: bit 0 accept, t in out_t.
AKR_TEST_API int32_t akr_host_tri_pretest(uint32_t n, const float* rays8, const float* tris9, float plane_shift, uint32_t* may, uint32_t* exact, float* out_t) {
    if (!rays8 || !tris9 || !may || !exact) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_tri_pretest: NULL argument");
    for (uint32_t k = 0; k < n; k++) {
        const float* r = rays8 + 8ull * k;
        const float* t = tris9 + 9ull * k;
        const vec3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]), A = mk3(t[0], t[1], t[2]), B = mk3(t[3], t[4], t[5]), C = mk3(t[6], t[7], t[8]);
        may[k] = tri_may_hit(o, d, A, B, C, r[6], r[7], plane_shift) ? 1u : 0u;
        float w[12], tt, u, v;
        woop_precompute(A, B, C, w);
        exact[k] = tri_test(o, d, make_float4(w[0], w[1], w[2], w[3]), make_float4(w[4], w[5], w[6], w[7]), make_float4(w[8], w[9], w[10], w[11]), r[6], r[7], tt, u, v) ? 1u : 0u;
        if (out_t) out_t[k] = tt;
    }
    return AKR_OK;
}
AKR_TEST_API int32_t akr_host_alias_table(const float* weights, uint32_t n, uint32_t* j, float* t, float* pdf) {
    if (!weights || !j || !t || !pdf || n == 0) return fail(AKR_ERR_INVALID_ARGUMENT, "akr_host_alias_table: bad argument");
    return guarded([&] {
        std::vector<float> w(weights, weights + n), p;
        std::vector<AliasEntry> e;
        build_alias_table(w, e, p);
        for (uint32_t i = 0; i < n; i++) { j[i] = e[i].j; t[i] = e[i].t; pdf[i] = p[i]; }
    });
}

// ------------------------------------------------------------------------------------------------ probes
AKR_TEST_API int32_t akr_probe_math(akr_context* ctx, uint32_t n, const float* x, float* s, float* c, float* l) {
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
AKR_TEST_API int32_t akr_probe_bsdf(akr_context* ctx, const akr_material_desc* m, const float* table, int32_t mode, const float* wo, uint32_t n,
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
AKR_TEST_API int32_t akr_probe_intersect(akr_context* ctx, akr_scene* scene, uint32_t n, const float* rays, uint32_t* hit_inst_prim, float* bary) {
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
AKR_TEST_API int32_t akr_probe_surface_interaction(akr_context* ctx, akr_scene* scene, uint32_t n, const uint32_t* inst_prim, const float* bary,
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

AKR_TEST_API int32_t akr_host_pmj02bn_tables(uint32_t* sets, uint16_t* bluenoise) {
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
AKR_TEST_API int32_t akr_host_decode_png(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba, uint64_t capacity) {
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
AKR_TEST_API int32_t akr_host_decode_jpeg(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba, uint64_t capacity) {
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
AKR_TEST_API int32_t akr_host_decode_tiff(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba, uint64_t capacity) {
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
AKR_TEST_API int32_t akr_host_decode_dds(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba, uint64_t capacity) {
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
AKR_TEST_API int32_t akr_host_decode_exr(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, float* rgba, uint64_t capacity_floats) {
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
AKR_TEST_API int32_t akr_probe_material_inputs_host(akr_scene* scene, uint32_t material, uint32_t color, uint32_t n, const float* uv, float* out26) {
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
AKR_TEST_API int32_t akr_probe_material_folded_host(akr_scene* scene, uint32_t material, uint32_t n, const float* uv, uint32_t* out64, float* alpha, float* emission3) {
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

AKR_TEST_API int32_t akr_probe_material_inputs(akr_context* ctx, akr_scene* scene, uint32_t material, uint32_t n, const float* uv, float* out26) {
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
#endif  // AKR_TEST_HOOKS
