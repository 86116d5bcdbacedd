/* akari_hip_test.h -- TEST HOOKS of libakari_hip.so. Not part of the drop-in boundary.
 *
 * include/akari_hip.h is what a host application binds. The entry points below exist so that the test suite can
 * compare single pieces of the library with the oracle -- the host's random number generators and table builders, the image decoders,
 * single device functions (elementary functions, BSDF, intersection, surface interaction, shader-graph evaluation) -- through the
 * same C ABI the product is called through. They are compiled into the library, and exported, only when it is built with
 * -DAKR_TEST_HOOKS=1: the in-tree build that `python -m akari_render_amd.build` makes for the tests sets it; a build with
 * AKR_SHIP=1 in the environment leaves them out (tests/test_abi.py checks both symbol sets against the library it loads).
 */
#ifndef AKARI_HIP_TEST_H
#define AKARI_HIP_TEST_H

#include "akari_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define AKR_TEST_API AKR_API

/* Host-side pieces exposed for known-answer tests (no GPU needed):
 *   akr_host_stdrng_u64      rand 0.8 StdRng::seed_from_u64(seed) then n x gen::<u64>() (sampler/mod.rs:150-151)
 *   akr_host_chacha_block    one ChaCha block with `rounds` rounds (RFC 7539 / zero-key vectors pin the core)
 *   akr_host_pcg32_states    init_pcg32_buffer_with_seed on the host: 2 x u64 (state, inc) per entry
 *   akr_host_pcg_start       the closed form the kernels use for sampler.start() = advance(16384)
 *   akr_host_alias_table     AliasTable::new (util/distribution.rs:35-78) */
AKR_TEST_API int32_t akr_host_stdrng_u64(uint64_t seed, uint32_t n, uint64_t *out);

AKR_TEST_API int32_t akr_host_chacha_block(const uint32_t *key8, uint64_t counter, uint64_t stream, int32_t rounds, uint32_t *out16);

AKR_TEST_API int32_t akr_host_pcg32_states(uint64_t seed, uint64_t n, uint64_t *out2n);

AKR_TEST_API int32_t akr_host_pcg_start(uint64_t *state, uint64_t inc);

AKR_TEST_API int32_t akr_host_sobol_dim1(uint32_t n, const uint32_t *index, uint32_t *by_loop, uint32_t *by_butterfly);

/* The sobol sampler's second dimension, bit-reversed: by the defining loop and by the butterfly the kernels use (csrc/device/drng.h). */
/* a[k] % d[k] the way the index-based samplers compute it (csrc/device/drng.h fastmod_u32: precomputed constant, no division). */
AKR_TEST_API int32_t akr_host_fastmod(uint32_t n, const uint32_t *a, const uint32_t *d, uint32_t *out);

/* Scenes kept as meshes + instances: the conservative reject of a candidate triangle (csrc/device/dinst.h tri_may_hit) next to the exact
 * test it stands in front of, on the host. rays8 = o.xyz d.xyz tmin tlimit, tris9 = world-space A B C; may / exact = 0 or 1 per item. */
AKR_TEST_API int32_t akr_host_tri_pretest(uint32_t n, const float *rays8, const float *tris9, float plane_shift, uint32_t *may, uint32_t *exact,
                                     float *out_t /* or NULL */);

AKR_TEST_API int32_t akr_host_alias_table(const float *weights, uint32_t n, uint32_t *j, float *t, float *pdf);

/* ---------------------------------------------------------------------------------------------------
 * Device-function probes (used by the parity tests to compare single device functions with the oracle;
 * they launch tiny kernels on the context's stream and block).
 * ------------------------------------------------------------------------------------------------- */
/* sin/cos/log of the kernels' elementary functions for n inputs. */
AKR_TEST_API int32_t akr_probe_math(akr_context *ctx, uint32_t n, const float *x, float *sin_out, float *cos_out, float *log_out);

/* BSDF of material `m` on a flat surface (normal +z, world == local; cf. akari_test.rs:16-439):
 * mode 0: in = wi (3 floats / item)  -> out = f.rgb, pdf (4 floats / item)
 * mode 1: in = u  (3 floats / item)  -> out = wi.xyz, f.rgb, pdf, valid (8 floats / item) */
AKR_TEST_API int32_t akr_probe_bsdf(akr_context *ctx, const akr_material_desc *m, const float *ggx_table4096, int32_t mode,
                               const float *wo, uint32_t n, const float *in, float *out);

/* Closest hit of n rays (o.xyz, d.xyz, tmin, tmax = 8 floats / ray) -> hit(0/1), inst, prim as u32 and u, v. */
AKR_TEST_API int32_t akr_probe_intersect(akr_context *ctx, akr_scene *scene, uint32_t n, const float *rays, uint32_t *hit_inst_prim,
                                    float *bary);

AKR_TEST_API int32_t akr_probe_surface_interaction(akr_context *ctx, akr_scene *scene, uint32_t n, const uint32_t *inst_prim,
                                              const float *bary, float *out);

/* SurfaceInteraction of (inst, prim, u, v): out 19 floats / item = p, ng, n, t, s, uv, area, material. */
/* The tables of the pmj02bn sampler as the library uses them: sets = u32[5 * 65536 * 2], bluenoise = u16[48 * 128 * 128]. */
AKR_TEST_API int32_t akr_host_pmj02bn_tables(uint32_t *sets, uint16_t *bluenoise);

/* The PNG reader of akr_scene_load (8/16-bit, all colour types, tRNS, Adam7 interlacing), image crate `to_rgba8` rules
 * (load.rs:583-604). Rows in file order. rgba == NULL: only the size is returned. */
AKR_TEST_API int32_t akr_host_decode_png(const uint8_t *data, uint64_t len, uint32_t *width, uint32_t *height, uint8_t *rgba, uint64_t capacity);

/* The JPEG reader of akr_scene_load (baseline + progressive Huffman, 8 bit, grey / YCbCr / RGB, any integer sampling
 * ratios, restart intervals). Same calling convention as akr_host_decode_png. */
AKR_TEST_API int32_t akr_host_decode_jpeg(const uint8_t *data, uint64_t len, uint32_t *width, uint32_t *height, uint8_t *rgba, uint64_t capacity);

/* The TIFF reader of akr_scene_load (classic TIFF, first image; strips or tiles; chunky 8 / 16-bit or float samples; grey,
 * grey + alpha, RGB, RGBA; none / LZW / deflate / PackBits; horizontal predictor) and the DDS reader (DXT1 / DXT3 / DXT5, top
 * mip level) -- the remaining two encoded formats of load.rs:585-592. Same calling convention as akr_host_decode_png. */
AKR_TEST_API int32_t akr_host_decode_tiff(const uint8_t *data, uint64_t len, uint32_t *width, uint32_t *height, uint8_t *rgba, uint64_t capacity);

AKR_TEST_API int32_t akr_host_decode_dds(const uint8_t *data, uint64_t len, uint32_t *width, uint32_t *height, uint8_t *rgba, uint64_t capacity);

/* The OpenEXR reader of akr_scene_load (single-part scanline; none / RLE / ZIPS / ZIP; half / float / uint channels R G B A
 * or Y), RGBA f32 out, rows in file order. rgba == NULL: only the size is returned. */
AKR_TEST_API int32_t akr_host_decode_exr(const uint8_t *data, uint64_t len, uint32_t *width, uint32_t *height, float *rgba, uint64_t capacity_floats);

/* The same on the host under the colour pipeline `color` (akr_color_pipeline_bits): the tables a session with that
 * akr_pt_config.color uses, evaluated by the code the kernels run. */
AKR_TEST_API int32_t akr_probe_material_inputs_host(akr_scene *scene, uint32_t material, uint32_t color, uint32_t n, const float *uv, float *out26);

/* The interpreter's result for `material` at n uv points on the host (default colour pipeline): the folded record (64 words each),
 * optionally the alpha of the base-colour node and emission_color * emission_strength -- the values per-scene code must reproduce. */
AKR_TEST_API int32_t akr_probe_material_folded_host(akr_scene *scene, uint32_t material, uint32_t n, const float *uv, uint32_t *out64, float *alpha, float *emission3);

/* Evaluated inputs of `material` (26 words each = akr_material_desc) at n uv points: shader-graph evaluation + texture
 * sampling on the device, or -- ctx == NULL -- the same code on the host. */
AKR_TEST_API int32_t akr_probe_material_inputs(akr_context *ctx, akr_scene *scene, uint32_t material, uint32_t n, const float *uv, float *out26);

#ifdef __cplusplus
}
#endif

#endif /* AKARI_HIP_TEST_H */
