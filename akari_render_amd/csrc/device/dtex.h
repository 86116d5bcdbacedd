// dtex.h -- textures and non-constant shader-graph inputs.
//
// The reference evaluates, per shading point, every node of the material's compiled bytecode
// (crates/akari_render/src/svm/eval.rs:97-269,363-380) and then builds the closure from the evaluated inputs
// (svm/surface/principled.rs:13-131). Constant inputs are folded on the host (host/scene_build.cpp); a material
// with at least one texture-fed input keeps a pruned node list in HBM: this file evaluates it at the hit's uv,
// overrides the fed inputs of the material's raw input record and folds the record exactly as the host does
// (fold_inputs below is the one definition both sides use).
//
// Texture sampling is LuisaCompute's `tex2d.sample` in the reference (third party, source absent): the definition
// here -- texel centres at +0.5, bilinear weights from the fractional part, lerp as a + (b - a) t, unorm8 as
// byte / 255 -- is the AKR-F32 restatement the oracle follows (oracle/or_tex.h).
#pragma once
#include "dbsdf.h"

namespace akr {

#ifndef AKR_TEX_FAST_UNORM
#define AKR_TEX_FAST_UNORM 1
#endif

// ---- images ------------------------------------------------------------------------------------------------------
enum : uint32_t { IMG_RGBA8 = 0, IMG_RGBA32F = 1 };
enum : uint32_t { TEXF_NEAREST = 0, TEXF_LINEAR = 1 };
enum : uint32_t { TEXA_REPEAT = 0, TEXA_CLIP = 1, TEXA_MIRROR = 2, TEXA_EXTEND = 3 };

struct DImage {  // 32 B
    uint32_t offset_lo, offset_hi;  // first texel, in 4-byte words from the start of the texel buffer
    uint32_t width, height;
    uint32_t format, filter, address, _pad;
};

struct TexVal {
    float x, y, z, w;
};
AKR_HD TexVal tv(float x, float y, float z, float w) { return TexVal{x, y, z, w}; }

// SamplerAddress (load.rs:684-689). Repeat and mirror first reduce the COORDINATE to one period in floating point
// (u - floor(u); the triangle wave of period 2), so that the texel indices the filter then asks for lie in [-1, n] and
// wrap with two compares -- no integer division per tap. Edge clamps the index, Zero reports "outside".
AKR_HD float tex_wrap_coord(float u, uint32_t mode) {
    if (mode == TEXA_REPEAT) return u - __builtin_floorf(u);
    if (mode == TEXA_MIRROR) {
        float t = u - 2.0f * __builtin_floorf(u * 0.5f);  // [0, 2]
        return t > 1.0f ? 2.0f - t : t;
    }
    return u;
}
AKR_HD bool tex_wrap(int& i, int n, uint32_t mode) {
    if (mode == TEXA_REPEAT) {
        i = i < 0 ? i + n : (i >= n ? i - n : i);
        i = i < 0 ? 0 : (i >= n ? n - 1 : i);  // NaN / out-of-range coordinates (never after tex_wrap_coord of a finite u)
    } else if (mode == TEXA_MIRROR || mode == TEXA_EXTEND) {
        i = i < 0 ? 0 : (i >= n ? n - 1 : i);
    } else {
        if (i < 0 || i >= n) return false;
    }
    return true;
}
// byte / 255 correctly rounded. Device: q = b y, r = b - 255 q (exact in an fma), q + r y with y = RN(1 / 255) -- the
// division algorithm with a reciprocal known in advance, three operations instead of the eleven of an IEEE division and the
// same bits for all 256 bytes (tests/test_textures.py checks the identity exhaustively; four taps x four channels per lookup).
AKR_HD float unorm8(uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && AKR_TEX_FAST_UNORM
    const float y = 0.003921568859368563f, fb = (float)b;
    const float q = fb * y;
    return __builtin_fmaf(__builtin_fmaf(-255.0f, q, fb), y, q);
#else
    return (float)b / 255.0f;
#endif
}
// The taps of one lookup, together: the address modes are applied to the two columns and the two rows once, the texel format
// is decided once, and the (up to four) loads are issued back to back before anything is decoded -- one memory round trip per
// lookup. (Fetching tap by tap, each behind its own address-mode and format branches, made them four dependent round trips:
// the bilinear filter cost 15 % of the textured room's run time.) A tap outside a clip-addressed image reads texel (0, 0) and is
// zeroed afterwards.
struct TexTaps {
    TexVal a, b, c, d;  // (i, j), (i + 1, j), (i, j + 1), (i + 1, j + 1)
};
AKR_HD TexVal tex_decode8(uint32_t p) { return tv(unorm8(p & 0xffu), unorm8((p >> 8) & 0xffu), unorm8((p >> 16) & 0xffu), unorm8(p >> 24)); }
AKR_HD TexVal tex_decode32f(const uint32_t* __restrict__ q) { return tv(u2f(q[0]), u2f(q[1]), u2f(q[2]), u2f(q[3])); }
template <bool FOUR>
AKR_HD TexTaps tex_fetch_taps(const uint32_t* __restrict__ texels, const DImage& im, int i, int j) {
    int i0 = i, i1 = i + 1, j0 = j, j1 = j + 1;
    const int w = (int)im.width, h = (int)im.height;
    const bool vi0 = tex_wrap(i0, w, im.address), vj0 = tex_wrap(j0, h, im.address);
    const bool vi1 = FOUR ? tex_wrap(i1, w, im.address) : false, vj1 = FOUR ? tex_wrap(j1, h, im.address) : false;
    if (!vi0) i0 = 0;
    if (!vi1) i1 = 0;
    if (!vj0) j0 = 0;
    if (!vj1) j1 = 0;
    const uint64_t base = ((uint64_t)im.offset_hi << 32) | im.offset_lo;
    const uint64_t r0 = (uint64_t)j0 * im.width, r1 = (uint64_t)j1 * im.width;
    const TexVal zero = tv(0, 0, 0, 0);
    TexTaps t;
    t.b = t.c = t.d = zero;
    if (im.format == IMG_RGBA8) {
        const uint32_t* q = texels + base;
        const uint32_t pa = q[r0 + (uint64_t)i0];
        uint32_t pb = 0, pc = 0, pd = 0;
        if (FOUR) { pb = q[r0 + (uint64_t)i1]; pc = q[r1 + (uint64_t)i0]; pd = q[r1 + (uint64_t)i1]; }
        t.a = tex_decode8(pa);
        if (FOUR) { t.b = tex_decode8(pb); t.c = tex_decode8(pc); t.d = tex_decode8(pd); }
    } else {
        const uint32_t* q = texels + base;
        const uint32_t* qa = q + 4 * (r0 + (uint64_t)i0);
        t.a = tex_decode32f(qa);
        if (FOUR) {
            t.b = tex_decode32f(q + 4 * (r0 + (uint64_t)i1));
            t.c = tex_decode32f(q + 4 * (r1 + (uint64_t)i0));
            t.d = tex_decode32f(q + 4 * (r1 + (uint64_t)i1));
        }
    }
    if (!(vi0 && vj0)) t.a = zero;
    if (FOUR) {
        if (!(vi1 && vj0)) t.b = zero;
        if (!(vi0 && vj1)) t.c = zero;
        if (!(vi1 && vj1)) t.d = zero;
    }
    return t;
}
AKR_HD int tex_floor_to_int(float x, float& fl) {
    if (!(x > -1.0e9f)) x = -1.0e9f;  // also NaN
    if (x > 1.0e9f) x = 1.0e9f;
    fl = __builtin_floorf(x);
    return (int)fl;
}
AKR_HD float tex_lerp(float a, float b, float t) { return a + (b - a) * t; }
AKR_HD TexVal tex_sample(const uint32_t* __restrict__ texels, const DImage& im, vec2 uv) {
    float x = tex_wrap_coord(uv.x, im.address) * (float)im.width, y = tex_wrap_coord(uv.y, im.address) * (float)im.height;
    float fx, fy;
    if (im.filter == TEXF_NEAREST) {
        int i = tex_floor_to_int(x, fx), j = tex_floor_to_int(y, fy);
        return tex_fetch_taps<false>(texels, im, i, j).a;
    }
    x = x - 0.5f;
    y = y - 0.5f;
    int i = tex_floor_to_int(x, fx), j = tex_floor_to_int(y, fy);
    float tx = x - fx, ty = y - fy;
    if (!(tx >= 0.0f)) tx = 0.0f;  // x was clamped / NaN
    if (!(ty >= 0.0f)) ty = 0.0f;
    if (tx > 1.0f) tx = 1.0f;
    if (ty > 1.0f) ty = 1.0f;
    const TexTaps t = tex_fetch_taps<true>(texels, im, i, j);
    const TexVal &a = t.a, &b = t.b, &c = t.c, &d = t.d;
    TexVal r0 = tv(tex_lerp(a.x, b.x, tx), tex_lerp(a.y, b.y, tx), tex_lerp(a.z, b.z, tx), tex_lerp(a.w, b.w, tx));
    TexVal r1 = tv(tex_lerp(c.x, d.x, tx), tex_lerp(c.y, d.y, tx), tex_lerp(c.z, d.z, tx), tex_lerp(c.w, d.w, tx));
    return tv(tex_lerp(r0.x, r1.x, ty), tex_lerp(r0.y, r1.y, ty), tex_lerp(r0.z, r1.z, ty), tex_lerp(r0.w, r1.w, ty));
}
// srgb_to_linear (color.rs:555-558)
AKR_HD float srgb_to_linear1(float s) { return s <= 0.04045f ? s / 12.92f : pow_f((s + 0.055f) / 1.055f, 2.4f); }

// ---- node list ---------------------------------------------------------------------------------------------------
enum : uint32_t {
    NODE_CONST = 0, NODE_RGB = 1, NODE_TEXCOORDS = 2, NODE_IMAGE = 3, NODE_MAPPING = 4, NODE_CHECKERBOARD = 5,
    NODE_SPECTRAL_UPLIFT = 6, NODE_SEPARATE_COLOR = 7, NODE_EXTRACT = 8, NODE_NORMAL_MAP = 9
};
constexpr uint32_t kNodeNone = 0xffffffffu;
constexpr uint32_t kMaxGraphNodes = 256;  // after pruning to the texture-fed inputs (host/scene_build.cpp); the values live in at most kTexMaxSlots slots, so the length of a list costs LDS staging space only
// A pruned node list is also register-allocated on the host: every node's value gets one of at most kTexMaxSlots value slots
// (its slot is free again after its last consumer), the arguments of its consumers name slots, and the inputs of the surface
// node it feeds are a bit mask. `op` of such a node: bits 0-7 the operation, bits 8-15 its slot (0xff: nobody reads the value
// back), bits 16-29 the mask of akr_material_input it feeds. On the device the slots live in LDS, strided by the workgroup
// size (kTexValStride lanes): the evaluation of a textured hit touches no scratch memory.
constexpr uint32_t kTexMaxSlots = 8;
constexpr uint32_t kTexValStride = 256;  // every kernel that evaluates graphs runs workgroups of (at most) 256 threads
constexpr uint32_t kTexNoSlot = 0xffu;
// DMaterial.tex_n_nodes: bits 0-15 the length of the pruned list, bits 16-31 the material's shader kind -- materials whose lists
// have the same shape (operations, argument topology, fed inputs, image formats) share one, the reference's `shader_kind`
// (svm/compiler.rs:16-76); a per-scene kernel switches on it (host/specialise.cpp), the interpreter ignores it.
constexpr uint32_t kTexCountMask = 0xffffu, kTexKindShift = 16;
struct DNode {                           // = akr_shader_node, 32 B
    uint32_t op;
    uint32_t arg[4];
    float k[3];
};
enum : uint32_t {
    IN_BASE_COLOR = 0, IN_METALLIC, IN_ROUGHNESS, IN_IOR, IN_SPECULAR_IOR_LEVEL, IN_SPECULAR_TINT, IN_TRANSMISSION_WEIGHT,
    IN_COAT_WEIGHT, IN_COAT_ROUGHNESS, IN_COAT_IOR, IN_COAT_TINT, IN_EMISSION_COLOR, IN_EMISSION_STRENGTH, IN_NORMAL, IN_COUNT
};
static_assert(IN_COUNT == 14, "DMaterial keeps the input map in 14 of its 16 spare words");

struct TexScene {  // the texture part of DScene
    const DNode* __restrict__ nodes;
    const DImage* __restrict__ images;
    const uint32_t* __restrict__ texels;
    const MatInputs* __restrict__ mat_inputs;  // raw (unfolded) inputs, one per material; constants already in the pipeline's space
    uint32_t color;                            // ColorPipeline bits (dbsdf.h COLOR_*)
    uint32_t val_offset_words;                 // where this launch's value slots start in the workgroup's dynamic LDS (set by the launcher)
};

// The nodes (svm/eval.rs:97-269), one function each: the interpreter below (eval_node) and the straight-line code generated per
// scene (host/specialise.cpp) call the same definitions. Values are float4, narrower types zero-extended, so the auto-convert
// rules of eval.rs:301-349 are component reads.
AKR_HD TexVal node_const(float k0, float k1, float k2) { return tv(k0, k1, k2, 0.0f); }
AKR_HD TexVal node_rgb(uint32_t color, float k0, float k1, float k2, bool tag_aces) {  // rgb_to_target_colorspace(rgb, node space, pipeline.rgb_colorspace), texture/mod.rs:9-30
    vec3 c = cs_convert(mk3(k0, k1, k2), tag_aces, (color & COLOR_RGB_ACES) != 0);
    return tv(c.x, c.y, c.z, 1.0f);
}
AKR_HD TexVal node_texcoords(vec2 uv) { return tv(uv.x, uv.y, 0.0f, 0.0f); }
AKR_HD TexVal node_image(const uint32_t* __restrict__ texels, const DImage& im, vec2 st, bool srgb) {
    TexVal v = tex_sample(texels, im, st);
    if (srgb) v = tv(srgb_to_linear1(v.x), srgb_to_linear1(v.y), srgb_to_linear1(v.z), v.w);
    return v;
}
AKR_HD TexVal node_mapping(TexVal a, TexVal loc, TexVal sc, uint32_t type) {
    if (type == 0) return tv(a.x * sc.x + loc.x, a.y * sc.y + loc.y, a.z * sc.z + loc.z, 0.0f);
    return tv((a.x - loc.x) / sc.x, (a.y - loc.y) / sc.y, (a.z - loc.z) / sc.z, 0.0f);
}
AKR_HD bool node_checker_first(vec2 st, float scale) {  // which of the two colours
    float fx, fy;
    int px = tex_floor_to_int((st.x * scale) * 2.0f, fx), py = tex_floor_to_int((st.y * scale) * 2.0f, fy);
    return ((px + py) & 1) == 0;
}
AKR_HD TexVal node_uplift(uint32_t color, TexVal a) {  // spectral_uplift: rgb_colorspace -> the space of color_repr, texture/mod.rs:31-43
    vec3 c = cs_convert(mk3(a.x, a.y, a.z), (color & COLOR_RGB_ACES) != 0, (color & COLOR_REPR_ACES) != 0);
    return tv(c.x, c.y, c.z, a.w);
}
AKR_HD TexVal node_extract(TexVal a, uint32_t f) { return f == 0 ? tv(a.x, 0, 0, 0) : f == 1 ? tv(a.y, 0, 0, 0) : f == 2 ? tv(a.z, 0, 0, 0) : tv(a.x, a.y, 0, 0); }
AKR_HD TexVal node_normal_map(TexVal a, float s) {
    float nx = 2.0f * a.x - 1.0f, ny = 2.0f * a.y - 1.0f, nz = 2.0f * a.z - 1.0f;
    if (s != 1.0f) { nx = nx * s; ny = ny * s; nz = nz * 1.0f; }
    return tv(nx, ny, nz, 0.0f);
}
// One node of a list. `get(a)` returns the value of argument node / slot `a`.
template <typename Get>
AKR_HD TexVal eval_node(const TexScene& ts, const DNode& nd, uint32_t op, vec2 uv, Get get) {
    TexVal v = tv(0, 0, 0, 0);
    switch (op) {
        case NODE_CONST: v = node_const(nd.k[0], nd.k[1], nd.k[2]); break;
        case NODE_RGB: v = node_rgb(ts.color, nd.k[0], nd.k[1], nd.k[2], nd.arg[0] == 1u); break;
        case NODE_TEXCOORDS: v = node_texcoords(uv); break;
        case NODE_IMAGE: {
            vec2 st = uv;
            if (nd.arg[1] != kNodeNone) { TexVal a = get(nd.arg[1]); st = mk2(a.x, a.y); }
            v = node_image(ts.texels, ts.images[nd.arg[0]], st, nd.arg[2] != 0);
            break;
        }
        case NODE_MAPPING: {
            TexVal a = get(nd.arg[0]), loc = get(nd.arg[1]), sc = get(nd.arg[2]);
            v = node_mapping(a, loc, sc, nd.arg[3]);
            break;
        }
        case NODE_CHECKERBOARD: {
            vec2 st = uv;
            if (nd.arg[0] != kNodeNone) { TexVal a = get(nd.arg[0]); st = mk2(a.x, a.y); }
            v = node_checker_first(st, get(nd.arg[1]).x) ? get(nd.arg[2]) : get(nd.arg[3]);
            break;
        }
        case NODE_SPECTRAL_UPLIFT: v = node_uplift(ts.color, get(nd.arg[0])); break;
        case NODE_SEPARATE_COLOR: v = get(nd.arg[0]); break;
        case NODE_EXTRACT: v = node_extract(get(nd.arg[0]), nd.arg[1]); break;
        case NODE_NORMAL_MAP: {
            TexVal a = get(nd.arg[0]);
            v = node_normal_map(a, get(nd.arg[1]).x);
            break;
        }
        default: break;
    }
    return v;
}
// eval_shader (eval.rs:363-380) over a node list whose arguments are NODE INDICES and whose values go to val[node]: the form
// the scene description arrives in (host/scene_build.cpp folds the constant inputs with it).
AKR_HD void eval_graph(const TexScene& ts, uint32_t first, uint32_t count, vec2 uv, TexVal* val) {
    for (uint32_t i = 0; i < count; i++) {
        const DNode nd = ts.nodes[first + i];
        val[i] = eval_node(ts, nd, nd.op & 0xffu, uv, [&](uint32_t a) { return val[a]; });
    }
}

// Overrides the texture-fed inputs of `in` with the node values (principled.rs:13-131 read rules: colours through
// eval_color_alpha -> xyz + alpha, scalars through eval_float_auto_convert -> x, normal through float3 -> xyz).
AKR_HD void apply_inputs(const uint32_t* __restrict__ map, const TexVal* val, MatInputs& in) {
    uint32_t n;
    if ((n = map[IN_BASE_COLOR]) != kNodeNone) { in.base_color[0] = val[n].x; in.base_color[1] = val[n].y; in.base_color[2] = val[n].z; in.base_alpha = val[n].w; }
    if ((n = map[IN_METALLIC]) != kNodeNone) in.metallic = val[n].x;
    if ((n = map[IN_ROUGHNESS]) != kNodeNone) in.roughness = val[n].x;
    if ((n = map[IN_IOR]) != kNodeNone) in.ior = val[n].x;
    if ((n = map[IN_SPECULAR_IOR_LEVEL]) != kNodeNone) in.specular_ior_level = val[n].x;
    if ((n = map[IN_SPECULAR_TINT]) != kNodeNone) { in.specular_tint[0] = val[n].x; in.specular_tint[1] = val[n].y; in.specular_tint[2] = val[n].z; }
    if ((n = map[IN_TRANSMISSION_WEIGHT]) != kNodeNone) in.transmission_weight = val[n].x;
    if ((n = map[IN_COAT_WEIGHT]) != kNodeNone) in.coat_weight = val[n].x;
    if ((n = map[IN_COAT_ROUGHNESS]) != kNodeNone) in.coat_roughness = val[n].x;
    if ((n = map[IN_COAT_IOR]) != kNodeNone) in.coat_ior = val[n].x;
    if ((n = map[IN_COAT_TINT]) != kNodeNone) { in.coat_tint[0] = val[n].x; in.coat_tint[1] = val[n].y; in.coat_tint[2] = val[n].z; }
    if ((n = map[IN_EMISSION_COLOR]) != kNodeNone) { in.emission_color[0] = val[n].x; in.emission_color[1] = val[n].y; in.emission_color[2] = val[n].z; }
    if ((n = map[IN_EMISSION_STRENGTH]) != kNodeNone) in.emission_strength = val[n].x;
    if ((n = map[IN_NORMAL]) != kNodeNone) { in.normal[0] = val[n].x; in.normal[1] = val[n].y; in.normal[2] = val[n].z; }
}

// One input of the surface node from a node value (principled.rs:13-131 read rules, as apply_inputs above).
AKR_HD void apply_fed(uint32_t feeds, TexVal v, MatInputs& in) {
    if (feeds & (1u << IN_BASE_COLOR)) { in.base_color[0] = v.x; in.base_color[1] = v.y; in.base_color[2] = v.z; in.base_alpha = v.w; }
    if (feeds & (1u << IN_METALLIC)) in.metallic = v.x;
    if (feeds & (1u << IN_ROUGHNESS)) in.roughness = v.x;
    if (feeds & (1u << IN_IOR)) in.ior = v.x;
    if (feeds & (1u << IN_SPECULAR_IOR_LEVEL)) in.specular_ior_level = v.x;
    if (feeds & (1u << IN_SPECULAR_TINT)) { in.specular_tint[0] = v.x; in.specular_tint[1] = v.y; in.specular_tint[2] = v.z; }
    if (feeds & (1u << IN_TRANSMISSION_WEIGHT)) in.transmission_weight = v.x;
    if (feeds & (1u << IN_COAT_WEIGHT)) in.coat_weight = v.x;
    if (feeds & (1u << IN_COAT_ROUGHNESS)) in.coat_roughness = v.x;
    if (feeds & (1u << IN_COAT_IOR)) in.coat_ior = v.x;
    if (feeds & (1u << IN_COAT_TINT)) { in.coat_tint[0] = v.x; in.coat_tint[1] = v.y; in.coat_tint[2] = v.z; }
    if (feeds & (1u << IN_EMISSION_COLOR)) { in.emission_color[0] = v.x; in.emission_color[1] = v.y; in.emission_color[2] = v.z; }
    if (feeds & (1u << IN_EMISSION_STRENGTH)) in.emission_strength = v.x;
    if (feeds & (1u << IN_NORMAL)) { in.normal[0] = v.x; in.normal[1] = v.y; in.normal[2] = v.z; }
}

// The value slots of one graph evaluation. Device: this lane's column of the workgroup's LDS block (TexScene.val_offset_words,
// set by the launcher; ds_read / ds_write_b128, no bank conflicts: consecutive lanes hold consecutive 16-byte values).
// Host (probes, tests): a local array.
struct TexSlots {
#if defined(__HIP_DEVICE_COMPILE__)
    TexVal* base;
    AKR_HD TexVal get(uint32_t s) const { return base[s * kTexValStride]; }
    AKR_HD void set(uint32_t s, TexVal v) { base[s * kTexValStride] = v; }
#else
    TexVal v_[kTexMaxSlots];
    TexVal get(uint32_t s) const { return v_[s]; }
    void set(uint32_t s, TexVal v) { v_[s] = v; }
#endif
};
AKR_HD TexSlots tex_slots(const TexScene& ts) {
    TexSlots st;
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint32_t akr_dynamic_lds[];
    st.base = reinterpret_cast<TexVal*>(akr_dynamic_lds + ts.val_offset_words) + threadIdx.x;
#else
    (void)ts;
#endif
    return st;
}
// The pruned, slot-allocated node list of a material at `uv`: every node's value into its slot, the inputs it feeds into `in`.
// (Tried and dropped: letting the lanes of a wave vote so that those whose next node is an image lookup wait for each other
// and sample together -- the lookup code then runs once per evaluation instead of once per loop position that holds one, but
// the extra trips cost more: 698 against 750 Msamples/s on the textured room, +1 % on its BVH variant.)
AKR_HD void eval_material_graph(const TexScene& ts, uint32_t first, uint32_t count, vec2 uv, MatInputs& in) {
    TexSlots st = tex_slots(ts);
    for (uint32_t i = 0; i < count; i++) {
        const DNode nd = ts.nodes[first + i];
        const uint32_t slot = (nd.op >> 8) & 0xffu, feeds = nd.op >> 16;
        const TexVal v = eval_node(ts, nd, nd.op & 0xffu, uv, [&](uint32_t a) { return st.get(a); });
        if (slot != kTexNoSlot) st.set(slot, v);
        if (feeds) apply_fed(feeds, v, in);
    }
}

#if defined(AKR_SPEC_GRAPHS)
// Per-scene kernels (hiprtc, host/specialise.cpp): the scene's node lists as straight-line code, one case per shader kind
// (svm/eval.rs:428-467 emits the same switch). The generated text defines
//   spec_material_at(ts, material, uv, m)   the whole of material_at below for an MF_TEXTURED material
//   spec_alpha(ts, m, material, uv)         w of the node feeding base_color (the alpha test, disect.h)
//   spec_emission(ts, m, material, uv)      emission_color * emission_strength with the fed ones evaluated (light samples, dpath.h)
#include "akr_scene_spec.h"
#endif

// The material at a shading point: the folded record as is, or -- for MF_TEXTURED materials -- its graph evaluated at
// `uv` and folded. `m` must hold the material's folded record on entry.
AKR_HD void material_at(const TexScene& ts, uint32_t material, vec2 uv, DMaterial& m) {
    if (!(m.flags & MF_TEXTURED)) return;
#if defined(AKR_DIAG_NO_GRAPH) && defined(__HIP_DEVICE_COMPILE__)  // diagnostic builds only (wrong images): what graph evaluation + re-folding cost
    return;
#endif
#if defined(AKR_SPEC_GRAPHS)
    spec_material_at(ts, material, uv, m);
#else
    const uint32_t first = m.tex_first_node, count = m.tex_n_nodes;
    uint32_t map[IN_COUNT];
    for (uint32_t i = 0; i < IN_COUNT; i++) map[i] = m.tex_input[i];
    MatInputs in = ts.mat_inputs[material];
#if !(defined(AKR_DIAG_NO_EVAL) && defined(__HIP_DEVICE_COMPILE__))  // diagnostic: re-fold the raw inputs without evaluating the graph
    eval_material_graph(ts, first, count & kTexCountMask, uv, in);
#endif
    const uint32_t keep = m.flags & (MF_TEXTURED | MF_ALPHA_TEXTURED);
    fold_inputs(in, m);
    m.flags |= keep;
    m.tex_first_node = first;
    m.tex_n_nodes = count;
    for (uint32_t i = 0; i < IN_COUNT; i++) m.tex_input[i] = map[i];
#endif
}
// w of the node feeding a texture-fed base colour at `uv` (the stochastic alpha test, scene.rs:49-86 / principled.rs:15-21)
AKR_HD float material_alpha_at(const TexScene& ts, const DMaterial& m, uint32_t material, vec2 uv) {
#if defined(AKR_SPEC_GRAPHS)
    return spec_alpha(ts, m, material, uv);
#else
    MatInputs in = ts.mat_inputs[material];
    eval_material_graph(ts, m.tex_first_node, m.tex_n_nodes & kTexCountMask, uv, in);
    return in.base_alpha;
#endif
}
// emission_color * emission_strength of a material whose emission inputs are texture-fed, at `uv` (AreaLight::sample_direct)
AKR_HD vec3 material_emission_inputs_at(const TexScene& ts, const DMaterial& m, uint32_t material, vec2 uv) {
#if defined(AKR_SPEC_GRAPHS)
    return spec_emission(ts, m, material, uv);
#else
    MatInputs in = ts.mat_inputs[material];
    eval_material_graph(ts, m.tex_first_node, m.tex_n_nodes & kTexCountMask, uv, in);
    return mk3(in.emission_color[0], in.emission_color[1], in.emission_color[2]) * in.emission_strength;
#endif
}

}  // namespace akr
