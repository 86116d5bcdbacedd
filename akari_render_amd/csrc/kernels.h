// kernels.h -- kernel parameter block and host-callable launchers (implemented in pt_kernels.hip).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif

#include "device/drng.h"
#include "device/dscene.h"

namespace akr {

// ---- build switches of k_pt_pass (pt_kernels.hip); each default is the winner of a same-box A/B run ----
#ifndef AKR_WALK_FD
#define AKR_WALK_FD 4    // exhaustive pair walk of the force_diffuse kernel: which form of the walk (disect.h: WALK; 4 measured best in round 4)
#endif
#ifndef AKR_WALK_FULL
#define AKR_WALK_FULL 4  // the same for the full-graph exhaustive kernels of scenes without textures
#endif
#ifndef AKR_WALK_TEX
#define AKR_WALK_TEX 4   // the same for the exhaustive kernels of scenes with textures (round 3: 0, the scalar-cache walk)
#endif
#ifndef AKR_WALK_FULL_UNROLL
#define AKR_WALK_FULL_UNROLL 1  // full-graph exhaustive kernels: two records per trip of the pair walk, as the force_diffuse kernel does
#endif
#ifndef AKR_PT_PARK_FULL
#define AKR_PT_PARK_FULL 0  // exhaustive full-graph kernels without textures: cold path state in LDS while a vertex is shaded (dpath.h: PARK)
#endif
#ifndef AKR_PT_PARK_BVH
#define AKR_PT_PARK_BVH 0   // the same for the BVH full-graph kernels without textures
#endif
#ifndef AKR_BVH_TILE
#define AKR_BVH_TILE 1  // BVH kernels: 1 = the top of the tree in LDS (disect.h: TILE), as many nodes as launch_pt_pass finds room for
#endif
#ifndef AKR_PT_XCD_BANDS
#define AKR_PT_XCD_BANDS 0  // BVH kernels: 1 = XCD x renders the x-th contiguous eighth of the launch's work items (pt_kernels.hip)
#endif
#ifndef AKR_PT_STRAGGLERS
#define AKR_PT_STRAGGLERS 8  // BVH kernels: n > 0 = an intersection phase ends when at most 1/n of the lanes that entered it are still
                             // tracing; those lanes keep their traversal and go on in the next phase (see k_pt_pass)
#endif
#ifndef AKR_PT_STRAGGLERS_TEX
#define AKR_PT_STRAGGLERS_TEX 0  // the same for the BVH kernels of scenes with textures (measured separately)
#endif
#ifndef AKR_PT_STRAGGLERS_INST
#define AKR_PT_STRAGGLERS_INST 8  // the same for the kernels of scenes kept as meshes + instances (dinst_trav.h trace_pair_inst)
#endif
#ifndef AKR_PT_PARK_TEX
#define AKR_PT_PARK_TEX 1   // the same for the full-graph kernels of scenes with textures (exhaustive and BVH)
#endif
// LDS columns per lane (one word per slot, slot s of lane i at word s * 256 + i): cold path state parked while a vertex is shaded
// (dpath.h: PARK), and a traversal carried over to the next intersection phase (pt_kernels.hip).
constexpr uint32_t kParkSlots = 16, kParkSlotsNoDefer = 13;  // dpath.h: PK_*
constexpr uint32_t kCarrySlots = 13;                         // pt_kernels.hip: a carried traversal
constexpr uint32_t kCarrySlotsInstanced = 16;                // ... of a scene kept as meshes + instances (dinst_trav.h)
constexpr size_t kBlueNoiseColumnBytes = 48 * 256 * 2;          // dpath.h pmj_bluenoise_stage: one u16 per array and lane
// What a k_pt_pass launch keeps in LDS beyond traversal stacks, staged tables and graph values, by kernel instantiation
// (BVH / force_diffuse / textured scene / conductor deferral): used by the launcher and by the host's staging decision.
struct PtLdsPlan {
    size_t recs_bytes, park_bytes, carry_bytes;
    bool tile;
};
inline PtLdsPlan pt_lds_plan(bool bvh, bool fd, bool tex, bool defer, uint32_t n_tris) {
    PtLdsPlan pl;
    const int walk = fd ? AKR_WALK_FD : (tex ? AKR_WALK_TEX : AKR_WALK_FULL);
    const bool recs_in_lds = !bvh && (walk == 1 || walk >= 3);
    const bool park = !fd && (tex ? AKR_PT_PARK_TEX != 0 : (bvh ? AKR_PT_PARK_BVH != 0 : AKR_PT_PARK_FULL != 0));
    const bool strag = bvh && (tex ? AKR_PT_STRAGGLERS_TEX : AKR_PT_STRAGGLERS) > 0;
    pl.recs_bytes = recs_in_lds ? (size_t)(n_tris + 2) * 48 : 0;
    pl.park_bytes = park ? (size_t)(defer ? kParkSlots : kParkSlotsNoDefer) * 256 * 4 : 0;
    pl.carry_bytes = strag ? (size_t)kCarrySlots * 256 * 4 : 0;
    pl.tile = bvh && !tex && AKR_BVH_TILE != 0;
    return pl;
}
// a workgroup's share of the CU's 160 KB: a quarter (four waves per SIMD), a third for the kernels of textured scenes (three)
inline size_t pt_lds_budget(bool tex) { return (tex ? 53 : 40) * 1024; }

// Launch counters (akr_pt_stats) are kStatStripes copies of 8 u64, a workgroup adding to copy blockIdx % kStatStripes: the wavefront
// schedule flushes them once per wave and ITERATION (32 k waves x 7 atomics on seven addresses per k_wf_shade launch serialise at the
// L2 -- measured: a large part of that kernel's time); the host sums the copies when it reads them.
constexpr uint32_t kStatStripes = 256;
// Everything one pass of the path tracer needs; passed by value as the kernel argument (lands in SGPRs).
struct PtParams {
    DScene sc;
    // PerspectiveCameraData (camera/mod.rs:105-118): raster->camera and camera->world, column-major
    float r2c[16];
    float c2w[16];
    uint32_t c2w_identity;
    uint32_t width, height;
    // pt::Config (pt.rs:916-929)
    uint32_t max_depth, rr_depth, use_nee, indirect_only, force_diffuse;
    int32_t debug_depth;
    int32_t pixel_offset[2];
    uint32_t filter_type;
    float filter_radius;
    uint32_t color;          // ColorPipeline bits (device/dbsdf.h COLOR_*): the space the path shades in
    uint32_t pass_spp;       // samples per pixel per pass (spp_per_pass)
    uint32_t n_passes;       // passes fused into this launch (>= 1)
    uint32_t last_pass_spp;  // samples of the launch's last pass (<= pass_spp)
    PcgStartConsts start;
    // per-pixel sampler states (Pcg32[N]), film accumulator (f32[7N], reference layout), counters (u64[8])
    Pcg32* states;
    float* film;
    uint64_t* counters;
    // sampler (sampler/mod.rs:282-295): 0 = independent (PCG32 state per pixel), 1 = pmj02bn (index-based: point sets +
    // blue-noise offsets; states[pix].state holds the pixel's sample index, .inc its coordinates)
    uint32_t sampler;
    uint32_t smp_seed, smp_spp, smp_w;      // Pmj02BnState.{seed, spp, w}
    uint64_t smp_mod_magic;                 // fastmod_magic(smp_spp): x % spp without a division (drng.h)
    const uint32_t* pmj_sets;               // [5][65536][2] u32 fixed point
    const uint16_t* bluenoise;              // [48][128][128] unorm16
    // Small scenes (exhaustive path): the tables the shading phase gathers from, staged in LDS by k_pt_pass. Bytes per table
    // in the order shade, normals, inst, materials, light_alias, area_alias, lights, light_pdf, area_pdf; 0 total = not staged.
    // ... then the texture tables of a TEX scene: pruned node lists, image headers, raw material inputs (12 entries in all).
    uint32_t stage_bytes[13];  // [12]: the GGX albedo table (full-graph exhaustive kernels)
    uint32_t stage_total;
    uint32_t simple_scene;   // no coat, no transmission, no normal map, no glass material, no textures: the full-graph kernels without that code
    uint32_t defer_metal;    // iterations with (iteration & defer_metal) != 0 put hits on "expensive" materials off by one iteration (pt_kernels.hip: DEFER)
    uint32_t defer_flags;    // ... expensive = (DMaterial.flags & defer_flags) != 0: MF_EVAL_METAL (the conductor lobe), MF_TEXTURED (a graph to evaluate)
    uint32_t tex_slots;      // TEX scenes: value slots per lane of the graph evaluation (LDS, after the launch's other blocks)
    uint32_t tile_offset;    // BVH kernels with a node tile (disect.h: TILE): word offset of the tile; its size is sc.bvh_tile_nodes
    uint32_t park_offset;    // kernels that park cold path state in LDS while shading (dpath.h: PARK): word offset of the columns
    uint32_t carry_offset;   // BVH kernels that let a wave's longest rays run on into the next iteration (pt_kernels.hip): their columns
    uint32_t bn_offset;      // pmj02bn sampler, k_pt_pass: word offset of the lanes' blue-noise columns (48 x 256 x 2 B, dpath.h), 0 = the table in HBM
    // wavefront schedule, option wf_sort: the ray queues are sorted by (Morton code of the origin in the scene's box, octant of the direction)
    uint32_t wf_sort;
    float sort_lo[3], sort_scale[3];  // cell = (o - lo) * scale, 128 cells per axis
    // work distribution
    uint32_t n_items;
    uint32_t shard_rank, shard_count;
    uint32_t tile_w, tile_h, tiles_x, tiles_y;
    const uint32_t* owned_tiles;  // shard_count > 1: the tiles (row-major ids) this rank owns, in Morton order (tile_owner below); else null
};

// Which rank owns tile (tx, ty) of a frame shared by `count` ranks: its position on the Z-order (Morton) curve, modulo the ranks
// ("tile t owned by GPU t mod G in Morton order"): 8 ranks each get one tile of every aligned 4 x 2 block of tiles, so every
// rank sees the same mix of cheap and expensive regions whatever the row length is. One definition for the kernels (dpath.h
// item_to_pixel through the session's owned_tiles list, gpt_kernels.hip), the host (api_pt.cpp, api_aux.cpp), and -- restated -- the
// oracle (akr_oracle.c or_pixel_owned) and the Python mirror (distributed.owned_pixel_mask).
AKR_HD uint32_t morton_spread16(uint32_t x) {
    x &= 0xffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
AKR_HD uint32_t tile_morton(uint32_t tx, uint32_t ty) { return morton_spread16(tx) | (morton_spread16(ty) << 1); }
AKR_HD uint32_t tile_owner(uint32_t tx, uint32_t ty, uint32_t count) { return tile_morton(tx, ty) % count; }

// Path state of the wavefront schedule (wf_kernels.hip): structure-of-arrays, one slot per pixel of the launch.
struct WfBuffers {
    float4 *ray_o, *ray_d;          // next closest-hit ray: o | exclude id ; d
    float4 *sh_o, *sh_d, *sh_c;     // pending shadow ray: o | exclude0 ; d | tmax ; contribution | exclude1
    float4* hit;                    // written by k_wf_trace: gid, u, v | occluded
    float4 *beta, *rad, *base;      // beta | prev_bsdf_pdf ; radiance | depth ; base | flags
    float4* film;                   // film accumulator of the slot's pixel: rgb | weight
    uint4 *rng, *misc;              // pcg state, dim, samples_done ; pass_idx, cur_spp, pcg inc
    uint32_t* queue_closest[2];     // ray queues (slot ids), double-buffered
    uint32_t* queue_shadow[2];
    uint32_t* key_closest[2];       // option wf_sort: sort key of every queue entry (wf_kernels.hip wf_ray_key), same indexing as the queues
    uint32_t* key_shadow[2];
    uint32_t* qcount;               // [4]: closest/shadow counts of queue 0, of queue 1
    uint32_t* qhead;                // next unclaimed ray id of the queue being traced
    uint32_t* n_active;             // slots still active after the last shade
    // Rays a trace launch did not finish (wf_kernels.hip: "carried rays"): a wave that finds the queue empty and has few lanes left saves those
    // lanes' traversals and ends; the ray goes on in the next trace launch, its slot is not shaded meanwhile. nullptr = every launch traces to the end.
    uint32_t* pend;                 // per slot: bit 0 = its closest-hit ray is carried, bit 1 = its shadow ray
    uint32_t* carry;                // per (kind, slot) carry_words words: best t, u, v, id | G, T, tbase, sp | leaf, pend_rec, pend_inst, - | the stack
    uint32_t carry_words, n_slots;  // (record of kind k, slot s at carry + (k * n_slots + s) * carry_words)
    uint32_t carry_queue;           // launches of fewer rays than this trace to the end
    uint32_t carry_lanes, carry_steps;  // a wave hands over when at most carry_lanes lanes are left and each has had carry_steps steps (16, 48; tests: 56, 4)
    // The slots [slot_base, slot_end) this set of queues and counters serves: a session's slots are divided into GROUPS, each with its own
    // queues, counters and stream, so that one group's kernels fill the chip while another's trace launch waits for its last rays
    // (host/api_pt.cpp wf_run). The state arrays above are the session's, indexed by slot.
    uint32_t slot_base, slot_end;
};

// Scratch and accumulators of the gpt integrator (gpt_kernels.hip), all f32 RGB: per-pixel splat slots of one sample (own,
// shifted[i]); with a reconstruction, the sums / sums of squares of the primal image and the (W+1) x (H+1) gradient images.
struct GptParams {
    float* own;
    float* shifted[4];
    float *acc_p, *acc_gx, *acc_gy, *sqr_p, *sqr_gx, *sqr_gy;
    uint32_t reconnect, stride, separate_weights, reconstruction;
    // Sharded render (akr_gpt_begin with an akr_shard): k_gpt_sample runs the pixels of `item_pixels` -- the rank's own tiles plus
    // the halo whose offset paths land in them -- and k_gpt_update folds only the pixels of the rank's own tiles.
    const uint32_t* item_pixels;  // pixel index per work item, or nullptr = PtParams' tile enumeration
    uint32_t shard_rank, shard_count, tile_w, tile_h, tiles_x;
};

// mcmc_opt integrator (mcmc_kernels.hip)
struct PssSample {  // mcmc_opt.rs:21-26
    float cur, backup;
    uint32_t last_modified, modified_backup;
};
struct MarkovState {  // mcmc_opt.rs:41-51
    uint32_t cur_pixel[2], chain_id;
    float cur_f, b;
    uint32_t b_cnt, n_accepted, n_mutations, cur_iter, last_large_iter;
};
struct McmcParams {
    PssSample* pss;          // [dim][n_chains]
    MarkovState* states;     // [n_chains]
    float4* cur_colors;      // [n_chains]
    Pcg32* rngs;             // [n_chains] the chains' independent samplers
    const Pcg32* seeds;      // [max(n_bootstrap, n_chains)] init_pcg32_buffer_with_seed(seed)
    float* fs;               // [n_bootstrap] bootstrap contributions
    const uint32_t* resampled;  // [n_chains] bootstrap path each chain starts from
    float* film;             // the film (7 N floats); the chains splat into its splat channels
    uint32_t n_chains, n_bootstrap, dim, width, height;
    uint32_t chain_begin, chain_count;  // the chains this launch runs: [chain_begin, chain_begin + chain_count) of the n_chains (a rank's share, akr_mcmc_render_shard)
    uint32_t exponential_mutation;
    float small_sigma, large_step_prob, image_mutation_prob, image_mutation_size;
};
#if !defined(__HIPCC_RTC__)  // host-callable launchers: not part of a per-scene kernel module
hipError_t launch_mcmc_bootstrap(const PtParams& p, const McmcParams& m, hipStream_t stream);
hipError_t launch_mcmc_init(const PtParams& p, const McmcParams& m, hipStream_t stream);
hipError_t launch_mcmc_advance(const PtParams& p, const McmcParams& m, uint32_t mutations_per_chain, float contribution, hipStream_t stream);
#endif

// Dynamic LDS of a launch that evaluates shader graphs = its own blocks (`base_bytes`: traversal stacks, staged tables), then
// tex_slots x 256 lanes x 16 B of value slots. Returns the parameter block with the slots' offset filled in and the total size.
inline PtParams with_tex_slots(const PtParams& p, size_t base_bytes, size_t& lds_bytes) {
    PtParams q = p;
    base_bytes = (base_bytes + 15) & ~(size_t)15;
    q.sc.tex.val_offset_words = (uint32_t)(base_bytes / 4);
    lds_bytes = base_bytes + (p.sc.tex.nodes != nullptr ? (size_t)p.tex_slots * kTexValStride * sizeof(TexVal) : 0);
    return q;
}
#if !defined(__HIPCC_RTC__)
// LDS layout of a k_pt_pass launch (pt_kernels.hip): the parameter block with the offsets filled in, the dynamic LDS size, the grid
PtParams pt_pass_layout(const PtParams& p, size_t& lds_bytes, uint32_t& blocks);
hipError_t launch_inst_share_bits(const DScene& sc, uint32_t* bits, uint32_t* mesh_tri_words, hipStream_t stream);  // pt_inst_kernels.hip: once per kept scene (DInst::share_bits)
hipError_t launch_pt_pass_inst(const PtParams& p, hipStream_t stream);  // pt_inst_kernels.hip: scenes kept as meshes + instances
// spec_fn: the per-scene kernel of the session (host/specialise.cpp) instead of the precompiled instantiation, or nullptr
hipError_t launch_pt_pass(const PtParams& p, hipStream_t stream, hipFunction_t spec_fn = nullptr);
hipError_t launch_gpt_sample(const PtParams& p, const GptParams& g, hipStream_t stream);
hipError_t launch_gpt_update(const GptParams& g, uint32_t W, uint32_t H, float* film, hipStream_t stream);
hipError_t launch_gpt_recon_init(const GptParams& g, uint32_t W, uint32_t H, float* old, float spp, hipStream_t stream);
hipError_t launch_gpt_recon(const GptParams& g, uint32_t W, uint32_t H, const float* old, float* cur, float scaling, float spp, hipStream_t stream);
hipError_t launch_aov(const PtParams& p, uint32_t spp, uint32_t aov, uint32_t remap, hipStream_t stream);
hipError_t launch_wf_init(const PtParams& p, const WfBuffers& wf, hipStream_t stream);
hipError_t launch_wf_shade(const PtParams& p, const WfBuffers& wf, uint32_t q_out, hipStream_t stream);
uint32_t wf_trace_blocks_per_cu(const PtParams& p);
hipError_t launch_wf_trace(const PtParams& p, const WfBuffers& wf, uint32_t q_in, uint32_t n_blocks, hipStream_t stream);
// wf_sort.hip: key-value radix sort of a ray queue (24-bit keys)
size_t wf_sort_temp_bytes(uint32_t n);
hipError_t wf_sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, hipStream_t stream);
hipError_t launch_probe_material(const PtParams& p, uint32_t material, uint32_t n, const float* uv, uint32_t* out, hipStream_t stream);
hipError_t launch_init_pcg32(const uint64_t* seeds, void* states, uint64_t n, hipStream_t stream);
hipError_t launch_film_resolve(const float* film, uint64_t n, float splat_scale, float* rgb, hipStream_t stream);
hipError_t launch_ggx_table(const uint64_t* seeds, float* table, uint32_t samples, hipStream_t stream);
hipError_t launch_probe_math(uint32_t n, const float* x, float* s, float* c, float* l, hipStream_t stream);
hipError_t launch_probe_bsdf(const DMaterial* m, const float* table, int mode, const float* wo, uint32_t n, const float* in, float* out,
                             hipStream_t stream);
hipError_t launch_probe_intersect(const PtParams& p, uint32_t n, const float* rays, uint32_t* out, float* bary, hipStream_t stream);
hipError_t launch_probe_si(const PtParams& p, uint32_t n, const uint32_t* inst_prim, const float* bary, float* out, hipStream_t stream);
#endif

}  // namespace akr
