// pt_launch.h -- k_pt_pass, the LDS plan of its launches and the launcher that picks the instantiation. Included by the two translation
// units that instantiate the kernel: pt_kernels.hip (the AKR-F32 contract: the default, and the verifier) and pt_kernels_relaxed.hip
// (the relaxed arithmetic tier, device/dmath.h AKR_ARITH_RELAXED; there everything below lives in namespace akr_rx).
#pragma once
#include <algorithm>
#include "device/pt_pass.h"

namespace akr {

template <bool BVH, bool FD, bool TEX, bool PMJ, bool STAGE, bool DEFER, bool SIMPLE = false>
__global__ __launch_bounds__(256, pt_pass_min_waves(BVH, FD, TEX)) void k_pt_pass(const PtParams p) {
    pt_pass_body<BVH, FD, TEX, PMJ, STAGE, DEFER, SIMPLE ? AB_SIMPLE : 0u>(p);
}

// Dynamic LDS of a k_pt_pass launch and where its blocks start: [traversal stacks][staged tables][triangle records (WALK 1)][node
// tile][park columns][carry columns][blue-noise columns (pmj02bn)][graph values]. Shared by the precompiled kernels, the per-scene
// kernels and the instanced-scene kernels (pt_inst_kernels.hip).
PtParams pt_pass_layout(const PtParams& p, size_t& lds, uint32_t& blocks) {
    blocks = (p.n_items + 255u) / 256u;
    const bool fd = p.force_diffuse != 0, tex = p.sc.tex.nodes != nullptr;
    const bool bvh = p.sc.bvh_nodes != nullptr, inst = p.sc.in2.on != 0;
    const PtLdsPlan plan = pt_lds_plan(bvh, fd, tex, p.defer_metal != 0, p.sc.n_tris);
    size_t base = (bvh ? p.sc.bvh_stack_depth * 256 * 4 : 0) + p.stage_total + plan.recs_bytes;
    base = (base + 15) & ~(size_t)15;
    PtParams pp = p;
    pp.tile_offset = (uint32_t)(base / 4);
    pp.sc.bvh_tile_nodes = 0;
    if (plan.tile && !inst) {
        // what is left of the workgroup's share of the CU's LDS after the launch's other blocks
        const size_t other = base + plan.park_bytes + plan.carry_bytes + (tex ? (size_t)p.tex_slots * kTexValStride * sizeof(TexVal) : 0);
        const size_t budget = pt_lds_budget(tex) - 256;
        if (other < budget) pp.sc.bvh_tile_nodes = (uint32_t)std::min<size_t>({(budget - other) / (kBvhNodeWords * 4), (size_t)p.sc.n_nodes, (size_t)1024});
        base += (size_t)pp.sc.bvh_tile_nodes * kBvhNodeWords * 4;
        base = (base + 15) & ~(size_t)15;
    }
    pp.park_offset = (uint32_t)(base / 4);
    base += plan.park_bytes;
    pp.carry_offset = (uint32_t)(base / 4);
    base += inst ? (AKR_PT_STRAGGLERS_INST > 0 ? (size_t)kCarrySlotsInstanced * 256 * 4 : 0) : plan.carry_bytes;
    pp.bn_offset = 0;
    {   // pmj02bn: the lanes' blue-noise columns, if the workgroup's share of the CU's LDS has room for them (exhaustive kernels of
        // small scenes: 24 KB next to ~13 KB of staged tables; the BVH kernels' traversal stacks leave none)
        const size_t slots = tex ? (size_t)p.tex_slots * kTexValStride * sizeof(TexVal) : 0;
        if (p.sampler == 1u && !bvh && p.bluenoise != nullptr && base + slots + kBlueNoiseColumnBytes <= pt_lds_budget(tex)) {
            base = (base + 15) & ~(size_t)15;
            pp.bn_offset = (uint32_t)(base / 4);
            base += kBlueNoiseColumnBytes;
        }
    }
    return with_tex_slots(pp, base, lds);
}
hipError_t launch_pt_pass(const PtParams& p, hipStream_t stream, hipFunction_t spec_fn) {
#if !AKR_ARITH_RELAXED
    if (p.sc.in2.on && !spec_fn) return launch_pt_pass_inst(p, stream);  // meshes + instances: pt_inst_kernels.hip (a per-scene kernel wraps the same body: below)
#else
    if (p.sc.in2.on) return hipErrorInvalidValue;  // (the host never sends a kept scene to the relaxed tier: api_pt.cpp)
#endif
    size_t lds;
    uint32_t blocks;
    const PtParams q = pt_pass_layout(p, lds, blocks);
    if (blocks == 0) return hipSuccess;
    const bool fd = p.force_diffuse != 0, tex = p.sc.tex.nodes != nullptr;
    const bool bvh = p.sc.bvh_nodes != nullptr;
    const bool stage = p.stage_total != 0;
    if (spec_fn) {  // the scene's own kernel (host/specialise.cpp): same parameter block, same LDS layout (p.tex_slots is 0: no value slots)
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)spec_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        void* args[] = {(void*)&q};
        return hipModuleLaunchKernel(spec_fn, blocks, 1, 1, 256, 1, 1, (unsigned)lds, stream, args, nullptr);
    }
    // a deep tree (up to 24 KB of stacks) + eight graph-value slots (32 KB) + the parked columns can pass the 64 KB a launch gets
    // without asking: the kernel is then allowed what it needs (a workgroup may have all 160 KB of the CU; fewer workgroups fit)
#define AKR_LAUNCH4(K)                                                                                                              \
    {                                                                                                                               \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        hipLaunchKernelGGL((K), dim3(blocks), dim3(256), lds, stream, q);                                                           \
    }
#define AKR_LAUNCH3(B, F, T, S, D, X)                                            \
    {                                                                           \
        if (p.sampler) AKR_LAUNCH4((k_pt_pass<B, F, T, true, S, D, X>))          \
        else AKR_LAUNCH4((k_pt_pass<B, F, T, false, S, D, X>))                  \
    }
    // the SIMPLE instantiations exist for the full-graph kernels of scenes without textures only
#define AKR_LAUNCH2(B, F, T, S, D)                                           \
    {                                                                        \
        if (!F && !T && p.simple_scene) AKR_LAUNCH3(B, F, T, S, D, (!F && !T)) \
        else AKR_LAUNCH3(B, F, T, S, D, false)                               \
    }
#define AKR_LAUNCH(B, F, T)                                                  \
    {                                                                        \
        if (!B && !F && p.defer_metal) AKR_LAUNCH2(false, false, T, true, true)   \
        else if (B && T && !F && p.defer_metal) {                                 \
            if (stage) AKR_LAUNCH2(B, false, T, true, true)                       \
            else AKR_LAUNCH2(B, false, T, false, true)                            \
        }                                                                         \
        else if (!B || stage) AKR_LAUNCH2(B, F, T, true, false)                  \
        else AKR_LAUNCH2(B, F, T, !B, false)                                     \
    }
    if (bvh) {
        if (tex) { if (fd) AKR_LAUNCH(true, true, true) else AKR_LAUNCH(true, false, true) }
        else { if (fd) AKR_LAUNCH(true, true, false) else AKR_LAUNCH(true, false, false) }
    } else {
        if (tex) { if (fd) AKR_LAUNCH(false, true, true) else AKR_LAUNCH(false, false, true) }
        else { if (fd) AKR_LAUNCH(false, true, false) else AKR_LAUNCH(false, false, false) }
    }
#undef AKR_LAUNCH4
#undef AKR_LAUNCH3
#undef AKR_LAUNCH2
#undef AKR_LAUNCH
    return hipGetLastError();
}
}  // namespace akr
