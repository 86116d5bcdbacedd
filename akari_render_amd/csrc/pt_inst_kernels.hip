// pt_inst_kernels.hip -- k_pt_pass for scenes kept as meshes + instances (host/scene_inst.cpp): the same persistent-lane path
// tracer over the two-level traversal of device/dinst_trav.h. BVH kernels without staged tables, deferral or absent-lobe masks:
// force_diffuse x textures x sampler family.
#include "device/pt_pass.h"

#ifndef AKR_PT_MIN_WAVES_INST
#define AKR_PT_MIN_WAVES_INST AKR_PT_MIN_WAVES_BVH
#endif
#ifndef AKR_PT_MIN_WAVES_INST_TEX
#define AKR_PT_MIN_WAVES_INST_TEX AKR_PT_MIN_WAVES_BVH_TEX
#endif

namespace akr {

template <bool FD, bool TEX, bool PMJ>
__global__ __launch_bounds__(256, TEX ? AKR_PT_MIN_WAVES_INST_TEX : AKR_PT_MIN_WAVES_INST) void k_pt_pass_inst(const PtParams p) {
    pt_pass_body<true, FD, TEX, PMJ, false, false, 0u, true>(p);
}

// One thread per instance-triangle, once per scene: the bit of an odd triangle that takes its even neighbour's plane row (dinst.h
// share_plane_row) -- what the flattening compiler decides per instance-triangle and resolve_pending used to decide at every candidate.
__global__ __launch_bounds__(256) void k_inst_share_bits(const DScene sc, uint32_t* __restrict__ bits, uint32_t* __restrict__ mesh_tri_words) {
    const uint32_t n_inst = sc.in2.n_instances;
    for (uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x; gid < sc.n_tris; gid += (uint64_t)gridDim.x * 256u) {
        uint32_t lo = 0, hi = n_inst;  // the instance of gid: the last one whose first id is <= gid (inst_tri_offset has n_inst + 1 entries)
        while (hi - lo > 1u) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            if (sc.inst_tri_offset[mid] <= gid) lo = mid; else hi = mid;
        }
        const uint32_t prim = (uint32_t)gid - sc.inst_tri_offset[lo];
        uint32_t pos;
        if ((prim & 1u) && inst_pair_shares(sc, lo, prim, pos)) {
            atomicOr(&bits[gid >> 5], 1u << ((uint32_t)gid & 31u));
            atomicOr(&mesh_tri_words[16ull * pos + 15u], kMeshTriShares);  // (nothing here reads that word)
        }
    }
}
hipError_t launch_inst_share_bits(const DScene& sc, uint32_t* bits, uint32_t* mesh_tri_words, hipStream_t stream) {
    if (sc.n_tris == 0 || sc.in2.n_instances == 0) return hipSuccess;
    const uint64_t blocks = ((uint64_t)sc.n_tris + 255u) / 256u;
    hipLaunchKernelGGL(k_inst_share_bits, dim3((uint32_t)(blocks < (1u << 20) ? blocks : (1u << 20))), dim3(256), 0, stream, sc, bits, mesh_tri_words);
    return hipGetLastError();
}

hipError_t launch_pt_pass_inst(const PtParams& p, hipStream_t stream) {
    size_t lds;
    uint32_t blocks;
    const PtParams q = pt_pass_layout(p, lds, blocks);
    if (blocks == 0) return hipSuccess;
    const bool fd = p.force_diffuse != 0, tex = p.sc.tex.nodes != nullptr, pmj = p.sampler != 0;
#define AKR_LAUNCH_INST(F, T, S)                                                                                                          \
    {                                                                                                                                   \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)(k_pt_pass_inst<F, T, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_pt_pass_inst<F, T, S>), dim3(blocks), dim3(256), lds, stream, q);                                            \
    }
    if (fd) {
        if (tex) { if (pmj) AKR_LAUNCH_INST(true, true, true) else AKR_LAUNCH_INST(true, true, false) }
        else { if (pmj) AKR_LAUNCH_INST(true, false, true) else AKR_LAUNCH_INST(true, false, false) }
    } else {
        if (tex) { if (pmj) AKR_LAUNCH_INST(false, true, true) else AKR_LAUNCH_INST(false, true, false) }
        else { if (pmj) AKR_LAUNCH_INST(false, false, true) else AKR_LAUNCH_INST(false, false, false) }
    }
#undef AKR_LAUNCH_INST
    return hipGetLastError();
}

}  // namespace akr
