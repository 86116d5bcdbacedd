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
