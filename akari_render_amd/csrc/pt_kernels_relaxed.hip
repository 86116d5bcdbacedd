// pt_kernels_relaxed.hip -- k_pt_pass in the RELAXED arithmetic tier (option `arith` = 1; device/dmath.h AKR_ARITH_RELAXED).
//
// The same source as pt_kernels.hip's instantiations -- pt_pass.h, every device header -- compiled a second time with
// -ffp-contract=fast, without correctly rounded division / square root, with denormals flushed (build.py RELAXED_FLAGS) and
// with AKR_ARITH_RELAXED = 1, inside another namespace so that neither the kernels nor a header's inline function can be
// mistaken for the contract-bound ones at link time. Films are the oracle's to relRMSE < 1e-3 (north_star's bar; measured
// ~1e-5, tests/test_gpu_relaxed.py), not to the bit: the bit-exact tier stays the default and is what verifies this one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#define AKR_ARITH_RELAXED 1
#define akr akr_rx
#include "pt_launch.h"
#undef akr

extern "C" hipError_t akr_launch_pt_pass_relaxed(const void* params, hipStream_t stream) {
    return akr_rx::launch_pt_pass(*static_cast<const akr_rx::PtParams*>(params), stream, nullptr);
}
