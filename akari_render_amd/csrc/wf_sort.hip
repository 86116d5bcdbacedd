// wf_sort.hip -- key-value sort of the wavefront schedule's ray queues (option wf_sort): ray ids ordered by
// (Morton code of the origin, octant of the direction) so that the 64 rays a trace wave claims walk the same part of the tree.
// The reference's author marked the same point in crates/akari_integrator/src/wfpt.rs:100-225 (a work queue per kernel, sorted by
// material there). First measurement: rocPRIM's device radix sort (ROCm's own primitives library, header-only); the model that
// priced the idea is tools/ray_sort_sim.cpp (profiles/r5_ray_sort_sim.json).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>

namespace akr {

size_t wf_sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, 24u, (hipStream_t) nullptr);
    return bytes;
}
hipError_t wf_sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 24u, stream);
}

}  // namespace akr
