// gather_bw.hip -- calibrates rocprofv3's HBM-side counters (FETCH_SIZE, TCC_EA0_RDREQ*, TCC_MISS) for the access pattern of the
// BVH traversal (device/disect.h trav_step): every lane issues four 16-byte loads from a RANDOM 64-byte-aligned record of a
// buffer far larger than the 256 MiB Infinity Cache, so the bytes a launch must move are known exactly.
// MI355X_MICROARCH.md (HBM) calibrates "FETCH_SIZE reports half the bytes" for wide coalesced streaming reads only and says
// other patterns must be calibrated on a known byte count -- this is that calibration; tools/pmc_bench.sh uses the factor
// profiles/r4_fetch_size_calibration.json records for the BVH configurations.
//
//   gather_bw <pattern> [buffer MiB = 4096] [records per lane = 64] [blocks = 8192] [dynamic LDS KiB per block = 0]
// (dynamic LDS bounds the occupancy: 40 KiB = four 256-thread blocks per CU = 4 waves per SIMD, the BVH kernel's)
// patterns:
//   stream     coalesced 16 B per lane, consecutive (the guide's calibrated case: factor 2 expected)
//   gather64   random 64-B-aligned record, four 16-B loads (= a node visit / triangle test of the traversal)
//   gather128  random 128-B-aligned pair of records, eight 16-B loads (what "two siblings in one line" would fetch)
//   gather16   random 64-B-aligned record, ONE 16-B load (how much of a line does a 16-B miss fetch?)
//   gather64x2 as gather64 with TWO independent records in flight per lane
//   gather32   random 64-B-aligned record, TWO 16-B loads (the first 32 B)
//   chain4_1k / chain4_2k / chain4_4k / chain4_64k   round 6: a random window of 1 / 2 / 4 / 64 KiB, then FOUR dependent
//              64-byte records at random places inside it -- what a traversal would fetch if a subtree's nodes were laid out as a treelet
//              aligned to such a window (does the rate of random records rise when consecutive dependent fetches share a DRAM page?)
// Prints one JSON line: known bytes, time, GB/s, loads. Run it under `rocprofv3 --kernel-trace --pmc <set>` (tools/gather_calib.sh).
// build: hipcc --offload-arch=gfx950 -O3 gather_bw.hip -o gather_bw
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__device__ __forceinline__ uint32_t mix(uint32_t x) {  // lowbias32
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int LOADS, int STRIDE16>  // LOADS 16-byte loads from a record aligned to STRIDE16 * 16 bytes
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ buf, uint64_t n_slots, uint32_t per_lane, uint32_t seed, uint32_t* out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0, h = mix(gid * 0x9e3779b9u + seed);
    for (uint32_t i = 0; i < per_lane; i++) {
        h = mix(h + i);
        // 64-bit multiply-shift: uniform slot in [0, n_slots)
        const uint64_t slot = ((uint64_t)h * n_slots) >> 32;
        const uint4* p = buf + slot * STRIDE16;
        uint4 w[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; j++) w[j] = p[j];
#pragma unroll
        for (int j = 0; j < LOADS; j++) acc ^= w[j].x ^ w[j].y ^ w[j].z ^ w[j].w;
        h ^= acc & 1u;  // the next address depends on the data, as a traversal's does (one record in flight per lane)
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// two independent random records in flight per lane (memory-level parallelism per lane: does the gather rate rise?)
__global__ __launch_bounds__(256) void k_gather_x2(const uint4* __restrict__ buf, uint64_t n_slots, uint32_t per_lane, uint32_t seed, uint32_t* out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0, h = mix(gid * 0x9e3779b9u + seed);
    for (uint32_t i = 0; i < per_lane; i += 2) {
        h = mix(h + i);
        const uint32_t h2 = mix(h ^ 0x5bd1e995u);
        const uint4* p = buf + (((uint64_t)h * n_slots) >> 32) * 4;
        const uint4* q = buf + (((uint64_t)h2 * n_slots) >> 32) * 4;
        uint4 w[8];
#pragma unroll
        for (int j = 0; j < 4; j++) { w[j] = p[j]; w[4 + j] = q[j]; }
#pragma unroll
        for (int j = 0; j < 8; j++) acc ^= w[j].x ^ w[j].y ^ w[j].z ^ w[j].w;
        h ^= acc & 1u;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// four dependent records inside one random window of WIN_RECORDS 64-byte records (aligned to the window's size)
template <uint32_t WIN_RECORDS>
__global__ __launch_bounds__(256) void k_chain4(const uint4* __restrict__ buf, uint64_t n_windows, uint32_t per_lane, uint32_t seed, uint32_t* out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0, h = mix(gid * 0x9e3779b9u + seed);
    for (uint32_t i = 0; i < per_lane; i += 4) {
        h = mix(h + i);
        const uint64_t win = ((uint64_t)h * n_windows) >> 32;
        const uint4* base = buf + win * (uint64_t)WIN_RECORDS * 4;
#pragma unroll 1
        for (uint32_t k = 0; k < 4; k++) {
            h = mix(h + k);
            const uint4* p = base + (h % WIN_RECORDS) * 4;
            const uint4 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3];
            acc ^= w0.x ^ w0.y ^ w0.z ^ w0.w ^ w1.x ^ w1.y ^ w1.z ^ w1.w ^ w2.x ^ w2.y ^ w2.z ^ w2.w ^ w3.x ^ w3.y ^ w3.z ^ w3.w;
            h ^= acc & 1u;  // the next record's place depends on this one's data
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ buf, uint64_t n16, uint32_t per_lane, uint32_t* out) {
    const uint64_t total = (uint64_t)gridDim.x * 256u;
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per_lane; k++, i += total) {
        const uint4 w = buf[i];  // main() keeps lanes x per_lane within the buffer
        acc ^= w.x ^ w.y ^ w.z ^ w.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_fill(uint4* buf, uint64_t n16) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n16; i += step) buf[i] = make_uint4((uint32_t)i, (uint32_t)(i >> 32), 0x01010101u, ~(uint32_t)i);
}

int main(int argc, char** argv) {
    const char* pat = argc > 1 ? argv[1] : "gather64";
    const uint64_t mib = argc > 2 ? strtoull(argv[2], nullptr, 10) : 4096;
    const uint32_t per_lane = argc > 3 ? (uint32_t)atoi(argv[3]) : 64;
    const uint32_t blocks = argc > 4 ? (uint32_t)atoi(argv[4]) : 8192;
    const size_t lds = (argc > 5 ? (size_t)atoi(argv[5]) : 0) << 10;
    const uint64_t bytes = mib << 20, n16 = bytes / 16;
    uint4* buf;
    uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, n16);
    (void)hipDeviceSynchronize();
    int loads = 4, stride16 = 4;
    if (!strcmp(pat, "gather128")) { loads = 8; stride16 = 8; }
    if (!strcmp(pat, "gather16")) loads = 1;
    if (!strcmp(pat, "gather32")) loads = 2;
    const bool stream = !strcmp(pat, "stream"), x2 = !strcmp(pat, "gather64x2");
    const uint32_t chain_win = !strcmp(pat, "chain4_1k") ? 16u : !strcmp(pat, "chain4_2k") ? 32u : !strcmp(pat, "chain4_4k") ? 64u : !strcmp(pat, "chain4_64k") ? 1024u : 0u;
    if (stream) loads = 1;
    const uint64_t n_slots = n16 / stride16;
    if (stream && (uint64_t)blocks * 256u * per_lane > n16) { fprintf(stderr, "stream: buffer too small\n"); return 1; }
    auto launch = [&](uint32_t seed) {
        if (chain_win == 16u) hipLaunchKernelGGL((k_chain4<16>), dim3(blocks), dim3(256), lds, 0, buf, n16 / (4 * 16), per_lane, seed, out);
        else if (chain_win == 32u) hipLaunchKernelGGL((k_chain4<32>), dim3(blocks), dim3(256), lds, 0, buf, n16 / (4 * 32), per_lane, seed, out);
        else if (chain_win == 64u) hipLaunchKernelGGL((k_chain4<64>), dim3(blocks), dim3(256), lds, 0, buf, n16 / (4 * 64), per_lane, seed, out);
        else if (chain_win == 1024u) hipLaunchKernelGGL((k_chain4<1024>), dim3(blocks), dim3(256), lds, 0, buf, n16 / (4 * 1024), per_lane, seed, out);
        else if (stream) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), lds, 0, buf, n16, per_lane, out);
        else if (x2) hipLaunchKernelGGL(k_gather_x2, dim3(blocks), dim3(256), lds, 0, buf, n_slots, per_lane, seed, out);
        else if (loads == 8) hipLaunchKernelGGL((k_gather<8, 8>), dim3(blocks), dim3(256), lds, 0, buf, n_slots, per_lane, seed, out);
        else if (loads == 4) hipLaunchKernelGGL((k_gather<4, 4>), dim3(blocks), dim3(256), lds, 0, buf, n_slots, per_lane, seed, out);
        else if (loads == 2) hipLaunchKernelGGL((k_gather<2, 4>), dim3(blocks), dim3(256), lds, 0, buf, n_slots, per_lane, seed, out);
        else hipLaunchKernelGGL((k_gather<1, 4>), dim3(blocks), dim3(256), lds, 0, buf, n_slots, per_lane, seed, out);
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(1);  // warm-up (also a counted dispatch under the profiler: the summary divides by the number of dispatches)
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(2);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double lanes = (double)blocks * 256.0;
    const double requested = lanes * per_lane * loads * 16.0;  // bytes the lanes asked for, per launch
    const double records = lanes * per_lane;
    printf("{\"records_per_s_G\": %.2f, ", stream ? 0.0 : records / (ms * 1e-3) / 1e9);
    printf("\"pattern\": \"%s\", \"buffer_mib\": %llu, \"records_per_lane\": %u, \"lanes\": %.0f, \"loads_per_record\": %d, \"record_align\": %d, "
           "\"requested_bytes_per_launch\": %.0f, \"launches\": 2, \"lds_bytes\": %zu, \"ms\": %.4f, \"requested_gbs\": %.1f}\n",
           pat, (unsigned long long)mib, per_lane, lanes, loads, stream ? 16 : stride16 * 16, requested, lds, ms, requested / (ms * 1e-3) / 1e9);
    return 0;
}
