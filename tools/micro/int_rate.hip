// int_rate.hip -- issue cost of the integer VALU instructions the index-based samplers are made of (xxhash32, Kensler's permutation,
// the Laine-Karras hash, bit reversal), next to a plain f32 multiply: cycles of one SIMD per wave64 instruction, four independent
// chains, 1 - 8 waves per SIMD. Measurement only (samplers).  build: hipcc --offload-arch=gfx950 -O3 int_rate.hip -o int_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int V>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned a) {
    unsigned x0 = threadIdx.x * 2654435761u + 1u, x1 = x0 ^ 0x9e3779b9u, x2 = x0 + 0x85ebca6bu, x3 = x0 * 3u, va = a | 1u;
    unsigned long long q0 = x0, q1 = x1, q2 = x2, q3 = x3;
    asm volatile("" : "+v"(va));
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(REP16("v_mul_f32 %0, %4, %0\nv_mul_f32 %1, %4, %1\nv_mul_f32 %2, %4, %2\nv_mul_f32 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va));
        if (V == 1) asm volatile(REP16("v_mul_lo_u32 %0, %4, %0\nv_mul_lo_u32 %1, %4, %1\nv_mul_lo_u32 %2, %4, %2\nv_mul_lo_u32 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va));
        if (V == 2) asm volatile(REP16("v_mul_hi_u32 %0, %4, %0\nv_mul_hi_u32 %1, %4, %1\nv_mul_hi_u32 %2, %4, %2\nv_mul_hi_u32 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va));
        if (V == 3) asm volatile(REP16("v_mul_u32_u24 %0, %4, %0\nv_mul_u32_u24 %1, %4, %1\nv_mul_u32_u24 %2, %4, %2\nv_mul_u32_u24 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va));
        if (V == 4) asm volatile(REP16("v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\nv_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3\n") : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(va), "v"(x0) : "vcc");
        if (V == 5) asm volatile(REP16("v_bfrev_b32 %0, %0\nv_bfrev_b32 %1, %1\nv_bfrev_b32 %2, %2\nv_bfrev_b32 %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 6) asm volatile(REP16("v_xor_b32 %0, %4, %0\nv_xor_b32 %1, %4, %1\nv_xor_b32 %2, %4, %2\nv_xor_b32 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va));
        if (V == 7) asm volatile(REP16("v_alignbit_b32 %0, %0, %0, 15\nv_alignbit_b32 %1, %1, %1, 15\nv_alignbit_b32 %2, %2, %2, 15\nv_alignbit_b32 %3, %3, %3, 15\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 8) asm volatile(REP16("v_lshrrev_b32 %0, 5, %0\nv_lshrrev_b32 %1, 5, %1\nv_lshrrev_b32 %2, 5, %2\nv_lshrrev_b32 %3, 5, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (V == 9) asm volatile(REP16("v_mul_lo_u32 %0, %4, %0\nv_mul_lo_u32 %1, %4, %1\nv_mul_lo_u32 %2, %4, %2\nv_mul_lo_u32 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(a));
        if (V == 10) asm volatile(REP16("v_mul_lo_u32 %0, %1, %0\n") : "+v"(x0) : "v"(va));
        if (V == 11) asm volatile(REP16("v_mad_u32_u24 %0, %4, %0, %5\nv_mad_u32_u24 %1, %4, %1, %5\nv_mad_u32_u24 %2, %4, %2, %5\nv_mad_u32_u24 %3, %4, %3, %5\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va), "v"(x0));
        if (V == 12) asm volatile(REP16("v_xad_u32 %0, %0, %4, %5\nv_xad_u32 %1, %1, %4, %5\nv_xad_u32 %2, %2, %4, %5\nv_xad_u32 %3, %3, %4, %5\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va), "v"(x1));
    }
    unsigned s = x0 + x1 + x2 + x3 + (unsigned)q0 + (unsigned)q1 + (unsigned)q2 + (unsigned)q3;
    if (s == 12345u) out[0] = s;
}
static const char* names[] = {"v_mul_f32 (reference)", "v_mul_lo_u32 vgpr", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u64_u32", "v_bfrev_b32", "v_xor_b32", "v_alignbit_b32 (rotate)",
                              "v_lshrrev_b32", "v_mul_lo_u32 sgpr", "v_mul_lo_u32 one dependent chain", "v_mad_u32_u24", "v_xad_u32"};
static const int per_iter[] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 64, 64};
template <int V>
void run(unsigned* d) {
    printf("%-34s", names[V]);
    for (int w : {1, 2, 4, 8}) {
        const int iters = 50000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<V>), dim3(256 * w), dim3(256), 0, 0, d, iters / 10, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V>), dim3(256 * w), dim3(256), 0, 0, d, iters, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double rate = (double)iters * per_iter[V] * w / (ms * 1e-3);
        printf("  w%d: %.2f", w, 2.4e9 / rate);
    }
    printf("\n");
}
int main() {
    unsigned* d;
    (void)hipMalloc(&d, 4);
    printf("cycles of one SIMD per wave64 instruction (2.4 GHz), four independent chains, w = waves per SIMD\n");
    run<0>(d); run<1>(d); run<2>(d); run<3>(d); run<4>(d); run<5>(d); run<6>(d); run<7>(d); run<8>(d); run<9>(d); run<10>(d); run<11>(d); run<12>(d);
    return 0;
}
