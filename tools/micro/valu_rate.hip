// valu_rate.hip -- what one SIMD of gfx950 retires per cycle in wave64 f32 VALU instructions, by operand kind (VGPR / SGPR /
// literal), dependency (one chain or four independent ones) and waves per SIMD. Measurement only: DESIGN.md prices the
// "VALU busy" figures of the kernels with it.  build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = (float)threadIdx.x * 1e-3f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, va = a, vb = b;
    asm volatile("" : "+v"(va), "+v"(vb));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x1, x2}, p2 = {x2, x3}, p3 = {x3, x0}, pa = {a, a}, sa = {a, a};
    asm volatile("" : "+v"(pa));
    // V >= 24: the same streams with only the low 32 lanes of the wave enabled -- does a SIMD-32 skip the empty half of a wave64?
    if (V >= 24) asm volatile("s_mov_b64 exec, 0xffffffff");
    for (int i = 0; i < iters; i++) {
        if (V == 24) asm volatile(REP16("v_mul_f32 %0, %4, %0\nv_mul_f32 %1, %4, %1\nv_mul_f32 %2, %4, %2\nv_mul_f32 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va));
        if (V == 25) asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va), "v"(vb));
        if (V == 26) asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(a), "v"(vb));
        if (V == 0) asm volatile(REP16("v_mul_f32 %0, %1, %0\n") : "+v"(x0) : "v"(va));
        if (V == 1) asm volatile(REP16("v_mul_f32 %0, %1, %0\n") : "+v"(x0) : "s"(a));
        if (V == 2) asm volatile(REP16("v_mul_f32 %0, 0x3f800347, %0\n") : "+v"(x0));
        if (V == 3) asm volatile(REP16("v_mul_f32 %0, %4, %0\nv_mul_f32 %1, %4, %1\nv_mul_f32 %2, %4, %2\nv_mul_f32 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va));
        if (V == 4) asm volatile(REP16("v_mul_f32 %0, %4, %0\nv_mul_f32 %1, %4, %1\nv_mul_f32 %2, %4, %2\nv_mul_f32 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(a));
        if (V == 5) asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "v"(va), "v"(vb));
        if (V == 6) asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n") : "+v"(x0) : "s"(a), "v"(vb));
        if (V == 7) asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va), "v"(vb));
        if (V == 8) asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(a), "v"(vb));
        if (V == 9) asm volatile(REP16("v_mul_f32 %0, %1, %0\nv_max_f32 %0, %0, %2\n") : "+v"(x0) : "s"(a), "v"(vb));
        if (V == 10) asm volatile(REP16("v_fma_f32 %0, %0, %4, %1\nv_fma_f32 %1, %1, %4, %2\nv_fma_f32 %2, %2, %4, %3\nv_fma_f32 %3, %3, %4, %0\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(a));
        if (V == 11) asm volatile(REP16("v_mul_f32 %0, %1, %0\nv_mul_f32 %0, %2, %0\n") : "+v"(x0) : "s"(a), "s"(b));
        if (V == 12) asm volatile(REP16("v_mul_f32 %0, %1, %0\ns_nop 0\n") : "+v"(x0) : "s"(a));
        if (V == 13) asm volatile(REP16("v_pk_mul_f32 %0, %1, %0\n") : "+v"(p0) : "v"(pa));
        if (V == 14) asm volatile(REP16("v_pk_mul_f32 %0, %1, %0\n") : "+v"(p0) : "s"(sa));
        if (V == 15) asm volatile(REP16("v_pk_mul_f32 %0, %4, %0\nv_pk_mul_f32 %1, %4, %1\nv_pk_mul_f32 %2, %4, %2\nv_pk_mul_f32 %3, %4, %3\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
        if (V == 16) asm volatile(REP16("v_pk_mul_f32 %0, %4, %0\nv_pk_mul_f32 %1, %4, %1\nv_pk_mul_f32 %2, %4, %2\nv_pk_mul_f32 %3, %4, %3\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(sa));
        if (V == 17) asm volatile(REP16("v_mul_f32 %0, %2, %0\nv_mul_f32 %1, %3, %1\n") : "+v"(x0), "+v"(x1) : "s"(a), "v"(va));
        if (V == 18) asm volatile(REP16("v_mul_f32 %0, %3, %0\nv_mul_f32 %1, %4, %1\nv_mul_f32 %2, %4, %2\n") : "+v"(x0), "+v"(x1), "+v"(x2) : "s"(a), "v"(va));
        if (V == 19) asm volatile(REP16("v_mul_f32 %0, %3, %0\nv_mul_f32 %1, %3, %1\nv_mul_f32 %2, %4, %2\n") : "+v"(x0), "+v"(x1), "+v"(x2) : "s"(a), "v"(va));
        if (V == 20) asm volatile(REP16("v_mov_b32 %4, %5\nv_mul_f32 %0, %4, %0\nv_mul_f32 %1, %4, %1\nv_mul_f32 %2, %4, %2\nv_mul_f32 %3, %4, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(va) : "s"(a));
        if (V == 21) asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(va), "s"(b));
        if (V == 22) asm volatile(REP16("v_mul_f32 %0, %2, %0\nv_mul_f32 %0, %3, %0\n") : "+v"(x0), "+v"(x1) : "s"(a), "v"(va));
        if (V == 23) asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %1, %1, %4, %5 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %2, %2, %4, %5 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %3, %3, %4, %5 op_sel_hi:[1,0,1]\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(sa), "v"(pa));
    }
    float s = x0 + x1 + x2 + x3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (s == 12345.678f) out[0] = s;
}
static const char* names[] = {"mul  dep      vgpr", "mul  dep      sgpr", "mul  dep      literal", "mul  4 chains vgpr", "mul  4 chains sgpr", "fma  dep      vgpr",
                              "fma  dep      1 sgpr", "fma  4 chains vgpr", "fma  4 chains 1 sgpr", "mul sgpr + max vgpr, dep", "fma ring of 4, 1 sgpr",
                              "mul dep, alternating 2 sgprs", "mul dep sgpr + s_nop", "pk_mul dep vgpr", "pk_mul dep sgpr pair", "pk_mul 4 chains vgpr", "pk_mul 4 chains sgpr pair",
                              "[mul sgpr, mul vgpr] 2 chains", "[sgpr, vgpr, vgpr] 3 chains", "[sgpr, sgpr, vgpr] 3 chains", "[v_mov s->v, 4 mul vgpr]",
                              "fma 4 chains sgpr in src2", "[mul sgpr, mul vgpr] ONE chain", "pk_fma 4 chains sgpr bcast",
                              "mul 4 chains vgpr, 32 lanes", "fma 4 chains vgpr, 32 lanes", "fma 4 chains 1 sgpr, 32 lanes"};
static const int per_iter[] = {16, 16, 16, 64, 64, 16, 16, 64, 64, 32, 64, 32, 16, 16, 16, 64, 64, 32, 48, 48, 80, 64, 32, 64, 64, 64, 64};
template <int V>
void run(float* d) {
    printf("%-32s", names[V]);
    for (int w : {1, 2, 4, 8}) {
        const int iters = 100000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<V>), dim3(256 * w), dim3(256), 0, 0, d, iters / 10, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V>), dim3(256 * w), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double rate = (double)iters * per_iter[V] * w / (ms * 1e-3);  // wave-instructions per second per SIMD
        printf("  w%d: %.2f cyc/inst", w, 2.4e9 / rate);
    }
    printf("\n");
}
int main() {
    float* d;
    (void)hipMalloc(&d, 4);
    printf("cycles of one SIMD per wave64 VALU instruction (2.4 GHz), w = waves per SIMD\n");
    run<0>(d); run<1>(d); run<2>(d); run<3>(d); run<4>(d); run<5>(d); run<6>(d); run<7>(d); run<8>(d); run<9>(d); run<10>(d); run<11>(d); run<12>(d); run<13>(d); run<14>(d); run<15>(d); run<16>(d); run<17>(d); run<18>(d); run<19>(d); run<20>(d); run<21>(d); run<22>(d); run<23>(d); run<24>(d); run<25>(d); run<26>(d);
    return 0;
}
