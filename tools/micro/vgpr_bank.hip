// vgpr_bank.hip -- does the cost of a wave64 f32 VALU instruction on gfx950 depend on WHICH VGPRs its sources are (register-file
// banks)?  valu_rate.hip measured 2.2 cycles for a 2-VGPR mul and 2.9 for a 3-VGPR fma with compiler-chosen registers; here the
// registers are named explicitly: sources v[a], v[b], v[c] are never written, destinations v20..v23 never read, so there are no
// dependencies and the only thing that varies is the operands' register numbers. Measurement only.
// build: hipcc --offload-arch=gfx950 -O3 vgpr_bank.hip -o vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"
#define FMA4(a, b, c) "v_fma_f32 v20, " a ", " b ", " c "\nv_fma_f32 v21, " a ", " b ", " c "\nv_fma_f32 v22, " a ", " b ", " c "\nv_fma_f32 v23, " a ", " b ", " c "\n"
#define MUL4(a, b) "v_mul_f32 v20, " a ", " b "\nv_mul_f32 v21, " a ", " b "\nv_mul_f32 v22, " a ", " b "\nv_mul_f32 v23, " a ", " b "\n"
// fmac: dst is also the third source
#define MAC4(a, b) "v_fmac_f32 v20, " a ", " b "\nv_fmac_f32 v21, " a ", " b "\nv_fmac_f32 v22, " a ", " b "\nv_fmac_f32 v23, " a ", " b "\n"
// packed: 64-bit operands (register pairs), two f32 fmas per instruction
#define PK4(a, b, c) "v_pk_fma_f32 v[20:21], " a ", " b ", " c "\nv_pk_fma_f32 v[22:23], " a ", " b ", " c "\nv_pk_fma_f32 v[24:25], " a ", " b ", " c "\nv_pk_fma_f32 v[26:27], " a ", " b ", " c "\n"
#define PK4M(a, b, c, m) "v_pk_fma_f32 v[20:21], " a ", " b ", " c " " m "\nv_pk_fma_f32 v[22:23], " a ", " b ", " c " " m "\nv_pk_fma_f32 v[24:25], " a ", " b ", " c " " m "\nv_pk_fma_f32 v[26:27], " a ", " b ", " c " " m "\n"
#define OP4(op, a, b, c) op " v20, " a ", " b ", " c "\n" op " v21, " a ", " b ", " c "\n" op " v22, " a ", " b ", " c "\n" op " v23, " a ", " b ", " c "\n"
#define MAC4D(d0, d1, d2, d3, a, b) "v_fmac_f32 " d0 ", " a ", " b "\nv_fmac_f32 " d1 ", " a ", " b "\nv_fmac_f32 " d2 ", " a ", " b "\nv_fmac_f32 " d3 ", " a ", " b "\n"

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds_pad[64];
    lds_pad[threadIdx.x & 63] = 0.0f;
    __syncthreads();
    asm volatile("v_mov_b32 v0, 1.0\nv_mov_b32 v1, 1.0\nv_mov_b32 v2, 0.5\nv_mov_b32 v3, 0.5\nv_mov_b32 v4, 1.0\nv_mov_b32 v5, 0.5\nv_mov_b32 v6, 0.5\nv_mov_b32 v7, 0.5\n"
                 "v_mov_b32 v8, 1.0\nv_mov_b32 v9, 0.5\nv_mov_b32 v10, 0.5\nv_mov_b32 v11, 0.5\nv_mov_b32 v12, 0.5\nv_mov_b32 v16, 0.5\nv_mov_b32 v17, 0.5\n"
                 "v_mov_b32 v19, 0\nv_mov_b32 v20, 0\nv_mov_b32 v21, 0\nv_mov_b32 v22, 0\nv_mov_b32 v23, 0\nv_mov_b32 v24, 0\nv_mov_b32 v25, 0\nv_mov_b32 v26, 0\nv_mov_b32 v27, 0\n" ::: CLOB);
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(REP16(MUL4("v0", "v1")) ::: CLOB);
        if (V == 1) asm volatile(REP16(MUL4("v0", "v4")) ::: CLOB);
        if (V == 2) asm volatile(REP16(MUL4("v0", "v2")) ::: CLOB);
        if (V == 3) asm volatile(REP16(MUL4("v0", "v8")) ::: CLOB);
        if (V == 4) asm volatile(REP16(MUL4("v0", "v16")) ::: CLOB);
        if (V == 5) asm volatile(REP16(FMA4("v0", "v1", "v2")) ::: CLOB);
        if (V == 6) asm volatile(REP16(FMA4("v0", "v4", "v2")) ::: CLOB);
        if (V == 7) asm volatile(REP16(FMA4("v0", "v1", "v4")) ::: CLOB);
        if (V == 8) asm volatile(REP16(FMA4("v0", "v4", "v8")) ::: CLOB);
        if (V == 9) asm volatile(REP16(FMA4("v0", "v1", "v5")) ::: CLOB);
        if (V == 10) asm volatile(REP16(FMA4("v0", "v1", "v3")) ::: CLOB);
        if (V == 11) asm volatile(REP16(FMA4("v0", "v2", "v6")) ::: CLOB);
        if (V == 12) asm volatile(REP16(FMA4("v0", "v0", "v1")) ::: CLOB);
        if (V == 13) asm volatile(REP16(MAC4("v0", "v1")) ::: CLOB);   // dsts v20..v23 = banks 0..3 if bank = n % 4
        if (V == 14) asm volatile(REP16(MAC4("v0", "v4")) ::: CLOB);
        if (V == 15) asm volatile(REP16(MAC4D("v20", "v24", "v20", "v24", "v1", "v2")) ::: CLOB);  // dst bank 0, srcs 1, 2 (two chains: dependent pairs)
        if (V == 16) asm volatile(REP16(MAC4D("v21", "v25", "v22", "v26", "v1", "v2")) ::: CLOB);  // dst banks 1 1 2 2 vs srcs 1 2
        if (V == 17) asm volatile(REP16(MAC4D("v20", "v23", "v24", "v27", "v1", "v2")) ::: CLOB);  // dst banks 0 3 0 3
        if (V == 18) asm volatile(REP16(FMA4("v0", "v1", "1.0")) ::: CLOB);
        if (V == 19) asm volatile(REP16(FMA4("v0", "v4", "1.0")) ::: CLOB);
        if (V == 20) asm volatile(REP16(FMA4("v0", "s4", "v1")) ::: CLOB, "s4");
        if (V == 21) asm volatile(REP16(FMA4("v0", "s4", "v4")) ::: CLOB, "s4");
        if (V == 22) asm volatile(REP16(PK4("v[0:1]", "v[2:3]", "v[4:5]")) ::: CLOB);    // pairs in banks (0,1) (2,3) (0,1)
        if (V == 23) asm volatile(REP16(PK4("v[0:1]", "v[4:5]", "v[2:3]")) ::: CLOB);    // src0, src1 in the same banks
        if (V == 24) asm volatile(REP16(PK4("v[0:1]", "v[4:5]", "v[8:9]")) ::: CLOB);    // all three
        if (V == 25) asm volatile(REP16(PK4("v[0:1]", "v[2:3]", "v[6:7]")) ::: CLOB);    // src1, src2 in the same banks
        if (V == 26) asm volatile(REP16(PK4M("v[0:1]", "v[2:3]", "v[4:5]", "op_sel:[0,0,0] op_sel_hi:[0,1,1]")) ::: CLOB);  // src0 low half broadcast
        if (V == 27) asm volatile(REP16(PK4M("v[0:1]", "v[4:5]", "v[2:3]", "op_sel:[1,0,0] op_sel_hi:[1,1,1]")) ::: CLOB);  // src0 high half broadcast, banks as 23
        if (V == 28) asm volatile(REP16(OP4("v_min3_f32", "v0", "v4", "v2")) ::: CLOB);
        if (V == 29) asm volatile(REP16(OP4("v_min3_f32", "v0", "v1", "v2")) ::: CLOB);
        if (V == 30) asm volatile(REP16(OP4("v_cndmask_b32_e64", "v0", "v4", "s[4:5]")) ::: CLOB, "s4", "s5");
        if (V == 31) asm volatile(REP16(OP4("v_cndmask_b32_e64", "v0", "v1", "s[4:5]")) ::: CLOB, "s4", "s5");
        if (V == 32) asm volatile(REP16(OP4("v_fma_f32", "v0", "v4", "v1")) ::: CLOB);   // src0/src1 same bank, src2 elsewhere
        if (V == 33) asm volatile(REP16(OP4("v_fma_f32", "v1", "v0", "v4")) ::: CLOB);   // src1/src2 same bank
        if (V == 34) asm volatile(REP16(OP4("v_fma_f32", "v0", "v1", "v4")) ::: CLOB);   // src0/src2 same bank (free in the first run)
        if (V == 35) asm volatile(REP16(OP4("v_fma_f32", "v0", "v4", "v0")) ::: CLOB);   // src0 == src2, src1 same bank
        if (V == 36) asm volatile(REP16("v_mul_f32 v20, v0, v1\nv_fma_f32 v21, v0, v4, v2\nv_mul_f32 v22, v0, v1\nv_fma_f32 v23, v0, v4, v2\n") ::: CLOB);  // conflicted fma between free muls
        if (V == 38) asm volatile(REP16(MAC4D("v21", "v23", "v25", "v27", "v0", "v4")) ::: CLOB);   // even, even, odd dst (= src2)
        if (V == 39) asm volatile(REP16(MAC4D("v20", "v22", "v24", "v26", "v0", "v4")) ::: CLOB);   // even, even, even
        if (V == 40) asm volatile(REP16(MAC4D("v20", "v22", "v24", "v26", "v1", "v5")) ::: CLOB);   // odd, odd, even
        if (V == 41) asm volatile(REP16(MAC4D("v21", "v23", "v25", "v27", "v1", "v5")) ::: CLOB);   // odd, odd, odd
        if (V == 42) asm volatile(REP16(OP4("v_fma_f32", "v0", "v2", "v1")) ::: CLOB);              // even, even, odd (banks mod 4: 0, 2, 1)
        if (V == 43) asm volatile(REP16(OP4("v_fma_f32", "v1", "v3", "v5")) ::: CLOB);              // odd, odd, odd
        if (V == 44) asm volatile(REP16(OP4("v_fma_f32", "v1", "v5", "v0")) ::: CLOB);              // odd, odd, even
        if (V == 45) asm volatile(REP16(OP4("v_fma_f32", "v0", "v2", "v4")) ::: CLOB);              // even x3, banks mod 4: 0, 2, 0
        if (V == 46) asm volatile(REP16("v_cndmask_b32_e32 v20, v0, v1, vcc\nv_cndmask_b32_e32 v21, v0, v1, vcc\nv_cndmask_b32_e32 v22, v0, v1, vcc\nv_cndmask_b32_e32 v23, v0, v1, vcc\n") ::: CLOB);
        if (V == 47) asm volatile(REP16("v_cndmask_b32_e32 v20, v0, v4, vcc\nv_cndmask_b32_e32 v21, v0, v4, vcc\nv_cndmask_b32_e32 v22, v0, v4, vcc\nv_cndmask_b32_e32 v23, v0, v4, vcc\n") ::: CLOB);
        if (V == 48) asm volatile(REP16("v_cmp_lt_f32_e32 vcc, v0, v1\nv_cmp_lt_f32_e32 vcc, v0, v2\nv_cmp_lt_f32_e32 vcc, v0, v3\nv_cmp_lt_f32_e32 vcc, v0, v4\n") ::: CLOB, "vcc");
        if (V == 49) asm volatile(REP16("v_cmp_lt_f32_e32 vcc, v0, v1\nv_cndmask_b32_e32 v20, v0, v1, vcc\nv_cmp_lt_f32_e32 vcc, v0, v2\nv_cndmask_b32_e32 v21, v0, v1, vcc\n") ::: CLOB, "vcc");
        if (V == 50) asm volatile(REP16("v_cmp_lt_f32_e32 vcc, v0, v1\ns_nop 1\nv_cndmask_b32_e32 v20, v0, v1, vcc\nv_cmp_lt_f32_e32 vcc, v0, v2\ns_nop 1\nv_cndmask_b32_e32 v21, v0, v1, vcc\n") ::: CLOB, "vcc");
        if (V == 51) asm volatile(REP16(OP4("v_bfi_b32", "v0", "v1", "v2")) ::: CLOB);
        if (V == 52) asm volatile(REP16(OP4("v_div_fixup_f32", "v0", "v1", "v2")) ::: CLOB);
        if (V == 53) asm volatile(REP16(OP4("v_div_fmas_f32", "v0", "v1", "v2")) ::: CLOB);
        if (V == 54) asm volatile(REP16("v_div_scale_f32 v20, vcc, v0, v1, v0\nv_div_scale_f32 v21, vcc, v0, v1, v0\nv_div_scale_f32 v22, vcc, v0, v1, v0\nv_div_scale_f32 v23, vcc, v0, v1, v0\n") ::: CLOB, "vcc");
        if (V == 55) asm volatile(REP16("v_rcp_f32_e32 v20, v0\nv_rcp_f32_e32 v21, v1\nv_rcp_f32_e32 v22, v2\nv_rcp_f32_e32 v23, v3\n") ::: CLOB);
        if (V == 56) asm volatile(REP16("v_rcp_f32_e32 v20, v0\nv_mul_f32 v21, v0, v1\nv_mul_f32 v22, v0, v1\nv_mul_f32 v23, v0, v1\n") ::: CLOB);   // does a transcendental overlap with VALU ops?
        if (V == 57) asm volatile(REP16("v_min_f32 v20, v0, v1\nv_min_f32 v21, v0, v1\nv_sub_f32 v22, v0, v1\nv_add_f32 v23, v0, v1\n") ::: CLOB);
        if (V == 58) asm volatile(REP16("v_max_f32 v20, v0, v0\nv_max_f32 v21, v1, v1\nv_max_f32 v22, v2, v2\nv_max_f32 v23, v3, v3\n") ::: CLOB);
        if (V == 59) asm volatile(REP16("v_cmp_eq_u32_e32 vcc, s4, v1\nv_cmp_eq_u32_e32 vcc, s4, v2\nv_cmp_eq_u32_e32 vcc, s4, v3\nv_cmp_eq_u32_e32 vcc, s4, v0\n") ::: CLOB, "vcc", "s4");
        if (V == 60) asm volatile(REP16("v_cmp_lt_f32_e64 s[6:7], v0, v1\nv_cmp_lt_f32_e64 s[6:7], v0, v2\nv_cmp_lt_f32_e64 s[6:7], v0, v3\nv_cmp_lt_f32_e64 s[6:7], v0, v4\n") ::: CLOB, "s6", "s7");
        if (V == 61) asm volatile(REP16("ds_read_b128 v[20:23], v19\nv_mul_f32 v24, v0, v1\nv_mul_f32 v25, v0, v1\nv_mul_f32 v26, v0, v1\ns_waitcnt lgkmcnt(0)\n") ::: CLOB);   // (address 0; 3 VALU per LDS read)
        if (V == 37) asm volatile(REP16("v_pk_mul_f32 v[20:21], v[0:1], v[4:5]\nv_pk_add_f32 v[22:23], v[0:1], v[4:5]\nv_pk_mul_f32 v[24:25], v[0:1], v[2:3]\nv_pk_add_f32 v[26:27], v[0:1], v[2:3]\n") ::: CLOB);
    }
    float s;
    asm volatile("v_add_f32 %0, v20, v21\nv_add_f32 %0, %0, v22\nv_add_f32 %0, %0, v23" : "=v"(s) :: CLOB);
    if (s == 12345.678f) out[0] = s;
}
static const char* names[] = {"mul v0 v1", "mul v0 v4", "mul v0 v2", "mul v0 v8", "mul v0 v16", "fma v0 v1 v2", "fma v0 v4 v2", "fma v0 v1 v4", "fma v0 v4 v8", "fma v0 v1 v5",
                              "fma v0 v1 v3", "fma v0 v2 v6", "fma v0 v0 v1", "fmac d20-23, v0 v1", "fmac d20-23, v0 v4", "fmac d{20,24} v1 v2", "fmac d{21,25,22,26} v1 v2",
                              "fmac d{20,23,24,27} v1 v2", "fma v0 v1 1.0", "fma v0 v4 1.0", "fma v0 s4 v1", "fma v0 s4 v4",
                              "pk_fma [0:1] [2:3] [4:5]", "pk_fma [0:1] [4:5] [2:3]", "pk_fma [0:1] [4:5] [8:9]", "pk_fma [0:1] [2:3] [6:7]", "pk_fma bcast lo src0", "pk_fma bcast hi src0, s0/s1 same",
                              "min3 v0 v4 v2", "min3 v0 v1 v2", "cndmask v0 v4 s[4:5]", "cndmask v0 v1 s[4:5]", "fma v0 v4 v1", "fma v1 v0 v4", "fma v0 v1 v4", "fma v0 v4 v0",
                              "[mul, fma-conflict] x2", "pk_mul/pk_add mix",
                              "fmac odd dst, v0 v4", "fmac even dst, v0 v4", "fmac even dst, v1 v5", "fmac odd dst, v1 v5", "fma v0 v2 v1", "fma v1 v3 v5", "fma v1 v5 v0", "fma v0 v2 v4",
                              "cndmask_e32 v0 v1 vcc", "cndmask_e32 v0 v4 vcc", "cmp_e32 -> vcc", "[cmp, cndmask] x2 (no nop!)", "[cmp, s_nop 1, cndmask] x2", "bfi v0 v1 v2", "div_fixup", "div_fmas", "div_scale",
                              "rcp x4", "[rcp, mul, mul, mul]", "[min, min, sub, add]", "max x,x (canonicalize)", "cmp_eq_u32 s4, v", "cmp_e64 -> sgpr pair", "[ds_read_b128 + 3 mul + wait]"};
template <int V>
void run(float* d) {
    printf("%-28s", names[V]);
    for (int w : {2, 4, 8}) {
        const int iters = 50000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<V>), dim3(256 * w), dim3(256), 0, 0, d, iters / 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V>), dim3(256 * w), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const int per_iter = V == 50 ? 64 : (V == 61 ? 64 : 64);  // (s_nop / s_waitcnt are not counted: cycles per VALU / LDS instruction of the block)
        const double rate = (double)iters * per_iter * w / (ms * 1e-3);
        printf("  w%d: %.2f", w, 2.4e9 / rate);
    }
    printf("\n");
}
int main() {
    float* d;
    (void)hipMalloc(&d, 4);
    printf("cycles of one SIMD per wave64 VALU instruction (2.4 GHz) by source registers, w = waves per SIMD\n");
    run<0>(d); run<1>(d); run<2>(d); run<3>(d); run<4>(d); run<5>(d); run<6>(d); run<7>(d); run<8>(d); run<9>(d); run<10>(d); run<11>(d); run<12>(d);
    run<13>(d); run<14>(d); run<15>(d); run<16>(d); run<17>(d); run<18>(d); run<19>(d); run<20>(d); run<21>(d);
    run<22>(d); run<23>(d); run<24>(d); run<25>(d); run<26>(d); run<27>(d); run<28>(d); run<29>(d); run<30>(d); run<31>(d); run<32>(d); run<33>(d); run<34>(d); run<35>(d); run<36>(d); run<37>(d);
    run<38>(d); run<39>(d); run<40>(d); run<41>(d); run<42>(d); run<43>(d); run<44>(d); run<45>(d); run<46>(d); run<47>(d); run<48>(d); run<49>(d); run<50>(d); run<51>(d); run<52>(d); run<53>(d); run<54>(d); run<55>(d); run<56>(d); run<57>(d); run<58>(d); run<59>(d); run<60>(d); run<61>(d);
    return 0;
}
