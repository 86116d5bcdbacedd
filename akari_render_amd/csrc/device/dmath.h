// dmath.h -- f32 vector type and elementary functions of the HIP path tracer (gfx950).
//
// Arithmetic contract (DESIGN.md "AKR-F32"): IEEE binary32, round-to-nearest-even, no implicit contraction
// (the library is built with -ffp-contract=off; fused multiply-adds appear only where fmaf is written),
// correctly rounded division and sqrt (-fhip-fp32-correctly-rounded-divide-sqrt), and sin/cos/log given by
// the polynomial kernels below instead of the device math library -- so a film rendered here is a pure
// function of (scene, config, seed), reproducible bit-for-bit on any IEEE machine that follows the same text.
#pragma once
#if !defined(__HIPCC_RTC__)  // hiprtc (per-scene kernels, host/specialise.cpp) brings the HIP runtime declarations and the fixed-width integers itself
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long int64_t;
typedef unsigned long uint64_t;
typedef unsigned long uintptr_t;
typedef unsigned long size_t;
#endif

#define AKR_HD __host__ __device__ __forceinline__
#define AKR_D __device__ __forceinline__

// The arithmetic tier of a translation unit. 0 (everything but pt_kernels_relaxed.hip): the AKR-F32 contract above, bit for bit the
// oracle's. 1 = RELAXED (option `arith`): the same algorithm to the tolerance north_star states (relRMSE < 1e-3
// against the oracle, tests/test_gpu_relaxed.py) instead of to the bit -- v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 (1 ulp) for the IEEE
// division and square root (11 and 16 instructions each under the contract), the hardware's sin / cos / log2 / exp2, and the
// compiler's contraction of a * b + c (that translation unit is built with -ffp-contract=fast, without correctly rounded
// divide / sqrt and with denormals flushed). Device code only: anything a relaxed translation unit compiles for the host keeps the contract.
#ifndef AKR_ARITH_RELAXED
#define AKR_ARITH_RELAXED 0
#endif
#if AKR_ARITH_RELAXED && defined(__HIP_DEVICE_COMPILE__)
#define AKR_RX 1
#else
#define AKR_RX 0
#endif
// parts of the relaxed tier that can be switched off one by one in a variant build (tools/arith_parts.sh: what each part buys and what it costs in flipped comparisons)
#ifndef AKR_RX_TRANS
#define AKR_RX_TRANS 1  // hardware sin / cos / log2 / exp2
#endif
#ifndef AKR_RX_RCP
#define AKR_RX_RCP 1    // v_rcp_f32 / v_rsq_f32 in the triangle test's plane solve and in vector normalisation
#endif

namespace akr {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;

struct vec2 {
    float x, y;
};
struct vec3 {
    float x, y, z;
};

AKR_HD vec2 mk2(float x, float y) { return vec2{x, y}; }
AKR_HD vec3 mk3(float x, float y, float z) { return vec3{x, y, z}; }
AKR_HD vec3 splat3(float v) { return vec3{v, v, v}; }
AKR_HD vec3 operator+(vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
AKR_HD vec3 operator-(vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
AKR_HD vec3 operator*(vec3 a, vec3 b) { return vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
AKR_HD vec3 operator*(vec3 a, float s) { return vec3{a.x * s, a.y * s, a.z * s}; }
AKR_HD vec3 operator-(vec3 a) { return vec3{-a.x, -a.y, -a.z}; }
AKR_HD float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
AKR_HD vec3 cross(vec3 a, vec3 b) { return vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// 1 / x, a / b, sqrt(x): correctly rounded under the contract; one hardware instruction (+ a multiply) in the relaxed tier
AKR_HD float rcp_f(float x) {
#if AKR_RX && AKR_RX_RCP
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
AKR_HD float div_f(float a, float b) {
#if AKR_RX && AKR_RX_RCP
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}
AKR_HD float sqrt_f(float x) {
#if AKR_RX
    return __builtin_amdgcn_sqrtf(x);
#else
    return __builtin_sqrtf(x);
#endif
}
AKR_HD float length2(vec3 a) { return dot(a, a); }
AKR_HD float length(vec3 a) { return sqrt_f(dot(a, a)); }
// vec / scalar := vec * (1 / scalar): one IEEE division
AKR_HD vec3 div_s(vec3 a, float s) {
    float inv = rcp_f(s);
    return a * inv;
}
AKR_HD vec3 normalize(vec3 a) {
#if AKR_RX && AKR_RX_RCP
    return a * __builtin_amdgcn_rsqf(dot(a, a));
#else
    return div_s(a, length(a));
#endif
}
AKR_HD float min_f(float a, float b) { return a < b ? a : b; }  // b when a is NaN
AKR_HD float max_f(float a, float b) { return a > b ? a : b; }  // b when a is NaN
AKR_HD float clamp_f(float x, float lo, float hi) { return min_f(max_f(x, lo), hi); }
AKR_HD float sqr(float x) { return x * x; }
AKR_HD float lerp_f(float a, float b, float t) { return a + (b - a) * t; }
AKR_HD vec3 lerp3(vec3 a, vec3 b, float t) { return vec3{lerp_f(a.x, b.x, t), lerp_f(a.y, b.y, t), lerp_f(a.z, b.z, t)}; }
AKR_HD float max3(vec3 a) { return max_f(max_f(a.x, a.y), a.z); }
AKR_HD float min3(vec3 a) { return min_f(min_f(a.x, a.y), a.z); }
AKR_HD float abs_f(float x) { return __builtin_fabsf(x); }

AKR_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
AKR_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
AKR_HD bool is_finite(float x) { return (f2u(x) & 0x7f800000u) != 0x7f800000u; }
AKR_HD bool is_nan(float x) { return x != x; }

// sin and cos of x (radians): Cody-Waite reduction by pi/2, then the Cephes single-precision kernels.
AKR_HD void sincos_f(float x, float& s_out, float& c_out) {
#if AKR_RX && AKR_RX_TRANS
    const float rev = x * 0.15915494309189535f;  // v_sin_f32 / v_cos_f32 take revolutions
    s_out = __builtin_amdgcn_sinf(rev);
    c_out = __builtin_amdgcn_cosf(rev);
    return;
#endif
    const float kTwoOverPi = 0.636619772367581343f;
    const float P1 = 1.5703125f, P2 = 4.837512969970703125e-4f, P3 = 7.54978995489188216e-8f;
    float kf = __builtin_rintf(x * kTwoOverPi);
    float r = __builtin_fmaf(-kf, P1, x);
    r = __builtin_fmaf(-kf, P2, r);
    r = __builtin_fmaf(-kf, P3, r);
    float z = r * r;
    float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sn = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cs = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    int k = (int)kf;
    float s = (k & 1) ? cs : sn;
    float c = (k & 1) ? sn : cs;
    if (k & 2) s = -s;
    if ((k + 1) & 2) c = -c;
    s_out = s;
    c_out = c;
}

// natural log (Cephes logf); log(0) = -inf, log(x < 0) = NaN
AKR_HD float log_f(float x) {
#if AKR_RX && AKR_RX_TRANS
    return __builtin_amdgcn_logf(x) * 0.6931471805599453f;  // v_log_f32 = log2; log2(0) = -inf, log2(x < 0) = NaN as below
#endif
    if (x == 0.0f) return -__builtin_inff();
    if (!(x > 0.0f)) return __builtin_nanf("");
    uint32_t ux = f2u(x);
    int e = (int)(ux >> 23) - 126;
    float m = u2f((ux & 0x007fffffu) | 0x3f000000u);
    if ((ux >> 23) == 0) {
        float xs = x * 16777216.0f;
        ux = f2u(xs);
        e = (int)(ux >> 23) - 126 - 24;
        m = u2f((ux & 0x007fffffu) | 0x3f000000u);
    }
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = (m * z) * p;
    float fe = (float)e;
    y = __builtin_fmaf(-2.12194440e-4f, fe, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(0.693359375f, fe, r);
    return r;
}

// e^x (Cephes expf: x = g + n ln2, degree-5 polynomial on g, scale by 2^n in two exact steps)
AKR_HD float exp_f(float x) {
#if AKR_RX && AKR_RX_TRANS
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
#endif
    if (x != x) return x;
    if (x > 88.72283905206835f) return __builtin_inff();
    if (x < -103.278929903431851103f) return 0.0f;
    float fn = __builtin_floorf(1.44269504088896341f * x + 0.5f);
    float g = __builtin_fmaf(-0.693359375f, fn, x);
    g = __builtin_fmaf(2.12194440e-4f, fn, g);
    float z = g * g;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, g, 1.3981999507e-3f);
    p = __builtin_fmaf(p, g, 8.3334519073e-3f);
    p = __builtin_fmaf(p, g, 4.1665795894e-2f);
    p = __builtin_fmaf(p, g, 1.6666665459e-1f);
    p = __builtin_fmaf(p, g, 5.0000001201e-1f);
    float r = __builtin_fmaf(p, z, g) + 1.0f;
    int n = (int)fn;                      // |n| <= 150
    int n1 = n / 2, n2 = n - n1;          // both in [-75, 75]: 2^n1 and 2^n2 are normal numbers
    r = r * u2f((uint32_t)(n1 + 127) << 23);
    return r * u2f((uint32_t)(n2 + 127) << 23);
}
// x^y for x > 0 as exp(y log x); pow_f(0, y > 0) = 0. This is what `powf` means in the AKR-F32 contract.
AKR_HD float pow_f(float x, float y) { return x == 0.0f ? 0.0f : exp_f(y * log_f(x)); }

// a*b - c*d with the rounding error of c*d folded back in (reference util/mod.rs:326-331)
AKR_HD float difference_of_products(float a, float b, float c, float d) {
    float cd = c * d;
    float diff = __builtin_fmaf(a, b, -cd);
    float err = __builtin_fmaf(-c, d, cd);
    return diff + err;
}

}  // namespace akr
