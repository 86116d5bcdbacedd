/* or_math.h -- f32 vector helpers and deterministic elementary functions for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing in the shipped HIP path may include,
 * link or call this file.
 *
 * Arithmetic contract ("AKR-F32", shared by definition -- not by code -- with the HIP kernels):
 *   - every operation is IEEE-754 binary32, round-to-nearest-even;
 *   - no implicit contraction: the file is built with -ffp-contract=off, a fused multiply-add
 *     happens only where fmaf() is written;
 *   - sin/cos/log are the polynomial kernels below (not libm), so that a GPU and a CPU that both
 *     follow this text produce identical bits;
 *   - dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z, vec/scalar = vec * (1/scalar).
 * The reference leaves these details to its JIT back end (LuisaCompute; source absent from
 * /root/reference), so they are pinned here instead.
 */
#ifndef OR_MATH_H
#define OR_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

#define OR_PI 3.14159265358979323846f
#define OR_INV_PI 0.31830988618379067154f /* std::f32::consts::FRAC_1_PI */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v2 V2(float x, float y) { v2 r = {x, y}; return r; }
static inline v3 v3add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v3neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float v3dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3len2(v3 a) { return v3dot(a, a); }
static inline float v3len(v3 a) { return sqrtf(v3dot(a, a)); }
/* vec / scalar := vec * (1/scalar) (one IEEE division) */
static inline v3 v3divs(v3 a, float s) { float inv = 1.0f / s; return v3scale(a, inv); }
static inline v3 v3normalize(v3 a) { return v3divs(a, v3len(a)); }
static inline float or_min(float a, float b) { return a < b ? a : b; } /* b if a is NaN */
static inline float or_max(float a, float b) { return a > b ? a : b; } /* b if a is NaN */
static inline float or_clamp(float x, float lo, float hi) { return or_min(or_max(x, lo), hi); }
static inline float or_sqr(float x) { return x * x; }
static inline float or_lerp(float a, float b, float t) { return a + (b - a) * t; }
static inline v3 v3lerp(v3 a, v3 b, float t) {
    return V3(or_lerp(a.x, b.x, t), or_lerp(a.y, b.y, t), or_lerp(a.z, b.z, t));
}
static inline float v3max(v3 a) { return or_max(or_max(a.x, a.y), a.z); }
static inline float v3min(v3 a) { return or_min(or_min(a.x, a.y), a.z); }
static inline int or_isfinite(float x) { return (f2u(x) & 0x7f800000u) != 0x7f800000u; }
static inline int or_isnan(float x) { return x != x; }

/* ---- deterministic sin/cos: Cody-Waite reduction by pi/2 + Cephes single-precision kernels ---- */
static inline void or_sincosf(float x, float *s_out, float *c_out) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float P1 = 1.5703125f;                /* pi/2 split in three parts */
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.54978995489188216e-8f;
    float kf = __builtin_rintf(x * TWO_OVER_PI);
    float r = fmaf(-kf, P1, x);
    r = fmaf(-kf, P2, r);
    r = fmaf(-kf, P3, r);
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sn = fmaf(r * z, ps, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cs = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    int k = (int)kf;
    float s = (k & 1) ? cs : sn;
    float c = (k & 1) ? sn : cs;
    if (k & 2) s = -s;
    if ((k + 1) & 2) c = -c;
    *s_out = s;
    *c_out = c;
}

/* ---- deterministic natural log (Cephes logf); log(0) = -inf, log(<0) = NaN ---- */
static inline float or_logf(float x) {
    if (x == 0.0f) return -INFINITY;
    if (!(x > 0.0f)) return NAN;
    uint32_t ux = f2u(x);
    int e = (int)(ux >> 23) - 126;                       /* x = m * 2^e, m in [0.5,1) (normal x) */
    float m = u2f((ux & 0x007fffffu) | 0x3f000000u);
    if ((ux >> 23) == 0) {                               /* subnormal: scale up by 2^24 first */
        float xs = x * 16777216.0f;
        ux = f2u(xs);
        e = (int)(ux >> 23) - 126 - 24;
        m = u2f((ux & 0x007fffffu) | 0x3f000000u);
    }
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = fmaf(p, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (m * z) * p;
    float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

/* util/mod.rs:326-331 difference_of_products(a,b,c,d) = a*b - c*d with one-ulp error term */
/* e^x: range reduction x = g + n ln2 (two-constant Cody-Waite), Cephes expf polynomial, exact two-step scaling.
 * Part of the AKR-F32 contract: this is what "exp" / "powf" mean on both sides. */
static inline float or_expf(float x) {
    if (or_isnan(x)) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.278929903431851103f) return 0.0f;
    float fn = floorf(1.44269504088896341f * x + 0.5f);
    float g = fmaf(-0.693359375f, fn, x);
    g = fmaf(2.12194440e-4f, fn, g);
    float z = g * g;
    float p = 1.9875691500e-4f;
    p = fmaf(p, g, 1.3981999507e-3f);
    p = fmaf(p, g, 8.3334519073e-3f);
    p = fmaf(p, g, 4.1665795894e-2f);
    p = fmaf(p, g, 1.6666665459e-1f);
    p = fmaf(p, g, 5.0000001201e-1f);
    float r = fmaf(p, z, g) + 1.0f;
    int n = (int)fn, n1 = n / 2, n2 = n - n1;
    r = r * u2f((uint32_t)(n1 + 127) << 23);
    return r * u2f((uint32_t)(n2 + 127) << 23);
}
static inline float or_powf(float x, float y) { return x == 0.0f ? 0.0f : or_expf(y * or_logf(x)); }

static inline float or_dop(float a, float b, float c, float d) {
    float cd = c * d;
    float diff = fmaf(a, b, -cd);
    float err = fmaf(-c, d, cd);
    return diff + err;
}
#endif
