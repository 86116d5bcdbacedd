// drng.h -- the reference's PCG32 variant (crates/akari_render/src/sampler/mod.rs:73-217) on the device,
// plus xxhash32_4 (util/hash.rs:44-60) and mix_bits (util/mod.rs:305-319).
#pragma once
#include "dmath.h"

namespace akr {

constexpr uint64_t kPcgMult = 0x5851f42d4c957f2dull;

struct Pcg32 {
    uint64_t state, inc;
};

AKR_HD uint32_t pcg_gen_u32(Pcg32& p) {  // sampler/mod.rs:101-113
    uint64_t old = p.state;
    p.state = old * kPcgMult + p.inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
}
AKR_HD Pcg32 pcg_new_seq_offset(uint64_t seq, uint64_t seed) {  // sampler/mod.rs:88-94
    Pcg32 p{0, (seq << 1) | 1u};
    pcg_gen_u32(p);
    p.state += seed;
    pcg_gen_u32(p);
    return p;
}
AKR_HD uint64_t mix_bits(uint64_t v) {
    v ^= v >> 31;
    v *= 0x7fb5d329728ea185ull;
    v ^= v >> 27;
    v *= 0x81dadef4bc2dd44dull;
    v ^= v >> 33;
    return v;
}
AKR_HD Pcg32 pcg_new_seq(uint64_t seq) { return pcg_new_seq_offset(seq, mix_bits(seq)); }

// sampler/mod.rs:115-131, restated verbatim: this is NOT the canonical PCG jump-ahead, but it is what
// defines the reference's sample stream (start() = advance(16384), drop = advance(-dim)).
AKR_HD void pcg_advance(Pcg32& p, int64_t idelta) {
    uint64_t cur_mult = kPcgMult, cur_plus = p.inc, acc_mult = 1, acc_plus = 0;
    uint64_t delta = (uint64_t)idelta;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus + cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    p.state = acc_mult * p.state + acc_plus;
}

// advance(16384) in closed form. 16384 = 1 << 14 has one set bit, so the loop above reduces to
//   state' = A * state + (A + C * inc),  A = MULT^(2^14),  C = prod_{k<14} (MULT^(2^k) + 1)   (mod 2^64)
// A and C are constants of the generator; they are computed once on the host with the loop itself
// (host/scene_build.cpp: pcg_start_constants) and checked against pcg_advance in the tests.
struct PcgStartConsts {
    uint64_t A, C;
};
AKR_HD void pcg_start(Pcg32& p, PcgStartConsts k) { p.state = k.A * p.state + (k.A + k.C * p.inc); }

AKR_HD float pcg_next_1d(Pcg32& p) {  // sampler/mod.rs:194-198; can return exactly 1.0
    uint32_t n = pcg_gen_u32(p);
    return (float)n * 2.3283064365386963e-10f;  // f32(1.0 / u32::MAX as f64) == 2^-32
}

// ---- Owen-scrambled Sobol' (0,2)-sequence, for the "sobol" sampler (no reference counterpart: akari_data only stubs the
// matrices). Dimension 0 = radical inverse base 2, dimension 1 = the Pascal-triangle generator matrix; nested uniform
// scrambling by the Laine-Karras hash between two bit reversals (Laine & Karras 2011; Burley 2020, "Practical Hash-based
// Owen Scrambling").
AKR_HD uint32_t reverse_bits32(uint32_t x) { return __builtin_bitreverse32(x); }  // v_bfrev_b32
AKR_HD uint32_t laine_karras(uint32_t x, uint32_t seed) {
    x += seed;
    x ^= x * 0x6c50b47cu;
    x ^= x * 0xb82f1e52u;
    x ^= x * 0xc7afe638u;
    x ^= x * 0x8d22f6e6u;
    return x;
}
AKR_HD uint32_t owen_scramble(uint32_t x, uint32_t seed) { return reverse_bits32(laine_karras(reverse_bits32(x), seed)); }
// owen_scramble of a value that is known bit-reversed: owen_scramble(reverse_bits32(r), seed) without the two reversals that cancel
AKR_HD uint32_t owen_scramble_of_reversed(uint32_t r, uint32_t seed) { return reverse_bits32(laine_karras(r, seed)); }
AKR_HD uint32_t sobol_dim1(uint32_t i) {
    uint32_t v = 0x80000000u, r = 0;
    for (; i; i >>= 1) {
        if (i & 1u) r ^= v;
        v ^= v >> 1;
    }
    return r;
}
// reverse_bits32(sobol_dim1(i)) without the loop. Column k of the generator matrix, reversed, is (1 + x)^k over GF(2): bit j of it
// is set iff j is a submask of k (Lucas). So bit j of the result is the parity of the set bits k of i with k a superset of j --
// the superset sum over the 5-bit lattice of bit POSITIONS, five butterfly steps on the word. (tests/test_sobol.py: equal to the
// loop for every index below 2^20 and random 32-bit ones.)
AKR_HD uint32_t sobol_dim1_reversed(uint32_t i) {
    i ^= (i >> 1) & 0x55555555u;
    i ^= (i >> 2) & 0x33333333u;
    i ^= (i >> 4) & 0x0f0f0f0fu;
    i ^= (i >> 8) & 0x00ff00ffu;
    i ^= (i >> 16) & 0x0000ffffu;
    return i;
}

// xxhash32_4(px, py, pz, pw) (util/hash.rs:44-60) in two halves: what does not depend on pz, and the rest. The index-based samplers
// hash (pixel, dimension, seed) for every dimension of a path; the compiler hoists the pixel's half out of the path loop by itself
// (keeping the halves in explicit per-lane fields cost 48 bytes of scratch in the force_diffuse kernel and 1.5 - 4 %: dropped).
AKR_HD uint32_t xxhash32_4_begin(uint32_t px, uint32_t py, uint32_t pw) {
    const uint32_t PRIME32_3 = 3266489917u, PRIME32_4 = 668265263u, PRIME32_5 = 374761393u;
    uint32_t h32 = pw + PRIME32_5 + px * PRIME32_3;
    h32 = PRIME32_4 * ((h32 << 17) | (h32 >> (32 - 17)));
    h32 = h32 + py * PRIME32_3;
    h32 = PRIME32_4 * ((h32 << 17) | (h32 >> (32 - 17)));
    return h32;
}
AKR_HD uint32_t xxhash32_4_end(uint32_t h32, uint32_t pz) {
    const uint32_t PRIME32_2 = 2246822519u, PRIME32_3 = 3266489917u, PRIME32_4 = 668265263u;
    h32 = h32 + pz * PRIME32_3;
    h32 = PRIME32_4 * ((h32 << 17) | (h32 >> (32 - 17)));
    h32 = PRIME32_2 * (h32 ^ (h32 >> 15));
    h32 = PRIME32_3 * (h32 ^ (h32 >> 13));
    return h32 ^ (h32 >> 16);
}
// a % d for a divisor known in advance (Lemire, Kaser & Kurz 2019, "Faster remainder by direct computation"): with
// M = floor((2^64 - 1) / d) + 1, a % d = floor(((M * a mod 2^64) * d) / 2^64) for every 32-bit a and d > 0. Four multiplies instead
// of the thirty-odd instructions of a division by a run-time value. (tests/test_host.py checks it against % .)
AKR_HD uint64_t fastmod_magic(uint32_t d) { return 0xffffffffffffffffull / d + 1ull; }
AKR_HD uint32_t fastmod_u32(uint32_t a, uint64_t M, uint32_t d) {
    const uint64_t low = M * (uint64_t)a;
    const uint64_t t = (uint64_t)(uint32_t)low * d;                   // low 32 bits of `low` times d
    const uint64_t u = (uint64_t)(uint32_t)(low >> 32) * d + (t >> 32);  // + high 32 bits times d: bits 32 .. 95 of low * d
    return (uint32_t)(u >> 32);
}

AKR_HD uint32_t xxhash32_4(uint32_t px, uint32_t py, uint32_t pz, uint32_t pw) { return xxhash32_4_end(xxhash32_4_begin(px, py, pw), pz); }

}  // namespace akr
