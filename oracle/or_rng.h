/* or_rng.h -- PCG32 (reference variant), StdRng(ChaCha12) seed stream, xxhash32_4, mix_bits.
 * TEST INFRASTRUCTURE ONLY. Follows crates/akari_render/src/sampler/mod.rs:73-217,
 * util/mod.rs:305-319, util/hash.rs:44-60 of the reference.
 */
#ifndef OR_RNG_H
#define OR_RNG_H
#include <stdint.h>

#define PCG32_MULT 0x5851f42d4c957f2dull /* sampler/mod.rs:84 */

typedef struct { uint64_t state, inc; } or_pcg32;

/* sampler/mod.rs:101-113 */
static inline uint32_t pcg_gen_u32(or_pcg32 *p) {
    uint64_t old = p->state;
    p->state = old * PCG32_MULT + p->inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
}
/* sampler/mod.rs:88-94 set_seq_offset */
static inline or_pcg32 pcg_new_seq_offset(uint64_t seq, uint64_t seed) {
    or_pcg32 p = {0, (seq << 1) | 1u};
    pcg_gen_u32(&p);
    p.state += seed;
    pcg_gen_u32(&p);
    return p;
}
/* util/mod.rs:305-319 */
static inline uint64_t or_mix_bits(uint64_t v) {
    v ^= v >> 31; v *= 0x7fb5d329728ea185ull;
    v ^= v >> 27; v *= 0x81dadef4bc2dd44dull;
    v ^= v >> 33;
    return v;
}
/* sampler/mod.rs:98-100 */
static inline or_pcg32 pcg_new_seq(uint64_t seq) { return pcg_new_seq_offset(seq, or_mix_bits(seq)); }

/* sampler/mod.rs:115-131. NOT the canonical PCG jump-ahead (acc_plus gets cur_mult + cur_plus added,
 * cur_plus is multiplied rather than accumulated) -- restated verbatim because it defines the stream. */
static inline void pcg_advance(or_pcg32 *p, int64_t idelta) {
    uint64_t cur_mult = PCG32_MULT, cur_plus = p->inc, acc_mult = 1, acc_plus = 0;
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
    p->state = acc_mult * p->state + acc_plus;
}
/* sampler/mod.rs:194-198: u32 -> f32 (round to nearest) times f32(1/u32::MAX); may return 1.0 */
static inline float pcg_next_1d(or_pcg32 *p) {
    uint32_t n = pcg_gen_u32(p);
    return (float)n * (float)(1.0 / 4294967295.0);
}

/* ---- rand 0.8.5 StdRng::seed_from_u64(seed) followed by gen::<u64>() (sampler/mod.rs:150-151) ----
 * Third-party algorithm restated from its published definition (rand_core 0.6 SeedableRng::seed_from_u64
 * = PCG32-XSH-RR fill of the 32-byte key; rand_chacha 0.3 ChaCha12Rng: 64-bit block counter in words
 * 12-13, stream id 0 in words 14-15, output words consumed in order, a u64 = lo word | hi word << 32).
 * The ChaCha core is pinned by the known-answer vectors in tests (RFC 7539 ChaCha20 block, and the
 * all-zero-key ChaCha8/12/20 keystreams). */
typedef struct { uint32_t key[8]; uint64_t counter; uint32_t buf[16]; int idx; } or_stdrng;

#define OR_ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define OR_QR(a, b, c, d)                                                                   \
    a += b; d ^= a; d = OR_ROTL32(d, 16); c += d; b ^= c; b = OR_ROTL32(b, 12);              \
    a += b; d ^= a; d = OR_ROTL32(d, 8);  c += d; b ^= c; b = OR_ROTL32(b, 7)

static inline void or_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                                   uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = s[i];
    for (int r = 0; r < rounds; r += 2) {
        OR_QR(x[0], x[4], x[8], x[12]); OR_QR(x[1], x[5], x[9], x[13]);
        OR_QR(x[2], x[6], x[10], x[14]); OR_QR(x[3], x[7], x[11], x[15]);
        OR_QR(x[0], x[5], x[10], x[15]); OR_QR(x[1], x[6], x[11], x[12]);
        OR_QR(x[2], x[7], x[8], x[13]); OR_QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
static inline void or_stdrng_seed_from_u64(or_stdrng *r, uint64_t state) {
    for (int i = 0; i < 8; i++) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        r->key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    r->counter = 0;
    r->idx = 16;
}
static inline uint32_t or_stdrng_next_u32(or_stdrng *r) {
    if (r->idx >= 16) { or_chacha_block(r->key, r->counter++, 0, 12, r->buf); r->idx = 0; }
    return r->buf[r->idx++];
}
static inline uint64_t or_stdrng_next_u64(or_stdrng *r) {
    uint64_t lo = or_stdrng_next_u32(r);
    uint64_t hi = or_stdrng_next_u32(r);
    return lo | (hi << 32);
}

/* util/hash.rs:44-60 */
static inline uint32_t or_xxhash32_4(uint32_t px, uint32_t py, uint32_t pz, uint32_t pw) {
    const uint32_t PRIME32_2 = 2246822519u, PRIME32_3 = 3266489917u, PRIME32_4 = 668265263u,
                   PRIME32_5 = 374761393u;
    uint32_t h32 = pw + PRIME32_5 + px * PRIME32_3;
    h32 = PRIME32_4 * ((h32 << 17) | (h32 >> (32 - 17)));
    h32 = h32 + py * PRIME32_3;
    h32 = PRIME32_4 * ((h32 << 17) | (h32 >> (32 - 17)));
    h32 = h32 + pz * PRIME32_3;
    h32 = PRIME32_4 * ((h32 << 17) | (h32 >> (32 - 17)));
    h32 = PRIME32_2 * (h32 ^ (h32 >> 15));
    h32 = PRIME32_3 * (h32 ^ (h32 >> 13));
    return h32 ^ (h32 >> 16);
}
#endif
