// stdrng.h -- the seed stream of `StdRng::seed_from_u64(seed)` followed by `gen::<u64>()`, which the reference
// uses to seed every per-pixel PCG32 (crates/akari_render/src/sampler/mod.rs:148-151; rand 0.8.5, rand_chacha
// 0.3.1, rand_core 0.6.4 per Cargo.lock:1762-1776). Third-party algorithm, restated from its published
// definition: seed_from_u64 expands the u64 with a PCG32 into a 256-bit ChaCha key; StdRng is ChaCha12 with a
// 64-bit block counter (words 12-13) and stream id 0; words are consumed in order and a u64 is lo | hi << 32.
#pragma once
#include <stdint.h>

namespace akr {

class StdRng {
   public:
    explicit StdRng(uint64_t seed) {
        uint64_t state = seed;
        for (int i = 0; i < 8; i++) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            key_[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
    }
    uint32_t next_u32() {
        if (idx_ >= 16) {
            block(counter_++);
            idx_ = 0;
        }
        return buf_[idx_++];
    }
    uint64_t next_u64() {
        uint64_t lo = next_u32();
        uint64_t hi = next_u32();
        return lo | (hi << 32);
    }
    // one ChaCha block with `rounds` rounds (exposed for the known-answer tests)
    static void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6],
                          key[7], (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
        uint32_t x[16];
        for (int i = 0; i < 16; i++) x[i] = s[i];
        auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 16);
            x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 12);
            x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 8);
            x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 7);
        };
        for (int r = 0; r < rounds; r += 2) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
    }

   private:
    void block(uint64_t counter) { chacha_block(key_, counter, 0, 12, buf_); }
    uint32_t key_[8];
    uint32_t buf_[16];
    uint64_t counter_ = 0;
    int idx_ = 16;
};

}  // namespace akr
