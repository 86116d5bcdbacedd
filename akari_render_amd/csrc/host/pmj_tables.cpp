// pmj_tables.cpp -- the two tables of the PMJ02BN sampler (crates/akari_render/src/sampler/mod.rs:329-470 reads them from
// akari_data::{pmj02bn, bluenoise}, copies of pbrt-v4's generated tables that are ABSENT from the reference tree here).
// Both are regenerated, same shapes and roles, different values (pmj02bn images are not comparable bit-for-bit
// with the reference's):
//   * 5 point sets x 65536 points, u32 fixed point: progressive multi-jittered (0,2) sequences. An Owen-scrambled Sobol'
//     (0,2)-sequence has exactly the stratification pmj02 asks for (every prefix of 2^k points is a (0,k,2)-net), so each
//     set is the first two Sobol' dimensions under a nested uniform scramble (Laine-Karras hash), one seed pair per set.
//   * 48 blue-noise dither arrays 128 x 128 (u16), void-and-cluster, generated offline by tools/make_bluenoise.py and
//     shipped as akari_render_amd/data/bluenoise_128x128x48_u16.bin next to the library.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene_build.h"

namespace akr {
namespace {
uint32_t reverse_bits(uint32_t x) {
    x = (x >> 16) | (x << 16);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    return x;
}
// Laine & Karras 2011, "Stratified sampling for stochastic transparency": a hash whose every output bit depends only on
// LOWER input bits; between two bit reversals it is a nested uniform (Owen) scramble.
uint32_t laine_karras(uint32_t x, uint32_t seed) {
    x += seed;
    x ^= x * 0x6c50b47cu;
    x ^= x * 0xb82f1e52u;
    x ^= x * 0xc7afe638u;
    x ^= x * 0x8d22f6e6u;
    return x;
}
uint32_t owen_scramble(uint32_t x, uint32_t seed) { return reverse_bits(laine_karras(reverse_bits(x), seed)); }
uint32_t sobol_dim1(uint32_t i) {  // second Sobol' dimension: generator matrix = Pascal triangle mod 2
    uint32_t v = 0x80000000u, r = 0;
    for (; i; i >>= 1) {
        if (i & 1u) r ^= v;
        v ^= v >> 1;
    }
    return r;
}
uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
std::string data_dir();
// a raw little-endian table file of exactly `bytes` bytes from the data directory; false when the file does not exist
bool read_table(const char* name, void* dst, size_t bytes, bool required) {
    const std::string path = data_dir() + "/" + name;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        if (required) throw std::runtime_error("cannot open '" + path + "' (table of the pmj02bn sampler; set AKR_DATA_DIR or run tools/make_bluenoise.py)");
        return false;
    }
    size_t got = std::fread(dst, 1, bytes, f);
    bool more = std::fgetc(f) != EOF;
    std::fclose(f);
    if (got != bytes || more) throw std::runtime_error("'" + path + "' has the wrong size");
    return true;
}
std::string library_dir() {
    Dl_info info;
    if (dladdr((const void*)&library_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t k = p.find_last_of('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}
std::string data_dir() {
    if (const char* e = std::getenv("AKR_DATA_DIR")) return e;
    return library_dir() + "/data";
}
}  // namespace

void make_pmj02_sets(std::vector<uint32_t>& out) {
    const uint32_t n_sets = 5, n = 65536;
    out.resize((size_t)n_sets * n * 2);
    // the reference's own table (akari_data::pmj02bn::PMJ02BN_SAMPLES dumped as raw u32 pairs), when somebody supplies it
    if (read_table("pmj02bn_5x65536x2_u32.bin", out.data(), out.size() * 4, false)) return;
    for (uint32_t s = 0; s < n_sets; s++) {
        const uint32_t sx = mix32(0x9e3779b9u * (2 * s + 1)), sy = mix32(0x85ebca6bu * (2 * s + 2));
        for (uint32_t i = 0; i < n; i++) {
            out[2 * ((size_t)s * n + i) + 0] = owen_scramble(reverse_bits(i), sx);
            out[2 * ((size_t)s * n + i) + 1] = owen_scramble(sobol_dim1(i), sy);
        }
    }
}

void load_bluenoise(std::vector<uint16_t>& out) {
    out.resize(48ull * 128 * 128);
    read_table("bluenoise_128x128x48_u16.bin", out.data(), out.size() * 2, true);
}

}  // namespace akr
