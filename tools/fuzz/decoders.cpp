#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>
#include <fstream>
#include <iostream>
#include "scene_build.h"
using namespace akr;
static std::vector<uint8_t> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
int main(int argc, char** argv) {
    // argv: kind file iterations
    std::string kind = argv[1];
    std::vector<uint8_t> base = slurp(argv[2]);
    int iters = atoi(argv[3]);
    std::mt19937 rng(1234);
    size_t ok = 0, err = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> d = base;
        int mode = rng() % 4;
        if (mode == 0) d.resize(rng() % (d.size() + 1));
        else if (mode == 1) { int n = 1 + rng() % 4; for (int i = 0; i < n; i++) d[rng() % d.size()] ^= (uint8_t)(1u << (rng() % 8)); }
        else if (mode == 2) { int n = 1 + rng() % 8; for (int i = 0; i < n; i++) d[rng() % d.size()] = (uint8_t)rng(); }
        else { size_t o = rng() % d.size(); uint32_t v = (rng() % 2) ? 0xffffffffu : (uint32_t)rng(); for (int i = 0; i < 4 && o + i < d.size(); i++) d[o + i] = (uint8_t)(v >> (8 * i)); }
        uint32_t w = 0, h = 0;
        try {
            if (kind == "exr") { std::vector<float> px; decode_exr(d.data(), d.size(), w, h, px); }
            else { std::vector<uint8_t> px;
                if (kind == "tiff") decode_tiff(d.data(), d.size(), w, h, px);
                else if (kind == "dds") decode_dds(d.data(), d.size(), w, h, px);
                else if (kind == "png") decode_png(d.data(), d.size(), w, h, px);
                else if (kind == "jpeg") decode_jpeg(d.data(), d.size(), w, h, px);
            }
            ok++;
        } catch (const std::exception&) { err++; }
    }
    printf("%s: %zu decoded, %zu refused\n", kind.c_str(), ok, err);
    return 0;
}
