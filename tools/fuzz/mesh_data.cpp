#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>
#include <fstream>
#include "scene_build.h"
using namespace akr;
extern "C" {
int32_t akr_pt_config_default(akr_pt_config* c) { std::memset(c, 0, sizeof *c); return 0; }
int32_t akr_aov_config_default(akr_aov_config* c) { std::memset(c, 0, sizeof *c); return 0; }
int32_t akr_gpt_config_default(akr_gpt_config* c) { std::memset(c, 0, sizeof *c); return 0; }
int32_t akr_mcmc_config_default(akr_mcmc_config* c) { std::memset(c, 0, sizeof *c); return 0; }
}
int main(int argc, char** argv) {
    std::string dir = argv[1];
    int iters = atoi(argv[2]);
    std::ifstream f(dir + "/Scene.bin.orig", std::ios::binary);
    std::vector<char> base((std::istreambuf_iterator<char>(f)), {});
    // every other run builds the 8-wide BVH for the 36 triangles too (round-2 advisor: one inf / NaN coordinate made the builder's
    // slot costs NaN and its greedy pairing write out of bounds; compile_scene now refuses non-finite corners before any build)
    std::mt19937 rng(7);
    size_t ok = 0, err = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<char> d = base;
        int n = 1 + rng() % 4;
        for (int i = 0; i < n; i++) {
            size_t o = (rng() % (d.size() / 4)) * 4;
            uint32_t vals[] = {0xffffffffu, 0x7fffffffu, 1000000u, 0x7f800000u, 0x7fc00000u, 0xff800000u, 0u, 36u};
            uint32_t v = vals[rng() % 8];
            std::memcpy(d.data() + o, &v, 4);
        }
        { std::ofstream o(dir + "/Scene.bin", std::ios::binary); o.write(d.data(), (std::streamsize)d.size()); }
        tuning_set("force_bvh", it & 1);
        try { FlatScene fs = load_scene_json(dir + "/scene.json"); CompiledScene cs; compile_scene(fs, cs); ok++; }
        catch (const std::exception&) { err++; }
    }
    printf("mesh data: %zu compiled, %zu refused\n", ok, err);
}
