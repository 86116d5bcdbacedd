#include <cstdio>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>
#include <fstream>
#include "scene_build.h"
using namespace akr;
int main(int argc, char** argv) {
    std::string dir = argv[1];
    int iters = atoi(argv[2]);
    std::ifstream f(dir + "/scene.json", std::ios::binary);
    std::string base((std::istreambuf_iterator<char>(f)), {});
    std::ifstream g(dir + "/pt.json", std::ios::binary);
    std::string mbase((std::istreambuf_iterator<char>(g)), {});
    std::mt19937 rng(99);
    size_t ok = 0, err = 0, mok = 0, merr = 0;
    const char* toks[] = {"{", "}", "[", "]", ",", ":", "\"", "-1", "1e999", "null", "true", "0", "4294967296", "\\u0000", "\"id\"", "-"};
    for (int it = 0; it < iters; it++) {
        std::string d = base;
        int n = 1 + rng() % 3;
        for (int i = 0; i < n; i++) {
            int mode = rng() % 6;
            size_t o = rng() % d.size();
            if (mode >= 4) {  // replace the number literal after position o
                size_t a = d.find_first_of("0123456789", o);
                if (a == std::string::npos) continue;
                size_t b = d.find_first_not_of("0123456789.eE+-", a);
                if (b == std::string::npos) continue;
                const char* nums[] = {"-1", "0", "4294967295", "1e30", "2147483648", "99999999999", "-0.0", "1e-40", "65536", "3"};
                d.replace(a, b - a, nums[rng() % 10]);
                continue;
            }
            if (mode == 0) d[o] = (char)(rng() % 96 + 32);
            else if (mode == 1) d.erase(o, 1 + rng() % 8);
            else if (mode == 2) d.insert(o, toks[rng() % 16]);
            else d.resize(o);
        }
        { std::ofstream o(dir + "/fuzz.json", std::ios::binary); o << d; }
        try { FlatScene fs = load_scene_json(dir + "/fuzz.json"); CompiledScene cs; compile_scene(fs, cs); ok++; }
        catch (const std::exception&) { err++; }
        std::string m = mbase;
        for (int i = 0; i < n; i++) {
            size_t o = rng() % m.size();
            int mode = rng() % 3;
            if (mode == 0) m[o] = (char)(rng() % 96 + 32); else if (mode == 1) m.erase(o, 1 + rng() % 6); else m.insert(o, toks[rng() % 16]);
        }
        try { auto t = parse_render_tasks(m, true); mok++; (void)t; } catch (const std::exception&) { merr++; }
    }
    printf("scene: %zu loaded, %zu refused; method: %zu parsed, %zu refused\n", ok, err, mok, merr);
}
// stand-ins for the four config defaults of api.cpp (zero-filled is enough for a parser fuzz)
#include <cstring>
extern "C" {
int32_t akr_pt_config_default(akr_pt_config* c) { std::memset(c, 0, sizeof *c); c->spp = 1; c->spp_per_pass = 1; return 0; }
int32_t akr_aov_config_default(akr_aov_config* c) { std::memset(c, 0, sizeof *c); return 0; }
int32_t akr_gpt_config_default(akr_gpt_config* c) { std::memset(c, 0, sizeof *c); return 0; }
int32_t akr_mcmc_config_default(akr_mcmc_config* c) { std::memset(c, 0, sizeof *c); return 0; }
}
