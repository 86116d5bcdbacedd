// akari-cli -- the reference's command line (crates/akari_api/src/bin/akari_cli.rs:8-95) over libakari_hip.so:
//   akari-cli -s scene.json -m method.json [-d <hip device ordinal>] [-v] [--save-intermediate] [--save-stats NAME]
//             [--resolution WxH] [--independent-sampler]
// -d accepts a HIP device ordinal (the reference's "cpu|cuda|dx|metal" back ends do not exist here; "hip" = 0).
// --gui is not supported. --independent-sampler renders method files that ask for pmj02bn (scenes/cbox/pt.json)
// with the independent sampler and the same seed.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "akari_hip.h"

static void usage() {
    std::puts("Usage: akari-cli -s <SCENE> -m <METHOD> [-d <DEVICE>] [-v] [--save-intermediate] [--save-stats <NAME>]\n"
              "                 [--resolution <W>x<H>] [--independent-sampler]\n"
              "  -s, --scene <SCENE>      Scene file to render (akari scene-graph JSON)\n"
              "  -m, --method <METHOD>    Render method config file (\"type\": \"pt\")\n"
              "  -d, --device <DEVICE>    HIP device ordinal (default 0)\n"
              "  -v, --verbose\n"
              "      --save-intermediate  write {name}-{spp}.exr after every pass\n"
              "      --save-stats <NAME>  write NAME.json (RenderStats) and use NAME for intermediate files");
}

// akari-cli --spec-compile <header file> <out.co> <arch> <flags> <min waves>: the library's helper process for per-scene kernels
// (host/specialise.cpp). The kernel is compiled HERE, in a process that holds nothing but this library and the ROCm installation's
// hiprtc, so that the host application's own copies of the ROCm compiler libraries (PyTorch ships its own) cannot change the code.
static int spec_compile_main(int argc, char** argv) {
    if (argc != 7) { std::fputs("usage: akari-cli --spec-compile <header> <out.co> <arch> <flags> <min_waves>\n", stderr); return 2; }
    std::ifstream hf(argv[2]);
    if (!hf) { std::fprintf(stderr, "akari-cli: cannot open %s\n", argv[2]); return 2; }
    std::stringstream ss;
    ss << hf.rdbuf();
    if (akr_host_spec_compile_text(ss.str().c_str(), (uint32_t)std::atoi(argv[5]), (uint32_t)std::atoi(argv[6]), argv[4], argv[3]) != AKR_OK) {
        std::fprintf(stderr, "%s\n", akr_last_error());
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--spec-compile") return spec_compile_main(argc, argv);
    std::string scene, method, name;
    int device = 0, verbose = 0, save_intermediate = 0, save_stats = 0, indep = 0;
    unsigned w = 0, h = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); std::exit(1); } return argv[++i]; };
        if (a == "-s" || a == "--scene") scene = next();
        else if (a == "-m" || a == "--method") method = next();
        else if (a == "-d" || a == "--device") { std::string d = next(); device = (d == "hip" || d == "gpu") ? 0 : std::atoi(d.c_str()); }
        else if (a == "-v" || a == "--verbose") verbose = 1;
        else if (a == "--save-intermediate") save_intermediate = 1;
        else if (a == "--save-stats") { name = next(); save_stats = 1; }
        else if (a == "--independent-sampler") indep = 1;
        else if (a == "--resolution") { if (std::sscanf(next(), "%ux%u", &w, &h) != 2) { usage(); return 1; } }
        else if (a == "--gui") { std::fputs("akari-cli: --gui is not supported by the HIP integrator\n", stderr); return 1; }
        else { usage(); return 1; }
    }
    if (scene.empty() || method.empty()) { usage(); return 1; }
    std::ifstream mf(method);
    if (!mf) { std::fprintf(stderr, "akari-cli: cannot open %s\n", method.c_str()); return 1; }
    std::stringstream ss;
    ss << mf.rdbuf();
    akr_context* ctx = nullptr;
    akr_scene* sc = nullptr;
    auto die = [&](const char* what) { std::fprintf(stderr, "akari-cli: %s: %s\n", what, akr_last_error()); std::exit(1); };
    if (akr_context_create(device, &ctx) != AKR_OK) die("device");
    if (akr_scene_load(ctx, scene.c_str(), w, h, &sc) != AKR_OK) die("scene");
    akr_render_session ses;
    ses.save_intermediate = save_intermediate;
    ses.save_stats = save_stats;
    ses.name = name.empty() ? nullptr : name.c_str();
    ses.override_sampler_independent = indep;
    ses.verbose = verbose;
    akr_pt_stats st;
    if (akr_render_task(ctx, sc, ss.str().c_str(), &ses, &st) != AKR_OK) die("render");
    std::printf("Rendering finished in %.2fs\n", st.kernel_ms * 1e-3);
    akr_scene_destroy(sc);
    akr_context_destroy(ctx);
    return 0;
}
